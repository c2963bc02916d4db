#!/bin/bash
# Collect the judged evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01      -> gpurun_out/<tag>/{bench_kernel_stats.csv, bench_pmc_summary.json, bench_line.json, ...}
# Counters are collected in their own passes, with --kernel-trace only (no sys/hip/hsa trace domains).
set -u
TAG=${1:-r00}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
export PYTHONUNBUFFERED=1 TMPDIR=/tmp
cd /tmp
rm -rf /tmp/prof_$TAG && mkdir -p /tmp/prof_$TAG
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats -o bench -- python $R/bench.py --steps 3 --warmup 1 --timed-only > $OUT/stats_bench.log 2>&1
cp /tmp/prof_$TAG/stats/bench_kernel_stats.csv $OUT/bench_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM"; do
  n=$(echo $c | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$TAG/pmc_$n -o bench -- python $R/bench.py --steps 1 --warmup 0 --timed-only > $OUT/pmc_$n.log 2>&1
done
python - <<PY
import csv, collections, json, re
def short(k):
    m = re.search(r'::(\w+)(<[^>]*>)?\(', k)
    return (m.group(1) + (m.group(2) or '')) if m else k[:40]
out = {}
for tag, n in [('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE'), ('sq', 'SQ_INSTS_VALU_MFMA_MOPS_BF16'), ('grbm', 'GRBM_GUI_ACTIVE')]:
    d = collections.OrderedDict()
    for r in csv.DictReader(open('/tmp/prof_$TAG/pmc_%s/bench_counter_collection.csv' % n)):
        k = short(r['Kernel_Name'])
        if not any(s in k for s in ('nerf_mlp_kernel', 'composite', 'sample_pdf')):
            continue
        d.setdefault((k, r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    out[tag] = [{'kernel': k[0], 'dispatch': k[1], **v} for k, v in d.items()]
sq = [x for x in out['sq'] if 'nerf_mlp_kernel' in x['kernel']][:2]
gr = [x for x in out['grbm'] if 'nerf_mlp_kernel' in x['kernel']][:2]
out['derived'] = [{'launch': i, 'mfma_pipe_busy_frac': a['SQ_VALU_MFMA_BUSY_CYCLES'] / (256 * 4 * b['GRBM_GUI_ACTIVE'] / 8),
                   'cycles_per_xcd': b['GRBM_GUI_ACTIVE'] / 8} for i, (a, b) in enumerate(zip(sq, gr))]
out['note'] = ("bench.py --steps 1 --warmup 0 --timed-only under rocprofv3 --pmc <one group per pass> --kernel-trace; FETCH_SIZE/WRITE_SIZE "
               "in KiB (FETCH_SIZE under-reports wide streaming reads by 2x on gfx950, MI355X_MICROARCH.md); dispatch order: coarse launch "
               "(81.92 M samples), fine launch (163.84 M samples)")
json.dump(out, open('$OUT/bench_pmc_summary.json', 'w'), indent=1)
print(json.dumps(out['derived']))
PY
cd $R
cp $OUT/bench_pmc_summary.json profiles/${TAG}_bench_pmc_summary.json 2>/dev/null
python bench.py --steps 5 --warmup 1 > $OUT/bench_line.json 2> $OUT/bench_stderr.log
tail -c 2500 $OUT/bench_line.json
# the labelled fast mode (i8x3): kernel stats + MFMA-pipe counters of its own timed-only run
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG/stats8 -o bench -- python $R/bench.py --steps 3 --warmup 1 --timed-only --precision i8x3 > $OUT/stats_bench_i8x3.log 2>&1
cp /tmp/prof_$TAG/stats8/bench_kernel_stats.csv $OUT/bench_i8x3_kernel_stats.csv
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_WAIT_ANY" FETCH_SIZE WRITE_SIZE; do
  n=$(echo $c | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_$TAG/pmc8_$n -o bench -- python $R/bench.py --steps 1 --warmup 0 --timed-only --precision i8x3 > $OUT/pmc8_$n.log 2>&1
done
cd /tmp
python - <<PY
import csv, collections, json
out = {}
for tag, n in [('sq', 'SQ_VALU_MFMA_BUSY_CYCLES'), ('grbm', 'GRBM_GUI_ACTIVE'), ('fetch', 'FETCH_SIZE'), ('write', 'WRITE_SIZE')]:
    d = collections.OrderedDict()
    for r in csv.DictReader(open('/tmp/prof_$TAG/pmc8_%s/bench_counter_collection.csv' % n)):
        if 'nerf_mlp_i8w_kernel' not in r['Kernel_Name']:
            continue
        d.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = float(r['Counter_Value'])
    out[tag] = [{'kernel': 'nerf_mlp_i8w_kernel', 'dispatch': k, **v} for k, v in d.items()]
out['derived'] = [{'launch': i, 'mfma_pipe_busy_frac': a['SQ_VALU_MFMA_BUSY_CYCLES'] / (256 * 4 * b['GRBM_GUI_ACTIVE'] / 8),
                   'cycles_per_xcd': b['GRBM_GUI_ACTIVE'] / 8} for i, (a, b) in enumerate(zip(out['sq'][:2], out['grbm'][:2]))]
out['note'] = "bench.py --steps 1 --warmup 0 --timed-only --precision i8x3 under rocprofv3 --pmc <one group per pass> --kernel-trace"
json.dump(out, open('$OUT/bench_i8x3_pmc_summary.json', 'w'), indent=1)
print(json.dumps(out['derived']))
PY
cd $R
# the in-kernel cycle buckets and the clock / power samples quoted in DESIGN.md section 6
python bench.py --steps 5 --warmup 1 --precision i8x3 --no-cpu-baseline > $OUT/bench_line_i8x3.json 2>> $OUT/bench_stderr.log
(python tools/mlp_profile.py; python tools/mlp_profile_i8.py) > $OUT/mlp_profile.log 2>&1
for p in bf16x3 i8x3 bf16; do tools/clock_watch.sh $p; done > $OUT/clock_watch.log 2>&1
python tools/mlp_power_probe.py > $OUT/mlp_power_probe.log 2>&1
for f in bench_line.json bench_line_i8x3.json mlp_profile.log clock_watch.log mlp_power_probe.log bench_kernel_stats.csv; do cp $OUT/$f profiles/${TAG}_$f 2>/dev/null; done
