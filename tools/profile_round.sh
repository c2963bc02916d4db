#!/bin/bash
# Collect the judged evidence for one round on the GPU box (run through gpurun):
#   tools/profile_round.sh r01      -> gpurun_out/<tag>/{bench_kernel_stats.csv, bench_pmc_summary.json, bench_line.json, ...}
# Counters are collected in their own passes, with --kernel-trace only (no sys/hip/hsa trace domains).
# The profiled command is `bench.py --timed-only` in the default (mixed) precision: every MLP launch rocprofv3 sees is a
# timed one -- nerf_sigma_f16t_kernel = the coarse (fp16x3, density only) launch, nerf_mlp_i8s_kernel = the fine (i8x3) launch.
# FULL=1 also collects the in-kernel cycle buckets, the clock / power samples and the zero-weight probe (minutes of GPU time).
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
        if not any(s in k for s in ('nerf_mlp', 'nerf_sigma', 'composite', 'sample_pdf')):
            continue
        d.setdefault((k, r['Dispatch_Id']), {})[r['Counter_Name']] = float(r['Counter_Value'])
    out[tag] = [{'kernel': k[0], 'dispatch': k[1], **v} for k, v in d.items()]
out['derived'] = []
for a, b in zip([x for x in out['sq'] if 'nerf_' in x['kernel']], [x for x in out['grbm'] if 'nerf_' in x['kernel']]):
    out['derived'].append({'kernel': a['kernel'], 'mfma_pipe_busy_frac': a['SQ_VALU_MFMA_BUSY_CYCLES'] / (256 * 4 * b['GRBM_GUI_ACTIVE'] / 8),
                           'cycles_per_xcd': b['GRBM_GUI_ACTIVE'] / 8})
out['note'] = ("bench.py --steps 1 --warmup 0 --timed-only (default precision: mixed) under rocprofv3 --pmc <one group per pass> "
               "--kernel-trace; FETCH_SIZE/WRITE_SIZE in KiB (FETCH_SIZE under-reports wide streaming reads by 2x on gfx950, "
               "MI355X_MICROARCH.md); one coarse launch (nerf_sigma_f16t_kernel, 81.92 M evaluations) and one fine launch "
               "(nerf_mlp_i8s_kernel, 163.84 M evaluations)")
json.dump(out, open('$OUT/bench_pmc_summary.json', 'w'), indent=1)
print(json.dumps(out['derived']))
PY
cd $R
cp $OUT/bench_pmc_summary.json profiles/${TAG}_bench_pmc_summary.json 2>/dev/null     # bench.py reads `traffic` from here
python bench.py --steps 5 --warmup 1 > $OUT/bench_line.json 2> $OUT/bench_stderr.log
tail -c 1500 $OUT/bench_line.json
cp $OUT/bench_kernel_stats.csv profiles/${TAG}_bench_kernel_stats.csv 2>/dev/null
cp $OUT/bench_line.json profiles/${TAG}_bench_line.json 2>/dev/null
if [ "${FULL:-0}" = "1" ]; then
  # the in-kernel cycle buckets and the clock / power samples quoted in DESIGN.md section 6
  (python tools/mlp_profile.py; python tools/mlp_profile_i8.py) > $OUT/mlp_profile.log 2>&1
  for p in fp16x3 i8x3 bf16; do tools/clock_watch.sh $p; done > $OUT/clock_watch.log 2>&1
  python tools/mlp_power_probe.py > $OUT/mlp_power_probe.log 2>&1
fi
