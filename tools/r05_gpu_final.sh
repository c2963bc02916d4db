#!/bin/bash
# round 5, the evidence run: full GPU suite, bench + PMC profile (tools/profile_round.sh r05), training benches and their profiles, smoke's parity lines,
# the configs, the data-parallel trainer.  Every step under its own timeout; results under gpurun_out/r05_final/ (and gpurun_out/r05/ for profile_round.sh)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_final
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 700 python -m pytest tests -q -m gpu > $OUT/test_all.log 2>&1; echo "gpu suite rc $?" > $OUT/progress.log
tail -n 3 $OUT/test_all.log
timeout 700 bash tools/profile_round.sh r05 > $OUT/profile_round.log 2>&1; echo "profile_round rc $?" >> $OUT/progress.log
for rep in 1 2 3; do
  for s16 in 1 0; do
    NEUMAN_TRAIN_STORE16=$s16 timeout 100 python tools/train_step_bench.py 2048 >> $OUT/train_step_store16_$s16.jsonl 2>> $OUT/train_step.err
  done
done
NEUMAN_TRAIN_STORE16=1 timeout 100 python tools/train_step_bench.py 4096 >> $OUT/train_step_store16_1.jsonl 2>> $OUT/train_step.err
for s16 in 1 0; do
  NEUMAN_TRAIN_STORE16=$s16 timeout 200 python tools/human_step_bench.py 2048 50 >> $OUT/human_step_store16_$s16.jsonl 2>> $OUT/human_step.err
done
echo "training benches done" >> $OUT/progress.log
rm -rf /tmp/prof_r05t && mkdir -p /tmp/prof_r05t
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05t/train -o train -- python $R/tools/train_step_bench.py 2048 > $R/$OUT/prof_train.log 2>&1 )
cp /tmp/prof_r05t/train/train_kernel_stats.csv $OUT/train_kernel_stats.csv 2>/dev/null
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05t/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05t/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_r05t/pmc_$c -o train -- python $R/tools/train_step_bench.py 2048 > /dev/null 2>&1 )
done
python - <<PY > $OUT/train_pmc_summary.json 2>> $OUT/train_step.err
import csv, collections, json, re
out = {}
for n in ('FETCH_SIZE', 'WRITE_SIZE'):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open('/tmp/prof_r05t/pmc_%s/train_counter_collection.csv' % n)):
        m = re.search(r'::(\w+)(<[^>]*>)?\(', r['Kernel_Name'])
        k = (m.group(1) + (m.group(2) or '')) if m else r['Kernel_Name'][:40]
        if any(s in k for s in ('nerf_mlp', 'wgrad', 'heads')):
            d[k].append(float(r['Counter_Value']))
    out[n] = {k: {'launches': len(v), 'mean_KiB': sum(v) / len(v), 'last4_KiB': v[-4:]} for k, v in d.items()}
out['note'] = "tools/train_step_bench.py 2048 under rocprofv3 --pmc (one counter per pass) --kernel-trace; per launch, KiB; FETCH_SIZE under-reports wide streaming reads by 2x on gfx950 (MI355X_MICROARCH.md); launches alternate coarse net (262144 evaluations) / fine net (524288)"
print(json.dumps(out, indent=1))
PY
echo "training profiles done" >> $OUT/progress.log
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/progress.log
grep "smoke parity" $OUT/smoke.log | cut -c1-400
timeout 300 python tools/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err; echo "configs rc $?" >> $OUT/progress.log
timeout 100 python tools/train_dp_bench.py --rays 2048 > $OUT/train_dp.jsonl 2> $OUT/train_dp.err
timeout 100 python tools/train_dp_bench.py --dist --rays 2048 >> $OUT/train_dp.jsonl 2>> $OUT/train_dp.err
timeout 150 python tools/train_dp_bench.py --gpus 2 --share-gpu --rays 2048 >> $OUT/train_dp.jsonl 2>> $OUT/train_dp.err
echo "all done" >> $OUT/progress.log
cat $OUT/progress.log
cat $OUT/train_step_store16_1.jsonl | cut -c1-160
cat $OUT/human_step_store16_1.jsonl | cut -c1-330
tail -c 600 gpurun_out/r05/bench_line.json
