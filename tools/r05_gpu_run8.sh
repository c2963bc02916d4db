#!/bin/bash
# round 5, GPU call 8: run-wise pre-reduction in the scatter kernels of the differentiable warp -- their tests first, under a short timeout
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run8
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 200 python -m pytest tests/test_hip_smpl_diff.py -q -m gpu -s -x > $OUT/test_smpl_diff.log 2>&1
rc=$?; echo "smpl_diff rc $rc" > $OUT/progress.log
grep "\[warp-apply\]\|\[bary\]" $OUT/test_smpl_diff.log | cut -c1-250; tail -n 5 $OUT/test_smpl_diff.log | cut -c1-250
if [ $rc -ne 0 ]; then echo "STOP: smpl_diff failed"; exit 1; fi
timeout 300 python -m pytest tests/test_hip_human_trainer.py tests/test_hip_human_loss_golden.py -q -m gpu > $OUT/test_human.log 2>&1
rc=$?; echo "human tests rc $rc" >> $OUT/progress.log
tail -n 5 $OUT/test_human.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "STOP: human tests failed"; exit 1; fi
timeout 200 python tools/human_step_bench.py 2048 50 > $OUT/human_step.jsonl 2> $OUT/human_step.err || { echo "STOP: human bench failed"; tail -5 $OUT/human_step.err; exit 1; }
cut -c1-330 $OUT/human_step.jsonl
rm -rf /tmp/prof_r05 && mkdir -p /tmp/prof_r05
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
python - <<'P'
import csv
rows = list(csv.DictReader(open('gpurun_out/r05_run8/human_kernel_stats.csv')))
print("launches per iteration", sum(int(r['Calls']) for r in rows) / 23.0, "kernel ms per iteration", sum(int(r['TotalDurationNs']) for r in rows) / 23e6)
for r in rows:
    if any(k in r['Name'] for k in ('warp_apply_backward', 'bary_backward', 'search_kernel', 'warp_apply_forward')):
        print(f"{int(r['TotalDurationNs']) / 23e6:7.3f} ms  x{int(r['Calls']) / 23:4.1f}  {r['Name'][:80]}")
P
