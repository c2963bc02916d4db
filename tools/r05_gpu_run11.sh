#!/bin/bash
# round 5, GPU call 11: the second view through the views head alone -- its tests first, under a short timeout; then the trainer's tests and its iteration
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run11
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 200 python -m pytest tests/test_hip_train16.py -q -m gpu -s -x -k "two_views or net16" > $OUT/test_two.log 2>&1
rc=$?; echo "two views rc $rc" > $OUT/progress.log
grep "\[train16\]" $OUT/test_two.log | cut -c1-250; tail -n 12 $OUT/test_two.log | cut -c1-250
if [ $rc -ne 0 ]; then echo "STOP: two views failed"; exit 1; fi
timeout 300 python -m pytest tests/test_hip_train16.py tests/test_hip_human_trainer.py tests/test_hip_human_loss_golden.py -q -m gpu > $OUT/test_human.log 2>&1
rc=$?; echo "human tests rc $rc" >> $OUT/progress.log
tail -n 5 $OUT/test_human.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "STOP: tests failed"; exit 1; fi
for tv in 1 0; do
  NEUMAN_TWO_VIEWS=$tv timeout 200 python tools/human_step_bench.py 2048 50 > $OUT/human_step_two_views_$tv.jsonl 2>> $OUT/human_step.err || { echo "STOP: human bench failed"; tail -5 $OUT/human_step.err; exit 1; }
done
cut -c1-330 $OUT/human_step_two_views_1.jsonl $OUT/human_step_two_views_0.jsonl
rm -rf /tmp/prof_r05 && mkdir -p /tmp/prof_r05
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
python - <<'P'
import csv
rows = list(csv.DictReader(open('gpurun_out/r05_run11/human_kernel_stats.csv')))
print("launches per iteration", sum(int(r['Calls']) for r in rows) / 23.0, "kernel ms per iteration", sum(int(r['TotalDurationNs']) for r in rows) / 23e6)
for r in sorted(rows, key=lambda r: -int(r['TotalDurationNs']))[:10]:
    print(f"{int(r['TotalDurationNs']) / 23e6:7.3f} ms  x{int(r['Calls']) / 23:4.1f}  {r['Name'][:80]}")
P
