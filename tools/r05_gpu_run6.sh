#!/bin/bash
# round 5, GPU call 6: the fused loss terms -- their unit tests first, alone and under a short timeout; the rest only if they pass
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run6
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 150 python -m pytest tests/test_hip_loss_ops.py -q -m gpu > $OUT/test_loss_ops.log 2>&1
rc=$?; echo "loss ops rc $rc" > $OUT/progress.log
tail -n 25 $OUT/test_loss_ops.log
if [ $rc -ne 0 ]; then echo "STOP: loss ops failed"; exit 1; fi
timeout 300 python -m pytest tests/test_hip_human_trainer.py tests/test_hip_human_loss_golden.py -q -m gpu -s > $OUT/test_human.log 2>&1
rc=$?; echo "human tests rc $rc" >> $OUT/progress.log
tail -n 25 $OUT/test_human.log | cut -c1-400
if [ $rc -ne 0 ]; then echo "STOP: human tests failed"; exit 1; fi
for fused in 1 0; do
  NEUMAN_FUSED_LOSS=$fused timeout 200 python tools/human_step_bench.py 2048 50 >> $OUT/human_step_fused_$fused.jsonl 2>> $OUT/human_step.err || { echo "STOP: human bench failed"; tail -5 $OUT/human_step.err; exit 1; }
done
cut -c1-330 $OUT/human_step_fused_1.jsonl $OUT/human_step_fused_0.jsonl
rm -rf /tmp/prof_r05 && mkdir -p /tmp/prof_r05
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
python - <<'P'
import csv
rows = list(csv.DictReader(open('gpurun_out/r05_run6/human_kernel_stats.csv')))
print("launches per iteration", sum(int(r['Calls']) for r in rows) / 23.0)
P
NEUMAN_LAUNCH_SOURCES=1 timeout 200 python tools/human_step_bench.py 2048 5 > $OUT/launch_sources.jsonl 2> $OUT/launch_sources.txt
grep -A75 "launch sources" $OUT/launch_sources.txt | cut -c1-260
