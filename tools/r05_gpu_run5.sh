#!/bin/bash
# round 5, GPU call 5: the offset nets' fused path first, alone and under a short timeout; everything else only if it passes
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run5
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 150 python -m pytest tests/test_hip_train16.py -q -m gpu -s -k "offset_net_on_the_fused" > $OUT/test_offset.log 2>&1
rc=$?; echo "offset rc $rc" > $OUT/progress.log
tail -n 6 $OUT/test_offset.log
if [ $rc -ne 0 ]; then echo "STOP: offset test failed"; exit 1; fi
timeout 300 python -m pytest tests/test_hip_train16.py -q -m gpu -s > $OUT/test_train16.log 2>&1; echo "train16 rc $?" >> $OUT/progress.log
tail -n 4 $OUT/test_train16.log
for s16 in 1 0; do
  NEUMAN_TRAIN_STORE16=$s16 timeout 200 python tools/human_step_bench.py 2048 50 >> $OUT/human_step_store16_$s16.jsonl 2>> $OUT/human_step.err || { echo "STOP: human bench failed"; tail -5 $OUT/human_step.err; exit 1; }
done
cat $OUT/human_step_store16_1.jsonl | cut -c1-330
NEUMAN_TRAIN_STORE16=1 timeout 100 python tools/train_step_bench.py 2048 >> $OUT/train_step_store16_1.jsonl 2>> $OUT/train_step.err
cat $OUT/train_step_store16_1.jsonl | cut -c1-200
rm -rf /tmp/prof_r05 && mkdir -p /tmp/prof_r05
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
timeout 400 python -m pytest tests/test_hip_human_trainer.py tests/test_hip_human_loss_golden.py tests/test_hip_train.py tests/test_hip_heads.py -q -m gpu -x > $OUT/test_human.log 2>&1; echo "human tests rc $?" >> $OUT/progress.log
tail -n 5 $OUT/test_human.log
