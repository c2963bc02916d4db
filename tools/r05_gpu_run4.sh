#!/bin/bash
# round 5, GPU call 4: full GPU suite + training benches + profiles
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run4
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 600 python -m pytest tests/test_hip_train16.py -q -m gpu -s > $OUT/test_train16.log 2>&1; echo "train16 rc $?" > $OUT/progress.log
for rep in 1 2; do
  for s16 in 1 0; do
    NEUMAN_TRAIN_STORE16=$s16 timeout 300 python tools/train_step_bench.py 2048 >> $OUT/train_step_store16_$s16.jsonl 2>> $OUT/train_step.err
  done
done
NEUMAN_TRAIN_STORE16=1 timeout 300 python tools/train_step_bench.py 4096 >> $OUT/train_step_store16_1.jsonl 2>> $OUT/train_step.err
for s16 in 1 0; do
  NEUMAN_TRAIN_STORE16=$s16 timeout 600 python tools/human_step_bench.py 2048 50 >> $OUT/human_step_store16_$s16.jsonl 2>> $OUT/human_step.err
done
echo "benches done" >> $OUT/progress.log
rm -rf /tmp/prof_r05 && mkdir -p /tmp/prof_r05
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/train -o train -- python $R/tools/train_step_bench.py 2048 > $R/$OUT/prof_train.log 2>&1 )
cp /tmp/prof_r05/train/train_kernel_stats.csv $OUT/train_kernel_stats.csv 2>/dev/null
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
echo "profiles done" >> $OUT/progress.log
timeout 1500 python -m pytest tests -q -m gpu > $OUT/test_all.log 2>&1; echo "all rc $?" >> $OUT/progress.log
tail -n 4 $OUT/test_train16.log
tail -n 15 $OUT/test_all.log
cat $OUT/train_step_store16_1.jsonl $OUT/train_step_store16_0.jsonl | cut -c1-200
cat $OUT/human_step_store16_1.jsonl | cut -c1-330
