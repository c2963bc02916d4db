#!/bin/bash
# round 5, GPU call 13: the search handle updated in place (nm_mesh_update), cached uploads -- the new test first; then warp / trainer tests and the iteration
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run13
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 200 python -m pytest tests/test_hip_smpl_diff.py -q -m gpu -x > $OUT/test_update.log 2>&1
rc=$?; echo "update rc $rc" > $OUT/progress.log
tail -n 12 $OUT/test_update.log | cut -c1-250
if [ $rc -ne 0 ]; then echo "STOP: mesh update failed"; exit 1; fi
timeout 500 python -m pytest tests/test_hip_ray_ops.py tests/test_hip_posed_golden.py tests/test_hip_human_trainer.py tests/test_hip_human_loss_golden.py tests/test_hip_train16.py tests/test_hip_frame.py -q -m gpu -x > $OUT/test_more.log 2>&1
rc=$?; echo "more tests rc $rc" >> $OUT/progress.log
tail -n 5 $OUT/test_more.log | cut -c1-300
if [ $rc -ne 0 ]; then echo "STOP: tests failed"; exit 1; fi
for tv in 1 0; do
  NEUMAN_TWO_VIEWS=$tv timeout 200 python tools/human_step_bench.py 2048 50 > $OUT/human_step_two_views_$tv.jsonl 2>> $OUT/human_step.err || { echo "STOP: human bench failed"; tail -5 $OUT/human_step.err; exit 1; }
done
cut -c1-330 $OUT/human_step_two_views_1.jsonl $OUT/human_step_two_views_0.jsonl
NEUMAN_TWO_VIEWS=1 NEUMAN_HOST_PROFILE=1 timeout 300 python tools/human_step_bench.py 2048 10 > $OUT/hp.jsonl 2> $OUT/hp.txt
grep -A16 "host profile" $OUT/hp.txt | cut -c1-170
