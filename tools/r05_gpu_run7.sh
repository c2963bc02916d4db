#!/bin/bash
# round 5, GPU call 7: the plain head on the i8 activation-stationary kernel -- its tests first, alone and under a short timeout
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run7
mkdir -p $OUT
export TMPDIR=/tmp
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 200 python -m pytest tests/test_hip_heads.py -q -m gpu -s -x > $OUT/test_heads.log 2>&1
rc=$?; echo "heads rc $rc" > $OUT/progress.log
grep "\[heads\]" $OUT/test_heads.log | cut -c1-250; tail -n 15 $OUT/test_heads.log | cut -c1-250
if [ $rc -ne 0 ]; then echo "STOP: heads failed"; exit 1; fi
timeout 300 python -m pytest tests/test_hip_i8_as.py tests/test_hip_mlp.py -q -m gpu -x > $OUT/test_i8.log 2>&1; echo "i8 rc $?" >> $OUT/progress.log
tail -n 4 $OUT/test_i8.log
timeout 120 python tools/plain_head_time.py > $OUT/plain_head_time.jsonl 2> $OUT/plain_head_time.err; tail -n 3 $OUT/plain_head_time.err; cat $OUT/plain_head_time.jsonl
