#!/bin/bash
# round 6, fifth GPU call: the whole GPU suite, smoke, kernel-level A/B of the deferred forward saves
mkdir -p gpurun_out
R=$(pwd)
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gputest_5.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_5.log
tail -8 gpurun_out/r06_gputest_5.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_5.log 2>&1; echo "smoke rc $?"
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_t && mkdir -p /tmp/prof_t
for d in 0 1; do
  NEUMAN_FWD_DEFER=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t/d$d -o train -- python $R/tools/train_step_bench.py > $R/gpurun_out/r06_train_defer$d.log 2>&1
  cp /tmp/prof_t/d$d/train_kernel_stats.csv $R/gpurun_out/r06_train_kernel_stats_defer$d.csv
done
for d in 0 1; do
  NEUMAN_FWD_DEFER=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t/e$d -o train -- python $R/tools/train_step_bench.py > /dev/null 2>&1
  cp /tmp/prof_t/e$d/train_kernel_stats.csv $R/gpurun_out/r06_train_kernel_stats_defer${d}_b.csv
done
cd $R
NEUMAN_BWD_HALF=1 python tools/train_step_bench.py 2>/dev/null | grep ms_per | cut -c1-260
python tools/train_step_bench.py 2>/dev/null | grep ms_per | cut -c1-260
head -3 gpurun_out/r06_train_kernel_stats_defer0.csv | cut -c1-200; head -3 gpurun_out/r06_train_kernel_stats_defer1.csv | cut -c1-200
