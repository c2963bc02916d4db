#!/bin/bash
# round 6, third GPU call: the tests the float64 near / far + runner-up foot touch, smoke, and kernel-level profiles of C4 / C5 and of a training step
mkdir -p gpurun_out
R=$(pwd)
python -m pytest tests/test_hip_ray_ops.py tests/test_hip_render.py tests/test_hip_posed_golden.py tests/test_hip_configs.py tests/test_hip_install_callers.py tests/test_hip_march.py tests/test_hip_fused.py \
  -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gputest_3.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_3.log
tail -12 gpurun_out/r06_gputest_3.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_3.log 2>&1; echo "smoke rc $?"
export TMPDIR=/tmp; cd /tmp; rm -rf /tmp/prof_c && mkdir -p /tmp/prof_c
for c in C4 C5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c/$c -o cfg -- python $R/tools/bench_configs.py --only $c > $R/gpurun_out/r06_cfg_$c.log 2>&1
  cp /tmp/prof_c/$c/cfg_kernel_stats.csv $R/gpurun_out/r06_cfg_${c}_kernel_stats.csv
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c/train -o train -- python $R/tools/train_step_bench.py > $R/gpurun_out/r06_train_step_0.log 2>&1
cp /tmp/prof_c/train/train_kernel_stats.csv $R/gpurun_out/r06_train_kernel_stats_0.csv
cd $R
tail -3 gpurun_out/r06_cfg_C5.log | cut -c1-400
tail -3 gpurun_out/r06_train_step_0.log | cut -c1-400
