#!/bin/bash
# round 6, fourth GPU call: near / far + posed tests on the reference's recorded rays; the training tests on the two-MFMA backward + deferred forward saves; A/B of a training step
mkdir -p gpurun_out
python -m pytest tests/test_hip_ray_ops.py tests/test_hip_render.py tests/test_hip_posed_golden.py tests/test_hip_configs.py tests/test_hip_install_callers.py \
  tests/test_hip_train.py tests/test_hip_train16.py tests/test_hip_human_loss_golden.py tests/test_hip_bkg_trainer.py tests/test_hip_human_trainer.py tests/test_hip_dp_train.py tests/test_hip_smpl_diff.py \
  -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gputest_4.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_4.log
tail -15 gpurun_out/r06_gputest_4.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_4.log 2>&1; echo "smoke rc $?"
for rep in 1 2; do
  for cfg in "NEUMAN_BWD_HALF=1 NEUMAN_FWD_DEFER=1" "NEUMAN_BWD_HALF=0 NEUMAN_FWD_DEFER=0" "NEUMAN_BWD_HALF=1 NEUMAN_FWD_DEFER=0" "NEUMAN_BWD_HALF=0 NEUMAN_FWD_DEFER=1"; do
    echo "== $cfg" >> gpurun_out/r06_train_ab.log
    env $cfg python tools/train_step_bench.py 2>/dev/null | grep ms_per_iteration | head -1 | cut -c1-220 >> gpurun_out/r06_train_ab.log
  done
done
cat gpurun_out/r06_train_ab.log
