timeout 600 python -m pytest tests/test_hip_smpl_diff.py tests/test_hip_human_loss_golden.py tests/test_hip_train.py -x -q -m gpu -k "not test_gemm" -s 2>&1 | grep "bary\|warp-apply\|passed\|failed\|Error" | tail -12
python tools/human_step_bench.py 2>/dev/null | grep "^{" | cut -c1-200
NEUMAN_BARY_KERNELS=0 python tools/human_step_bench.py 2>/dev/null | grep "^{" | cut -c1-200
