timeout 600 python -m pytest tests/test_hip_ray_ops.py tests/test_hip_warp.py tests/test_hip_posed_golden.py -x -q -m gpu 2>&1 | tail -3
python tools/human_step_bench.py 2>/dev/null | grep "^{" | cut -c1-160
python tools/bench_configs.py 2>/dev/null | grep '^{' | cut -c1-200
