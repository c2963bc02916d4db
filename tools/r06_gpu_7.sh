#!/bin/bash
# round 6: the compact multi-actor merge: its test and an A/B of the C5 frame
mkdir -p gpurun_out
python -m pytest tests/test_hip_configs.py tests/test_hip_posed_golden.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2; do
  for c in 1 0; do
    echo "== NEUMAN_MULTI_COMPACT=$c" >> gpurun_out/r06_c5_compact_ab.log
    NEUMAN_MULTI_COMPACT=$c python tools/bench_configs.py --only C5 2>/dev/null | grep config | cut -c1-200 >> gpurun_out/r06_c5_compact_ab.log
  done
done
cat gpurun_out/r06_c5_compact_ab.log
