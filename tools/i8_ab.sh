#!/bin/bash
# A/B of the two i8x3 schedules on one box: parity tests of the i8x3 path with the resident-weight kernel, then the timed frame with either.
#   gpurun -- 'bash tools/i8_ab.sh r02x'
TAG=${1:-i8ab}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p "$OUT"
cd "$R"
export PYTHONUNBUFFERED=1
NEUMAN_I8_KERNEL=r timeout 300 python -m pytest tests/test_hip_mlp.py -m gpu -q -s -k "i8x3 or sigma_only or composited" > "$OUT/tests_r.log" 2>&1
echo "tests rc $?" >> "$OUT/tests_r.log"
tail -5 "$OUT/tests_r.log"
for k in w r w r; do
  NEUMAN_I8_KERNEL=$k timeout 300 python bench.py --steps 3 --warmup 1 --timed-only > "$OUT/bench_$k.json" 2> "$OUT/bench_$k.err"
  echo "kernel $k rc $?"
  python - "$OUT/bench_$k.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print("  rays/s %.4g  ms %.1f  fine %s %.1f ms frac %.3f  coarse %.1f ms" % (d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline_coarse']['avg_launch_ms']))
except Exception as e:
    print("  no result", e)
PY
done
