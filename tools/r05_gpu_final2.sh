#!/bin/bash
# round 5, the closing run on the committed tree: full GPU suite, smoke, the bench line, the configs.  Every step under its own timeout.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_final2
mkdir -p $OUT
export TMPDIR=/tmp PYTHONUNBUFFERED=1
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 900 python -m pytest tests -q -m gpu > $OUT/test_all.log 2>&1; echo "gpu suite rc $?" > $OUT/progress.log
tail -n 3 $OUT/test_all.log
timeout 120 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc $?" >> $OUT/progress.log
tail -n 2 $OUT/smoke.log | cut -c1-300
timeout 400 python bench.py > $OUT/bench_line.json 2> $OUT/bench.err; echo "bench rc $?" >> $OUT/progress.log
python - <<'P'
import json
d = json.loads(open('gpurun_out/r05_final2/bench_line.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step', 'n_gpus')}, d['roofline']['frac'], d['roofline']['avg_launch_ms'], {k: v for k, v in d.get('train_iteration', {}).items() if 'ms' in k})
P
timeout 300 python tools/bench_configs.py > $OUT/bench_configs.jsonl 2> $OUT/bench_configs.err; echo "configs rc $?" >> $OUT/progress.log
cut -c1-200 $OUT/bench_configs.jsonl
cat $OUT/progress.log
