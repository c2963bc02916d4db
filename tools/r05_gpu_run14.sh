#!/bin/bash
# round 5, GPU call 14: the human iteration's kernel profile after the host stalls went (two views on), and the background trainer's lines on the same box
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/r05_run14
mkdir -p $OUT
export TMPDIR=/tmp
R=$(pwd)
python ml-neuman_amd/build.py > $OUT/build.log 2>&1
timeout 200 python tools/human_step_bench.py 2048 50 > $OUT/human_step.jsonl 2> $OUT/human_step.err || { echo "STOP: human bench failed"; tail -5 $OUT/human_step.err; exit 1; }
cut -c1-330 $OUT/human_step.jsonl
rm -rf /tmp/prof_r05 && mkdir -p /tmp/prof_r05
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_r05/human -o human -- python $R/tools/human_step_bench.py 2048 20 > $R/$OUT/prof_human.log 2>&1 )
cp /tmp/prof_r05/human/human_kernel_stats.csv $OUT/human_kernel_stats.csv 2>/dev/null
python - <<'P'
import csv
rows = list(csv.DictReader(open('gpurun_out/r05_run14/human_kernel_stats.csv')))
print("launches per iteration", sum(int(r['Calls']) for r in rows) / 23.0, "kernel ms per iteration", sum(int(r['TotalDurationNs']) for r in rows) / 23e6)
for r in sorted(rows, key=lambda r: -int(r['TotalDurationNs']))[:12]:
    print(f"{int(r['TotalDurationNs']) / 23e6:7.3f} ms  x{int(r['Calls']) / 23:4.1f}  {r['Name'][:80]}")
P
for s16 in 1; do
  timeout 100 python tools/train_step_bench.py 2048 > $OUT/train_step.jsonl 2> $OUT/train_step.err
done
cut -c1-200 $OUT/train_step.jsonl
NEUMAN_LAUNCH_SOURCES=1 timeout 200 python tools/human_step_bench.py 2048 5 > $OUT/launch_sources.jsonl 2> $OUT/launch_sources.txt
grep -A40 "launch sources" $OUT/launch_sources.txt | cut -c1-220
