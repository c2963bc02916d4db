for i in 1 2; do python tools/train_step_bench.py 2>/dev/null | grep "^{" | cut -c60-140; done
NEUMAN_TRAIN_FUSED_BWD=0 python tools/train_step_bench.py 2>/dev/null | grep "^{" | cut -c60-140
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $GRAFT_REPO_ROOT/tools/train_step_bench.py > /dev/null 2>&1
python - <<PY
import csv
rows=list(csv.DictReader(open('/tmp/tr/tr_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms',tot/1e6, 'launches', sum(int(r['Calls']) for r in rows))
for r in rows[:12]:
    print(f"{float(r['Percentage']):6.2f}% {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Calls']:>5} avg {float(r['AverageNs'])/1e3:8.1f} us  {r['Name'][:100]}")
PY
