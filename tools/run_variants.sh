export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
python tools/train_step_bench.py 2>/dev/null | grep "^{" > $R/gpurun_out/r04/train_step.jsonl
python tools/train_step_bench.py 4096 2>/dev/null | grep "^{" >> $R/gpurun_out/r04/train_step.jsonl
NEUMAN_TRAIN_FUSED_BWD=0 python tools/train_step_bench.py 2>/dev/null | grep "^{" >> $R/gpurun_out/r04/train_step.jsonl
python tools/human_step_bench.py 2>/dev/null | grep "^{" > $R/gpurun_out/r04/human_step.jsonl
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $R/tools/train_step_bench.py > /dev/null 2>&1
cp /tmp/tr/tr_kernel_stats.csv $R/gpurun_out/r04/train_kernel_stats.csv
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hs -o hs -- python $R/tools/human_step_bench.py > /dev/null 2>&1
cp /tmp/hs/hs_kernel_stats.csv $R/gpurun_out/r04/human_step_kernel_stats.csv
cut -c1-220 $R/gpurun_out/r04/train_step.jsonl $R/gpurun_out/r04/human_step.jsonl
python - <<PY
import csv
for f in ('/tmp/tr/tr_kernel_stats.csv','/tmp/hs/hs_kernel_stats.csv'):
    rows=list(csv.DictReader(open(f)))
    print(f, 'total ms', sum(float(r['TotalDurationNs']) for r in rows)/1e6, 'launches', sum(int(r['Calls']) for r in rows))
PY
