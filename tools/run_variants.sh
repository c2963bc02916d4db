export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o tr -- python $R/tools/train_step_bench.py 2>/dev/null | grep '^{' > $R/gpurun_out/r04/train_step.jsonl
cp /tmp/tr/tr_kernel_stats.csv $R/gpurun_out/r04/train_kernel_stats.csv
cat $R/gpurun_out/r04/train_step.jsonl | cut -c1-600
