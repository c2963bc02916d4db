export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg -o cfg -- python $R/tools/bench_configs.py 2>/dev/null | grep '^{' > $R/gpurun_out/r04/bench_configs.jsonl
cp /tmp/cfg/cfg_kernel_stats.csv $R/gpurun_out/r04/configs_kernel_stats.csv
cd $R
timeout 300 python tools/human_step_bench.py 2>/dev/null | grep '^{' > gpurun_out/r04/human_step.jsonl
cat gpurun_out/r04/bench_configs.jsonl | cut -c1-250; cat gpurun_out/r04/human_step.jsonl | cut -c1-400
