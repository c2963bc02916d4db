export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r04
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cfg -o cfg -- python $R/tools/bench_configs.py 2>/dev/null | grep '^{' > $R/gpurun_out/r04/bench_configs.jsonl
cp /tmp/cfg/cfg_kernel_stats.csv $R/gpurun_out/r04/configs_kernel_stats.csv
cut -c1-200 $R/gpurun_out/r04/bench_configs.jsonl
