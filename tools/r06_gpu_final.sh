#!/bin/bash
# round 6, closing run on the final tree: GPU suite, smoke, the round's profile (bench line + rocprofv3 stats + PMC passes), configs, trainer iterations
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gputest_final.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_final.log
tail -4 gpurun_out/r06_gputest_final.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_final.log 2>&1; echo "smoke rc $?"
bash tools/profile_round.sh r06 > gpurun_out/r06_profile_round.log 2>&1; echo "profile rc $?"
python tools/bench_configs.py 2>/dev/null | grep '^{' > gpurun_out/r06_bench_configs.jsonl; cat gpurun_out/r06_bench_configs.jsonl | cut -c1-200
python tools/train_step_bench.py 2>/dev/null | grep '^{' > gpurun_out/r06_train_step.jsonl; python tools/train_step_bench.py 4096 2>/dev/null | grep '^{' >> gpurun_out/r06_train_step.jsonl; cut -c1-200 gpurun_out/r06_train_step.jsonl
python tools/human_step_bench.py 2>/dev/null | grep '^{' > gpurun_out/r06_human_step.jsonl; cut -c1-300 gpurun_out/r06_human_step.jsonl
tail -c 600 gpurun_out/r06/bench_line.json
