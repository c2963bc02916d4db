#!/bin/bash
# round 6, first GPU call: the whole GPU suite, smoke, one default bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > gpurun_out/r06_gputest_1.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_1.log
tail -5 gpurun_out/r06_gputest_1.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_1.log 2>&1; echo "smoke rc $?"
python bench.py > gpurun_out/r06_bench_1.json 2> gpurun_out/r06_bench_1.err; echo "bench rc $?"
tail -c 1500 gpurun_out/r06_bench_1.json
