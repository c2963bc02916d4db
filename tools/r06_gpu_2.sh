#!/bin/bash
# round 6, second GPU call: the whole GPU suite without -x, smoke
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gputest_2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_2.log
tail -15 gpurun_out/r06_gputest_2.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_2.log 2>&1; echo "smoke rc $?"
