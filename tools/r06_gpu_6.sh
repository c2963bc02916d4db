#!/bin/bash
# round 6: the whole GPU suite + smoke on the current tree
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -s -p no:cacheprovider > gpurun_out/r06_gputest_6.log 2>&1; echo "pytest rc $?" >> gpurun_out/r06_gputest_6.log
tail -8 gpurun_out/r06_gputest_6.log
python __graft_entry__.py smoke > gpurun_out/r06_smoke_6.log 2>&1; echo "smoke rc $?"
