for v in "$@"; do NEUMAN_HIP_LIB=$PWD/ml-neuman_amd/lib/exp/libneuman_hip_$v.so python tools/bwd_time.py 2>&1 | grep chain; done
python tools/bwd_time.py 2>&1 | grep chain
