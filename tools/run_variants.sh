for v in "$@"; do NEUMAN_HIP_LIB=$PWD/ml-neuman_amd/lib/exp/libneuman_hip_$v.so NEUMAN_I8_KERNEL=t python tools/i8_time.py 2>&1 | grep kernel; done
NEUMAN_I8_KERNEL=t python tools/i8_time.py 2>&1 | grep kernel
python tools/i8_time.py 2>&1 | grep kernel
