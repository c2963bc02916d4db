timeout 200 python tools/f16t_debug.py 2>&1 | grep -v '"equal": true\|bit_identical": true' | tail -8
NEUMAN_SIGMA_KERNEL=t python tools/coarse_time.py 2>&1 | grep coarse
for nd in 0 2; do
NEUMAN_F16T_NDIR=$nd NEUMAN_HIP_LIB=$PWD/ml-neuman_amd/lib/exp/libneuman_hip_f16ndir$nd.so timeout 200 python tools/f16t_debug.py 2>&1 | tail -1
NEUMAN_F16T_NDIR=$nd NEUMAN_HIP_LIB=$PWD/ml-neuman_amd/lib/exp/libneuman_hip_f16ndir$nd.so NEUMAN_SIGMA_KERNEL=t python tools/coarse_time.py 2>&1 | grep coarse
done
python tools/coarse_time.py 2>&1 | grep coarse
