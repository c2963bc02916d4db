# bit-identity of the activation-stationary i8 kernel against the wave-specialised one, then the fine-launch time (and of lib/exp variants)
timeout 120 python tools/i8s_check.py > gpurun_out/chk_w.log 2>&1
NEUMAN_I8_KERNEL=as timeout 120 python tools/i8s_check.py 2>&1 | grep -c "bit-identical"
NEUMAN_I8_KERNEL=as timeout 120 python tools/i8_time.py 2>&1 | tail -1
for v in "$@"; do
  NEUMAN_HIP_LIB=$PWD/ml-neuman_amd/lib/exp/libneuman_hip_$v.so NEUMAN_I8_KERNEL=as timeout 120 python tools/i8s_check.py 2>&1 | grep -c "bit-identical"
  NEUMAN_HIP_LIB=$PWD/ml-neuman_amd/lib/exp/libneuman_hip_$v.so NEUMAN_I8_KERNEL=as timeout 120 python tools/i8_time.py 2>&1 | grep -v amdgpu.ids | tail -3
done
