#!/bin/bash
# AddressSanitizer over the HOST side of libneuman_hip.so ("ASAN host build of the shim").
#   tools/asan_host.sh [pytest args ...]      default: the CPU-runnable packer / ABI tests; on a GPU box pass e.g. `-m gpu tests/test_hip_mlp.py`
# Every .hip file is compiled with -fsanitize=address -fno-gpu-sanitize (host code instrumented, device code untouched: device ASAN needs xnack),
# linked against the shared ASAN runtime of ROCm's clang, and loaded through NEUMAN_HIP_LIB with that runtime preloaded.  What it covers: the weight
# packers (2.4 MB images assembled with computed offsets), argument marshalling, handle lifetime, every host-side table the launchers build.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
B=${ASAN_BUILD_DIR:-/tmp/neuman_asan_build}
mkdir -p $B $R/ml-neuman_amd/lib/exp
cd $R/ml-neuman_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -fsanitize=address -fno-gpu-sanitize -fno-omit-frame-pointer \
      -c $f -o $B/${f%.hip}.o 2> $B/${f%.hip}.err &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address -fno-gpu-sanitize -shared-libasan -o $R/ml-neuman_amd/lib/exp/libneuman_hip_asan.so $B/*.o
ASAN=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
cd $R
ARGS=${@:-tests/test_mlp_pack.py tests/test_abi.py}
LD_PRELOAD=$ASAN ASAN_OPTIONS=${ASAN_OPTIONS:-detect_leaks=0:protect_shadow_gap=0:abort_on_error=0:halt_on_error=1:allocator_may_return_null=1} NEUMAN_HIP_LIB=$R/ml-neuman_amd/lib/exp/libneuman_hip_asan.so \
    python -m pytest -q -x $ARGS
