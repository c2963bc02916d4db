// Shared helpers for libneuman_hip.so (gfx950 only; no CUDA / multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/neuman_hip.h"

namespace nm {

void set_error(const char* fmt, ...);
// hipGetLastError() after a launch; sets the error string on failure.
int check_launch(const char* what);
int check_hip(hipError_t e, const char* what);

constexpr int kWave = 64;  // CDNA wavefront width

inline hipStream_t as_stream(nm_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

}  // namespace nm

// entry points accept null pointers for an empty batch (R == 0): an empty torch tensor has data_ptr() == 0
#define NM_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            nm::set_error(__VA_ARGS__); \
            return NM_ERR_ARG;         \
        }                              \
    } while (0)

// ---- 16-bit storage of a training step's backward pass (mlp_bwd.hip writes, train.hip's wgrad16 kernels read) -------------------
// dZ is kept as fp16 of dZ * s, s the power of two that puts `amax` -- the largest magnitude entering the chain (d_feat, d_raw), measured on
// the device by nm_absmax -- into [2, 4): 2^13 of headroom for growth down the trunk before fp16 saturates (values are clamped, never
// inf), and entries 2^-16 below the top are still normal numbers.  Every kernel derives s from the same device scalar.
__device__ __forceinline__ float nm_dz_scale(float amax) {
    const unsigned e = (__float_as_uint(amax) >> 23) & 0xffu;             // biased exponent: amax = m 2^(e - 127), m in [1, 2)
    if (e == 0u || e == 255u) return 1.f;                                   // zero / subnormal / not finite: nothing to scale
    int se = 255 - (int)e;                                                  // s = 2^(1 - (e - 127)) -> biased exponent 127 + 128 - e
    se = se < 1 ? 1 : (se > 254 ? 254 : se);
    return __uint_as_float((unsigned)se << 23);
}
constexpr float kNmAct16Scale = 32.f;                                       // saved activations: fp16 of 32 x value (mlp_device.h kF16ActScale)

// ---- wave-level primitives (64 lanes) -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// inclusive prefix product / sum / max across the 64 lanes (Hillis-Steele over shuffles)
__device__ __forceinline__ float wave_scan_mul(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v *= t;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_add(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ __forceinline__ float wave_scan_max(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        float t = __shfl_up(v, o, 64);
        if (lane >= o) v = fmaxf(v, t);
    }
    return v;
}
