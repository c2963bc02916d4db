// Front-to-back marching of the shading pass: early ray termination + compaction of the live rays (BASELINE north star;
// the reference evaluates every sample of every ray, utils/render_utils.py:139-151).
//
// A pass of S_total sorted samples per ray is evaluated in chunks of S samples, nearest first.  After a chunk, each live
// ray's transmittance T = prod (1 - alpha_i + 1e-10) over everything evaluated so far (the factors of raw2outputs,
// render_utils.py:86-95) is updated by nm_transmittance_chunk; rays with T < eps are dropped from the list (ballot /
// prefix-sum compaction, nm_compact_hits with the predicate eps < T) and the next chunk's MLP launch
// (nm_mlp_forward_ray_chunk, csrc/mlp.hip in_mode 2) runs over the compacted (live rays x chunk samples) batch only.
// The records of samples that are never evaluated stay zero (sigma = 0: weight exactly 0 in nm_composite), so the
// composited colour differs from the full evaluation's by at most the weight that was cut off, sum_{dropped} w_i <= T < eps,
// per channel.  Nothing returns to the host between chunks: the list length lives on the device.
#include "common.h"

namespace {

// one wave per live ray; lanes over the chunk's samples.  GIVEN_DZ: `z` holds the samples' INTERVALS instead of their positions
// (a list that will be merged with others before it is composited: the interval behind a sample ends at its successor in the
// MERGED order, render_utils.py:330-345)
template <bool GIVEN_DZ>
__global__ __launch_bounds__(256) void transmittance_chunk_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                  const float* __restrict__ rays_d, const int* __restrict__ ray_idx,
                                                                  const int* __restrict__ n_rays_dev, int n_rays, int s0, int S, int S_total,
                                                                  float* __restrict__ T) {
    const int lane = threadIdx.x & 63;
    const int wave = (int)((blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6);
    const int nwaves = (int)((gridDim.x * (int64_t)blockDim.x) >> 6);
    const int n = n_rays_dev ? *n_rays_dev : n_rays;
    for (int j = wave; j < n; j += nwaves) {
        const int64_t r = ray_idx ? ray_idx[j] : j;
        const float* d = rays_d + r * 3;
        const float dn = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);                       // render_utils.py:88
        const float* zr = z + r * S_total;
        const float4* rw = reinterpret_cast<const float4*>(raw) + r * S_total;
        float prod = 1.f;
        for (int t = lane; t < S; t += 64) {
            const int i = s0 + t;
            const float dist = (GIVEN_DZ ? zr[i] : (i + 1 < S_total ? zr[i + 1] - zr[i] : 1e10f)) * dn;   // render_utils.py:85-88
            const float alpha = 1.f - expf(-fmaxf(rw[i].w, 0.f) * dist);                       // :94
            prod *= 1.f - alpha + 1e-10f;                                                      // :95
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) prod *= __shfl_xor(prod, o, 64);
        if (lane == 0) T[r] *= prod;
    }
}

}  // namespace

extern "C" {

int nm_transmittance_chunk(const float* raw, const float* z_vals, const float* rays_d, const int32_t* ray_idx, const int32_t* n_rays_dev,
                           int64_t n_rays, int s0, int S, int S_total, float* T, nm_stream_t stream) {
    NM_REQUIRE(n_rays == 0 || (raw && z_vals && rays_d && T), "nm_transmittance_chunk: null pointer");
    NM_REQUIRE(n_rays >= 0 && n_rays < (1ll << 31) && S >= 1 && s0 >= 0 && s0 + S <= S_total, "nm_transmittance_chunk: bad sizes (s0=%d S=%d S_total=%d)",
               s0, S, S_total);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15) == 0, "nm_transmittance_chunk: raw must be 16-byte aligned");
    if (n_rays == 0) return NM_OK;
    int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(transmittance_chunk_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, nm::as_stream(stream), raw, z_vals, rays_d, ray_idx,
                       n_rays_dev, (int)n_rays, s0, S, S_total, T);
    return nm::check_launch("transmittance_chunk_kernel");
}

int nm_transmittance_chunk_dz(const float* raw, const float* dz, const float* rays_d, const int32_t* ray_idx, const int32_t* n_rays_dev,
                              int64_t n_rays, int s0, int S, int S_total, float* T, nm_stream_t stream) {
    NM_REQUIRE(n_rays == 0 || (raw && dz && rays_d && T), "nm_transmittance_chunk_dz: null pointer");
    NM_REQUIRE(n_rays >= 0 && n_rays < (1ll << 31) && S >= 1 && s0 >= 0 && s0 + S <= S_total, "nm_transmittance_chunk_dz: bad sizes (s0=%d S=%d S_total=%d)",
               s0, S, S_total);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15) == 0, "nm_transmittance_chunk_dz: raw must be 16-byte aligned");
    if (n_rays == 0) return NM_OK;
    int64_t blocks = (n_rays + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(transmittance_chunk_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, nm::as_stream(stream), raw, dz, rays_d, ray_idx,
                       n_rays_dev, (int)n_rays, s0, S, S_total, T);
    return nm::check_launch("transmittance_chunk_kernel<dz>");
}

}  // extern "C"
