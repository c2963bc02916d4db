// SMPL-guided ray bounds and hit-ray compaction for gfx950.
//
// a2 geometry_guided_near_far (reference utils/ray_utils.py:197-233): the reference materialises four
// [R,V,3] f32 tensors (4 x 169 MB at R=2048, V=6890).  Here one lane owns one ray and keeps min/max
// in registers; the vertex index is wave-uniform so the vertex stream is read through the scalar
// cache (s_load) and never occupies vector-memory bandwidth.  HBM traffic = 24 B in + 8 B out per ray.
//
// a3 compaction (reference utils/render_utils.py:199-212): boolean-mask indexing becomes a wave ballot
// + popcount prefix inside the block and a two-level prefix sum across blocks; order is ascending ray
// index like the reference's mask.
#include "common.h"

namespace {

// One lane per ray, the vertex index wave-uniform (the vertex stream comes through the scalar cache).  Rays far from the body skip the
// vertex loop: every workgroup first bounds the cloud by a sphere (box centre, largest distance: 2 x V / 256 vertex reads per lane, against
// V in the loop), and a WAVE whose rays all pass that sphere at more than radius + tau -- with a margin far above float32 rounding --
// cannot reach any vertex's tau-sphere: every discriminant below is negative there, so the loop would leave near = +inf, far = -inf, which
// is what the wave writes without running it (bit-identical; only for unit directions, as the renderers' rays are: the reference's
// discriminant is a distance only then).  A frame's hit rays cluster: in the hybrid configurations 4 of 5 waves skip.  The waves
// that do run the loop apply the same test per cluster of 64 consecutive vertices.
constexpr int kMaxClusters = 512;                       // 64-vertex clusters: meshes up to 32 768 vertices (SMPL: 108)
__global__ __launch_bounds__(256) void near_far_kernel(const float* __restrict__ origin, const float* __restrict__ direction,
                                                       int64_t R, const float* __restrict__ verts, int V, float tau2, double tau2d,
                                                       float* __restrict__ near, float* __restrict__ far) {
    __shared__ float red[6][4];
    __shared__ float red_r[4];
    __shared__ float cl[kMaxClusters][4];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = threadIdx.x; v < V; v += 256)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float x = verts[v * 3 + c];
            lo[c] = fminf(lo[c], x);
            hi[c] = fmaxf(hi[c], x);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[c] = fminf(lo[c], __shfl_xor(lo[c], o, 64));
            hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o, 64));
        }
        if (lane == 0) { red[c][wv] = lo[c]; red[3 + c][wv] = hi[c]; }
    }
    __syncthreads();
    float cen[3];
#pragma unroll
    for (int c = 0; c < 3; ++c)
        cen[c] = 0.5f * (fminf(fminf(red[c][0], red[c][1]), fminf(red[c][2], red[c][3])) + fmaxf(fmaxf(red[3 + c][0], red[3 + c][1]), fmaxf(red[3 + c][2], red[3 + c][3])));
    float rad2 = 0.f;
    for (int v = threadIdx.x; v < V; v += 256) {
        const float a = verts[v * 3 + 0] - cen[0], b = verts[v * 3 + 1] - cen[1], c = verts[v * 3 + 2] - cen[2];
        rad2 = fmaxf(rad2, a * a + b * b + c * c);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) rad2 = fmaxf(rad2, __shfl_xor(rad2, o, 64));
    if (lane == 0) red_r[wv] = rad2;
    __syncthreads();
    const float rad = sqrtf(fmaxf(fmaxf(red_r[0], red_r[1]), fmaxf(red_r[2], red_r[3])));
    // bounding spheres of the clusters of 64 consecutive vertices (box centre, largest distance), one cluster per wave at a time
    const bool use_cl = (V + 63) / 64 <= kMaxClusters;
    const int ncl = use_cl ? (V + 63) / 64 : 1;
    if (use_cl) {
        for (int k = wv; k < ncl; k += 4) {
            const int v = 64 * k + lane;
            const bool in = v < V;
            const float x = in ? verts[v * 3] : 0.f, y = in ? verts[v * 3 + 1] : 0.f, z = in ? verts[v * 3 + 2] : 0.f;
            float l3[3] = {in ? x : INFINITY, in ? y : INFINITY, in ? z : INFINITY}, h3[3] = {in ? x : -INFINITY, in ? y : -INFINITY, in ? z : -INFINITY};
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    l3[c] = fminf(l3[c], __shfl_xor(l3[c], o, 64));
                    h3[c] = fmaxf(h3[c], __shfl_xor(h3[c], o, 64));
                }
            const float mx = 0.5f * (l3[0] + h3[0]), my = 0.5f * (l3[1] + h3[1]), mz = 0.5f * (l3[2] + h3[2]);
            float r2 = in ? (x - mx) * (x - mx) + (y - my) * (y - my) + (z - mz) * (z - mz) : 0.f;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r2 = fmaxf(r2, __shfl_xor(r2, o, 64));
            if (lane == 0) { cl[k][0] = mx; cl[k][1] = my; cl[k][2] = mz; cl[k][3] = sqrtf(r2); }
        }
    } else if (threadIdx.x == 0) {
        cl[0][0] = cen[0]; cl[0][1] = cen[1]; cl[0][2] = cen[2]; cl[0][3] = rad;
    }
    __syncthreads();

    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const int64_t r = i < R ? i : R - 1;
    const float ox = origin[r * 3 + 0], oy = origin[r * 3 + 1], oz = origin[r * 3 + 2];
    const float dx = direction[r * 3 + 0], dy = direction[r * 3 + 1], dz = direction[r * 3 + 2];
    float n = INFINITY, f = -INFINITY;                                     // ray_utils.py:213-218 (NaN -> +-inf)
    bool may_hit = true;
    {
        const float cx = cen[0] - ox, cy = cen[1] - oy, cz = cen[2] - oz;
        const float dd = dx * dx + dy * dy + dz * dz, cc = cx * cx + cy * cy + cz * cz, cd = cx * dx + cy * dy + cz * dz;
        const float reach = (rad + sqrtf(tau2)) * 1.001f + 1e-3f * (1.f + sqrtf(cc));
        const bool unit = fabsf(dd - 1.f) < 1e-6f && cc < 1e6f;                // (|d| = 1 to float32 rounding; the margins hold to |c| ~ 1e3)
        may_hit = !(unit && cc - cd * cd > reach * reach * 1.01f) || !(rad == rad) || !(cc == cc);      // (non-finite anything: run the loop)
    }
    if (__any(may_hit)) {
        // the same test per CLUSTER of 64 consecutive vertices (their bounding spheres: cl[], built below by this workgroup): a wave skips the
        // clusters none of its rays can reach -- the vertices it does visit are visited in ascending order with the reference's arithmetic
        const float tau = sqrtf(tau2);
        const float ddq = dx * dx + dy * dy + dz * dz;
        const bool unit_d = fabsf(ddq - 1.f) < 1e-6f;
        for (int k = 0; k < ncl; ++k) {
            const float cx = cl[k][0] - ox, cy = cl[k][1] - oy, cz = cl[k][2] - oz, rk = cl[k][3];
            const float cc = cx * cx + cy * cy + cz * cz, cd = cx * dx + cy * dy + cz * dz;
            const float reach = (rk + tau) * 1.001f + 1e-3f * (1.f + sqrtf(cc));
            const bool skip = use_cl && unit_d && cc < 1e6f && cc - cd * cd > reach * reach * 1.01f && rk == rk;
            if (!__any(!skip)) continue;
            const int v1 = use_cl ? (64 * k + 64 < V ? 64 * k + 64 : V) : V;
#pragma unroll 4
            for (int v = use_cl ? 64 * k : 0; v < v1; ++v) {
                // The reference's expression :209-211 -- tau^2 - (|v - o|^2 - z0^2) -- cancels two numbers of size |v - o|^2 ~ 10 down to ~ tau^2 =
                // 0.04: in float32 the bracket is right to ~1e-6 and the root divides that by 2 dz, so near / far of a float32 evaluation carry
                // 3e-5 (grazing rays 1e-3) -- the reference's own torch and numpy branches differ by that much, and every sample of the ray moves
                // with them.  The differences v - o are exact in float32 for points this close; the products and the cancellation are done in
                // float64 here (same rate as scalar float32 FMAs on gfx950), so near / far are the values of a float64 evaluation of the
                // reference's formula rounded once: the device's bounds sit 1e-7 from the exact ones instead of adding a float32 error of their own
                const float vxf = verts[v * 3 + 0] - ox, vyf = verts[v * 3 + 1] - oy, vzf = verts[v * 3 + 2] - oz;  // :209
                const double vx = vxf, vy = vyf, vz = vzf;
                const double z0 = fma(vz, (double)dz, fma(vy, (double)dy, vx * (double)dx));                       // :210
                const double disc = tau2d - (fma(vz, vz, fma(vy, vy, vx * vx)) - z0 * z0);                  // :211
                if (disc >= 0.0) {                                                  // sqrt(negative) = NaN -> dropped
                    const float dzv = sqrtf((float)disc), z0f = (float)z0;
                    n = fminf(n, z0f - dzv);
                    f = fmaxf(f, z0f + dzv);
                }
            }
        }
    }
    if (i < R) {
        near[i] = n;
        far[i] = f;
    }
}

constexpr int kCompactBlock = 256;

// pass 1: per-block number of hits
__global__ __launch_bounds__(kCompactBlock) void count_hits_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                                                   int64_t R, int32_t* __restrict__ block_counts) {
    __shared__ int wave_cnt[kCompactBlock / 64];
    const int64_t i = blockIdx.x * (int64_t)kCompactBlock + threadIdx.x;
    const bool hit = i < R && near[i] < far[i];
    const unsigned long long b = __ballot(hit);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(b);
    __syncthreads();
    if (threadIdx.x == 0) {
        int c = 0;
        for (int w = 0; w < kCompactBlock / 64; ++w) c += wave_cnt[w];
        block_counts[blockIdx.x] = c;
    }
}

// pass 2: exclusive scan of the block counts (single block), totals into counts[0..1]
__global__ __launch_bounds__(1024) void scan_blocks_kernel(int32_t* __restrict__ block_counts, int nblocks, int64_t R,
                                                           int32_t* __restrict__ counts) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x;
        const int v = i < nblocks ? block_counts[i] : 0;
        int inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wave_tot[wid] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wid; ++w) woff += wave_tot[w];
        const int carry = carry_s;
        if (i < nblocks) block_counts[i] = carry + woff + inc - v;         // exclusive
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = carry + woff + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        counts[0] = carry_s;
        counts[1] = (int32_t)(R - carry_s);
    }
}

// pass 3: write the compacted index lists
__global__ __launch_bounds__(kCompactBlock) void write_hits_kernel(const float* __restrict__ near, const float* __restrict__ far,
                                                                   int64_t R, const int32_t* __restrict__ block_offsets,
                                                                   int32_t* __restrict__ hit_idx, int32_t* __restrict__ miss_idx) {
    __shared__ int wave_cnt[kCompactBlock / 64];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t i = blockIdx.x * (int64_t)kCompactBlock + threadIdx.x;
    const bool in = i < R;
    const bool hit = in && near[i] < far[i];
    const unsigned long long b = __ballot(hit);
    if (lane == 0) wave_cnt[wid] = __popcll(b);
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wid; ++w) woff += wave_cnt[w];
    const int rank = woff + __popcll(b & ((1ull << lane) - 1ull));          // hits before this lane in the block
    const int hoff = block_offsets[blockIdx.x];
    if (hit) {
        hit_idx[hoff + rank] = (int32_t)i;
    } else if (in && miss_idx) {
        const int64_t before = blockIdx.x * (int64_t)kCompactBlock + threadIdx.x;   // rays before this one
        miss_idx[before - (hoff + rank)] = (int32_t)i;
    }
}

}  // namespace

extern "C" {

int nm_near_far(const float* origin, const float* direction, int64_t R, const float* verts, int V, double geo_threshold,
                float* near, float* far, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (origin && direction && verts && near && far), "nm_near_far: null pointer");
    NM_REQUIRE(R >= 0 && V >= 1, "nm_near_far: bad sizes");
    if (R == 0) return NM_OK;
    const float tau2 = (float)(geo_threshold * geo_threshold);             // python float ** 2 (ray_utils.py:211): float32 for the skip tests' margins,
    const double tau2d = geo_threshold * geo_threshold;                    // the double itself in the discriminant
    const int blocks = (int)((R + 255) / 256);
    hipLaunchKernelGGL(near_far_kernel, dim3(blocks), dim3(256), 0, nm::as_stream(stream), origin, direction, R, verts, V,
                       tau2, tau2d, near, far);
    return nm::check_launch("near_far_kernel");
}

int64_t nm_compact_workspace_ints(int64_t R) { return (R + kCompactBlock - 1) / kCompactBlock + 2; }

int nm_compact_hits(const float* near, const float* far, int64_t R, int32_t* hit_idx, int32_t* miss_idx, int32_t* counts,
                    int32_t* workspace, nm_stream_t stream) {
    NM_REQUIRE(near && far && hit_idx && counts && workspace, "nm_compact_hits: null pointer");
    NM_REQUIRE(R >= 0 && R < (1ll << 31), "nm_compact_hits: R out of range");
    hipStream_t st = nm::as_stream(stream);
    const int nblocks = (int)((R + kCompactBlock - 1) / kCompactBlock);
    if (nblocks > 0) {
        hipLaunchKernelGGL(count_hits_kernel, dim3(nblocks), dim3(kCompactBlock), 0, st, near, far, R, workspace);
        if (int e = nm::check_launch("count_hits_kernel")) return e;
    }
    hipLaunchKernelGGL(scan_blocks_kernel, dim3(1), dim3(1024), 0, st, workspace, nblocks, R, counts);
    if (int e = nm::check_launch("scan_blocks_kernel")) return e;
    if (nblocks > 0) {
        hipLaunchKernelGGL(write_hits_kernel, dim3(nblocks), dim3(kCompactBlock), 0, st, near, far, R, workspace, hit_idx,
                           miss_idx);
        if (int e = nm::check_launch("write_hits_kernel")) return e;
    }
    return NM_OK;
}

}  // extern "C"
