// Per-ray kernels of the NeuMan hot path for gfx950: stratified sampling, alpha compositing,
// hierarchical (inverse-CDF) resampling with a fused sorted merge, two-list merge, row gather/scatter.
//
// All of them are HBM-bound, one wavefront (64 lanes) per ray with the ray's sample state staged in
// LDS; reads and writes of a ray's S samples are contiguous so every wave instruction is a coalesced
// 256 B - 1 KiB transaction.  Built with -ffp-contract=off: the reference evaluates `a*b + c` as two
// roundings (torch elementwise ops), so nothing here may be fused unless written as fmaf().
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int kRayWavesPerBlock = 4;

// ------------------------------------------------------------------------------------------------
// a4 ray_to_samples (reference utils/ray_utils.py:96-135)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lerp_z(float near, float far, float t, int lindisp) {
    if (!lindisp) return near * (1.f - t) + far * t;                      // ray_utils.py:112
    return 1.f / (1.f / near * (1.f - t) + 1.f / far * t);                 // ray_utils.py:114
}

__global__ __launch_bounds__(256) void ray_to_samples_kernel(
    const float* __restrict__ origin, const float* __restrict__ direction, const float* __restrict__ near,
    const float* __restrict__ far, int64_t R, int S, const float* __restrict__ t_vals, int lindisp,
    const float* __restrict__ t_rand, float* __restrict__ pts, float* __restrict__ dirs, float* __restrict__ z_vals) {
    const int64_t total = R * (int64_t)S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / S;
        const int s = (int)(i - r * S);
        const float n = near[r], f = far[r];
        float z = lerp_z(n, f, t_vals[s], lindisp);
        if (t_rand) {                                                       // ray_utils.py:116-129
            const float zm = s > 0 ? lerp_z(n, f, t_vals[s - 1], lindisp) : z;
            const float zp = s < S - 1 ? lerp_z(n, f, t_vals[s + 1], lindisp) : z;
            const float lower = s > 0 ? .5f * (z + zm) : z;
            const float upper = s < S - 1 ? .5f * (zp + z) : z;
            z = lower + (upper - lower) * t_rand[i];
        }
        z_vals[i] = z;
        const float dx = direction[r * 3 + 0], dy = direction[r * 3 + 1], dz = direction[r * 3 + 2];
        if (pts) {                                                          // ray_utils.py:131
            pts[i * 3 + 0] = origin[r * 3 + 0] + dx * z;
            pts[i * 3 + 1] = origin[r * 3 + 1] + dy * z;
            pts[i * 3 + 2] = origin[r * 3 + 2] + dz * z;
        }
        if (dirs) {                                                         // ray_utils.py:132
            dirs[i * 3 + 0] = dx;
            dirs[i * 3 + 1] = dy;
            dirs[i * 3 + 2] = dz;
        }
    }
}

__global__ __launch_bounds__(256) void z_to_points_kernel(const float* __restrict__ origin, const float* __restrict__ direction,
                                                          const float* __restrict__ z_vals, int64_t R, int S,
                                                          float* __restrict__ pts, float* __restrict__ dirs) {
    const int64_t total = R * (int64_t)S;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / S;
        const float z = z_vals[i];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float d = direction[r * 3 + c];
            if (pts) pts[i * 3 + c] = origin[r * 3 + c] + d * z;           // ray_utils.py:153
            if (dirs) dirs[i * 3 + c] = d;                                  // ray_utils.py:155
        }
    }
}

// ------------------------------------------------------------------------------------------------
// a5 raw2outputs (reference utils/render_utils.py:69-105): one wave per ray.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ int upper_bound_lds(const float* a, int n, float v) {  // first i with a[i] > v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] > v) hi = mid; else lo = mid + 1;
    }
    return lo;
}
__device__ __forceinline__ int lower_bound_lds(const float* a, int n, float v) {  // first i with a[i] >= v
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}


struct CompositeSums {
    float r, g, b, d, a;
};
// One ray's compositing (render_utils.py:85-100), the wave's lanes across its S samples in chunks of 64.  raw_at(s) / z_at(s): the
// s-th record and depth of the list (global memory, or a merged list staged in LDS); w_out(s, w): called with every weight.  Every
// kernel that composites goes through this one body, so they all produce the same bits.
template <class RawAt, class ZAt, class WOut>
__device__ __forceinline__ CompositeSums composite_ray(int S, float dnorm, int lane, const float* noise_row, RawAt raw_at, ZAt z_at, WOut w_out) {
    double t_carry = 1.0;
    float sr = 0.f, sg = 0.f, sb = 0.f, sd = 0.f, sa = 0.f;
    for (int c0 = 0; c0 < S; c0 += 64) {
        const int s = c0 + lane;
        const bool valid = s < S;
        float w = 0.f, f = 1.f;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
        float z = 0.f;
        if (valid) {
            q = raw_at(s);
            z = z_at(s);
            float dist = (s + 1 < S) ? (z_at(s + 1) - z) : 1e10f;       // render_utils.py:85-86
            dist = dist * dnorm;
            float sigma = q.w;
            if (noise_row) sigma = sigma + noise_row[s];               // render_utils.py:93-94
            const float alpha = 1.f - expf(-fmaxf(sigma, 0.f) * dist);  // render_utils.py:81
            w = alpha;
            f = 1.f - alpha + 1e-10f;                                   // render_utils.py:95
        }
        // transmittance: running product in f64, rounded to f32 per entry -- order independent, and what torch's
        // CPU cumprod computes for f32 inputs (the weights feed the inverse-CDF step function)
        double incl = (double)f;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const double t = __shfl_up(incl, o, 64);
            if (lane >= o) incl *= t;
        }
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = 1.0;
        w = w * (float)(t_carry * excl);
        t_carry = t_carry * __shfl(incl, 63, 64);
        if (valid) {
            w_out(s, w);
            sr += w * sigmoidf_(q.x);                                   // render_utils.py:90, 96
            sg += w * sigmoidf_(q.y);
            sb += w * sigmoidf_(q.z);
            sd += w * z;                                                // render_utils.py:98
            sa += w;                                                    // render_utils.py:100
        }
    }
    CompositeSums c;
    c.r = wave_sum(sr); c.g = wave_sum(sg); c.b = wave_sum(sb); c.d = wave_sum(sd); c.a = wave_sum(sa);
    return c;
}
__device__ __forceinline__ void composite_store(CompositeSums c, int white_bkg, int64_t r, float* rgb, float* disp, float* acc, float* depth) {
    if (white_bkg) {                                                    // render_utils.py:102-103
        const float bg = 1.f - c.a;
        c.r = c.r + bg; c.g = c.g + bg; c.b = c.b + bg;
    }
    rgb[r * 3 + 0] = c.r; rgb[r * 3 + 1] = c.g; rgb[r * 3 + 2] = c.b;
    depth[r] = c.d;
    acc[r] = c.a;
    if (disp) {
        const float q = c.d / c.a;                                      // NaN when acc == 0, as torch.max propagates it
        const float m = (q != q) ? q : fmaxf(1e-10f, q);                // render_utils.py:99
        disp[r] = 1.f / m;
    }
}

__global__ __launch_bounds__(64 * kRayWavesPerBlock) void composite_kernel(
    const float4* __restrict__ raw, const float* __restrict__ z_vals, const float* __restrict__ rays_d, int64_t R, int S,
    int white_bkg, const float* __restrict__ noise, float* __restrict__ rgb, float* __restrict__ disp,
    float* __restrict__ acc, float* __restrict__ weights, float* __restrict__ depth) {
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = blockIdx.x * (int64_t)kRayWavesPerBlock + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * kRayWavesPerBlock;
    for (int64_t r = wave0; r < R; r += nwaves) {
        const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
        const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);            // render_utils.py:88
        const float* zr = z_vals + r * S;
        const float4* rr = raw + r * S;
        float* wr = weights ? weights + r * S : nullptr;
        const CompositeSums c = composite_ray(S, dnorm, lane, noise ? noise + r * S : nullptr, [&](int s) { return rr[s]; }, [&](int s) { return zr[s]; },
                                              [&](int s, float w) { if (wr) wr[s] = w; });
        if (lane == 0) composite_store(c, white_bkg, r, rgb, disp, acc, depth);
    }
}

// ------------------------------------------------------------------------------------------------
// a13 + a5 as ONE kernel: k sorted lists per ray -> merged order -> compositing, the merged list never written to HBM
// (reference utils/render_utils.py:330-345, 441-456: sort(cat(z lists)), the three-index gather of cat(raw lists), raw2outputs).
// One wave per ray.  The lists' z are staged in LDS; every sample's position in the merged order is its own index plus, per other
// list, the number of that list's samples before it (binary search; on equal z the EARLIER list first = what merging list by list with
// nm_merge_sorted gives); its record goes to that position of an LDS copy of the merged list, and composite_ray runs on the copy.
// rows[l] (nullable): list l's arrays are indexed by rows[l][ray] instead of ray (the background list of the hit rays in place).
// ------------------------------------------------------------------------------------------------
constexpr int kMaxMergeLists = 4;
struct MergeLists {
    const float* z[kMaxMergeLists];
    const float4* raw[kMaxMergeLists];
    const int32_t* rows[kMaxMergeLists];
    int S[kMaxMergeLists];
    int k, S_total;
};
__global__ __launch_bounds__(256) void merge_composite_kernel(const MergeLists L, int64_t R, const float* __restrict__ rays_d, int white_bkg,
                                                              float* __restrict__ rgb, float* __restrict__ depth, float* __restrict__ acc) {
    extern __shared__ float lds_f[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const int St = L.S_total;
    // per wave: the lists' z, concatenated [St] | merged z [St] | merged source (list << 16 | index) [St].  The records themselves are
    // read from global memory in merged order by the compositing pass (runs of one list: coalesced), once each.
    // (Measured, 524 288 rays x (320 + 3 x 192): 9.0 ms against 16.6 ms for three nm_merge_sorted + nm_composite; staging the records in
    //  LDS at their merged position as well -- 24 B per sample, 7 waves per CU instead of 12 -- 13.8 ms: the kernel is bound by the latency
    //  of the per-ray chains (searches, the f64 transmittance scan), i.e. by the waves in flight, not by bytes.)
    float* lz = lds_f + (size_t)wib * 3 * St;
    float* mz = lz + St;
    unsigned* msrc = reinterpret_cast<unsigned*>(mz + St);
    int off[kMaxMergeLists + 1];
    off[0] = 0;
#pragma unroll
    for (int l = 0; l < kMaxMergeLists; ++l) off[l + 1] = off[l] + L.S[l];
    int steps = 0;                                                 // binary-search steps: enough for the longest list
#pragma unroll
    for (int l = 0; l < kMaxMergeLists; ++l)
        while ((1 << steps) <= L.S[l]) ++steps;
    for (int64_t r0 = blockIdx.x * (int64_t)wpb; r0 < R; r0 += (int64_t)gridDim.x * wpb) {
        const bool live = r0 + wib < R;
        const int64_t r = live ? r0 + wib : R - 1;
        const float4* rbase[kMaxMergeLists];
#pragma unroll
        for (int l = 0; l < kMaxMergeLists; ++l) {
            rbase[l] = nullptr;
            if (l < L.k) {
                const int64_t row = L.rows[l] ? (int64_t)L.rows[l][r] : r;
                const float* zr = L.z[l] + row * L.S[l];
                rbase[l] = L.raw[l] + row * L.S[l];
                for (int i = lane; i < L.S[l]; i += 64) lz[off[l] + i] = zr[i];
            }
        }
        __syncthreads();
        // position in the merged order = own index + per other list the number of its samples that come first (an earlier list's
        // equal samples do, a later list's do not).  Branch-free searches with a fixed number of steps, the lists side by side and two
        // elements per lane, so that the LDS round trips of a step overlap instead of forming one chain per list.
        for (int e0 = 0; e0 < St; e0 += 128) {
            int e[2], l_of[2], cnt[2][kMaxMergeLists];
            float v[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                e[q] = e0 + 64 * q + lane;
                const int ec = e[q] < St ? e[q] : St - 1;
                v[q] = lz[ec];
                l_of[q] = (ec >= off[1]) + (ec >= off[2]) + (ec >= off[3]);
#pragma unroll
                for (int m = 0; m < kMaxMergeLists; ++m) cnt[q][m] = 0;
            }
            for (int st = steps - 1; st >= 0; --st) {
                const int half = 1 << st;
                float x[2][kMaxMergeLists];
                // all eight probes of the step are issued before any is used (an unused list has S = 0: its count stays 0)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int m = 0; m < kMaxMergeLists; ++m) {
                        const int t = cnt[q][m] + half;
                        x[q][m] = lz[off[m] + (t <= L.S[m] ? t - 1 : 0)];
                    }
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int m = 0; m < kMaxMergeLists; ++m) {
                        const int t = cnt[q][m] + half;
                        const bool first = m < l_of[q] ? x[q][m] <= v[q] : x[q][m] < v[q];
                        cnt[q][m] = (t <= L.S[m] && first) ? t : cnt[q][m];
                    }
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                if (e[q] >= St) continue;
                int k = e[q] - off[l_of[q]];
#pragma unroll
                for (int m = 0; m < kMaxMergeLists; ++m)
                    if (m < L.k && m != l_of[q]) k += cnt[q][m];
                mz[k] = v[q];
                msrc[k] = ((unsigned)l_of[q] << 16) | (unsigned)(e[q] - off[l_of[q]]);
            }
        }
        __syncthreads();
        const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
        const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
        const CompositeSums c = composite_ray(St, dnorm, lane, nullptr,
                                              [&](int s) {
                                                  const unsigned src = msrc[s];
                                                  const unsigned l = src >> 16;
                                                  const float4* base = l == 0 ? rbase[0] : l == 1 ? rbase[1] : l == 2 ? rbase[2] : rbase[3];
                                                  return base[src & 0xffffu];
                                              },
                                              [&](int s) { return mz[s]; }, [&](int, float) {});
        if (lane == 0 && live) composite_store(c, white_bkg, r, rgb, nullptr, acc, depth);
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// a6/a7 sample_pdf (det) + ray_to_importance_samples (reference utils/ray_utils.py:138-194)
// one wave per ray; per-wave LDS: bins[B] | cdf[B] | zs[N]   (z itself is re-read from global / L1)
// ------------------------------------------------------------------------------------------------
constexpr int kMaxImportanceChunks = 8;     // importance samples per ray <= 512 when merged with the old samples
// MODE 0: bins/weights given (sample_pdf).  MODE 1: derive from z/w (importance), optionally merge.  MODE 2: as MODE 1 with the
// compositing weights computed HERE from the pass's raw output (raw2outputs' weights, composite_ray: the same bits as nm_composite
// writes) -- the coarse tail `raw2outputs -> sample_pdf -> sort(cat)` of a two-pass render (render_utils.py:139-147) as one kernel
// that reads sigma once and writes only the merged sample positions (the weights too when w_out != NULL).
template <int MODE>
__global__ __launch_bounds__(64 * kRayWavesPerBlock) void sample_pdf_kernel(
    const float* __restrict__ in_a /* bins [R,B] | z [R,S] */, const float* __restrict__ in_w /* weights [R,B-1] | w [R,S] */,
    int64_t R, int B, const float* __restrict__ u, int N, int including_old, float* __restrict__ out,
    const float4* __restrict__ raw = nullptr, const float* __restrict__ rays_d = nullptr, float* __restrict__ w_out = nullptr) {
    extern __shared__ float lds_f[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    const int S = B + 1;                       // MODE 1: samples per ray
    const int per_wave = 2 * B + N + (MODE >= 1 ? S : 0) + (MODE == 2 ? S : 0);
    float* bins = lds_f + wib * per_wave;
    float* cdf = bins + B;
    float* zs = cdf + B;
    float* zl = zs + N;                        // MODE 1, 2: the ray's coarse z
    float* wl = zl + S;                        // MODE 2: the ray's compositing weights
    const int nW = B - 1;                      // number of pdf weights
    // block-uniform trip count so __syncthreads() is legal; a wave past the end idles on r = R-1 without storing
    for (int64_t r0 = blockIdx.x * (int64_t)kRayWavesPerBlock; r0 < R; r0 += (int64_t)gridDim.x * kRayWavesPerBlock) {
        const bool live = r0 + wib < R;
        const int64_t r = live ? r0 + wib : R - 1;
        // ---- bins and (weights + 1e-5)
        // The inverse-CDF lookup below is a step function of the f32 cdf entries, so the two
        // reductions are done the way that is independent of the reduction ORDER: the normaliser is the correctly
        // rounded f32 sum (f64 accumulate) and the running sum accumulates in f64 and rounds every entry to f32 -- which
        // is what torch's CPU cumsum does (acc_type<float> = double) and what oracle/ray_ops.py restates.
        double wsum_d = 0.0;
        if (MODE == 0) {
            for (int i = lane; i < B; i += 64) bins[i] = in_a[r * B + i];
            for (int i = lane; i < nW; i += 64) wsum_d += (double)(in_w[r * nW + i] + 1e-5f);
        } else {
            const float* zr = in_a + r * S;
            for (int i = lane; i < S; i += 64) zl[i] = zr[i];
            for (int i = lane; i < B; i += 64) bins[i] = .5f * (zr[i + 1] + zr[i]);     // ray_utils.py:148
            if (MODE == 2) {
                const float dx = rays_d[r * 3 + 0], dy = rays_d[r * 3 + 1], dz = rays_d[r * 3 + 2];
                const float dnorm = sqrtf(dx * dx + dy * dy + dz * dz);
                const float4* rr = raw + r * S;
                float* wr = (w_out && live) ? w_out + r * S : nullptr;
                (void)composite_ray(S, dnorm, lane, nullptr, [&](int s_) { return rr[s_]; }, [&](int s_) { return zr[s_]; },
                                    [&](int s_, float w_) { wl[s_] = w_; if (wr) wr[s_] = w_; });
                __syncthreads();
            }
            for (int i = lane; i < nW; i += 64) wsum_d += (double)((MODE == 2 ? wl[1 + i] : in_w[r * S + 1 + i]) + 1e-5f);   // weights[..., 1:-1], :149
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) wsum_d += __shfl_xor(wsum_d, o, 64);
        const float wsum = (float)wsum_d;                                                // ray_utils.py:167
        // ---- cdf = [0, cumsum(pdf)]                                                   // ray_utils.py:168-169
        double carry = 0.0;
        if (lane == 0) cdf[0] = 0.f;
        for (int c0 = 0; c0 < nW; c0 += 64) {
            const int i = c0 + lane;
            float p = 0.f;
            if (i < nW) {
                const float wv = (MODE == 0 ? in_w[r * nW + i] : MODE == 2 ? wl[1 + i] : in_w[r * S + 1 + i]) + 1e-5f;
                p = wv / wsum;
            }
            double inc = (double)p;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double t = __shfl_up(inc, o, 64);
                if (lane >= o) inc += t;
            }
            if (i < nW) cdf[i + 1] = (float)(carry + inc);
            carry = carry + __shfl(inc, 63, 64);
        }
        __syncthreads();
        // ---- invert the cdf at u                                                      // ray_utils.py:180-192
        bool inverted = false;                                   // MODE 1: some sample is smaller than its predecessor
        float prev_last = -INFINITY;
        for (int c0 = 0; c0 < N; c0 += 64) {
            const int j = c0 + lane;
            float smp = INFINITY;
            if (j < N) {
                const float uj = u[j];
                const int inds = upper_bound_lds(cdf, B, uj);                            // searchsorted(right=True)
                const int below = max(0, inds - 1);
                const int above = min(B - 1, inds);
                const float cdf0 = cdf[below], cdf1 = cdf[above];
                const float b0 = bins[below], b1 = bins[above];
                float denom = cdf1 - cdf0;
                if (denom < 1e-5f) denom = 1.f;
                const float t = (uj - cdf0) / denom;
                smp = b0 + t * (b1 - b0);
            }
            if (MODE >= 1 && including_old) {
                float prev = __shfl_up(smp, 1, 64);
                if (lane == 0) prev = prev_last;
                inverted |= j < N && smp < prev;
                prev_last = __shfl(smp, 63, 64);
                if (j < N) zs[j] = smp;
            } else if (j < N && live) {
                out[r * N + j] = smp;
            }
        }
        if (MODE >= 1 && including_old && __any(inverted)) {
            // The inverse-CDF output is monotone up to f32 rounding; the reference's torch.sort (ray_utils.py:151) puts the
            // rare inversions in order, and so does this exact, stable rank sort (a wave-uniform branch taken only by rays
            // that have one; N <= 64 * kMaxImportanceChunks is checked on the host).  Values and ranks of every chunk are
            // held in registers until all ranks are known; a wave's DS operations execute in program order, so the loads
            // are done before the first store.
            float v[kMaxImportanceChunks];
            int rk[kMaxImportanceChunks];
#pragma unroll
            for (int c = 0; c < kMaxImportanceChunks; ++c) {
                const int j = c * 64 + lane;
                v[c] = j < N ? zs[j] : INFINITY;
                rk[c] = 0;
            }
            for (int k = 0; k < N; ++k) {
                const float x = zs[k];
#pragma unroll
                for (int c = 0; c < kMaxImportanceChunks; ++c) rk[c] += (x < v[c] || (x == v[c] && k < c * 64 + lane)) ? 1 : 0;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int c = 0; c < kMaxImportanceChunks; ++c)
                if (c * 64 + lane < N) zs[rk[c]] = v[c];
        }
        __syncthreads();
        if (MODE >= 1 && including_old && live) {
            // ---- rank merge of z (S, sorted) and zs (N, sorted) == sort(cat)              // ray_utils.py:151-152
            float* o = out + r * (int64_t)(S + N);
            for (int i = lane; i < S; i += 64) {
                const float v = zl[i];
                o[i + lower_bound_lds(zs, N, v)] = v;
            }
            for (int j = lane; j < N; j += 64) {
                const float v = zs[j];
                o[j + upper_bound_lds(zl, S, v)] = v;
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// a13 two-list sorted merge with raw gather (reference utils/render_utils.py:330-337, 441-448)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * kRayWavesPerBlock) void merge_sorted_kernel(
    const float* __restrict__ za, const float4* __restrict__ rawa, int Sa, const float* __restrict__ zb,
    const float4* __restrict__ rawb, int Sb, int64_t R, float* __restrict__ z_out, float4* __restrict__ raw_out) {
    extern __shared__ float lds_f[];
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    float* a = lds_f + wib * (Sa + Sb);
    float* b = a + Sa;
    for (int64_t r0 = blockIdx.x * (int64_t)kRayWavesPerBlock; r0 < R; r0 += (int64_t)gridDim.x * kRayWavesPerBlock) {
        const bool live = r0 + wib < R;
        const int64_t r = live ? r0 + wib : R - 1;
        for (int i = lane; i < Sa; i += 64) a[i] = za[r * Sa + i];
        for (int i = lane; i < Sb; i += 64) b[i] = zb[r * Sb + i];
        __syncthreads();
        const int64_t ob = r * (int64_t)(Sa + Sb);
        for (int i = lane; i < Sa && live; i += 64) {         // ties: list a first (stable)
            const float v = a[i];
            const int k = i + lower_bound_lds(b, Sb, v);
            z_out[ob + k] = v;
            raw_out[ob + k] = rawa[r * Sa + i];
        }
        for (int j = lane; j < Sb && live; j += 64) {
            const float v = b[j];
            const int k = j + upper_bound_lds(a, Sa, v);
            z_out[ob + k] = v;
            raw_out[ob + k] = rawb[r * Sb + j];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------
// row gather / scatter (boolean-mask indexing, reference utils/render_utils.py:206-212, 231-233)
// ------------------------------------------------------------------------------------------------
template <bool SCATTER>
__global__ __launch_bounds__(256) void rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                   const int32_t* __restrict__ n_dev, int64_t n_max, int width,
                                                   float* __restrict__ dst) {
    const int64_t n = n_dev ? min((int64_t)n_dev[0], n_max) : n_max;
    const int64_t total = n * width;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / width;
        const int c = (int)(i - row * width);
        const int64_t j = idx[row];
        if (SCATTER) dst[j * width + c] = src[i];
        else dst[i] = src[j * width + c];
    }
}

// ------------------------------------------------------------------------------------------------
// For every sample of k <= 4 sorted lists per ray: the distance to its successor in the MERGED (stable, earlier list first) order of the
// reference's sort(cat(...)) (render_utils.py:330-337, 441-448); the last sample of the merged list gets raw2outputs' 1e10 (:86).  The
// successor of sample i of list a is the nearest of: its own list's sample i + 1, the first sample >= z of every LATER list (equal values
// of a later list come after) and the first sample > z of every EARLIER list.  One thread per sample, binary searches in the other rows
// (a row is at most a few hundred floats: L1 / L2 resident); nothing is sorted, nothing is written but dz.  Bound: HBM (4 B in, 4 B out).
// ------------------------------------------------------------------------------------------------
struct IntervalLists {
    const float* z[kMaxMergeLists];
    float* dz[kMaxMergeLists];
    int S[kMaxMergeLists];
    int k, S_total;
};
__global__ __launch_bounds__(256) void merged_intervals_kernel(const IntervalLists L, int64_t R) {
    const int64_t total = R * L.S_total;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ray = t / L.S_total;
        int i = (int)(t - ray * L.S_total), a = 0;
        while (i >= L.S[a]) { i -= L.S[a]; ++a; }
        const float* za = L.z[a] + ray * L.S[a];
        const float z = za[i];
        float next = i + 1 < L.S[a] ? za[i + 1] : INFINITY;
        for (int b = 0; b < L.k; ++b) {
            if (b == a) continue;
            const float* zb = L.z[b] + ray * L.S[b];
            int lo = 0, hi = L.S[b];                       // first index whose sample comes after (a, i) in the stable order
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float v = zb[mid];
                if (b > a ? v >= z : v > z) hi = mid; else lo = mid + 1;
            }
            if (lo < L.S[b]) next = fminf(next, zb[lo]);
        }
        L.dz[a][ray * L.S[a] + i] = next == INFINITY ? 1e10f : next - z;
    }
}


inline int grid_for(int64_t work_items, int per_block, int max_blocks = 256 * 16) {
    int64_t b = (work_items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > max_blocks) b = max_blocks;
    return (int)b;
}

}  // namespace

extern "C" {

int nm_ray_to_samples(const float* origin, const float* direction, const float* near, const float* far, int64_t R, int S,
                      const float* t_vals, int lindisp, const float* t_rand, float* pts, float* dirs, float* z_vals,
                      nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (origin && direction && near && far && t_vals && z_vals), "nm_ray_to_samples: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1, "nm_ray_to_samples: bad sizes R=%lld S=%d", (long long)R, S);
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(ray_to_samples_kernel, dim3(grid_for(R * S, 256)), dim3(256), 0, nm::as_stream(stream), origin,
                       direction, near, far, R, S, t_vals, lindisp, t_rand, pts, dirs, z_vals);
    return nm::check_launch("ray_to_samples_kernel");
}

int nm_z_to_points(const float* origin, const float* direction, const float* z_vals, int64_t R, int S, float* pts,
                   float* dirs, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (origin && direction && z_vals), "nm_z_to_points: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1, "nm_z_to_points: bad sizes");
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(z_to_points_kernel, dim3(grid_for(R * S, 256)), dim3(256), 0, nm::as_stream(stream), origin,
                       direction, z_vals, R, S, pts, dirs);
    return nm::check_launch("z_to_points_kernel");
}

int nm_composite(const float* raw, const float* z_vals, const float* rays_d, int64_t R, int S, int white_bkg,
                 const float* noise, float* rgb, float* disp, float* acc, float* weights, float* depth,
                 nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (raw && z_vals && rays_d && rgb && acc && depth), "nm_composite: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1, "nm_composite: bad sizes R=%lld S=%d", (long long)R, S);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15) == 0, "nm_composite: raw must be 16-byte aligned");
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(composite_kernel, dim3(grid_for(R, kRayWavesPerBlock)), dim3(64 * kRayWavesPerBlock), 0,
                       nm::as_stream(stream), reinterpret_cast<const float4*>(raw), z_vals, rays_d, R, S, white_bkg, noise,
                       rgb, disp, acc, weights, depth);
    return nm::check_launch("composite_kernel");
}

int nm_sample_pdf(const float* bins, const float* weights, int64_t R, int B, const float* u, int N, float* samples,
                  nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (bins && weights && u && samples), "nm_sample_pdf: null pointer");
    NM_REQUIRE(R >= 0 && B >= 2 && N >= 1, "nm_sample_pdf: bad sizes");
    const size_t lds = (size_t)kRayWavesPerBlock * (2 * B + N) * sizeof(float);
    NM_REQUIRE(lds <= 64 * 1024, "nm_sample_pdf: B=%d N=%d exceed the per-wave LDS budget", B, N);
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(sample_pdf_kernel<0>, dim3(grid_for(R, kRayWavesPerBlock)), dim3(64 * kRayWavesPerBlock), lds,
                       nm::as_stream(stream), bins, weights, R, B, u, N, 0, samples);
    return nm::check_launch("sample_pdf_kernel<0>");
}

int nm_importance_z(const float* z_vals, const float* weights, int64_t R, int S, const float* u, int N,
                    int including_old, float* z_out, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (z_vals && weights && u && z_out), "nm_importance_z: null pointer");
    NM_REQUIRE(R >= 0 && S >= 3 && N >= 1, "nm_importance_z: bad sizes S=%d N=%d", S, N);
    const int B = S - 1;
    const size_t lds = (size_t)kRayWavesPerBlock * (2 * B + N + S) * sizeof(float);
    NM_REQUIRE(lds <= 64 * 1024, "nm_importance_z: S=%d N=%d exceed the per-wave LDS budget", S, N);
    NM_REQUIRE(!including_old || N <= 64 * kMaxImportanceChunks, "nm_importance_z: N=%d importance samples exceed %d when merged with the old ones", N,
               64 * kMaxImportanceChunks);
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(sample_pdf_kernel<1>, dim3(grid_for(R, kRayWavesPerBlock)), dim3(64 * kRayWavesPerBlock), lds,
                       nm::as_stream(stream), z_vals, weights, R, B, u, N, including_old, z_out);
    return nm::check_launch("sample_pdf_kernel<1>");
}

int nm_importance_from_raw(const float* raw, const float* z_vals, const float* rays_d, int64_t R, int S, const float* u, int N, float* z_out,
                           float* weights_out, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (raw && z_vals && rays_d && u && z_out), "nm_importance_from_raw: null pointer");
    NM_REQUIRE(R >= 0 && S >= 3 && N >= 1, "nm_importance_from_raw: bad sizes S=%d N=%d", S, N);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(raw) & 15) == 0, "nm_importance_from_raw: raw must be 16-byte aligned");
    const int B = S - 1;
    const size_t lds = (size_t)kRayWavesPerBlock * (2 * B + N + 2 * S) * sizeof(float);
    NM_REQUIRE(lds <= 64 * 1024, "nm_importance_from_raw: S=%d N=%d exceed the per-wave LDS budget", S, N);
    NM_REQUIRE(N <= 64 * kMaxImportanceChunks, "nm_importance_from_raw: N=%d importance samples exceed %d", N, 64 * kMaxImportanceChunks);
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(sample_pdf_kernel<2>, dim3(grid_for(R, kRayWavesPerBlock)), dim3(64 * kRayWavesPerBlock), lds, nm::as_stream(stream), z_vals,
                       (const float*)nullptr, R, B, u, N, 1, z_out, reinterpret_cast<const float4*>(raw), rays_d, weights_out);
    return nm::check_launch("sample_pdf_kernel<2>");
}

int nm_merge_composite_lists(int k, const float* const* z, const float* const* raw, const int32_t* const* rows, const int* S, int64_t R,
                             const float* rays_d, int white_bkg, float* rgb, float* depth, float* acc, nm_stream_t stream) {
    NM_REQUIRE(k >= 1 && k <= kMaxMergeLists && z && raw && S, "nm_merge_composite_lists: 1 <= k <= %d lists (k=%d)", kMaxMergeLists, k);
    NM_REQUIRE(R == 0 || (rays_d && rgb && depth && acc), "nm_merge_composite_lists: null pointer");
    MergeLists L;
    L.k = k;
    L.S_total = 0;
    for (int l = 0; l < kMaxMergeLists; ++l) {
        const bool on = l < k;
        L.z[l] = on ? z[l] : nullptr;
        L.raw[l] = on ? reinterpret_cast<const float4*>(raw[l]) : nullptr;
        L.rows[l] = (on && rows) ? rows[l] : nullptr;
        L.S[l] = on ? S[l] : 0;
        if (on) {
            NM_REQUIRE(R == 0 || (z[l] && raw[l]), "nm_merge_composite_lists: list %d is null", l);
            NM_REQUIRE(S[l] >= 1, "nm_merge_composite_lists: list %d is empty", l);
            NM_REQUIRE((reinterpret_cast<uintptr_t>(raw[l]) & 15) == 0, "nm_merge_composite_lists: raw arrays must be 16-byte aligned");
            L.S_total += S[l];
        }
    }
    if (R == 0) return NM_OK;
    const size_t per_wave = (size_t)L.S_total * 12;
    NM_REQUIRE(per_wave <= 64 * 1024 && L.S_total < 65536, "nm_merge_composite_lists: %d merged samples exceed the per-wave LDS budget", L.S_total);
    int wpb = (int)((64 * 1024) / per_wave);
    if (wpb > 4) wpb = 4;
    hipLaunchKernelGGL(merge_composite_kernel, dim3(grid_for(R, wpb)), dim3(64 * wpb), per_wave * wpb, nm::as_stream(stream), L, R, rays_d, white_bkg, rgb,
                       depth, acc);
    return nm::check_launch("merge_composite_kernel");
}

int nm_merged_intervals(int k, const float* const* z, const int* S, int64_t R, float* const* dz, nm_stream_t stream) {
    NM_REQUIRE(k >= 1 && k <= kMaxMergeLists && z && S && dz, "nm_merged_intervals: 1 <= k <= %d lists (k=%d)", kMaxMergeLists, k);
    IntervalLists L;
    L.k = k;
    L.S_total = 0;
    for (int l = 0; l < kMaxMergeLists; ++l) {
        const bool on = l < k;
        L.z[l] = on ? z[l] : nullptr;
        L.dz[l] = on ? dz[l] : nullptr;
        L.S[l] = on ? S[l] : 0;
        if (on) {
            NM_REQUIRE(S[l] >= 1, "nm_merged_intervals: list %d is empty", l);
            NM_REQUIRE(R == 0 || (z[l] && dz[l]), "nm_merged_intervals: list %d is null", l);
            L.S_total += S[l];
        }
    }
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(merged_intervals_kernel, dim3(grid_for(R * L.S_total, 256)), dim3(256), 0, nm::as_stream(stream), L, R);
    return nm::check_launch("merged_intervals_kernel");
}

int nm_merge_sorted(const float* za, const float* rawa, int Sa, const float* zb, const float* rawb, int Sb, int64_t R,
                    float* z_out, float* raw_out, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (za && rawa && zb && rawb && z_out && raw_out), "nm_merge_sorted: null pointer");
    NM_REQUIRE(R >= 0 && Sa >= 1 && Sb >= 1, "nm_merge_sorted: bad sizes");
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(rawa) | reinterpret_cast<uintptr_t>(rawb) | reinterpret_cast<uintptr_t>(raw_out)) & 15) == 0,
               "nm_merge_sorted: raw arrays must be 16-byte aligned");
    const size_t lds = (size_t)kRayWavesPerBlock * (Sa + Sb) * sizeof(float);
    NM_REQUIRE(lds <= 64 * 1024, "nm_merge_sorted: Sa+Sb=%d exceeds the per-wave LDS budget", Sa + Sb);
    if (R == 0) return NM_OK;
    hipLaunchKernelGGL(merge_sorted_kernel, dim3(grid_for(R, kRayWavesPerBlock)), dim3(64 * kRayWavesPerBlock), lds,
                       nm::as_stream(stream), za, reinterpret_cast<const float4*>(rawa), Sa, zb,
                       reinterpret_cast<const float4*>(rawb), Sb, R, z_out, reinterpret_cast<float4*>(raw_out));
    return nm::check_launch("merge_sorted_kernel");
}

int nm_gather_rows(const float* src, const int32_t* idx, const int32_t* n_dev, int64_t n_max, int width, float* dst,
                   nm_stream_t stream) {
    NM_REQUIRE(n_max == 0 || (src && idx && dst), "nm_gather_rows: null pointer");
    NM_REQUIRE(n_max >= 0 && width >= 1, "nm_gather_rows: bad sizes");
    if (n_max == 0) return NM_OK;
    hipLaunchKernelGGL(rows_kernel<false>, dim3(grid_for(n_max * width, 256)), dim3(256), 0, nm::as_stream(stream), src,
                       idx, n_dev, n_max, width, dst);
    return nm::check_launch("rows_kernel<gather>");
}

int nm_scatter_rows(const float* src, const int32_t* idx, const int32_t* n_dev, int64_t n_max, int width, float* dst,
                    nm_stream_t stream) {
    NM_REQUIRE(n_max == 0 || (src && idx && dst), "nm_scatter_rows: null pointer");
    NM_REQUIRE(n_max >= 0 && width >= 1, "nm_scatter_rows: bad sizes");
    if (n_max == 0) return NM_OK;
    hipLaunchKernelGGL(rows_kernel<true>, dim3(grid_for(n_max * width, 256)), dim3(256), 0, nm::as_stream(stream), src,
                       idx, n_dev, n_max, width, dst);
    return nm::check_launch("rows_kernel<scatter>");
}

}  // extern "C"
