// The scalar regularisers of the human trainer's loss (reference trainers/human_nerf_trainer.py:280-380) as single passes: value AND gradient of a term in
// one kernel + one deterministic finalisation, where torch's elementwise algebra takes 10-25 launches per term and as many again in autograd's backward pass.
//   nm_loss_bimodal    mean(-log(e^-|y| + e^-|1 - y|) + HARD_SURFACE_OFFSET), y = clamp(x, 0, 1)      the sharp-edge / hard-surface priors (:368-379)
//   nm_loss_pair_mse   mse of sigmoid(rgb) (colour range, :280-290) or of tanh(relu(sigma)) (symmetry, :292-304) between two raw network outputs
//   nm_loss_shape      the SMPL shape prior (:305-343): occupancy 1 inside the canonical body for the rays' samples and for dummy points, 0 outside
//                      weighted by the distance from the surface -- three masked means
// Every kernel writes the gradient of ITS term with respect to its inputs beside the value (the term enters the total with weight 1; autograd's backward
// multiplies by whatever arrives).  Sums: per-workgroup partials in float64, combined in workgroup order by one workgroup -- deterministic.
#include <math.h>

#include "common.h"

namespace {

constexpr int kLossThreads = 256;
constexpr int kLossMaxBlocks = 1024;

__device__ __forceinline__ double block_sum(double v, double* sh) {     // sum over the 256 threads, result valid in thread 0
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    __syncthreads();
    return r;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }
// torch.relu / torch.clamp hand a NaN on (fmaxf would turn it into 0): a NaN density must reach the loss, where the trainer's NaN guard
// (human_nerf_trainer.py:476-478) sees it
__device__ __forceinline__ float relu_nan(float x) { return x <= 0.f ? 0.f : x; }
__device__ __forceinline__ float clamp01_nan(float x) { return x < 0.f ? 0.f : (x > 1.f ? 1.f : x); }

__global__ __launch_bounds__(kLossThreads) void bimodal_kernel(const float* __restrict__ x, int64_t n, int clamp01, float offset, double* __restrict__ partial,
                                                               float* __restrict__ dx) {
    __shared__ double sh[4];
    double acc = 0.0;
    const float inv_n = 1.f / (float)n;
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLossThreads) {
        const float xv = x[i];
        const float y = clamp01 ? clamp01_nan(xv) : xv;
        const float u = expf(-fabsf(y)), v = expf(-fabsf(1.f - y));
        acc += (double)(-logf(u + v) + offset);
        const float pass = clamp01 ? ((xv >= 0.f && xv <= 1.f) ? 1.f : 0.f) : 1.f;      // torch.clamp's backward: inside the closed interval
        dx[i] = pass * (u * sgn(y) - v * sgn(1.f - y)) / (u + v) * inv_n;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// mode 0: columns 0..2 through a sigmoid (mean over 3 n); mode 1: column 3 through tanh(relu(.)) (mean over n).  da / db [n][4] get the whole row.
__global__ __launch_bounds__(kLossThreads) void pair_mse_kernel(int mode, const float4* __restrict__ a, const float4* __restrict__ b, int64_t n, float scale,
                                                                double* __restrict__ partial, float4* __restrict__ da, float4* __restrict__ db) {
    __shared__ double sh[4];
    double acc = 0.0;
    const float k = scale * 2.f / (float)(mode == 0 ? 3 * n : n);
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kLossThreads) {
        const float4 av = a[i], bv = b[i];
        float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
        if (mode == 0) {
            const float ax[3] = {av.x, av.y, av.z}, bx[3] = {bv.x, bv.y, bv.z};
            float g1[3], g2[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float sa = 1.f / (1.f + expf(-ax[c])), sb = 1.f / (1.f + expf(-bx[c]));
                const float d = sa - sb;
                acc += (double)(d * d);
                g1[c] = k * d * sa * (1.f - sa);
                g2[c] = -k * d * sb * (1.f - sb);
            }
            ga = make_float4(g1[0], g1[1], g1[2], 0.f);
            gb = make_float4(g2[0], g2[1], g2[2], 0.f);
        } else {
            const float ta = tanhf(relu_nan(av.w)), tb = tanhf(relu_nan(bv.w));
            const float d = ta - tb;
            acc += (double)(d * d);
            ga.w = av.w > 0.f ? k * d * (1.f - ta * ta) : 0.f;
            gb.w = bv.w > 0.f ? -k * d * (1.f - tb * tb) : 0.f;
        }
        da[i] = ga;
        db[i] = gb;
    }
    const double s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

// the sum of `blocks` partials with stride `stride`, by one wave: lane l adds partials l, l + 64, ... in order, the 64 lane sums are added in a fixed
// tree -- the same order on every run
__device__ __forceinline__ double wave_total(const double* __restrict__ partial, int blocks, int stride) {
    double s = 0.0;
    for (int i = threadIdx.x; i < blocks; i += 64) s += partial[(int64_t)i * stride];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_down(s, off, 64);
    return __shfl(s, 0, 64);
}

// out[0] = scale * (sum of the partials) / denom
__global__ __launch_bounds__(64) void loss_finish_kernel(const double* __restrict__ partial, int blocks, double scale_over_denom, float* __restrict__ out) {
    const double s = wave_total(partial, blocks, 1);
    if (threadIdx.x == 0) out[0] = (float)(s * scale_over_denom);
}

// ---- shape prior: sums S1 (pred inside), S2 (dummy inside), S3 (dummy outside, weighted) and the three counts
__global__ __launch_bounds__(kLossThreads) void shape_sums_kernel(const float4* __restrict__ pred, const float* __restrict__ dist_h, int64_t nh,
                                                                  const float4* __restrict__ dummy, const float* __restrict__ dist_d, int64_t nd, float factor,
                                                                  float exponent, double* __restrict__ partial /*[blocks][6]*/) {
    __shared__ double sh[4];
    double s1 = 0, s2 = 0, s3 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < nh; i += (int64_t)gridDim.x * kLossThreads) {
        if (dist_h[i] < 0.f) {
            const float e = expf(-relu_nan(pred[i].w));              // 1 - occupancy
            s1 += (double)(e * e);
            c1 += 1.0;
        }
    }
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < nd; i += (int64_t)gridDim.x * kLossThreads) {
        const float dd = dist_d[i];
        const float e = expf(-relu_nan(dummy[i].w));
        if (dd < 0.f) { s2 += (double)(e * e); c2 += 1.0; }
        if (dd > 0.f) { s3 += (double)fabsf((1.f - e) * powf(fabsf(dd) * factor, exponent)); c3 += 1.0; }
    }
    double v[6] = {s1, s2, s3, c1, c2, c3};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double r = block_sum(v[k], sh);
        if (threadIdx.x == 0) partial[blockIdx.x * 6 + k] = r;
    }
}
// totals [6] and the value: out[0] = w_smpl S1 / max(C1, 1) + w_dummy (S2 / max(C2, 1) + S3 / max(C3, 1)); norm[3] = the three weights / counts
__global__ __launch_bounds__(64) void shape_finish_kernel(const double* __restrict__ partial, int blocks, float w_smpl, float w_dummy, float* __restrict__ out,
                                                          float* __restrict__ norm) {
    double t[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) t[k] = wave_total(partial + k, blocks, 6);
    if (threadIdx.x == 0) {
        const double n1 = t[3] > 1.0 ? t[3] : 1.0, n2 = t[4] > 1.0 ? t[4] : 1.0, n3 = t[5] > 1.0 ? t[5] : 1.0;
        out[0] = (float)(w_smpl * t[0] / n1 + w_dummy * (t[1] / n2 + t[2] / n3));
        norm[0] = (float)(w_smpl / n1); norm[1] = (float)(w_dummy / n2); norm[2] = (float)(w_dummy / n3);
    }
}
__global__ __launch_bounds__(kLossThreads) void shape_grad_kernel(const float4* __restrict__ pred, const float* __restrict__ dist_h, int64_t nh,
                                                                  const float4* __restrict__ dummy, const float* __restrict__ dist_d, int64_t nd, float factor,
                                                                  float exponent, const float* __restrict__ norm, float4* __restrict__ d_pred,
                                                                  float4* __restrict__ d_dummy) {
    const float k1 = norm[0], k2 = norm[1], k3 = norm[2];
    for (int64_t i = (int64_t)blockIdx.x * kLossThreads + threadIdx.x; i < nh + nd; i += (int64_t)gridDim.x * kLossThreads) {
        if (i < nh) {
            const float s = pred[i].w;
            float g = 0.f;
            if (dist_h[i] < 0.f && s > 0.f) { const float e = expf(-s); g = -2.f * k1 * e * e; }
            d_pred[i] = make_float4(0.f, 0.f, 0.f, g);
        } else {
            const int64_t j = i - nh;
            const float s = dummy[j].w, dd = dist_d[j];
            float g = 0.f;
            if (s > 0.f) {
                const float e = expf(-s);
                if (dd < 0.f) g = -2.f * k2 * e * e;
                if (dd > 0.f) { const float f = powf(fabsf(dd) * factor, exponent); g = (1.f - e) * f != 0.f ? k3 * f * e : 0.f; }
            }
            d_dummy[j] = make_float4(0.f, 0.f, 0.f, g);
        }
    }
}

inline int loss_blocks(int64_t n) {
    int64_t b = (n + kLossThreads - 1) / kLossThreads;
    return (int)(b < 1 ? 1 : (b > kLossMaxBlocks ? kLossMaxBlocks : b));
}

}  // namespace

extern "C" {

int64_t nm_loss_workspace_doubles(void) { return (int64_t)kLossMaxBlocks * 6; }

int nm_loss_bimodal(const float* x, int64_t n, int clamp01, float offset, float* loss, float* dx, double* workspace, nm_stream_t stream) {
    NM_REQUIRE(n >= 1 && x && loss && dx && workspace, "nm_loss_bimodal: bad arguments");
    hipStream_t st = nm::as_stream(stream);
    const int blocks = loss_blocks(n);
    hipLaunchKernelGGL(bimodal_kernel, dim3(blocks), dim3(kLossThreads), 0, st, x, n, clamp01, offset, workspace, dx);
    if (int rc = nm::check_launch("bimodal_kernel")) return rc;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, 1.0 / (double)n, loss);
    return nm::check_launch("loss_finish_kernel");
}

int nm_loss_pair_mse(int mode, const float* a_raw, const float* b_raw, int64_t n, float scale, float* loss, float* da_raw, float* db_raw, double* workspace,
                     nm_stream_t stream) {
    NM_REQUIRE((mode == 0 || mode == 1) && n >= 1 && a_raw && b_raw && loss && da_raw && db_raw && workspace, "nm_loss_pair_mse: bad arguments");
    NM_REQUIRE((((uintptr_t)a_raw | (uintptr_t)b_raw | (uintptr_t)da_raw | (uintptr_t)db_raw) & 15) == 0, "nm_loss_pair_mse: rows must be 16-byte aligned");
    hipStream_t st = nm::as_stream(stream);
    const int blocks = loss_blocks(n);
    hipLaunchKernelGGL(pair_mse_kernel, dim3(blocks), dim3(kLossThreads), 0, st, mode, reinterpret_cast<const float4*>(a_raw), reinterpret_cast<const float4*>(b_raw), n,
                       scale, workspace, reinterpret_cast<float4*>(da_raw), reinterpret_cast<float4*>(db_raw));
    if (int rc = nm::check_launch("pair_mse_kernel")) return rc;
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, (double)scale / (double)(mode == 0 ? 3 * n : n), loss);
    return nm::check_launch("loss_finish_kernel");
}

int nm_loss_shape(const float* pred_raw, const float* dist_h, int64_t nh, const float* dummy_raw, const float* dist_d, int64_t nd, float w_smpl, float w_dummy,
                  float outside_factor, float exponent, float* loss, float* d_pred_raw, float* d_dummy_raw, double* workspace, float* norm3, nm_stream_t stream) {
    NM_REQUIRE(nh >= 1 && nd >= 0 && pred_raw && dist_h && loss && d_pred_raw && workspace && norm3, "nm_loss_shape: bad arguments");
    NM_REQUIRE(nd == 0 || (dummy_raw && dist_d && d_dummy_raw), "nm_loss_shape: dummy points without their arrays");
    NM_REQUIRE((((uintptr_t)pred_raw | (uintptr_t)dummy_raw | (uintptr_t)d_pred_raw | (uintptr_t)d_dummy_raw) & 15) == 0, "nm_loss_shape: rows must be 16-byte aligned");
    hipStream_t st = nm::as_stream(stream);
    const int blocks = loss_blocks(nh > nd ? nh : nd);
    hipLaunchKernelGGL(shape_sums_kernel, dim3(blocks), dim3(kLossThreads), 0, st, reinterpret_cast<const float4*>(pred_raw), dist_h, nh,
                       reinterpret_cast<const float4*>(dummy_raw), dist_d, nd, outside_factor, exponent, workspace);
    if (int rc = nm::check_launch("shape_sums_kernel")) return rc;
    hipLaunchKernelGGL(shape_finish_kernel, dim3(1), dim3(64), 0, st, workspace, blocks, w_smpl, w_dummy, loss, norm3);
    if (int rc = nm::check_launch("shape_finish_kernel")) return rc;
    hipLaunchKernelGGL(shape_grad_kernel, dim3(loss_blocks(nh + nd)), dim3(kLossThreads), 0, st, reinterpret_cast<const float4*>(pred_raw), dist_h, nh,
                       reinterpret_cast<const float4*>(dummy_raw), dist_d, nd, outside_factor, exponent, norm3, reinterpret_cast<float4*>(d_pred_raw),
                       reinterpret_cast<float4*>(d_dummy_raw));
    return nm::check_launch("shape_grad_kernel");
}

}  // extern "C"
