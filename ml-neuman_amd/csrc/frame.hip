// Frame ingress / egress around the ray-march path, HBM-bound element kernels:
//   * a1 ray generation on the device  -- reference utils/ray_utils.py:23-38, geometry/pcd_projector.py:85-120, 209-227
//   * float frame -> uint8 pixels       -- what imageio.imsave does to the renderers' f32 output (render_test_views.py:83-88)
//   * PSNR of two uint8 frames          -- skimage.metrics.peak_signal_noise_ratio as called at render_test_views.py:35
#include "common.h"

namespace {

struct CamParams {
    double kinv[9];   // inverse intrinsic matrix, row-major
    double c2w[16];   // camera-to-world, row-major
};

// One thread per ray.  All arithmetic in f64 exactly as the reference's numpy chain:
//   cam = Kinv [x, y, 1] (* depth 1); world = c2w [cam, 1]; world /= world_w
//   mode 0 (shot_all_rays): d = world - centre, d /= |d| in f64, cast to f32
//   mode 1 (shot_rays, f64 cam2world): world is cast to f32 first, then the same in f64 (numpy promotes to the centre's type)
//   mode 2 (shot_rays, f32 cam2world -- the reference's CameraPose builds its matrix from f32 quaternions): world is cast
//           to f32 and the subtraction, the norm ((d0^2 + d1^2) + d2^2, sqrt) and the division all run in f32
// The products are summed left to right without contraction (the library is built with -ffp-contract=off); BLAS may
// contract differently inside its dgemm, which moves a result by at most one f64 ulp -- invisible after the f32 cast
// except on a rounding tie.
__device__ __forceinline__ void shoot(double x, double y, const double* kinv, const double* c2w, int mode, float* __restrict__ origin,
                                      float* __restrict__ direction) {
    double cam[3], w4[4];
#pragma unroll
    for (int r = 0; r < 3; ++r) cam[r] = kinv[3 * r] * x + kinv[3 * r + 1] * y + kinv[3 * r + 2];
#pragma unroll
    for (int r = 0; r < 4; ++r) w4[r] = c2w[4 * r] * cam[0] + c2w[4 * r + 1] * cam[1] + c2w[4 * r + 2] * cam[2] + c2w[4 * r + 3];
    if (mode == 2) {
        float f[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) f[r] = (float)(w4[r] / w4[3]) - (float)c2w[4 * r + 3];
        const float norm = sqrtf((f[0] * f[0] + f[1] * f[1]) + f[2] * f[2]);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            direction[r] = f[r] / norm;
            origin[r] = (float)c2w[4 * r + 3];
        }
        return;
    }
    double d[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        double p = w4[r] / w4[3];
        if (mode == 1) p = (double)(float)p;
        d[r] = p - c2w[4 * r + 3];
    }
    const double norm = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        direction[r] = (float)(d[r] / norm);                  // numpy divides by the norm, it does not multiply by a reciprocal
        origin[r] = (float)c2w[4 * r + 3];
    }
}

__global__ __launch_bounds__(256) void shot_rays_kernel(const int32_t* __restrict__ xy, int64_t n, int width, int mode, CamParams P,
                                                        float* __restrict__ origin, float* __restrict__ direction) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double x, y;
    if (xy) {
        x = (double)xy[2 * i];
        y = (double)xy[2 * i + 1];
    } else {                                                   // full grid, row-major: np.meshgrid(linspace(0,w-1), linspace(0,h-1))
        y = (double)(i / width);
        x = (double)(i - (i / width) * width);
    }
    shoot(x, y, P.kinv, P.c2w, mode, origin + 3 * i, direction + 3 * i);
}

// The same per ray with the camera picked by cam_id[i] from a device table of [n_cams][25] doubles (Kinv 3x3, then cam2world 4x4):
// a training batch draws its rays from many captures at once (datasets/background_rays.py:47-101 loops over them on the host)
__global__ __launch_bounds__(256) void shot_rays_cams_kernel(const int32_t* __restrict__ xy, const int32_t* __restrict__ cam_id, int64_t n,
                                                             int mode, const double* __restrict__ cams, int n_cams,
                                                             float* __restrict__ origin, float* __restrict__ direction) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int c = cam_id[i];
    c = c < 0 ? 0 : (c >= n_cams ? n_cams - 1 : c);
    double P[25];
#pragma unroll
    for (int k = 0; k < 25; ++k) P[k] = cams[(int64_t)c * 25 + k];
    shoot((double)xy[2 * i], (double)xy[2 * i + 1], P, P + 9, mode, origin + 3 * i, direction + 3 * i);
}

// imageio (v2 `image_as_uint`, bitdepth 8) on a float image: clip to [0, 1], then uint8(x * 255 + 0.499999999) in f64
__global__ __launch_bounds__(256) void to_uint8_kernel(const float* __restrict__ src, int64_t n, uint8_t* __restrict__ dst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double v = (double)src[i];
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);                   // NaN falls through both comparisons and converts to 0
    dst[i] = (uint8_t)(int)(v * 255.0 + 0.499999999);
}

// sum of squared differences of two uint8 arrays, exact in integers: per-block partial sums then one atomic each
__global__ __launch_bounds__(256) void ssd_u8_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int64_t n,
                                                     unsigned long long* __restrict__ out) {
    unsigned long long s = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int d = (int)a[i] - (int)b[i];
        s += (unsigned long long)(d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ unsigned long long part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, part[0] + part[1] + part[2] + part[3]);
}

// skimage.metrics.structural_similarity(pred, gt, multichannel=True) for uint8 images, its defaults (render_test_views.py:33):
// 7 x 7 uniform window, K1 = 0.01, K2 = 0.03, data range 255, sample covariance (x 49/48), the mean of the SSIM map over the image
// cropped by 3 pixels per side, then the mean over channels.  The five window sums are exact integers; the map is f64 as skimage
// computes it.  One thread per (pixel of the cropped region, channel); per-block partial sums, then one block adds them in order.
constexpr int kSsimWin = 7, kSsimPad = 3;
__global__ __launch_bounds__(256) void ssim_partial_kernel(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, int H, int W, int C,
                                                           double* __restrict__ partial) {
    const int hh = H - 2 * kSsimPad, ww = W - 2 * kSsimPad;
    const int64_t n = (int64_t)hh * ww * C;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int x = (int)((i / C) % ww) + kSsimPad, y = (int)(i / ((int64_t)C * ww)) + kSsimPad;
        int sx = 0, sy = 0, sxx = 0, syy = 0, sxy = 0;
        for (int dy = -kSsimPad; dy <= kSsimPad; ++dy)
            for (int dx = -kSsimPad; dx <= kSsimPad; ++dx) {
                const int64_t o = ((int64_t)(y + dy) * W + (x + dx)) * C + c;
                const int p = a[o], q = b[o];
                sx += p; sy += q; sxx += p * p; syy += q * q; sxy += p * q;
            }
        const double NP = kSsimWin * kSsimWin, cov = NP / (NP - 1.0);
        const double ux = sx / NP, uy = sy / NP, uxx = sxx / NP, uyy = syy / NP, uxy = sxy / NP;
        const double vx = cov * (uxx - ux * ux), vy = cov * (uyy - uy * uy), vxy = cov * (uxy - ux * uy);
        const double C1 = (0.01 * 255.0) * (0.01 * 255.0), C2 = (0.03 * 255.0) * (0.03 * 255.0);
        s += ((2.0 * ux * uy + C1) * (2.0 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__global__ void ssim_final_kernel(const double* __restrict__ partial, int blocks, double n, double* __restrict__ out) {
    double s = 0.0;
    for (int i = 0; i < blocks; ++i) s += partial[i];
    out[0] = s / n;
}

}  // namespace

extern "C" {

int nm_shot_rays(const int32_t* xy, int64_t n, int width, int mode, const double* inv_intrinsic, const double* cam2world,
                 float* origin, float* direction, nm_stream_t stream) {
    NM_REQUIRE(n >= 0, "nm_shot_rays: negative n");
    NM_REQUIRE(mode >= 0 && mode <= 2, "nm_shot_rays: mode %d (0 = shot_all_rays, 1 / 2 = shot_rays with an f64 / f32 pose)", mode);
    NM_REQUIRE(inv_intrinsic && cam2world, "nm_shot_rays: null camera matrices");
    NM_REQUIRE(xy || width >= 1, "nm_shot_rays: the full-grid form needs the image width");
    NM_REQUIRE(n == 0 || (origin && direction), "nm_shot_rays: null output");
    if (n == 0) return NM_OK;
    CamParams P;
    for (int i = 0; i < 9; ++i) P.kinv[i] = inv_intrinsic[i];
    for (int i = 0; i < 16; ++i) P.c2w[i] = cam2world[i];
    hipLaunchKernelGGL(shot_rays_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nm::as_stream(stream), xy, n, width, mode, P,
                       origin, direction);
    return nm::check_launch("shot_rays_kernel");
}

int nm_shot_rays_cams(const int32_t* xy, const int32_t* cam_id, int64_t n, int mode, const double* cams, int n_cams, float* origin,
                      float* direction, nm_stream_t stream) {
    NM_REQUIRE(n >= 0, "nm_shot_rays_cams: negative n");
    NM_REQUIRE(mode == 1 || mode == 2, "nm_shot_rays_cams: mode %d (1 / 2 = shot_rays with an f64 / f32 pose)", mode);
    NM_REQUIRE(cams && n_cams >= 1, "nm_shot_rays_cams: no camera table");
    NM_REQUIRE(n == 0 || (xy && cam_id && origin && direction), "nm_shot_rays_cams: null pointer");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(shot_rays_cams_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nm::as_stream(stream), xy, cam_id, n, mode,
                       cams, n_cams, origin, direction);
    return nm::check_launch("shot_rays_cams_kernel");
}

int nm_frame_to_uint8(const float* src, int64_t n, uint8_t* dst, nm_stream_t stream) {
    NM_REQUIRE(n >= 0, "nm_frame_to_uint8: negative n");
    NM_REQUIRE(n == 0 || (src && dst), "nm_frame_to_uint8: null pointer");
    if (n == 0) return NM_OK;
    hipLaunchKernelGGL(to_uint8_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nm::as_stream(stream), src, n, dst);
    return nm::check_launch("to_uint8_kernel");
}

int nm_ssd_u8(const uint8_t* a, const uint8_t* b, int64_t n, uint64_t* ssd, nm_stream_t stream) {
    NM_REQUIRE(n >= 0, "nm_ssd_u8: negative n");
    NM_REQUIRE(ssd && (n == 0 || (a && b)), "nm_ssd_u8: null pointer");
    if (int rc = nm::check_hip(hipMemsetAsync(ssd, 0, sizeof(uint64_t), nm::as_stream(stream)), "nm_ssd_u8: memset")) return rc;
    if (n == 0) return NM_OK;
    const int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    hipLaunchKernelGGL(ssd_u8_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, nm::as_stream(stream), a, b, n,
                       reinterpret_cast<unsigned long long*>(ssd));
    return nm::check_launch("ssd_u8_kernel");
}

int nm_ssim_u8(const uint8_t* a, const uint8_t* b, int H, int W, int C, double* ssim, double* workspace, nm_stream_t stream) {
    NM_REQUIRE(H >= kSsimWin && W >= kSsimWin && C >= 1, "nm_ssim_u8: the image must be at least 7 x 7 (H=%d W=%d C=%d)", H, W, C);
    NM_REQUIRE(a && b && ssim && workspace, "nm_ssim_u8: null pointer");
    hipStream_t st = nm::as_stream(stream);
    const int64_t n = (int64_t)(H - 2 * kSsimPad) * (W - 2 * kSsimPad) * C;
    const int blocks = (int)((n + 255) / 256 < NM_SSIM_WORKSPACE_DOUBLES ? (n + 255) / 256 : NM_SSIM_WORKSPACE_DOUBLES);
    hipLaunchKernelGGL(ssim_partial_kernel, dim3(blocks), dim3(256), 0, st, a, b, H, W, C, workspace);
    hipLaunchKernelGGL(ssim_final_kernel, dim3(1), dim3(1), 0, st, workspace, blocks, (double)n, ssim);
    return nm::check_launch("ssim kernels");
}

}  // extern "C"
