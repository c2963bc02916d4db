// NM_PREC_FP32: exact-f32 evaluation of the same PE + NeRF MLP on the vector ALU (fmaf chains in
// ascending k, f32 accumulate) -- an independent implementation (natural weight layout, no MFMA, no
// split, no k-slot permutation) used to validate the MFMA kernel at sizes the CPU oracle cannot reach.
// Follows reference models/vanilla.py:82-92, 120-152, 162-166.  Not a performance path.
#include "common.h"
#include "mlp_launch.h"

namespace {

constexpr int kRefTile = 64;   // samples per workgroup (one per lane)
constexpr int kRefThreads = 256;

struct RefArgs {
    const float* wt;          // transposed weights, layer l at wt + off[l], [K_l][N_l] row-major
    const float* bias;        // natural biases, layer l at bias + boff[l]
    int off[12];
    int boff[12];
    const float* petab;
    const float* pts; const float* dirs;
    const float* origin; const float* direction; const float* z;
    float* out; float* dbg;
    int64_t n; int S; int in_mode; int stop_stage; float sigma_scale;
    int pe_kind, pos_nfreq, dir_nfreq;
    int plain_head;           // use_viewdirs=False: layer 10 is output_linear (256 -> 4), layers 8, 9, 11 do not exist
};
// layer ids: 0..7 pts_linears, 8 views, 9 feature, 10 alpha, 11 rgb   (reference state_dict order)

__device__ __forceinline__ float ref_pe_feature(int p, float x0, float x1, float x2, int kind, int nfreq, const float* __restrict__ tab) {
    if (p < 3) return p == 0 ? x0 : (p == 1 ? x1 : x2);
    const int m = p - 3;
    if (m >= 6 * nfreq) return 0.f;
    if (kind == NM_PE_POSENC) {
        const int b = m / 6, r = m - 6 * b;
        const int dim = r >= 3 ? r - 3 : r;
        const float xv = dim == 0 ? x0 : (dim == 1 ? x1 : x2);
        const float a = xv * tab[b];
        return r >= 3 ? cosf(a) : sinf(a);
    }
    const int n3 = 3 * nfreq;
    const bool is_cos = m >= n3;
    const float* b = tab + 3 * (is_cos ? m - n3 : m);
    const float a = fmaf(x2, b[2], fmaf(x1, b[1], x0 * b[0]));
    return is_cos ? cosf(a) : sinf(a);
}

template <int NPT>
__device__ __forceinline__ void dense(float (&acc)[NPT], const float* __restrict__ wt, int N, int n0, const float* act, int K, int s) {
    for (int k = 0; k < K; ++k) {
        const float a = act[k * kRefTile + s];
        const float* wr = wt + (int64_t)k * N + n0;     // wave-uniform address -> scalar loads
#pragma unroll
        for (int i = 0; i < NPT; ++i) acc[i] = fmaf(a, wr[i], acc[i]);
    }
}

__global__ __launch_bounds__(kRefThreads) void nerf_mlp_ref_kernel(const RefArgs a) {
    __shared__ float pe[64 * kRefTile];
    __shared__ float dpe[32 * kRefTile];
    __shared__ float h[256 * kRefTile];
    const int tid = threadIdx.x, s = tid & 63;
    const int grp = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t ntiles = (a.n + kRefTile - 1) / kRefTile;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kRefTile;
        int64_t i = base + s;
        const bool live = i < a.n;
        if (!live) i = a.n - 1;
        float x0, x1, x2, d0, d1, d2;
        if (a.in_mode == 0) {
            x0 = a.pts[i * 3]; x1 = a.pts[i * 3 + 1]; x2 = a.pts[i * 3 + 2];
            d0 = a.dirs[i * 3]; d1 = a.dirs[i * 3 + 1]; d2 = a.dirs[i * 3 + 2];
        } else {
            const int64_t r = i / a.S;
            d0 = a.direction[r * 3]; d1 = a.direction[r * 3 + 1]; d2 = a.direction[r * 3 + 2];
            const float zz = a.z[i];
            x0 = a.origin[r * 3] + d0 * zz; x1 = a.origin[r * 3 + 1] + d1 * zz; x2 = a.origin[r * 3 + 2] + d2 * zz;
        }
        for (int p = grp; p < 64; p += 4) pe[p * kRefTile + s] = ref_pe_feature(p, x0, x1, x2, a.pe_kind, a.pos_nfreq, a.petab);
        for (int p = grp; p < 32; p += 4) dpe[p * kRefTile + s] = ref_pe_feature(p, d0, d1, d2, a.pe_kind, a.dir_nfreq, a.petab + 96);
        __syncthreads();
        if (a.stop_stage == -1) {
            if (live && grp == 0) for (int p = 0; p < 64; ++p) a.dbg[i * 64 + p] = pe[p * kRefTile + s];
            __syncthreads();
            continue;
        }
        const int kpe = 3 + 6 * a.pos_nfreq, kdpe = 3 + 6 * a.dir_nfreq;    // 63, 27
        bool stopped = false;
        float acc[64];
#pragma unroll 1
        for (int l = 0; l < 8; ++l) {
            const int n0 = 64 * grp;
#pragma unroll
            for (int j = 0; j < 64; ++j) acc[j] = a.bias[a.boff[l] + n0 + j];
            const float* wt = a.wt + a.off[l];
            if (l == 0) dense<64>(acc, wt, 256, n0, pe, kpe, s);
            else if (l == 5) {                                              // cat([x_pe, h]), vanilla.py:131
                dense<64>(acc, wt, 256, n0, pe, kpe, s);
                dense<64>(acc, wt + (int64_t)kpe * 256, 256, n0, h, 256, s);
            } else dense<64>(acc, wt, 256, n0, h, 256, s);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 64; ++j) h[(n0 + j) * kRefTile + s] = fmaxf(acc[j], 0.f);
            __syncthreads();
            if (a.stop_stage == l) {
                if (live) for (int j = 0; j < 64; ++j) a.dbg[i * 256 + n0 + j] = h[(n0 + j) * kRefTile + s];
                stopped = true;
                break;
            }
        }
        if (stopped) { __syncthreads(); continue; }
        if (a.plain_head) {                                                 // outputs = output_linear(h), vanilla.py:145
            if (grp == 0) {
                float c[4] = {a.bias[a.boff[10]], a.bias[a.boff[10] + 1], a.bias[a.boff[10] + 2], a.bias[a.boff[10] + 3]};
                dense<4>(c, a.wt + a.off[10], 4, 0, h, 256, s);
                if (live) reinterpret_cast<float4*>(a.out)[i] = make_float4(c[0], c[1], c[2], c[3] * a.sigma_scale);
            }
            __syncthreads();
            continue;
        }
        // alpha (vanilla.py:135) then feature (:136)
        float sigma = 0.f;
        if (grp == 0) {
            float a1[1] = {a.bias[a.boff[10]]};
            dense<1>(a1, a.wt + a.off[10], 1, 0, h, 256, s);
            sigma = a1[0];
        }
        {
            const int n0 = 64 * grp;
#pragma unroll
            for (int j = 0; j < 64; ++j) acc[j] = a.bias[a.boff[9] + n0 + j];
            dense<64>(acc, a.wt + a.off[9], 256, n0, h, 256, s);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 64; ++j) h[(n0 + j) * kRefTile + s] = acc[j];
            __syncthreads();
            if (a.stop_stage == 8) {
                if (live) for (int j = 0; j < 64; ++j) a.dbg[i * 256 + n0 + j] = acc[j];
                __syncthreads();
                continue;
            }
        }
        // views: cat([feature, d_pe]) -> 128, relu (vanilla.py:137-141)
        {
            const int n0 = 32 * grp;
            float v[32];
#pragma unroll
            for (int j = 0; j < 32; ++j) v[j] = a.bias[a.boff[8] + n0 + j];
            dense<32>(v, a.wt + a.off[8], 128, n0, h, 256, s);
            dense<32>(v, a.wt + a.off[8] + 256 * 128, 128, n0, dpe, kdpe, s);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 32; ++j) h[(n0 + j) * kRefTile + s] = fmaxf(v[j], 0.f);
            __syncthreads();
            if (a.stop_stage == 9) {
                if (live) for (int j = 0; j < 32; ++j) a.dbg[i * 128 + n0 + j] = h[(n0 + j) * kRefTile + s];
                __syncthreads();
                continue;
            }
        }
        if (grp == 0) {                                                     // rgb (vanilla.py:143-144)
            float c[3] = {a.bias[a.boff[11]], a.bias[a.boff[11] + 1], a.bias[a.boff[11] + 2]};
            dense<3>(c, a.wt + a.off[11], 3, 0, h, 128, s);
            if (live) reinterpret_cast<float4*>(a.out)[i] = make_float4(c[0], c[1], c[2], sigma * a.sigma_scale);
        }
        __syncthreads();
    }
}

}  // namespace

namespace nm {

int launch_mlp_ref(const RefLaunch& L, const float* pts, const float* dirs, const float* origin, const float* direction,
                   const float* z, int64_t n, int S, int in_mode, int stop_stage, float sigma_scale, float* out, float* dbg,
                   hipStream_t stream) {
    RefArgs a;
    a.wt = L.wt; a.bias = L.bias; a.petab = L.petab;
    for (int i = 0; i < 12; ++i) { a.off[i] = L.off[i]; a.boff[i] = L.boff[i]; }
    a.pts = pts; a.dirs = dirs; a.origin = origin; a.direction = direction; a.z = z;
    a.out = out; a.dbg = dbg; a.n = n; a.S = S; a.in_mode = in_mode; a.stop_stage = stop_stage; a.sigma_scale = sigma_scale;
    a.pe_kind = L.pe_kind; a.pos_nfreq = L.pos_nfreq; a.dir_nfreq = L.dir_nfreq;
    a.plain_head = L.plain_head;
    const int64_t ntiles = (n + kRefTile - 1) / kRefTile;
    const int grid = (int)(ntiles < 2048 ? ntiles : 2048);
    hipLaunchKernelGGL(nerf_mlp_ref_kernel, dim3(grid), dim3(kRefThreads), 0, stream, a);
    return check_launch("nerf_mlp_ref_kernel");
}

}  // namespace nm
