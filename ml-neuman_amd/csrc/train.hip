// SURVEY 8f-1, first slice: the device primitives of a training step of the background NeRF
// (reference trainers/vanilla_nerf_trainer.py:45-96 -- forward with saved activations, torch autograd's backward).
//   * nm_gemm_f32            C = op(A) op(B) on the f32 MFMA (v_mfma_f32_32x32x2_f32: true float32 products and accumulation,
//                            like the reference's sgemm), with the epilogues a dense layer needs in either direction: + bias,
//                            ReLU (forward), x (saved activation > 0) (backward-data), accumulate (the skip connection and
//                            the two-input views layer are two GEMMs into one output).  Backward-weights (dW = dZ^T A, a
//                            reduction over all samples into a 256 x 256 output) runs split-K with a deterministic second pass.
//   * nm_pe_encode           models/vanilla.py:60-92 as a stand-alone kernel (the rendering path has it fused in K4)
//   * nm_composite_backward  d loss / d raw through raw2outputs (utils/render_utils.py:69-105)
// The layer loop lives in neuman_hip/train.py (the reference's is Python as well).  This is the correctness slice: the f32 MFMA
// peaks at 157 TFLOP/s, a split-bf16 version on the 2.5 PFLOP/s pipe is the follow-up.
#include <math.h>

#include "common.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 16;

struct GemmArgs {
    const float* A; const float* B; float* C;
    const float* bias; const float* mask;
    float* partial;                 // split-K: raw accumulators to partial[z][M][N] instead of C
    float* colsum;                  // NM_GEMM_COLSUM: [ceil(M/64)][N] column sums of the stored output, one band per wave row-block
    int M, N, K, lda, ldb, ldc, ldmask, k_per_split, flags;
};

// One 128 x 16 operand tile -> registers (two float4 per thread), zero beyond the edges.
//   KMAJOR source: element (k, d) at S[k * ld + d]   (a [K, DIM] row-major array: direct copy)
//   else:          element (k, d) at S[d * ld + k]   (a [DIM, K] row-major array: transposed on the way into LDS)
template <bool KMAJOR>
__device__ __forceinline__ void tile_load(const float* __restrict__ S, int ld, int DIM, int d0, int k0, int kend, int tid, float4 (&r)[2]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        r[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (KMAJOR) {
            const int k = k0 + (tid >> 5) + 8 * h, d = d0 + 4 * (tid & 31);
            if (k < kend && d < DIM) r[h] = *reinterpret_cast<const float4*>(S + (int64_t)k * ld + d);
        } else {                                                       // four lanes share a row's 64 B: 16 cache lines per wave load, not 64
            const int d = d0 + (tid >> 2) + 64 * h, k = k0 + 4 * (tid & 3);
            if (d < DIM && k < kend) r[h] = *reinterpret_cast<const float4*>(S + (int64_t)d * ld + k);
        }
    }
}
constexpr int LDT = BM + 4;             // LDS row stride: the transposing stores below hit every bank twice (the minimum for 64 lanes)
template <bool KMAJOR>
__device__ __forceinline__ void tile_store(float (*T)[LDT], int tid, const float4 (&r)[2]) {     // T[k][d]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        if (KMAJOR) {
            *reinterpret_cast<float4*>(&T[(tid >> 5) + 8 * h][4 * (tid & 31)]) = r[h];
        } else {
            const int d = (tid >> 2) + 64 * h, k = 4 * (tid & 3);
            T[k][d] = r[h].x; T[k + 1][d] = r[h].y; T[k + 2][d] = r[h].z; T[k + 3][d] = r[h].w;
        }
    }
}

// The epilogue both GEMM kernels share: a wave's 2 x 2 accumulator blocks -> C (or the split-K partials), with the dense-layer
// options.  Accumulator element v of a 32 x 32 block: row 8 (v / 4) + 4 (lane / 32) + v % 4, column lane % 32.
__device__ __forceinline__ void store_tile(const GemmArgs& g, const floatx16 (&acc)[2][2], int m0, int n0, int wm, int wn, int lane) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wn + 32 * j + (lane & 31);
            if (col >= g.N) continue;
            const int row0 = m0 + wm + 32 * i + 4 * (lane >> 5);
            if (g.partial) {
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int row = row0 + 8 * (v >> 2) + (v & 3);
                    if (row < g.M) g.partial[((int64_t)blockIdx.z * g.M + row) * g.N + col] = acc[i][j][v];
                }
                continue;
            }
            // what the epilogue reads goes to registers first: sixteen loads in flight, not sixteen round trips
            float cv[16], mv[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = row0 + 8 * (v >> 2) + (v & 3);
                cv[v] = ((g.flags & NM_GEMM_ACCUMULATE) && row < g.M) ? g.C[(int64_t)row * g.ldc + col] : 0.f;
                mv[v] = ((g.flags & NM_GEMM_MASK) && row < g.M) ? g.mask[(int64_t)row * g.ldmask + col] : 1.f;
            }
            const float bias = (g.flags & NM_GEMM_BIAS) ? g.bias[col] : 0.f;
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int row = row0 + 8 * (v >> 2) + (v & 3);
                if (row >= g.M) continue;
                float x = acc[i][j][v] + cv[v] + bias;
                if (g.flags & NM_GEMM_RELU) x = fmaxf(x, 0.f);
                g.C[(int64_t)row * g.ldc + col] = mv[v] > 0.f ? x : 0.f;
            }
        }
}

// The same epilogue with 16-byte accesses: each wave passes its 64 x 64 accumulator tile through LDS (free once the k-loop is
// over), half of it at a time -- 32 rows x 64 columns = 8 KB per wave --, and reads it back row-wise, so that a lane holds four
// consecutive columns of one row: the mask / C loads and the stores become b128 accesses of 256 contiguous bytes per row (a
// quarter of the instructions of the column-per-lane layout, whole cache lines).  Needs 16-byte aligned C / mask / partial rows
// (ldc, ldmask multiples of 4; N is one already); `scratch` = 8192 floats of LDS no wave reads any more (caller synchronised).
__device__ __forceinline__ bool wide_ok(const GemmArgs& g) {
    const bool c_ok = g.partial ? (((uintptr_t)g.partial & 15) == 0) : ((((uintptr_t)g.C & 15) == 0) && (g.ldc & 3) == 0);
    const bool m_ok = !(g.flags & NM_GEMM_MASK) || ((((uintptr_t)g.mask & 15) == 0) && (g.ldmask & 3) == 0);
    const bool b_ok = !(g.flags & NM_GEMM_BIAS) || (((uintptr_t)g.bias & 15) == 0);
    return c_ok && m_ok && b_ok;
}
__device__ __forceinline__ void store_tile_wide(const GemmArgs& g, const floatx16 (&acc)[2][2], int m0, int n0, int wm, int wn, int w, int lane,
                                                float* scratch) {
    float* T = scratch + w * 2048;                                  // this wave's [32][64] window
    const int c4 = 4 * (lane & 15), rsub = lane >> 4;               // read-back: columns c4..c4+3 of rows rsub + 4 k
    const int col = n0 + wn + c4;
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if ((g.flags & NM_GEMM_BIAS) && col < g.N) bias = *reinterpret_cast<const float4*>(g.bias + col);
    float cs[4] = {0.f, 0.f, 0.f, 0.f};                             // NM_GEMM_COLSUM: this lane's four columns over the rows it stores
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) T[(8 * (v >> 2) + 4 * (lane >> 5) + (v & 3)) * 64 + 32 * j + (lane & 31)] = acc[i][j][v];
        // (a wave reads only what it wrote itself: LDS operations of one wave complete in order, no barrier)
        const int row_base = m0 + wm + 32 * i + rsub;
#pragma unroll
        for (int half = 0; half < 2; ++half) {                      // four rows' worth of loads in flight at a time
            float4 cv[4], mv[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = row_base + 4 * (4 * half + k);
                const bool in = row < g.M && col < g.N;
                cv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                mv[k] = make_float4(1.f, 1.f, 1.f, 1.f);
                if (!g.partial && (g.flags & NM_GEMM_ACCUMULATE) && in) cv[k] = *reinterpret_cast<const float4*>(g.C + (int64_t)row * g.ldc + col);
                if (!g.partial && (g.flags & NM_GEMM_MASK) && in) mv[k] = *reinterpret_cast<const float4*>(g.mask + (int64_t)row * g.ldmask + col);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int rr = rsub + 4 * (4 * half + k), row = row_base + 4 * (4 * half + k);
                if (row >= g.M || col >= g.N) continue;
                const float4 a = *reinterpret_cast<const float4*>(T + rr * 64 + c4);
                if (g.partial) {
                    *reinterpret_cast<float4*>(g.partial + ((int64_t)blockIdx.z * g.M + row) * g.N + col) = a;
                    continue;
                }
                float x[4] = {a.x + cv[k].x + bias.x, a.y + cv[k].y + bias.y, a.z + cv[k].z + bias.z, a.w + cv[k].w + bias.w};
                const float m[4] = {mv[k].x, mv[k].y, mv[k].z, mv[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (g.flags & NM_GEMM_RELU) x[e] = fmaxf(x[e], 0.f);
                    x[e] = m[e] > 0.f ? x[e] : 0.f;
                }
                *reinterpret_cast<float4*>(g.C + (int64_t)row * g.ldc + col) = make_float4(x[0], x[1], x[2], x[3]);
#pragma unroll
                for (int e = 0; e < 4; ++e) cs[e] += x[e];             // rows ascending within the lane: 16 rows of the wave's 64
            }
        }
    }
    if (g.colsum) {                                                  // the four lanes sharing these columns (rows rsub = 0..3 mod 4), then one store
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            cs[e] += __shfl_xor(cs[e], 16, 64);
            cs[e] += __shfl_xor(cs[e], 32, 64);
        }
        if (lane < 16 && col < g.N && m0 + wm < g.M)
            *reinterpret_cast<float4*>(g.colsum + (int64_t)((m0 + wm) >> 6) * g.N + col) = make_float4(cs[0], cs[1], cs[2], cs[3]);
    }
}

// 128 x 128 output tile per workgroup of four waves (64 x 64 each = 2 x 2 MFMA blocks of 32 x 32), K in steps of 16 through a
// double-buffered LDS tile pair; the next step's global loads are in flight during the current step's MFMAs.
template <bool A_KMAJOR, bool B_KMAJOR>
__global__ __launch_bounds__(256, 3) void gemm_f32_kernel(const GemmArgs g) {
    __shared__ float smem[2 * 2 * BK * LDT];                       // As | Bs; reused by the epilogue
    float (*As)[BK][LDT] = reinterpret_cast<float (*)[BK][LDT]>(smem);
    float (*Bs)[BK][LDT] = reinterpret_cast<float (*)[BK][LDT]>(smem + 2 * BK * LDT);
    static_assert(2 * 2 * BK * LDT >= 4 * 2048, "the epilogue needs 8192 floats of LDS");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    float4 ra[2], rb[2];
    tile_load<A_KMAJOR>(g.A, g.lda, g.M, m0, kbeg, kend, tid, ra);
    tile_load<B_KMAJOR>(g.B, g.ldb, g.N, n0, kbeg, kend, tid, rb);
    tile_store<A_KMAJOR>(As[0], tid, ra);
    tile_store<B_KMAJOR>(Bs[0], tid, rb);
    __syncthreads();
    int buf = 0;
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
        const bool more = k0 + BK < kend;
        if (more) {
            tile_load<A_KMAJOR>(g.A, g.lda, g.M, m0, k0 + BK, kend, tid, ra);
            tile_load<B_KMAJOR>(g.B, g.ldb, g.N, n0, k0 + BK, kend, tid, rb);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const int kr = kk + (lane >> 5), c = lane & 31;
            const float a0 = As[buf][kr][wm + c], a1 = As[buf][kr][wm + 32 + c];
            const float b0 = Bs[buf][kr][wn + c], b1 = Bs[buf][kr][wn + 32 + c];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (more) {
            tile_store<A_KMAJOR>(As[buf ^ 1], tid, ra);
            tile_store<B_KMAJOR>(Bs[buf ^ 1], tid, rb);
        }
        __syncthreads();
        buf ^= 1;
    }
    if (wide_ok(g)) store_tile_wide(g, acc, m0, n0, wm, wn, w, lane, smem);
    else store_tile(g, acc, m0, n0, wm, wn, lane);
}

// ---- the same product on the bf16 MFMA, each float32 operand split into bf16 hi + lo (RNE both times; x - hi is exact in f32)
// and hi*hi + hi*lo + lo*hi accumulated in f32: 2^-17 relative per product, 5x the f32 MFMA's rate at peak.
// LDS holds the operand tiles already split and in MFMA fragment order: T[k-octet][row] = 8 consecutive-k bf16 (16 B), which is
// what lane (row = l % 32, octet = l / 32) of v_mfma_f32_32x32x16_bf16 consumes -- one ds_read_b128 per fragment, conflict-free.
constexpr int XK = 32;                  // k per tile: two MFMA k-steps

typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <bool HALF>
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo) {
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2 a = {v[2 * p], v[2 * p + 1]};
        if (HALF) {                                                  // fp16 parts (11 + 11 significand bits); saturate instead of inf - inf
            a.x = __builtin_amdgcn_fmed3f(a.x, -65504.f, 65504.f);
            a.y = __builtin_amdgcn_fmed3f(a.y, -65504.f, 65504.f);
            const f16x2 hb = __builtin_convertvector(a, f16x2);
            const f32x2 hf = __builtin_convertvector(hb, f32x2);
            const f32x2 r = {a.x - hf.x, a.y - hf.y};
            const f16x2 lb = __builtin_convertvector(r, f16x2);
            h[p] = __builtin_bit_cast(unsigned, hb);
            l[p] = __builtin_bit_cast(unsigned, lb);
            continue;
        }
        const bf16x2 hb = __builtin_convertvector(a, bf16x2);
        const f32x2 hf = __builtin_convertvector(hb, f32x2);
        const f32x2 r = {a.x - hf.x, a.y - hf.y};
        const bf16x2 lb = __builtin_convertvector(r, bf16x2);
        h[p] = __builtin_bit_cast(unsigned, hb);
        l[p] = __builtin_bit_cast(unsigned, lb);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

// one 128 x 32 operand tile -> registers: two (row, k-octet) pairs of 8 floats per thread, zero beyond the edges.
//   KMAJOR (element (k, d) at S[k * ld + d]): lanes run along d, eight strided 4-byte loads each -- 256 B contiguous per instruction
//   else   (element (k, d) at S[d * ld + k]): four lanes share a row's 128 B (one cache line), so a wave instruction touches 16
//          lines; one lane per row (64 lines per instruction) left the kernel bound by the texture path, not by its MFMAs
template <bool KMAJOR>
__device__ __forceinline__ void xtile_item(int tid, int h, int& row, int& octet) {
    if (KMAJOR) { row = tid & 127; octet = (tid >> 7) + 2 * h; }
    else { const int item = tid + 256 * h; row = item >> 2; octet = item & 3; }
}
template <bool KMAJOR>
__device__ __forceinline__ void xtile_load(const float* __restrict__ S, int ld, int DIM, int d0, int k0, int kend, int tid, float (&r)[2][8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int row, octet;
        xtile_item<KMAJOR>(tid, h, row, octet);
        const int d = d0 + row, k = k0 + 8 * octet;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[h][j] = 0.f;
        if (d >= DIM) continue;
        if (KMAJOR) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (k + j < kend) r[h][j] = S[(int64_t)(k + j) * ld + d];
        } else {
            if (k < kend) *reinterpret_cast<float4*>(&r[h][0]) = *reinterpret_cast<const float4*>(S + (int64_t)d * ld + k);
            if (k + 4 < kend) *reinterpret_cast<float4*>(&r[h][4]) = *reinterpret_cast<const float4*>(S + (int64_t)d * ld + k + 4);
        }
    }
}
template <bool KMAJOR, bool HALF>
__device__ __forceinline__ void xtile_store(uint4 (*Th)[BM], uint4 (*Tl)[BM], int tid, const float (&r)[2][8]) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        int row, octet;
        xtile_item<KMAJOR>(tid, h, row, octet);
        uint4 hi, lo;
        split8<HALF>(r[h], hi, lo);
        Th[octet][row] = hi;
        Tl[octet][row] = lo;
    }
}

template <bool A_KMAJOR, bool B_KMAJOR, bool HALF>
__global__ __launch_bounds__(256, 3) void gemm_split_kernel(const GemmArgs g) {
    __shared__ uint4 smem[4][XK / 8][BM];                           // Ah | Al | Bh | Bl (BM == BN); reused by the epilogue
    uint4 (*Ah)[BM] = smem[0], (*Al)[BM] = smem[1], (*Bh)[BM] = smem[2], (*Bl)[BM] = smem[3];
    static_assert(sizeof(smem) >= 4 * 2048 * sizeof(float) && BM == BN, "the epilogue needs 8192 floats of LDS");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * g.k_per_split;
    const int kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
    const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
    floatx16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    float ra[2][8], rb[2][8];
    xtile_load<A_KMAJOR>(g.A, g.lda, g.M, m0, kbeg, kend, tid, ra);
    xtile_load<B_KMAJOR>(g.B, g.ldb, g.N, n0, kbeg, kend, tid, rb);
    for (int k0 = kbeg; k0 < kend; k0 += XK) {
        __syncthreads();                                               // the previous tile's fragments have been read
        xtile_store<A_KMAJOR, HALF>(Ah, Al, tid, ra);
        xtile_store<B_KMAJOR, HALF>(Bh, Bl, tid, rb);
        __syncthreads();
        if (k0 + XK < kend) {                                          // next tile's loads fly during this tile's MFMAs
            xtile_load<A_KMAJOR>(g.A, g.lda, g.M, m0, k0 + XK, kend, tid, ra);
            xtile_load<B_KMAJOR>(g.B, g.ldb, g.N, n0, k0 + XK, kend, tid, rb);
        }
#pragma unroll
        for (int s = 0; s < XK / 16; ++s) {
            const int o = 2 * s + (lane >> 5), c = lane & 31;
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, Ah[o][wm + 32 * i + c]);
                al[i] = __builtin_bit_cast(bf16x8, Al[o][wm + 32 * i + c]);
                bh[i] = __builtin_bit_cast(bf16x8, Bh[o][wn + 32 * i + c]);
                bl[i] = __builtin_bit_cast(bf16x8, Bl[o][wn + 32 * i + c]);
            }
            auto mm = [](bf16x8 x, bf16x8 y, floatx16 c) {          // (the registers hold fp16 bit patterns when HALF)
                if (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
            };
            // the three products of one k-step, each over the four accumulator blocks: a block's next MFMA never waits for its last
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mm(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mm(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mm(ah[i], bh[j], acc[i][j]);
        }
    }
    __syncthreads();                                                   // every wave is done with the operand tiles
    if (wide_ok(g)) store_tile_wide(g, acc, m0, n0, wm, wn, w, lane, reinterpret_cast<float*>(smem));
    else store_tile(g, acc, m0, n0, wm, wn, lane);
}

// Backward-weights of a 256-wide layer, dW [256, 256] = dZ^T A over all n samples (both operands K-major: [n, 256] row-major), as ONE
// 256 x 256 tile per workgroup: every operand element is read from HBM exactly once (the 128 x 128 tiling of gemm_split_kernel reads
// each operand row for both of its tile columns), 512 threads = 8 waves of 128 x 64 (4 x 2 accumulator blocks = 128 registers), a
// contiguous range of samples per workgroup (split-K over the grid, partial[z][256][256] -> splitk_reduce_kernel).  A step = 32 samples:
// coalesced 4-byte loads along the features, eight samples of one feature per lane (the transpose happens in registers), split into
// 16-bit parts, one ds_write_b128 per part -- then two MFMA k-steps x 3 products x 8 blocks per wave.
template <bool HALF>
__global__ __launch_bounds__(512, 2) void wgrad256_kernel(const GemmArgs g) {
    __shared__ uint4 smem[4][XK / 8][256];                          // Ah | Al | Bh | Bl: [octet of 8 samples][feature]
    uint4 (*Ah)[256] = smem[0], (*Al)[256] = smem[1], (*Bh)[256] = smem[2], (*Bl)[256] = smem[3];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int kbeg = blockIdx.x * g.k_per_split;
    const int kend = kbeg + g.k_per_split < g.K ? kbeg + g.k_per_split : g.K;
    const int wm = (w >> 2) * 128, wn = (w & 3) * 64;
    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    // items of a step: (matrix, octet, feature) = 2 x 4 x 256; thread t takes items t + 512 q, q = 0..3: feature = t & 255, octet = (t >> 8) + 2 (q & 1),
    // matrix = q >> 1 -- consecutive lanes read consecutive features of one sample row (256 B per wave instruction)
    float r[4][8];
    const int f = tid & 255, o2 = tid >> 8;
    auto load = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float* S = (q >> 1) ? g.B : g.A;
            const int ld = (q >> 1) ? g.ldb : g.lda;
            const int k = k0 + 8 * (o2 + 2 * (q & 1));
#pragma unroll
            for (int j = 0; j < 8; ++j) r[q][j] = (k + j < kend) ? S[(int64_t)(k + j) * ld + f] : 0.f;
        }
    };
    if (kbeg < kend) load(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += XK) {
        __syncthreads();                                               // the previous step's fragments have been read
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint4 hi, lo;
            split8<HALF>(r[q], hi, lo);
            const int o = o2 + 2 * (q & 1);
            if (q >> 1) { Bh[o][f] = hi; Bl[o][f] = lo; }
            else { Ah[o][f] = hi; Al[o][f] = lo; }
        }
        __syncthreads();
        if (k0 + XK < kend) load(k0 + XK);                             // the next step's loads fly during this step's MFMAs
#pragma unroll
        for (int s = 0; s < XK / 16; ++s) {
            const int o = 2 * s + (lane >> 5), c = lane & 31;
            bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ah[i] = __builtin_bit_cast(bf16x8, Ah[o][wm + 32 * i + c]);
                al[i] = __builtin_bit_cast(bf16x8, Al[o][wm + 32 * i + c]);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = __builtin_bit_cast(bf16x8, Bh[o][wn + 32 * j + c]);
                bl[j] = __builtin_bit_cast(bf16x8, Bl[o][wn + 32 * j + c]);
            }
            auto mm = [](bf16x8 x, bf16x8 y, floatx16 c) {
                if (HALF) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, x), __builtin_bit_cast(f16x8, y), c, 0, 0, 0);
                return __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, c, 0, 0, 0);
            };
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mm(al[i], bh[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mm(ah[i], bl[j], acc[i][j]);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = mm(ah[i], bh[j], acc[i][j]);
        }
    }
    // accumulator block (i, j): register v of lane (g, c) = row wm + 32 i + (v & 3) + 8 (v >> 2) + 4 g, column wn + 32 j + c
    float* P = g.partial + (int64_t)blockIdx.x * 256 * 256;
    const int gq = lane >> 5, c = lane & 31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) P[(wm + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * gq) * 256 + wn + 32 * j + c] = acc[i][j][v];
}
constexpr int kWgradSplits = 256;
__host__ inline bool wgrad256_ok(int mode, int a_kmajor, int b_kmajor, int M, int N, int K, int flags) {
    return mode != 0 && a_kmajor && b_kmajor && M == 256 && N == 256 && K >= 16384 && !(flags & ~NM_GEMM_ACCUMULATE);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// 16-bit operands (round 5): backward-weights of the 256-wide layers from the fp16 copies the fused forward / backward-data chain keep
// (nm_mlp_forward_save16: 32 x activation; nm_mlp_backward_chain16: dZ x nm_dz_scale(amax); both [n][256] in k-slot order) -- 1 KB per
// sample and product instead of 2 KB, ONE MFMA per product instead of three, no conversions.  Rounding: 2^-12 relative per element,
// independent from sample to sample, under a sum over >= 32 k samples.
//
// One 256 x 256 tile per workgroup (8 waves of 128 x 64, as wgrad256_kernel), a contiguous range of samples per workgroup, a step = 64
// samples: thread (operand, octet o, chunk c) loads chunk c (8 features, 16 B) of the 8 samples of octet o -- a wave instruction reads two
// whole 512-byte rows --, transposes the 8 x 8 halves in registers and writes 8 fragments "8 samples of one feature" (the MFMA operand
// of a product whose K runs over samples) to T[operand][octet][position e 32 + c]: conflict-free both ways.  The accumulators' rows /
// columns are those positions; wgrad16_reduce_kernel sums the splits in order, scales by 1 / (32 s) and stores to the natural [out][in].
struct Wgrad16Args {
    const uint4* P[8]; const uint4* Q[8];       // per product: dZ16 [n][MP / 8] uint4 rows; activation16 [n][NQ / 8] uint4 rows
    float* C[8]; int ldc[8];                    // per product: the [MP][ncols] gradient (row stride ldc)
    float* partial;                             // [nprod][nsplit][MP][NQ]
    const float* amax;                          // the device scalar dZ16's scale derives from (nm_dz_scale)
    int64_t n;
    int k_per_split, nsplit, ncols;             // ncols <= NQ columns are stored
};

// MP = 256: dZ of a trunk layer / feature_linear; 128: of the views layer (both k-slot order).  NQ = 256: a hidden operand (k-slot order);
// NQ = 64: an encoded input (natural order, zero beyond the encoding), fp16 of 32 x value too.  8 waves tile the [MP][NQ] output:
//   (256, 256) 2 x 4 of 128 x 64    (256, 64) 8 x 1 of 32 x 64    (128, 256) 2 x 4 of 64 x 64    (128, 64) 4 x 2 of 32 x 32
template <int MP, int NQ>
__global__ __launch_bounds__(512, 2) void wgrad16_kernel(const Wgrad16Args g) {
    constexpr int PC = MP / 8, QC = NQ / 8;                         // 16-byte chunks per P / Q row
    constexpr int WN = NQ == 256 ? 4 : (MP == 256 ? 1 : 2);         // waves along the columns
    constexpr int WM = 8 / WN;
    constexpr int MI = MP / (32 * WM), NJ = NQ / (32 * WN);         // 32 x 32 blocks per wave
    __shared__ uint4 T[2][8][256];                                  // [operand][octet][position]: the first MP / NQ positions of a row
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int prod = blockIdx.y;
    const int op = tid >> 8, t8 = tid & 255;
    const int rowlen = op ? QC : PC;
    const bool loader = t8 < 8 * rowlen;
    const int c = t8 % rowlen, o = (t8 / rowlen) & 7;
    const uint4* __restrict__ src = op ? g.Q[prod] : g.P[prod];
    const int64_t kbeg = (int64_t)blockIdx.x * g.k_per_split;
    const int64_t kend = kbeg + g.k_per_split < g.n ? kbeg + g.k_per_split : g.n;
    const int wm = (w / WN) * (32 * MI), wn = (w % WN) * (32 * NJ);
    floatx16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    unsigned r[8][4];
    auto load = [&](int64_t k0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t k = k0 + 8 * o + j;
            uint4 t = make_uint4(0u, 0u, 0u, 0u);
            if (loader && k < kend) t = src[k * rowlen + c];
            r[j][0] = t.x; r[j][1] = t.y; r[j][2] = t.z; r[j][3] = t.w;
        }
    };
    if (kbeg < kend) load(kbeg);
    for (int64_t k0 = kbeg; k0 < kend; k0 += 64) {
        __syncthreads();                                               // the previous step's fragments have been read
        if (loader) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {                               // element e of the chunk, samples 0..7 -> one fragment
                unsigned q[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    const unsigned a = r[2 * m][e >> 1], b = r[2 * m + 1][e >> 1];
                    q[m] = (e & 1) ? ((a >> 16) | (b & 0xffff0000u)) : ((a & 0xffffu) | (b << 16));
                }
                T[op][o][e * rowlen + c] = make_uint4(q[0], q[1], q[2], q[3]);
            }
        }
        __syncthreads();
        if (k0 + 64 < kend) load(k0 + 64);                             // the next step's loads fly during this step's MFMAs
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int oct = 2 * s + (lane >> 5), cc = lane & 31;
            f16x8 a[MI], b[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) a[i] = __builtin_bit_cast(f16x8, T[0][oct][wm + 32 * i + cc]);
#pragma unroll
            for (int j = 0; j < NJ; ++j) b[j] = __builtin_bit_cast(f16x8, T[1][oct][wn + 32 * j + cc]);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    }
    float* P = g.partial + ((int64_t)prod * g.nsplit + blockIdx.x) * (MP * NQ);
    const int gq = lane >> 5, cc = lane & 31;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) P[(wm + 32 * i + (v & 3) + 8 * (v >> 2) + 4 * gq) * NQ + wn + 32 * j + cc] = acc[i][j][v];
}

// position p = e (W / 8) + c of a fragment row / column of a W-wide k-slot-order operand -> the feature it is (k-slot (chunk c, element e), mlp_layout.h)
template <int W>
__device__ __forceinline__ int pos_feature(int p) {
    const int c = p % (W / 8), e = p / (W / 8);
    return 32 * (c >> 2) + 8 * (2 * ((c >> 1) & 1) + (e >> 2)) + 4 * (c & 1) + (e & 3);
}
template <int MP, int NQ>
__global__ __launch_bounds__(256) void wgrad16_reduce_kernel(const Wgrad16Args g) {
    const int prod = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;                          // (row position, column position): coalesced reads of the partials
    const float* part = g.partial + (int64_t)prod * g.nsplit * (MP * NQ) + i;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};                 // eight interleaved partial sums, combined in a fixed order (as splitk_reduce_kernel)
    int z = 0;
    for (; z + 8 <= g.nsplit; z += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] += part[(int64_t)(z + k) * (MP * NQ)];
    for (int k = 0; z < g.nsplit; ++z, ++k) p[k] += part[(int64_t)z * (MP * NQ)];
    const float sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    const float inv = 1.f / (nm_dz_scale(*g.amax) * kNmAct16Scale);        // (a power of two: exact)
    const int pc = i % NQ;
    const int col = NQ == 256 ? pos_feature<256>(pc) : 8 * (pc % (NQ / 8)) + pc / (NQ / 8);     // (natural-order operand: position e (NQ / 8) + c = column 8 c + e)
    if (col < g.ncols) g.C[prod][(int64_t)pos_feature<MP>(i / NQ) * g.ldc[prod] + col] = sum * inv;
}

// alpha_linear's weight gradient: out[f] = sum_n d_raw[n][3] H7[n][f] from the fp16 copy (k-slot order, x 32): a thread owns two slots of a
// band of rows; the bands are summed in order by splitk_reduce_kernel
constexpr int kAlphaRows = 512;
__global__ __launch_bounds__(256) void wgrad_alpha16_kernel(const float* __restrict__ d_raw, const unsigned* __restrict__ h16, int64_t n, float* __restrict__ partial) {
    const int t = threadIdx.x & 127, half = threadIdx.x >> 7;               // slots 2 t, 2 t + 1; rows of parity `half`
    const int64_t r0 = (int64_t)blockIdx.x * kAlphaRows, r1 = r0 + kAlphaRows < n ? r0 + kAlphaRows : n;
    float a0 = 0.f, a1 = 0.f;
    for (int64_t k = r0 + half; k < r1; k += 2) {
        const unsigned v = h16[k * 128 + t];
        const float ds = d_raw[k * 4 + 3];
        a0 = fmaf(ds, (float)__builtin_bit_cast(_Float16, (unsigned short)(v & 0xffffu)), a0);
        a1 = fmaf(ds, (float)__builtin_bit_cast(_Float16, (unsigned short)(v >> 16)), a1);
    }
    __shared__ float sh[2][256];
    sh[half][2 * t] = a0;
    sh[half][2 * t + 1] = a1;
    __syncthreads();
    if (threadIdx.x < 256) {                                            // slot p = 8 c + e -> feature; x 1/32
        const int p = threadIdx.x, c = p >> 3, e = p & 7;
        const int f = 32 * (c >> 2) + 8 * (2 * ((c >> 1) & 1) + (e >> 2)) + 4 * (c & 1) + (e & 3);
        partial[(int64_t)blockIdx.x * 256 + f] = (sh[0][p] + sh[1][p]) * (1.f / kNmAct16Scale);
    }
}

// The 4-row heads of a training step in ONE pass over d_raw, the fp16 copy of H_7 and hv: alpha_linear's weight gradient (sum_n d sigma H7),
// rgb_linear's (sum_n d rgb_k hv), the column sums of d_raw (their bias gradients) and max |d_raw| (the scale source of the backward-data pass).
// Thread (row group r of 8, lane c of 32) of a band of kHeadRows rows: chunk c of H7's row (8 halves) and columns 4 c .. 4 c + 3 of hv's; the
// eight row groups are summed through LDS in order, the bands by splitk_reduce_kernel.
constexpr int kHeadRows = 512;
constexpr int kHeadOut = 256 + 3 * 128 + 4;
__global__ __launch_bounds__(256) void wgrad_heads16_kernel(const float* __restrict__ d_raw, const uint4* __restrict__ h16, const float* __restrict__ hv, int64_t n,
                                                            float* __restrict__ partial, unsigned* __restrict__ amax) {
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kHeadRows, r1 = r0 + kHeadRows < n ? r0 + kHeadRows : n;
    float aa[8], ar[3][4], ab[4] = {0.f, 0.f, 0.f, 0.f}, m = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) aa[e] = 0.f;
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) ar[k][j] = 0.f;
    for (int64_t row = r0 + r; row < r1; row += 8) {
        const float4 dr = *reinterpret_cast<const float4*>(d_raw + row * 4);
        const uint4 h = h16[row * 32 + c];
        const float4 v = *reinterpret_cast<const float4*>(hv + row * 128 + 4 * c);
        const unsigned hw[4] = {h.x, h.y, h.z, h.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            aa[2 * p] = fmaf(dr.w, (float)__builtin_bit_cast(_Float16, (unsigned short)(hw[p] & 0xffffu)), aa[2 * p]);
            aa[2 * p + 1] = fmaf(dr.w, (float)__builtin_bit_cast(_Float16, (unsigned short)(hw[p] >> 16)), aa[2 * p + 1]);
        }
        const float d3[3] = {dr.x, dr.y, dr.z};
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ar[k][0] = fmaf(d3[k], v.x, ar[k][0]); ar[k][1] = fmaf(d3[k], v.y, ar[k][1]);
            ar[k][2] = fmaf(d3[k], v.z, ar[k][2]); ar[k][3] = fmaf(d3[k], v.w, ar[k][3]);
        }
        ab[0] += dr.x; ab[1] += dr.y; ab[2] += dr.z; ab[3] += dr.w;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(dr.x), fabsf(dr.y)), fmaxf(fabsf(dr.z), fabsf(dr.w))));
    }
    __shared__ float sh[8][kHeadOut];
#pragma unroll
    for (int e = 0; e < 8; ++e) sh[r][8 * c + e] = aa[e];                // slot order 8 c + e
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[r][256 + 128 * k + 4 * c + j] = ar[k][j];
    if (c == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[r][640 + j] = ab[j];
    }
    __syncthreads();
    float* P = partial + (int64_t)blockIdx.x * kHeadOut;
    for (int i = threadIdx.x; i < kHeadOut; i += 256) {
        float s = sh[0][i];
#pragma unroll
        for (int q = 1; q < 8; ++q) s += sh[q][i];
        if (i < 256) {                                                  // slot p = 8 c + e -> feature; x 1/32
            const int cc = i >> 3, e = i & 7;
            P[32 * (cc >> 2) + 8 * (2 * ((cc >> 1) & 1) + (e >> 2)) + 4 * (cc & 1) + (e & 3)] = s * (1.f / kNmAct16Scale);
        } else P[i] = s;
    }
    if (amax) {
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax, __float_as_uint(m));
    }
}

// ... and of the plain-head net (output_linear [4][256] straight off layer 7): out[k][f] = sum_n d_out[n][k] H7[n][f], the column sums of d_out, max |d_out|
constexpr int kOutOut = 4 * 256 + 4;
__global__ __launch_bounds__(256) void wgrad_out16_kernel(const float* __restrict__ d_out, const uint4* __restrict__ h16, int64_t n, float* __restrict__ partial,
                                                          unsigned* __restrict__ amax) {
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5;
    const int64_t r0 = (int64_t)blockIdx.x * kHeadRows, r1 = r0 + kHeadRows < n ? r0 + kHeadRows : n;
    float aa[4][8], ab[4] = {0.f, 0.f, 0.f, 0.f}, m = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) aa[k][e] = 0.f;
    for (int64_t row = r0 + r; row < r1; row += 8) {
        const float4 dr = *reinterpret_cast<const float4*>(d_out + row * 4);
        const uint4 h = h16[row * 32 + c];
        const unsigned hw[4] = {h.x, h.y, h.z, h.w};
        const float d4[4] = {dr.x, dr.y, dr.z, dr.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const float lo = (float)__builtin_bit_cast(_Float16, (unsigned short)(hw[p] & 0xffffu)), hi = (float)__builtin_bit_cast(_Float16, (unsigned short)(hw[p] >> 16));
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                aa[k][2 * p] = fmaf(d4[k], lo, aa[k][2 * p]);
                aa[k][2 * p + 1] = fmaf(d4[k], hi, aa[k][2 * p + 1]);
            }
        }
        ab[0] += dr.x; ab[1] += dr.y; ab[2] += dr.z; ab[3] += dr.w;
        m = fmaxf(m, fmaxf(fmaxf(fabsf(dr.x), fabsf(dr.y)), fmaxf(fabsf(dr.z), fabsf(dr.w))));
    }
    __shared__ float sh[8][kOutOut];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) sh[r][256 * k + 8 * c + e] = aa[k][e];
    if (c == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sh[r][1024 + j] = ab[j];
    }
    __syncthreads();
    float* P = partial + (int64_t)blockIdx.x * kOutOut;
    for (int i = threadIdx.x; i < kOutOut; i += 256) {
        float s = sh[0][i];
#pragma unroll
        for (int q = 1; q < 8; ++q) s += sh[q][i];
        if (i < 1024) {
            const int p = i & 255, cc = p >> 3, e = p & 7;
            P[(i & ~255) + 32 * (cc >> 2) + 8 * (2 * ((cc >> 1) & 1) + (e >> 2)) + 4 * (cc & 1) + (e & 3)] = s * (1.f / kNmAct16Scale);
        } else P[i] = s;
    }
    if (amax) {
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(amax, __float_as_uint(m));
    }
}

// largest magnitude of x[0 .. count): *out = max(*out, ...) (bit pattern of a non-negative float: unsigned order = float order)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t count, unsigned* __restrict__ out) {
    float m = 0.f;
    const int64_t n4 = count >> 2;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}

// second pass of split-K: C (+)= sum over the splits, in split order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, int splits, int M, int N, float* __restrict__ C, int ldc,
                                                            int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * N) return;
    // eight independent partial sums (splits z = k mod 8), combined in a fixed order: deterministic, and eight loads in flight
    // instead of a chain of `splits` dependent adds behind one load each
    const int64_t MN = (int64_t)M * N;
    float p[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int z = 0;
    for (; z + 8 <= splits; z += 8)
#pragma unroll
        for (int k = 0; k < 8; ++k) p[k] += partial[(int64_t)(z + k) * MN + i];
    for (int k = 0; z < splits; ++z, ++k) p[k] += partial[(int64_t)z * MN + i];
    const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    float* c = C + (i / N) * ldc + (i % N);
    *c = accumulate ? *c + s : s;
}

// column sums of X [n, W] (bias gradients): each workgroup sums a band of kColsumRows rows, one column per thread (coalesced
// along the row); a second kernel adds the bands
constexpr int kColsumRows = 256;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int64_t n, int W, int ld, float* __restrict__ partial) {
    const int64_t r0 = (int64_t)blockIdx.x * kColsumRows;
    const int64_t r1 = r0 + kColsumRows < n ? r0 + kColsumRows : n;
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};          // eight loads in flight
        int64_t r = r0;
        for (; r + 8 <= r1; r += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += X[(r + j) * ld + c];
        for (; r < r1; ++r) s[0] += X[r * ld + c];
        partial[(int64_t)blockIdx.x * W + c] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
    }
}

// second stage: one wave per column adds the bands (lane-strided partial sums in band order, then a butterfly)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ partial, int bands, int W, float* __restrict__ out) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= W) return;
    float s = 0.f;
    for (int b = lane; b < bands; b += 64) s += partial[(int64_t)b * W + c];
    s = wave_sum(s);
    if (lane == 0) out[c] = s;
}

int pick_splits(int M, int N, int K) {
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
    if (tiles >= 256 || K < 4096) return 1;
    int s = (512 + tiles - 1) / tiles;                       // ~2 workgroups per CU
    const int maxs = (K + 2047) / 2048;                      // at least 2048 of K per split
    s = s < maxs ? s : maxs;
    return s < 1 ? 1 : s;
}

// [x, sin(f0 x), cos(f0 x), sin(f1 x), ...] (posenc, models/vanilla.py:60-79) or [x, sin(x B^T), cos(x B^T)] (rotate, :83-89),
// then zeros up to `ld`.  One thread per output element.
template <bool HALF>                                                    // HALF: out is fp16 of 32 x value (the operand of nm_wgrad16); ones_col >= 0: that column is 1
__global__ __launch_bounds__(256) void pe_encode_kernel(const float* __restrict__ x, int64_t n, int D, int kind, int nfreq,
                                                        const float* __restrict__ tab, float* __restrict__ out, int ld, int ones_col = -1) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * ld) return;
    const int64_t r = i / ld;
    const int p = (int)(i % ld);
    float v = 0.f;
    if (p < D) v = x[r * D + p];
    else if (p - D < 2 * D * nfreq) {
        const int m = p - D;
        if (kind == NM_PE_POSENC) {                                  // D = 3 (points, directions) or 4 (point + time: the offset net)
            const int b = m / (2 * D), q = m - 2 * D * b, dim = q >= D ? q - D : q;
            const float a = x[r * D + dim] * tab[b];
            v = q >= D ? cosf(a) : sinf(a);
        } else {
            const float x0 = x[r * 3], x1 = x[r * 3 + 1], x2 = x[r * 3 + 2];
            const int n3 = 3 * nfreq;
            const bool is_cos = m >= n3;
            const float* b = tab + 3 * (is_cos ? m - n3 : m);
            const float a = fmaf(x2, b[2], fmaf(x1, b[1], x0 * b[0]));
            v = is_cos ? cosf(a) : sinf(a);
        }
    }
    if (HALF && p == ones_col) v = 1.f;
    if (HALF) reinterpret_cast<_Float16*>(out)[i] = (_Float16)(v * kNmAct16Scale);
    else out[i] = v;
}

// adjoint of pe_encode_kernel: dx [n,3] from the gradient g [n,ld] of the encoded features, one thread per row
constexpr int kPeRows = 128;
__global__ __launch_bounds__(kPeRows) void pe_backward_kernel(const float* __restrict__ x, int64_t n, int D, int kind, int nfreq,
                                                              const float* __restrict__ tab, const float* __restrict__ g, int ld, float* __restrict__ dx) {
    // a workgroup takes 128 rows: their gradient rows go through LDS with coalesced loads (row stride ld + 1: a thread walking its own
    // row then hits every bank once per wave), one sincos per (row, band, coordinate) instead of a sin and a cos
    extern __shared__ float tile[];                                  // [128][ld + 1]
    const int64_t r0 = (int64_t)blockIdx.x * kPeRows;
    const int rows = (int)(n - r0 < kPeRows ? n - r0 : kPeRows);
    const int lds = ld + 1;
    for (int i = threadIdx.x; i < rows * ld; i += kPeRows) tile[(i / ld) * lds + i % ld] = g[r0 * ld + i];
    __syncthreads();
    if ((int)threadIdx.x >= rows) return;
    const int64_t r = r0 + threadIdx.x;
    float xv[4] = {0.f, 0.f, 0.f, 0.f}, d[4] = {0.f, 0.f, 0.f, 0.f};
    const float* gr = tile + threadIdx.x * lds;
    for (int k = 0; k < D; ++k) { xv[k] = x[r * D + k]; d[k] = gr[k]; }
    if (kind == NM_PE_POSENC) {
        for (int b = 0; b < nfreq; ++b) {
            const float f = tab[b];
            for (int k = 0; k < D; ++k) {
                float sn, cs;
                sincosf(xv[k] * f, &sn, &cs);
                d[k] += f * (gr[D + 2 * D * b + k] * cs - gr[D + 2 * D * b + D + k] * sn);
            }
        }
    } else {
        const int n3 = 3 * nfreq;
        for (int m = 0; m < n3; ++m) {
            const float* b = tab + 3 * m;
            float sn, cs;
            sincosf(fmaf(xv[2], b[2], fmaf(xv[1], b[1], xv[0] * b[0])), &sn, &cs);
            const float t = gr[3 + m] * cs - gr[3 + n3 + m] * sn;
#pragma unroll
            for (int k = 0; k < 3; ++k) d[k] += b[k] * t;
        }
    }
    for (int k = 0; k < D; ++k) dx[r * D + k] = d[k];
}

// d loss / d raw through raw2outputs, one thread per ray; the transmittance scan and its adjoint run in f64.
//   w_i = a_i T_i, T_i = prod_{j<i} u_j, u_j = 1 - a_j + 1e-10, a_i = 1 - exp(-relu(sigma_i) dist_i)
//   G_i = dL/dw_i = g_rgb . c_i + g_acc' + g_depth z_i + g_w_i      (white background: g_acc' = g_acc - sum_k g_rgb_k)
//   dL/da_i = G_i T_i - (sum_{m>i} G_m w_m) / u_i;   da_i/dsigma_i = dist_i (1 - a_i) [sigma_i > 0]
__global__ __launch_bounds__(64) void composite_backward_kernel(const float* __restrict__ raw, const float* __restrict__ z, const float* __restrict__ rays_d,
                                                                int64_t R, int S, int white_bkg, const float* __restrict__ g_rgb,
                                                                const float* __restrict__ g_acc, const float* __restrict__ g_depth,
                                                                const float* __restrict__ g_w, float* __restrict__ d_raw) {
    const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= R) return;
    const float dn = sqrtf(rays_d[r * 3] * rays_d[r * 3] + rays_d[r * 3 + 1] * rays_d[r * 3 + 1] + rays_d[r * 3 + 2] * rays_d[r * 3 + 2]);
    const float gr0 = g_rgb ? g_rgb[r * 3] : 0.f, gr1 = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gr2 = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
    const double ga = (double)(g_acc ? g_acc[r] : 0.f) - (white_bkg ? (double)gr0 + (double)gr1 + (double)gr2 : 0.0);
    const double gd = g_depth ? (double)g_depth[r] : 0.0;
    const float* rw = raw + r * S * 4;
    const float* zz = z + r * S;
    float* dr = d_raw + r * S * 4;
    // forward scan: T_i = prod_{j<i} u_j is kept per sample (as a double in the first 8 bytes of the sample's own d_raw
    // record, which the backward scan overwrites after reading it).  Recovering T_i by dividing T_S back would give 0 / u = 0
    // for every sample in front of a run of saturated ones (u = 1e-10 each: T_S underflows after ~31 of them, or beyond ~709
    // nats of optical depth) -- exactly the rays late training produces.
    {
        double T = 1.0;
        for (int i = 0; i < S; ++i) {
            const float dist = (i + 1 < S ? zz[i + 1] - zz[i] : 1e10f) * dn;
            const float a = 1.f - expf(-fmaxf(rw[i * 4 + 3], 0.f) * dist);
            *reinterpret_cast<double*>(dr + i * 4) = T;
            T *= (double)(1.f - a + 1e-10f);
        }
    }
    // backward scan: carry the suffix sum  sum_{m>i} G_m w_m
    double suffix = 0.0;
    for (int i = S - 1; i >= 0; --i) {
        const float sg = rw[i * 4 + 3];
        const float dist = (i + 1 < S ? zz[i + 1] - zz[i] : 1e10f) * dn;
        const float e = expf(-fmaxf(sg, 0.f) * dist);
        const float a = 1.f - e;
        const double u = (double)(1.f - a + 1e-10f);
        const double T = *reinterpret_cast<const double*>(dr + i * 4);   // T_i
        const double wgt = (double)a * T;
        float c[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) c[k] = 1.f / (1.f + expf(-rw[i * 4 + k]));
        const double G = (double)gr0 * c[0] + (double)gr1 * c[1] + (double)gr2 * c[2] + ga + gd * (double)zz[i] + (g_w ? (double)g_w[r * S + i] : 0.0);
        const double da = G * T - suffix / u;
        suffix += G * wgt;
        dr[i * 4] = (float)(wgt * (double)gr0 * (double)(c[0] * (1.f - c[0])));
        dr[i * 4 + 1] = (float)(wgt * (double)gr1 * (double)(c[1] * (1.f - c[1])));
        dr[i * 4 + 2] = (float)(wgt * (double)gr2 * (double)(c[2] * (1.f - c[2])));
        dr[i * 4 + 3] = sg > 0.f ? (float)(da * (double)dist * (double)e) : 0.f;
    }
}

// The same adjoint with one WAVE per ray (a training batch is a few thousand rays: one lane per ray leaves the GPU idle).  Lane l
// owns the consecutive samples [l c, l c + c), c = ceil(S / 64) <= kCbMax: the transmittance in front of its run is an exclusive
// product scan across the lanes, the sum behind its run an exclusive suffix sum, both in f64 through shuffles; inside the run the
// two recurrences of the serial kernel.  Same formulas, the products / sums associated differently (f64: 1e-16).
constexpr int kCbMax = 16;                                           // S <= 1024 here; longer rays take the serial kernel
__device__ __forceinline__ double shfl_up_f64(double v, int d) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_up((int)(b & 0xffffffffll), d, 64), hi = __shfl_up((int)(b >> 32), d, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double shfl_down_f64(double v, int d) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_down((int)(b & 0xffffffffll), d, 64), hi = __shfl_down((int)(b >> 32), d, 64);
    return __longlong_as_double(((long long)hi << 32) | (unsigned)lo);
}
__global__ __launch_bounds__(256) void composite_backward_wave_kernel(const float* __restrict__ raw, const float* __restrict__ z,
                                                                     const float* __restrict__ rays_d, int64_t R, int S, int white_bkg,
                                                                     const float* __restrict__ g_rgb, const float* __restrict__ g_acc,
                                                                     const float* __restrict__ g_depth, const float* __restrict__ g_w,
                                                                     float* __restrict__ d_raw) {
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (r >= R) return;
    const float dn = sqrtf(rays_d[r * 3] * rays_d[r * 3] + rays_d[r * 3 + 1] * rays_d[r * 3 + 1] + rays_d[r * 3 + 2] * rays_d[r * 3 + 2]);
    const float gr0 = g_rgb ? g_rgb[r * 3] : 0.f, gr1 = g_rgb ? g_rgb[r * 3 + 1] : 0.f, gr2 = g_rgb ? g_rgb[r * 3 + 2] : 0.f;
    const double ga = (double)(g_acc ? g_acc[r] : 0.f) - (white_bkg ? (double)gr0 + (double)gr1 + (double)gr2 : 0.0);
    const double gd = g_depth ? (double)g_depth[r] : 0.0;
    const float* rw = raw + r * S * 4;
    const float* zz = z + r * S;
    float* dr = d_raw + r * S * 4;
    const int c = (S + 63) / 64, i0 = lane * c, i1 = i0 + c < S ? i0 + c : S;
    // this lane's run: u_i and the run's product
    float4 rec[kCbMax];
    float dist[kCbMax], e[kCbMax];
    double P = 1.0;
#pragma unroll
    for (int k = 0; k < kCbMax; ++k) {
        const int i = i0 + k;
        if (k < c && i < i1) {
            rec[k] = reinterpret_cast<const float4*>(rw)[i];
            dist[k] = (i + 1 < S ? zz[i + 1] - zz[i] : 1e10f) * dn;
            e[k] = expf(-fmaxf(rec[k].w, 0.f) * dist[k]);
            P *= (double)(1.f - (1.f - e[k]) + 1e-10f);
        }
    }
    // exclusive product scan over the lanes: T in front of this lane's run
    double incl = P;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = shfl_up_f64(incl, o);
        if (lane >= o) incl *= t;
    }
    double T = shfl_up_f64(incl, 1);
    if (lane == 0) T = 1.0;
    // forward inside the run: T_i, G_i w_i; the run's sum
    double Ti[kCbMax], Gi[kCbMax], Q = 0.0;
#pragma unroll
    for (int k = 0; k < kCbMax; ++k) {
        const int i = i0 + k;
        if (k < c && i < i1) {
            const float a = 1.f - e[k];
            Ti[k] = T;
            float cs[3] = {1.f / (1.f + expf(-rec[k].x)), 1.f / (1.f + expf(-rec[k].y)), 1.f / (1.f + expf(-rec[k].z))};
            Gi[k] = (double)gr0 * cs[0] + (double)gr1 * cs[1] + (double)gr2 * cs[2] + ga + gd * (double)zz[i] + (g_w ? (double)g_w[r * S + i] : 0.0);
            Q += Gi[k] * ((double)a * T);
            T *= (double)(1.f - a + 1e-10f);
        }
    }
    // exclusive suffix sum over the lanes: sum of G w behind this lane's run
    double inclS = Q;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = shfl_down_f64(inclS, o);
        if (lane + o < 64) inclS += t;
    }
    double suffix = shfl_down_f64(inclS, 1);
    if (lane == 63) suffix = 0.0;
    // backward inside the run
#pragma unroll
    for (int k = kCbMax - 1; k >= 0; --k) {
        const int i = i0 + k;
        if (k < c && i < i1) {
            const float a = 1.f - e[k];
            const double u = (double)(1.f - a + 1e-10f);
            const double wgt = (double)a * Ti[k];
            const double da = Gi[k] * Ti[k] - suffix / u;
            suffix += Gi[k] * wgt;
            float cs[3] = {1.f / (1.f + expf(-rec[k].x)), 1.f / (1.f + expf(-rec[k].y)), 1.f / (1.f + expf(-rec[k].z))};
            reinterpret_cast<float4*>(dr)[i] = make_float4((float)(wgt * (double)gr0 * (double)(cs[0] * (1.f - cs[0]))),
                                                         (float)(wgt * (double)gr1 * (double)(cs[1] * (1.f - cs[1]))),
                                                         (float)(wgt * (double)gr2 * (double)(cs[2] * (1.f - cs[2]))),
                                                         rec[k].w > 0.f ? (float)(da * (double)dist[k] * (double)e[k]) : 0.f);
        }
    }
}

}  // namespace

template <int MP, int NQ>
static int wgrad16_launch(const Wgrad16Args& g, int nprod, hipStream_t st) {
    hipLaunchKernelGGL((wgrad16_kernel<MP, NQ>), dim3(g.nsplit, nprod), dim3(512), 0, st, g);
    if (int rc = nm::check_launch("wgrad16_kernel")) return rc;
    hipLaunchKernelGGL((wgrad16_reduce_kernel<MP, NQ>), dim3(MP * NQ / 256, nprod), dim3(256), 0, st, g);
    return nm::check_launch("wgrad16_reduce_kernel");
}
extern "C" {

int64_t nm_gemm_workspace_floats(int M, int N, int K) {
    const int s = pick_splits(M, N, K);
    if (M == 256 && N == 256 && K >= 16384) return (int64_t)(s > kWgradSplits ? s : kWgradSplits) * M * N;     // (wgrad256_kernel's partials)
    return s > 1 ? (int64_t)s * M * N : 0;
}

static int gemm_dispatch(int mode, int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                         int ldc, const float* bias, const float* mask, int ldmask, int flags, float* workspace, int64_t workspace_floats,
                         nm_stream_t stream) {
    NM_REQUIRE(M >= 0 && N >= 0 && K >= 0, "nm_gemm_f32: negative size");
    if (M == 0 || N == 0) return NM_OK;
    NM_REQUIRE(A && B && C, "nm_gemm_f32: null pointer");
    NM_REQUIRE(M % 4 == 0 && N % 4 == 0 && K % 4 == 0 && lda % 4 == 0 && ldb % 4 == 0, "nm_gemm_f32: M, N, K, lda, ldb must be multiples of 4 "
               "(M=%d N=%d K=%d lda=%d ldb=%d): pad with zeros", M, N, K, lda, ldb);
    NM_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0, "nm_gemm_f32: A and B must be 16-byte aligned");
    NM_REQUIRE(lda >= (a_kmajor ? M : K) && ldb >= (b_kmajor ? N : K) && ldc >= N, "nm_gemm_f32: leading dimension too small");
    NM_REQUIRE(!(flags & NM_GEMM_BIAS) || bias, "nm_gemm_f32: NM_GEMM_BIAS without a bias vector");
    NM_REQUIRE(!(flags & NM_GEMM_MASK) || (mask && ldmask >= N), "nm_gemm_f32: NM_GEMM_MASK without a mask array");
    hipStream_t st = nm::as_stream(stream);
    GemmArgs g;
    g.A = A; g.B = B; g.C = C; g.bias = bias; g.mask = mask; g.partial = nullptr; g.colsum = nullptr;
    g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ldmask = ldmask; g.flags = flags;
    if (wgrad256_ok(mode, a_kmajor, b_kmajor, M, N, K, flags)) {          // backward-weights of a 256-wide layer: one 256 x 256 tile per workgroup
        NM_REQUIRE(workspace && workspace_floats >= (int64_t)kWgradSplits * M * N, "nm_gemm: needs %lld floats of workspace (nm_gemm_workspace_floats)",
                   (long long)kWgradSplits * M * N);
        g.partial = workspace;
        g.k_per_split = ((K + kWgradSplits - 1) / kWgradSplits + XK - 1) / XK * XK;
        const int nsplit = (K + g.k_per_split - 1) / g.k_per_split;
        if (mode == 1) hipLaunchKernelGGL((wgrad256_kernel<false>), dim3(nsplit), dim3(512), 0, st, g);
        else hipLaunchKernelGGL((wgrad256_kernel<true>), dim3(nsplit), dim3(512), 0, st, g);
        if (int rc = nm::check_launch("wgrad256_kernel")) return rc;
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, nsplit, M, N, C, ldc,
                           (flags & NM_GEMM_ACCUMULATE) ? 1 : 0);
        return nm::check_launch("splitk_reduce_kernel");
    }
    int splits = pick_splits(M, N, K);
    if (flags & NM_GEMM_COLSUM) {
        NM_REQUIRE(splits == 1, "nm_gemm: NM_GEMM_COLSUM on a split-K product (M=%d N=%d K=%d)", M, N, K);
        NM_REQUIRE(((uintptr_t)C & 15) == 0 && (ldc & 3) == 0 && (!(flags & NM_GEMM_MASK) || ((((uintptr_t)mask & 15) == 0) && (ldmask & 3) == 0)) &&
                       (!(flags & NM_GEMM_BIAS) || ((uintptr_t)bias & 15) == 0),
                   "nm_gemm: NM_GEMM_COLSUM needs 16-byte aligned C / mask / bias rows");
        NM_REQUIRE(workspace && workspace_floats >= (int64_t)((M + 63) / 64) * N, "nm_gemm: NM_GEMM_COLSUM needs %lld floats of workspace",
                   (long long)((M + 63) / 64) * N);
        g.colsum = workspace;
    }
    if (splits > 1) {
        NM_REQUIRE(!(flags & ~NM_GEMM_ACCUMULATE), "nm_gemm_f32: a split-K product (M=%d N=%d K=%d) takes no epilogue but ACCUMULATE", M, N, K);
        NM_REQUIRE(workspace && workspace_floats >= (int64_t)splits * M * N, "nm_gemm_f32: needs %lld floats of workspace (nm_gemm_workspace_floats)",
                   (long long)splits * M * N);
        g.partial = workspace;
    }
    g.k_per_split = ((K + splits - 1) / splits + XK - 1) / XK * XK;
    if (g.k_per_split < XK) g.k_per_split = XK;
    splits = (K + g.k_per_split - 1) / g.k_per_split;
    if (splits < 1) splits = 1;
    const dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, splits);
    if (mode == 1) {
        if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<true, true, false>), grid, dim3(256), 0, st, g);
        else if (a_kmajor) hipLaunchKernelGGL((gemm_split_kernel<true, false, false>), grid, dim3(256), 0, st, g);
        else if (b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<false, true, false>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_split_kernel<false, false, false>), grid, dim3(256), 0, st, g);
    } else if (mode == 2) {
        if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<true, true, true>), grid, dim3(256), 0, st, g);
        else if (a_kmajor) hipLaunchKernelGGL((gemm_split_kernel<true, false, true>), grid, dim3(256), 0, st, g);
        else if (b_kmajor) hipLaunchKernelGGL((gemm_split_kernel<false, true, true>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_split_kernel<false, false, true>), grid, dim3(256), 0, st, g);
    } else {
        if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, dim3(256), 0, st, g);
        else if (a_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, dim3(256), 0, st, g);
        else if (b_kmajor) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, dim3(256), 0, st, g);
    }
    if (int rc = nm::check_launch("gemm kernel")) return rc;
    if (g.partial) {
        const int64_t n = (int64_t)M * N;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, workspace, splits, M, N, C, ldc,
                           (flags & NM_GEMM_ACCUMULATE) ? 1 : 0);
        return nm::check_launch("splitk_reduce_kernel");
    }
    return NM_OK;
}

int nm_gemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                const float* bias, const float* mask, int ldmask, int flags, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    return gemm_dispatch(0, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, ldmask, flags, workspace, workspace_floats, stream);
}

int nm_gemm_bf16x3(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, const float* mask, int ldmask, int flags, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    return gemm_dispatch(1, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, ldmask, flags, workspace, workspace_floats, stream);
}

int nm_gemm_fp16x3(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C, int ldc,
                   const float* bias, const float* mask, int ldmask, int flags, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    return gemm_dispatch(2, a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, bias, mask, ldmask, flags, workspace, workspace_floats, stream);
}

int nm_absmax(const float* x, int64_t count, float* out_max, nm_stream_t stream) {
    NM_REQUIRE(count >= 0 && out_max, "nm_absmax: bad arguments");
    if (count == 0) return NM_OK;
    NM_REQUIRE(x && ((uintptr_t)x & 15) == 0, "nm_absmax: x must be a 16-byte aligned device array");
    int64_t blocks = (count / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, nm::as_stream(stream), x, count, reinterpret_cast<unsigned*>(out_max));
    return nm::check_launch("absmax_kernel");
}

static void wgrad16_split(int nprod, int64_t n, int& k_per_split, int& nsplit) {
    int want = 1024 / (nprod < 1 ? 1 : nprod);                            // ~1024 workgroups over all products: four rounds of the chip, few partials
    if (want < 64) want = 64;
    if (want > 256) want = 256;
    int64_t per = ((n + want - 1) / want + 63) / 64 * 64;
    if (per < 64) per = 64;
    k_per_split = (int)per;
    nsplit = (int)((n + per - 1) / per);
}
int64_t nm_wgrad16_workspace_floats(int nprod, int64_t n, int p_cols, int q_cols) {
    int k, s;
    wgrad16_split(nprod, n, k, s);
    return (int64_t)nprod * s * (p_cols > 128 ? 256 : 128) * (q_cols > 64 ? 256 : 64);
}
int nm_wgrad16(int nprod, int p_cols, int q_cols, const uint16_t* const* dz16, const uint16_t* const* act16, float* const* dW, const int* ldw, int64_t n,
               const float* amax, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    NM_REQUIRE(nprod >= 1 && nprod <= 8 && n >= 1 && n < ((int64_t)1 << 31), "nm_wgrad16: nprod %d (1..8), n %lld", nprod, (long long)n);
    NM_REQUIRE(p_cols == 256 || p_cols == 128, "nm_wgrad16: p_cols %d (256: a trunk layer / feature_linear; 128: the views layer)", p_cols);
    NM_REQUIRE(q_cols == 256 || (q_cols >= 1 && q_cols <= 64), "nm_wgrad16: q_cols %d (256: a hidden operand; <= 64: an encoded input in rows of 64)", q_cols);
    NM_REQUIRE(dz16 && act16 && dW && ldw && amax && workspace, "nm_wgrad16: null pointer");
    const int NQ = q_cols == 256 ? 256 : 64;
    Wgrad16Args g;
    for (int i = 0; i < 8; ++i) { g.P[i] = nullptr; g.Q[i] = nullptr; g.C[i] = nullptr; g.ldc[i] = 0; }
    for (int i = 0; i < nprod; ++i) {
        NM_REQUIRE(dz16[i] && act16[i] && dW[i] && ldw[i] >= q_cols, "nm_wgrad16: product %d: null pointer or ldw < q_cols", i);
        NM_REQUIRE((((uintptr_t)dz16[i] | (uintptr_t)act16[i]) & 15) == 0, "nm_wgrad16: operands must be 16-byte aligned");
        g.P[i] = reinterpret_cast<const uint4*>(dz16[i]); g.Q[i] = reinterpret_cast<const uint4*>(act16[i]); g.C[i] = dW[i]; g.ldc[i] = ldw[i];
    }
    wgrad16_split(nprod, n, g.k_per_split, g.nsplit);
    NM_REQUIRE(workspace_floats >= (int64_t)nprod * g.nsplit * p_cols * NQ, "nm_wgrad16: needs %lld floats of workspace (nm_wgrad16_workspace_floats)",
               (long long)nprod * g.nsplit * p_cols * NQ);
    g.partial = workspace; g.amax = amax; g.n = n; g.ncols = q_cols;
    hipStream_t st = nm::as_stream(stream);
    if (p_cols == 256) return NQ == 256 ? wgrad16_launch<256, 256>(g, nprod, st) : wgrad16_launch<256, 64>(g, nprod, st);
    return NQ == 256 ? wgrad16_launch<128, 256>(g, nprod, st) : wgrad16_launch<128, 64>(g, nprod, st);
}

int64_t nm_wgrad_heads16_workspace_floats(int64_t n) { return ((n + kHeadRows - 1) / kHeadRows) * kHeadOut; }
int nm_wgrad_heads16(const float* d_raw, const uint16_t* h16_7, const float* hv, int64_t n, float* out644, float* amax, float* workspace,
                     int64_t workspace_floats, nm_stream_t stream) {
    NM_REQUIRE(n >= 1 && d_raw && h16_7 && hv && out644 && workspace, "nm_wgrad_heads16: bad arguments");
    NM_REQUIRE((((uintptr_t)d_raw | (uintptr_t)h16_7 | (uintptr_t)hv) & 15) == 0, "nm_wgrad_heads16: inputs must be 16-byte aligned");
    const int bands = (int)((n + kHeadRows - 1) / kHeadRows);
    NM_REQUIRE(workspace_floats >= (int64_t)bands * kHeadOut, "nm_wgrad_heads16: needs %lld floats of workspace", (long long)bands * kHeadOut);
    hipStream_t st = nm::as_stream(stream);
    hipLaunchKernelGGL(wgrad_heads16_kernel, dim3(bands), dim3(256), 0, st, d_raw, reinterpret_cast<const uint4*>(h16_7), hv, n, workspace,
                       reinterpret_cast<unsigned*>(amax));
    if (int rc = nm::check_launch("wgrad_heads16_kernel")) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((kHeadOut + 255) / 256), dim3(256), 0, st, workspace, bands, 1, kHeadOut, out644, kHeadOut, 0);
    return nm::check_launch("splitk_reduce_kernel");
}

int64_t nm_wgrad_out16_workspace_floats(int64_t n) { return ((n + kHeadRows - 1) / kHeadRows) * kOutOut; }
int nm_wgrad_out16(const float* d_out, const uint16_t* h16_7, int64_t n, float* out1028, float* amax, float* workspace, int64_t workspace_floats,
                   nm_stream_t stream) {
    NM_REQUIRE(n >= 1 && d_out && h16_7 && out1028 && workspace, "nm_wgrad_out16: bad arguments");
    NM_REQUIRE((((uintptr_t)d_out | (uintptr_t)h16_7) & 15) == 0, "nm_wgrad_out16: inputs must be 16-byte aligned");
    const int bands = (int)((n + kHeadRows - 1) / kHeadRows);
    NM_REQUIRE(workspace_floats >= (int64_t)bands * kOutOut, "nm_wgrad_out16: needs %lld floats of workspace", (long long)bands * kOutOut);
    hipStream_t st = nm::as_stream(stream);
    hipLaunchKernelGGL(wgrad_out16_kernel, dim3(bands), dim3(256), 0, st, d_out, reinterpret_cast<const uint4*>(h16_7), n, workspace, reinterpret_cast<unsigned*>(amax));
    if (int rc = nm::check_launch("wgrad_out16_kernel")) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((kOutOut + 255) / 256), dim3(256), 0, st, workspace, bands, 1, kOutOut, out1028, kOutOut, 0);
    return nm::check_launch("splitk_reduce_kernel");
}

int64_t nm_wgrad_alpha16_workspace_floats(int64_t n) { return ((n + kAlphaRows - 1) / kAlphaRows) * 256; }
int nm_wgrad_alpha16(const float* d_raw, const uint16_t* h16, int64_t n, float* out, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    NM_REQUIRE(n >= 1 && d_raw && h16 && out && workspace, "nm_wgrad_alpha16: bad arguments");
    const int bands = (int)((n + kAlphaRows - 1) / kAlphaRows);
    NM_REQUIRE(workspace_floats >= (int64_t)bands * 256, "nm_wgrad_alpha16: needs %lld floats of workspace", (long long)bands * 256);
    hipStream_t st = nm::as_stream(stream);
    hipLaunchKernelGGL(wgrad_alpha16_kernel, dim3(bands), dim3(256), 0, st, d_raw, reinterpret_cast<const unsigned*>(h16), n, workspace);
    if (int rc = nm::check_launch("wgrad_alpha16_kernel")) return rc;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(1), dim3(256), 0, st, workspace, bands, 1, 256, out, 256, 0);
    return nm::check_launch("splitk_reduce_kernel");
}

int64_t nm_colsum_workspace_floats(int64_t n, int W) { return ((n + kColsumRows - 1) / kColsumRows) * W; }

int nm_colsum(const float* X, int64_t n, int W, int ld, float* out, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    NM_REQUIRE(n >= 0 && W >= 1 && ld >= W, "nm_colsum: bad sizes n=%lld W=%d ld=%d", (long long)n, W, ld);
    NM_REQUIRE(out && (n == 0 || X), "nm_colsum: null pointer");
    hipStream_t st = nm::as_stream(stream);
    if (n == 0) return nm::check_hip(hipMemsetAsync(out, 0, (size_t)W * 4, st), "nm_colsum: memset");
    const int bands = (int)((n + kColsumRows - 1) / kColsumRows);
    NM_REQUIRE(workspace && workspace_floats >= (int64_t)bands * W, "nm_colsum: needs %lld floats of workspace (nm_colsum_workspace_floats)",
               (long long)bands * W);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(bands), dim3(256), 0, st, X, n, W, ld, workspace);
    if (int rc = nm::check_launch("colsum_partial_kernel")) return rc;
    hipLaunchKernelGGL(colsum_final_kernel, dim3((W + 3) / 4), dim3(256), 0, st, workspace, bands, W, out);
    return nm::check_launch("colsum_final_kernel");
}

int nm_pe_encode(const float* x, int64_t n, int dims, int kind, int n_freqs, const float* table, float* out, int ld, nm_stream_t stream) {
    NM_REQUIRE(dims == 3 || (dims == 4 && kind == NM_PE_POSENC), "nm_pe_encode: dims %d (3, or 4 with the posenc mapping)", dims);
    NM_REQUIRE(n >= 0 && n_freqs >= 0 && ld >= dims + 2 * dims * n_freqs, "nm_pe_encode: bad sizes n=%lld n_freqs=%d ld=%d", (long long)n, n_freqs, ld);
    NM_REQUIRE(kind == NM_PE_POSENC || kind == NM_PE_ROTATE, "nm_pe_encode: mapping %d", kind);
    if (n == 0) return NM_OK;
    NM_REQUIRE(x && out && (table || n_freqs == 0), "nm_pe_encode: null pointer");
    const int64_t total = n * ld;
    hipLaunchKernelGGL(pe_encode_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nm::as_stream(stream), x, n, dims, kind, n_freqs, table, out, ld, -1);
    return nm::check_launch("pe_encode_kernel");
}

int nm_pe_encode16(const float* x, int64_t n, int dims, int kind, int n_freqs, const float* table, uint16_t* out, int ld, int ones_col, nm_stream_t stream) {
    NM_REQUIRE(ones_col < ld && (ones_col < 0 || ones_col >= dims + 2 * dims * n_freqs), "nm_pe_encode16: ones_col %d must lie in the padding of a row", ones_col);
    NM_REQUIRE(dims == 3 || (dims == 4 && kind == NM_PE_POSENC), "nm_pe_encode16: dims %d (3, or 4 with the posenc mapping)", dims);
    NM_REQUIRE(n >= 0 && n_freqs >= 0 && ld >= dims + 2 * dims * n_freqs, "nm_pe_encode16: bad sizes n=%lld n_freqs=%d ld=%d", (long long)n, n_freqs, ld);
    NM_REQUIRE(kind == NM_PE_POSENC || kind == NM_PE_ROTATE, "nm_pe_encode16: mapping %d", kind);
    if (n == 0) return NM_OK;
    NM_REQUIRE(x && out && (table || n_freqs == 0), "nm_pe_encode16: null pointer");
    const int64_t total = n * ld;
    hipLaunchKernelGGL(pe_encode_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, nm::as_stream(stream), x, n, dims, kind, n_freqs, table,
                       reinterpret_cast<float*>(out), ld, ones_col);
    return nm::check_launch("pe_encode_kernel");
}

int nm_pe_backward(const float* x, int64_t n, int dims, int kind, int n_freqs, const float* table, const float* g, int ld, float* dx,
                   nm_stream_t stream) {
    NM_REQUIRE(dims == 3 || (dims == 4 && kind == NM_PE_POSENC), "nm_pe_backward: dims %d (3, or 4 with the posenc mapping)", dims);
    NM_REQUIRE(n >= 0 && n_freqs >= 0 && ld >= dims + 2 * dims * n_freqs, "nm_pe_backward: bad sizes n=%lld n_freqs=%d ld=%d", (long long)n, n_freqs, ld);
    NM_REQUIRE(kind == NM_PE_POSENC || kind == NM_PE_ROTATE, "nm_pe_backward: mapping %d", kind);
    if (n == 0) return NM_OK;
    NM_REQUIRE(x && g && dx && (table || n_freqs == 0), "nm_pe_backward: null pointer");
    NM_REQUIRE(ld <= 120, "nm_pe_backward: ld %d (rows of at most 120 encoded features)", ld);
    hipLaunchKernelGGL(pe_backward_kernel, dim3((unsigned)((n + kPeRows - 1) / kPeRows)), dim3(kPeRows), (size_t)kPeRows * (ld + 1) * sizeof(float), nm::as_stream(stream), x, n, dims, kind,
                       n_freqs, table, g, ld, dx);
    return nm::check_launch("pe_backward_kernel");
}

int nm_composite_backward(const float* raw, const float* z_vals, const float* rays_d, int64_t R, int S, int white_bkg, const float* g_rgb,
                          const float* g_acc, const float* g_depth, const float* g_weights, float* d_raw, nm_stream_t stream) {
    NM_REQUIRE(R >= 0 && S >= 1, "nm_composite_backward: bad sizes R=%lld S=%d", (long long)R, S);
    if (R == 0) return NM_OK;
    NM_REQUIRE(raw && z_vals && rays_d && d_raw, "nm_composite_backward: null pointer");
    if (S <= 64 * kCbMax && (((uintptr_t)raw | (uintptr_t)d_raw) & 15) == 0) {
        hipLaunchKernelGGL(composite_backward_wave_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, nm::as_stream(stream), raw, z_vals, rays_d, R, S,
                           white_bkg, g_rgb, g_acc, g_depth, g_weights, d_raw);
        return nm::check_launch("composite_backward_wave_kernel");
    }
    hipLaunchKernelGGL(composite_backward_kernel, dim3((unsigned)((R + 63) / 64)), dim3(64), 0, nm::as_stream(stream), raw, z_vals, rays_d, R, S,
                       white_bkg, g_rgb, g_acc, g_depth, g_weights, d_raw);
    return nm::check_launch("composite_backward_kernel");
}

}  // extern "C"
