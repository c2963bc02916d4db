// The backward-data chain of the 8 x 256 trunk in ONE kernel: dZ of a 128-sample tile stays in LDS from layer 7 down to layer 0.
//
// What it replaces: the loop of neuman_hip/train.py _MLP.backward that turns the gradient of layer i's pre-activation into layer i - 1's --
// autograd's adjoint of models/vanilla.py:126-131 (h = relu(linear_i(h)); the skip concatenation before layer 5) inside the training steps of
// trainers/vanilla_nerf_trainer.py:45-96 and trainers/human_nerf_trainer.py:382-446 -- seven products [n, 256] x [256, 256] through HBM
// with a column-sum pass each (21 launches per net), as
//
//     dZ_{i-1} = (dZ_i W_i[:, hidden part]) * (H_{i-1} > 0),        db_{i-1} = column sums of dZ_{i-1},        i = 7 .. 1
//
// in nerf_mlp_kernel's frame (mlp.hip): one workgroup of 8 waves per 128-sample tile, dZ in LDS as split bf16 (hi | lo: the arithmetic of
// the GEMM chain's backward products, nm_gemm_bf16x3: range-safe for gradients of any magnitude), wave w owns output features 32 w .. 32 w + 31
// of all 128 samples, the transposed weights stream from L2 as packed MFMA fragments (bwd_pack_kernel: from the LIVE parameters, every call).
// Per stage the epilogue masks with the saved activation (read once, f32), stores the f32 copy the weight-gradient product reads, reduces
// the tile's column sums inside the wave (its 32 features' 128 samples are all its own) and writes the split bf16 operand of the next stage.
// The weight-gradient products stay separate launches: their 256 x 256 accumulators per layer do not fit a workgroup beside this.
#include "mlp_device.h"
#include <stdlib.h>

namespace {

constexpr int kBwdStages = 7;                                       // trunk layers 7 .. 1
constexpr int kBwdSlots = kBwdStages + 1;                           // + the head stage (feature_linear), image slot 0
constexpr int kBwdStageBytes = 8 * 16 * nm::kStepBytes;            // 8 output blocks x 16 k-steps
constexpr int kBwdVBytes = 8 * 8 * nm::kStepBytes;                  // the views-back stage d_feat = d_hv W_views[:, :256]: 8 output blocks x 8 k-steps (K = 128)
constexpr int kBwdVOff = kBwdSlots * kBwdStageBytes;
constexpr int kBwdImageBytes = kBwdVOff + kBwdVBytes;
constexpr int kBwdPadBytes = 4 * nm::kStepBytes;                   // the weight pipeline prefetches two steps past a run's end

struct BwdArgs {
    const uint4* wpack;        // [8][8 blocks][16 steps] fragments, split bf16: slot 0 = feature_linear's W^T, slot 1 + j = layer 7 - j's W^T (hidden columns)
    const float* dz_top;       // [n][256] gradient of layer 7's pre-activation (masked by H7 > 0 already); HEAD: unused
    const float* d_feat;       // HEAD: [n][256] gradient of feature_linear's output (MODE 2: nullable, ADDED to the one the kernel forms), d_raw [n][4] (column 3 = d sigma), w_alpha [256]:
    const float* d_raw;        //   the first stage forms dZ_7 = (d_feat W_f + d sigma w_alpha) * (H_7 > 0) itself
    const float* w_alpha;
    const float* acts;         // [9][n][256] saved post-activation outputs of layers 0..7 (+ feature): stage j masks with acts[6 - j]
    const unsigned* bits;      // nullable: [8][n][8] their signs (nm_mlp_forward_save_bits): read instead of acts, 1/32 of the bytes
    float* dz_out;             // [NS][n][256]: stage s's output (HEAD: dZ_7, dZ_6 .. dZ_0; else dZ_6 .. dZ_0)
    float* colsum;             // [tiles][NS][256] per-tile column sums of dz_out[s]
    int64_t n;
    // 16-bit form (nm_mlp_backward_chain16): dz16 != nullptr replaces dz_out's float32 copies by fp16 of dZ * nm_dz_scale(*amax) in k-slot order
    // (chunk c, element e <-> feature slot_feature(c, e): the order the lanes hold them, and the saved activations' of nm_mlp_forward_save16);
    // layers listed in dz32 also keep a float32 copy (natural order) for the products that still want one
    uint4* dz16;               // [NS][n][32]
    uint4* dfeat16;            // HEAD, nullable: d_feat the same way, [n][32]
    const float* amax;
    float* dz32[8];            // by layer, nullable each
    // the whole backward pass of the net from d_raw (MODE 2, nm_mlp_backward_net16): the views layer's adjoint is formed in the kernel --
    //     d_hv = (d_rgb W_rgb) * (hv > 0)   [128]   ->   d_feat = d_hv W_views[:, :256]   ->   the chain above
    const unsigned* hvbits;    // [n][4] signs of the views layer's output (nm_mlp_forward_save16)
    const float* w_rgb;        // rgb_linear.weight [3][128]
    uint4* dhv16;              // [n][16] fp16 of d_hv * scale, k-slot order of a 128-wide row
    float* dhv32;              // nullable: [n][128] float32, natural order (the view-direction gradient's product)
};

__device__ __forceinline__ unsigned pack_f16(float a, float b, float sc) {
    f32x2 v = {__builtin_amdgcn_fmed3f(a * sc, -65504.f, 65504.f), __builtin_amdgcn_fmed3f(b * sc, -65504.f, 65504.f)};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2));
}

// HALF (modes 2 and 3, the 16-bit form only): the operand of every stage is dZ * s as ONE fp16 number -- the very value the stage stores for the
// weight-gradient products -- against split-fp16 weights (W^T * 2^8 as hi + lo, bwd_pack_kernel): two MFMAs per k-step instead of three and half the LDS
// operand traffic.  What is propagated is then exactly what is stored: the step's weight gradients are the gradients of ONE chain of fp16-rounded dZ's
// (float32 accumulation everywhere), instead of float32-class dZ's whose fp16 copies feed the products.  Accumulators carry 2^8 s dZ; `amax` scales as before.
template <int MODE, bool HALF = false>                              // 0: from dz_top = dZ_7; 1: from d_feat / d_raw; 2: from d_raw alone (the views layer's adjoint too);
                                                                    // 3: the plain-head net from d_out [n][4]: stage 0 = d_out W_out (K = 4, as two k-steps), layers 7 .. 0
__global__ __launch_bounds__(kThreads, 2) void nerf_mlp_bwd_kernel(const BwdArgs a) {
    constexpr bool HEAD = MODE >= 1, NET = MODE == 2, PLAIN = MODE == 3;
    constexpr int NS = HEAD ? kBwdSlots : kBwdStages;               // stages of a tile; stage s reads image slot s + (HEAD ? 0 : 1)
    static_assert(!HALF || NET || PLAIN, "the two-MFMA form exists for the whole-pass modes");
    __shared__ uint4 lds[LDS_U4];
    constexpr int PREC = HALF ? kPrecF16W2 : NM_PREC_BF16X3;
    constexpr float kWScale = 256.f;                                 // HALF: the image holds W^T * 2^8 (mlp_device.h: lo parts of small weights stay normal numbers)
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, s = lane & 31;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(a.wpack), 0, kBwdImageBytes + kBwdPadBytes, 0x00020000);
    const int voff = lane * 16;
    auto wo = [](int j, int blk) { return (j + (HEAD ? 0 : 1)) * kBwdStageBytes + blk * 16 * nm::kStepBytes; };
    const int64_t ntiles = (a.n + kTileM - 1) / kTileM;
    const float sc16 = a.dz16 ? nm_dz_scale(*a.amax) : 1.f;
    const float pk_sc = HALF ? 1.f : sc16;                           // HALF: the values in flight are s dZ already
    const float out_sc = HALF ? 1.f / sc16 : 1.f;                    //       and float32 results (copies, column sums) are divided by s (a power of two: exact)
    auto unscale = [&](f32x16 (&acc)[4]) {                           // 2^8 s dZ -> s dZ
        if (HALF) {
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] *= (1.f / kWScale);
        }
    };
    const int vo = kBwdVOff + w * 8 * nm::kStepBytes;               // this wave's block of the views-back stage
    const int first = NET ? vo : wo(0, w);                          // the first run of a tile
    WPre W;
    w_prefetch<PREC>(W, wsrc, voff, first);
#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileM;
        if (NET) {
            // ---- d_hv of the tile: item = (chunk c of the 128-wide row, sample): 8 features slot_feature(c, e), d_hv = sum_k d_rgb[k] W_rgb[k][f], masked
#pragma unroll 1
            for (int item = tid; item < 16 * kTileM; item += kThreads) {
                const int c = item >> 7, row = item & (kTileM - 1);
                const int64_t i = base + row;
                const int f0 = nm::slot_feature(c, 0);                  // features f0 .. f0 + 3 and f0 + 8 .. f0 + 11
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (i < a.n) {
                    const float4 dr = *reinterpret_cast<const float4*>(a.d_raw + i * 4);
                    const unsigned byte = (a.hvbits[i * 4 + (c >> 2)] >> (16 * (c & 1) + 8 - 8 * ((c >> 1) & 1))) & 0xffu;   // element e = bit 7 - e
                    const float d3[3] = {dr.x, dr.y, dr.z};
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        const float4 wa = *reinterpret_cast<const float4*>(a.w_rgb + k * 128 + f0), wb = *reinterpret_cast<const float4*>(a.w_rgb + k * 128 + f0 + 8);
                        v[0] = fmaf(d3[k], wa.x, v[0]); v[1] = fmaf(d3[k], wa.y, v[1]); v[2] = fmaf(d3[k], wa.z, v[2]); v[3] = fmaf(d3[k], wa.w, v[3]);
                        v[4] = fmaf(d3[k], wb.x, v[4]); v[5] = fmaf(d3[k], wb.y, v[5]); v[6] = fmaf(d3[k], wb.z, v[6]); v[7] = fmaf(d3[k], wb.w, v[7]);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = ((byte >> (7 - e)) & 1u) ? v[e] : 0.f;
                    a.dhv16[i * 16 + c] = make_uint4(pack_f16(v[0], v[1], sc16), pack_f16(v[2], v[3], sc16), pack_f16(v[4], v[5], sc16), pack_f16(v[6], v[7], sc16));
                    if (a.dhv32) {
                        *reinterpret_cast<float4*>(a.dhv32 + i * 128 + f0) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(a.dhv32 + i * 128 + f0 + 8) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                }
                if (HALF) {
                    lds[H_BASE + c * kChunkU4 + row] = make_uint4(pack_f16(v[0], v[1], sc16), pack_f16(v[2], v[3], sc16), pack_f16(v[4], v[5], sc16), pack_f16(v[6], v[7], sc16));
                } else {
                    uint4 hi, lo;
                    split8<false, false>(v, hi, lo);
                    lds[H_BASE + c * kChunkU4 + row] = hi;
                    lds[H_BASE + c * kChunkU4 + kLoU4 + row] = lo;
                }
            }
            __syncthreads();
            // ---- views-back stage: d_feat [32 w .. 32 w + 31][128 samples] = W_views[:, :256]^T block x d_hv (K = 128), on top of a.d_feat when given:
            // what a second evaluation of the views head on the same features (train.py's second view) sends back to them
            f32x16 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int64_t row = base + 32 * mb + s;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (a.d_feat && row < a.n) t = *reinterpret_cast<const float4*>(a.d_feat + row * 256 + 32 * w + 4 * g + 8 * q);
                    const float in_sc = HALF ? kWScale * sc16 : 1.f;
                    acc[mb][4 * q] = t.x * in_sc; acc[mb][4 * q + 1] = t.y * in_sc; acc[mb][4 * q + 2] = t.z * in_sc; acc[mb][4 * q + 3] = t.w * in_sc;
                }
            }
            k_run<4, PREC>(acc, W, wsrc, voff, vo, wo(0, w), lds + H_BASE + g * kChunkU4 + s, 8);
            unscale(acc);
            uint4 nx[4][2];                                              // HALF: the next stage's operand = the stored fp16 values
            float cs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[r] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int64_t row = base + 32 * mb + s;
                const bool live = row < a.n;
                unsigned pk[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    pk[2 * q] = pack_f16(acc[mb][4 * q], acc[mb][4 * q + 1], pk_sc);
                    pk[2 * q + 1] = pack_f16(acc[mb][4 * q + 2], acc[mb][4 * q + 3], pk_sc);
#pragma unroll
                    for (int j = 0; j < 4; ++j) cs[4 * q + j] += live ? acc[mb][4 * q + j] : 0.f;
                }
                nx[mb][0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                nx[mb][1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                if (live && a.dfeat16) {
                    uint4* o = a.dfeat16 + row * 32 + 4 * w + g;
                    o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    o[2] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = cs[r];
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                cs[r] = v * out_sc;
            }
            if (s == 0) {                                               // feature_linear's bias gradient: column NS of the per-tile sums
                float* o = a.colsum + ((int64_t)tile * (NS + 1) + NS) * 256 + 32 * w + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(o + 8 * q) = make_float4(cs[4 * q], cs[4 * q + 1], cs[4 * q + 2], cs[4 * q + 3]);
            }
            if (HALF) {
                __syncthreads();                                       // every wave has finished reading d_hv
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                    for (int qp = 0; qp < 2; ++qp) lds[H_BASE + (4 * w + 2 * qp + g) * kChunkU4 + 32 * mb + s] = nx[mb][qp];
            } else {
                ActRegs<4> ar;
                convert_act<4, false, NM_PREC_BF16X3>(acc, ar);
                __syncthreads();                                       // every wave has finished reading d_hv
                write_act<4, NM_PREC_BF16X3>(ar, lds, w, 0, g, s);
            }
            __syncthreads();
        } else {
        // ---- dZ_7 of the tile -> split bf16 in LDS, k-slot order (chunk c, element e) = feature slot_feature(c, e): two runs of 4 features
#pragma unroll 1
        for (int item = tid; item < (PLAIN ? 4 : nm::kHChunks) * kTileM; item += kThreads) {
            const int c = item >> 7, row = item & (kTileM - 1);
            const int64_t i = base + row;
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (PLAIN) {                                              // k-slots 0..3 of chunk 0 = d_out's four columns; the rest of the two k-steps zero
                if (c == 0 && i < a.n) {
                    const float4 dr = *reinterpret_cast<const float4*>(a.d_raw + i * 4);
                    v[0] = dr.x; v[1] = dr.y; v[2] = dr.z; v[3] = dr.w;
                }
            } else if (i < a.n) {
                const float* src = (HEAD ? a.d_feat : a.dz_top) + i * 256 + nm::slot_feature(c, 0);
                const float4 lo4 = *reinterpret_cast<const float4*>(src), hi4 = *reinterpret_cast<const float4*>(src + 8);
                v[0] = lo4.x; v[1] = lo4.y; v[2] = lo4.z; v[3] = lo4.w; v[4] = hi4.x; v[5] = hi4.y; v[6] = hi4.z; v[7] = hi4.w;
            }
            if (HALF) {
                lds[H_BASE + c * kChunkU4 + row] = make_uint4(pack_f16(v[0], v[1], sc16), pack_f16(v[2], v[3], sc16), pack_f16(v[4], v[5], sc16), pack_f16(v[6], v[7], sc16));
            } else {
                uint4 hi, lo;
                split8<false, false>(v, hi, lo);
                lds[H_BASE + c * kChunkU4 + row] = hi;
                lds[H_BASE + c * kChunkU4 + kLoU4 + row] = lo;
            }
            if (HEAD && a.dfeat16 && i < a.n)                         // (v[e] = feature slot_feature(c, e): the chunk as it stands)
                a.dfeat16[i * 32 + c] = make_uint4(pack_f16(v[0], v[1], sc16), pack_f16(v[2], v[3], sc16), pack_f16(v[4], v[5], sc16), pack_f16(v[6], v[7], sc16));
        }
        __syncthreads();
        }
#pragma unroll 1
        for (int j = 0; j < NS; ++j) {
            f32x16 acc[4];
#pragma unroll
            for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mb][r] = 0.f;
            k_run<4, PREC>(acc, W, wsrc, voff, wo(j, w), j + 1 < NS ? wo(j + 1, w) : first, lds + H_BASE + g * kChunkU4 + s, (PLAIN && j == 0) ? 2 : 16);
            unscale(acc);
            uint4 nx[4][2];
            // ---- mask with the saved activation of layer 6 - j, store the f32 copy, column sums
            const int layer = NS - 1 - j;                                // the layer whose saved output masks this stage's result (HEAD, j = 0: 7)
            const float* mask = a.acts + (int64_t)layer * a.n * 256;
            const unsigned* mbits = a.bits ? a.bits + (int64_t)layer * a.n * 8 + w : nullptr;
            float wa[16];
            if (HEAD && !PLAIN && j == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 t = *reinterpret_cast<const float4*>(a.w_alpha + 32 * w + 4 * g + 8 * q);
                    wa[4 * q] = t.x; wa[4 * q + 1] = t.y; wa[4 * q + 2] = t.z; wa[4 * q + 3] = t.w;
                }
            }
            float* out = a.dz16 ? a.dz32[layer] : a.dz_out + (int64_t)j * a.n * 256;   // (16-bit form: float32 only where asked for)
            uint4* out16 = a.dz16 ? a.dz16 + (int64_t)j * a.n * 32 : nullptr;
            float cs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) cs[r] = 0.f;
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const int64_t row = base + 32 * mb + s;
                const bool live = row < a.n;
                const int64_t off = (live ? row : 0) * 256 + 32 * w + 4 * g;
                if (HEAD && !PLAIN && j == 0) {                           // + d sigma x alpha_linear's row (models/vanilla.py:133)
                    const float ds = live ? a.d_raw[row * 4 + 3] * (HALF ? sc16 : 1.f) : 0.f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mb][r] = fmaf(ds, wa[r], acc[mb][r]);
                }
                const unsigned word = (mbits && live) ? mbits[row * 8] >> (16 * g) : 0u;    // bit 15 - (4 q + j) of it: register 4 q + j of this lane
                unsigned pk[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bool k0, k1, k2, k3;
                    if (mbits) {
                        k0 = (word >> (15 - 4 * q)) & 1u; k1 = (word >> (14 - 4 * q)) & 1u; k2 = (word >> (13 - 4 * q)) & 1u; k3 = (word >> (12 - 4 * q)) & 1u;
                    } else {
                        const float4 m = *reinterpret_cast<const float4*>(mask + off + 8 * q);
                        k0 = live && m.x > 0.f; k1 = live && m.y > 0.f; k2 = live && m.z > 0.f; k3 = live && m.w > 0.f;
                    }
                    float4 v = make_float4(acc[mb][4 * q], acc[mb][4 * q + 1], acc[mb][4 * q + 2], acc[mb][4 * q + 3]);
                    v.x = k0 ? v.x : 0.f;
                    v.y = k1 ? v.y : 0.f;
                    v.z = k2 ? v.z : 0.f;
                    v.w = k3 ? v.w : 0.f;
                    acc[mb][4 * q] = v.x; acc[mb][4 * q + 1] = v.y; acc[mb][4 * q + 2] = v.z; acc[mb][4 * q + 3] = v.w;
                    if (live && out) *reinterpret_cast<float4*>(out + off + 8 * q) = HALF ? make_float4(v.x * out_sc, v.y * out_sc, v.z * out_sc, v.w * out_sc) : v;
                    pk[2 * q] = pack_f16(v.x, v.y, pk_sc);
                    pk[2 * q + 1] = pack_f16(v.z, v.w, pk_sc);
                    cs[4 * q] += v.x; cs[4 * q + 1] += v.y; cs[4 * q + 2] += v.z; cs[4 * q + 3] += v.w;
                }
                nx[mb][0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                nx[mb][1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                if (live && out16) {                                      // registers 0..7 = chunk 4 w + g, 8..15 = chunk 4 w + 2 + g of the row
                    uint4* o = out16 + row * 32 + 4 * w + g;
                    o[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
                    o[2] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {                         // over the 32 samples of a lane half (fixed order: deterministic)
                float v = cs[r];
#pragma unroll
                for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
                cs[r] = v * out_sc;
            }
            if (s == 0) {
                float* o = a.colsum + ((int64_t)tile * (NS + (NET ? 1 : 0)) + j) * 256 + 32 * w + 4 * g;
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(o + 8 * q) = make_float4(cs[4 * q], cs[4 * q + 1], cs[4 * q + 2], cs[4 * q + 3]);
            }
            if (j + 1 < NS) {
                if (HALF) {
                    __syncthreads();                               // every wave has finished reading this stage's operand
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
                        for (int qp = 0; qp < 2; ++qp) lds[H_BASE + (4 * w + 2 * qp + g) * kChunkU4 + 32 * mb + s] = nx[mb][qp];
                } else {
                    ActRegs<4> ar;
                    convert_act<4, false, NM_PREC_BF16X3>(acc, ar);
                    __syncthreads();                               // every wave has finished reading this stage's operand
                    write_act<4, NM_PREC_BF16X3>(ar, lds, w, 0, g, s);
                }
            }
            __syncthreads();
        }
    }
}

// the transposed hidden weights of layers 7 .. 1 as MFMA A-operand fragments (mlp_layout.h: lane (g, s) of k-step t of output block nb holds
// output feature 32 nb + s, k-slots (chunk 2 t + g, e = 0 .. 7)), split bf16: stage j multiplies dZ of layer i = 7 - j, so its "output
// feature" is an INPUT feature of layer i (skip layer 5: behind the encoding columns) and its k index an OUTPUT feature of layer i
// half: split fp16 of W^T * 2^8 (nerf_mlp_bwd_kernel<., true>) instead of split bf16 of W^T
__global__ __launch_bounds__(256) void bwd_pack_kernel(nm::DevParams P, int kpe, int kdir, int mode, int half, uint8_t* __restrict__ img) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int step = gid >> 6, lane = gid & 63;
    if (step >= kBwdSlots * 8 * 16 + 8 * 8) return;
    const float* Wi;
    int K, col, t;
    if (step >= kBwdSlots * 8 * 16) {                                   // the views-back stage (mode 2): W_views [128][256 + kdir], its feature columns
        if (mode != 2) return;
        const int v = step - kBwdSlots * 8 * 16;
        t = v % 8;
        Wi = P.p[nm::P_VIEWS_W]; K = 256 + kdir; col = 32 * (v / 8) + (lane & 31);
    } else {
        const int slot = step / 128, nb = (step % 128) / 16;
        t = step % 16;
        if (slot == 0 && mode < 1) return;
        if (slot == 0 && mode == 3) {                                   // plain head: stage 0 multiplies d_out [4] by output_linear.weight [4][256]: k-slot e < 4 of chunk 0 only
            if (t >= 2) return;
            const float* Wo = P.p[nm::P_OUT_W];
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (2 * t + (lane >> 5) == 0 && e < 4) ? Wo[e * 256 + 32 * nb + (lane & 31)] : 0.f;
            uint4 hi, lo;
            if (half) split8<false, true>(v, hi, lo, 256.f);
            else split8<false, false>(v, hi, lo);
            uint4* dst = reinterpret_cast<uint4*>(img + (int64_t)step * nm::kStepBytes) + lane;
            dst[0] = hi;
            dst[64] = lo;
            return;
        }
        const int i = 8 - slot;                                         // layer; slot 0: feature_linear
        Wi = slot == 0 ? P.p[nm::P_FEAT_W] : P.p[nm::P_PTS_W + 2 * i];
        K = i == 5 ? kpe + 256 : 256; col = (i == 5 ? kpe : 0) + 32 * nb + (lane & 31);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = Wi[(int64_t)nm::slot_feature(2 * t + (lane >> 5), e) * K + col];
    uint4 hi, lo;
    if (half) split8<false, true>(v, hi, lo, 256.f);
    else split8<false, false>(v, hi, lo);
    uint4* dst = reinterpret_cast<uint4*>(img + (int64_t)step * nm::kStepBytes) + lane;
    dst[0] = hi;
    dst[64] = lo;
}

// gb[j][f] = sum over tiles of colsum[tile][j][f]: 32 columns per workgroup, 32 interleaved row groups, fixed order
__global__ __launch_bounds__(1024) void bwd_colsum_kernel(const float* __restrict__ colsum, int64_t ntiles, int width, float* __restrict__ gb) {
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5, c = blockIdx.x * 32 + cx;
    float acc = 0.f;
    for (int64_t t = ry; t < ntiles; t += 32) acc += colsum[t * width + c];
    __shared__ float part[32][33];
    part[ry][cx] = acc;
    __syncthreads();
    if (ry == 0) {
        float v = part[0][cx];
#pragma unroll
        for (int k = 1; k < 32; ++k) v += part[k][cx];
        gb[c] = v;
    }
}

}  // namespace

namespace nm {

int64_t mlp_bwd_image_bytes() { return kBwdImageBytes + kBwdPadBytes; }

int launch_mlp_bwd(const DevParams& P, int kpe, uint8_t* image, const float* dz_top, const float* d_feat, const float* d_raw, const float* acts,
                   const unsigned* relu_bits, int64_t n, float* dz_out, float* colsum, float* gb, hipStream_t stream, const Bwd16* h) {
    const bool net = h && h->hvbits;                                   // the whole backward pass from d_raw
    const bool plain = h && h->plain;
    const bool head = d_feat != nullptr || net || plain;
    const int mode = plain ? 3 : (net ? 2 : (head ? 1 : 0));
    // the two-MFMA form (single fp16 dZ against split-fp16 weights) is OPT-IN (NEUMAN_BWD_HALF=1): it takes 7 % off a training iteration (measured, round 6:
    // 9.6 -> 8.9 ms) but what it propagates is dZ rounded to fp16 at EVERY layer -- 2^-12 per element, and a gradient that is itself a cancelling sum does not
    // average that away: 3e-4 of the largest entry in tests/test_hip_train16.py's float64 comparison, against the 2e-5 the split-bf16 x3 default holds
    static const bool half_ok = [] { const char* e = getenv("NEUMAN_BWD_HALF"); return e && e[0] == '1'; }();
    const bool half = half_ok && (net || plain) && h && h->dz16 && h->amax;
    hipLaunchKernelGGL(bwd_pack_kernel, dim3(((kBwdSlots * 8 * 16 + 8 * 8) * 64 + 255) / 256), dim3(256), 0, stream, P, kpe, h ? h->kdir : 0, mode, half ? 1 : 0, image);
    BwdArgs a;
    a.wpack = reinterpret_cast<const uint4*>(image);
    a.dz_top = dz_top; a.d_feat = d_feat; a.d_raw = d_raw; a.w_alpha = P.p[P_ALPHA_W];
    a.acts = acts; a.bits = relu_bits; a.dz_out = dz_out; a.colsum = colsum; a.n = n;
    a.dz16 = h ? reinterpret_cast<uint4*>(h->dz16) : nullptr;
    a.dfeat16 = h ? reinterpret_cast<uint4*>(h->dfeat16) : nullptr;
    a.amax = h ? h->amax : nullptr;
    for (int i = 0; i < 8; ++i) a.dz32[i] = h ? h->dz32[i] : nullptr;
    a.hvbits = net ? h->hvbits : nullptr; a.w_rgb = P.p[P_RGB_W];
    a.dhv16 = net ? reinterpret_cast<uint4*>(h->dhv16) : nullptr; a.dhv32 = net ? h->dhv32 : nullptr;
    const int64_t ntiles = (n + kTileM - 1) / kTileM;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    const int ns = (head ? kBwdSlots : kBwdStages) + (net ? 1 : 0);     // rows of the bias-gradient block: the stages (+ feature_linear's)
    if (plain && half) hipLaunchKernelGGL((nerf_mlp_bwd_kernel<3, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (net && half) hipLaunchKernelGGL((nerf_mlp_bwd_kernel<2, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (plain) hipLaunchKernelGGL(nerf_mlp_bwd_kernel<3>, dim3(grid), dim3(kThreads), 0, stream, a);
    else if (net) hipLaunchKernelGGL(nerf_mlp_bwd_kernel<2>, dim3(grid), dim3(kThreads), 0, stream, a);
    else if (head) hipLaunchKernelGGL(nerf_mlp_bwd_kernel<1>, dim3(grid), dim3(kThreads), 0, stream, a);
    else hipLaunchKernelGGL(nerf_mlp_bwd_kernel<0>, dim3(grid), dim3(kThreads), 0, stream, a);
    hipLaunchKernelGGL(bwd_colsum_kernel, dim3(ns * 256 / 32), dim3(1024), 0, stream, colsum, ntiles, ns * 256, gb);
    return check_launch("nerf_mlp_bwd_kernel");
}

}  // namespace nm
