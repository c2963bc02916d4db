// Fused positional-encoding + 8x256 NeRF MLP for gfx950 (MI355X), the roofline kernel of the path.
// Replaces reference models/vanilla.py Embedder.forward (:82-92), NeRF.forward (:120-152) and
// Joiner.forward (:162-166); optionally also ray_to_samples' point construction (ray_utils.py:131).
//
// Design (see DESIGN.md "K4"):
//   * one workgroup = 8 waves (512 threads) = one tile of 128 samples; grid-stride over tiles
//     (persistent: the 2.3 MB split-bf16 weight image stays resident in every XCD's 4 MB L2);
//   * activations never leave the CU: they sit in LDS as split bf16 (hi | lo arrays, 128 KB for the
//     256-wide hidden state + 32 KB for the position / direction encodings = all 160 KB of the CU);
//   * each layer is D[feature][sample] = W * X on v_mfma_f32_32x32x16_bf16; wave w owns output
//     features 32w..32w+31 for all 128 samples (4 accumulator tiles = 64 VGPRs), so every weight
//     fragment is fetched by exactly one wave, straight from L2 into VGPRs in MFMA A-operand order
//     (pre-packed on the host: one coalesced 1 KB load per wave instruction, no LDS staging);
//   * parity mode (NM_PREC_BF16X3): x = xh + xl, w = wh + wl, acc += wh*xl + wl*xh + wh*xh in f32
//     (three MFMAs; the dropped wl*xl term is <= 2^-16 relative); NM_PREC_BF16 issues only wh*xh;
//   * the epilogue adds nothing (bias is the accumulator's initial value), applies ReLU, splits to
//     hi/lo with v_cvt_pk_bf16_f32 and writes one ds_write_b128 per 8 features in exactly the k-slot
//     order the next layer's packed weights expect (mlp_layout.h).
#include "mlp_device.h"
#include <stdlib.h>

namespace {


// SAVE 1: float32 copies of the activations (nm_mlp_forward_save); 2: fp16 copies of the trunk's (nm_mlp_forward_save16).  (Round 6 tried issuing each stage's
// copies one stage later, read back from LDS after the next k-loop, on the theory that the stores' acknowledgements stall the weight prefetch through the shared
// in-order counter: the launch took 1578.0 / 1580.2 us against 1573.8 / 1582.9 -- no effect; profiles/r06_train_experiments.md.)
template <int PREC, bool PROF, int SAVE = 0>
__global__ __launch_bounds__(kThreads, 2) void nerf_mlp_kernel(const MlpArgs a_in) {
    const MlpArgs a = resolve_args(a_in);
    unsigned long long pr[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = PROF ? __builtin_readcyclecounter() : 0;
#define NM_TICK(b)                                                   \
    if (PROF) {                                                      \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        pr[b] += t_now - t_prev;                                     \
        t_prev = t_now;                                              \
    }
    __shared__ uint4 lds[LDS_U4];
    constexpr bool F16 = PREC == NM_PREC_FP16X3;
    // accumulators of the fp16 mode carry Y * 2^(k_stage + 5): per-stage factors 2^-k (-> stored activations) and 2^-(k+5)
    // (-> outputs) follow the bias table of that mode's image (exact to undo; wave-uniform scalar loads)
    const float* f16tab = a.bias + nm::kBiasFloats;
    auto acc2act = [&](int st) { return F16 ? f16tab[st] : 1.f; };
    auto acc2out = [&](int st) { return F16 ? f16tab[nm::kStages + st] : 1.f; };
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, s = lane & 31;
    // SAVE: a block's accumulators as float32 activations, natural feature order: lane (g, s) holds features 32 blk + 8 q + 4 g + j of
    // sample row0 + 32 mb + s -- one 16-byte store per (mb, q); the two lane halves of a sample write adjacent 16 bytes
    auto save_block = [&](float* dst, int ld, const f32x16& acc, int blk, int64_t row, float scale, bool relu) {
        if (row >= a.n) return;
        float* o = dst + row * ld + 32 * blk + 4 * g;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float4 v = make_float4(acc[4 * q] * scale, acc[4 * q + 1] * scale, acc[4 * q + 2] * scale, acc[4 * q + 3] * scale);
            if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
            *reinterpret_cast<float4*>(o + 8 * q) = v;
        }
    };
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(a.wpack), 0, (int)(nm::kWeightBytes + nm::kWeightPadBytes), 0x00020000);
    const int voff = lane * 16;                                   // the only per-lane part of a weight address
    const int64_t ntiles = (a.n + kTileM - 1) / kTileM;

    // wave-uniform offsets of this wave's weight streams (bytes into the image)
    auto wo = [](int st, int blk) { return (int)nm::stage_w_off(st) + blk * nm::stage_shape(st).steps * nm::kStepBytes; };
    const int so_s0 = wo(0, w);
    const int so_s8a = wo(8, 8), so_s9 = wo(9, w & 3), so_s10 = wo(10, 0);
    // pad slots of the encodings (63; 27..31) are never written by the octave path: give them a finite value once
    for (int i = tid; i < nm::kPeChunks * kChunkU4; i += kThreads) lds[P_BASE + i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    WPre W;
    BiasRegs B;
    w_prefetch<PREC>(W, wsrc, voff, so_s0);
    bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);

#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileM;
        // ---------------- position PE -> P
        fill_pe_any<F16>(lds, false, a, base, tid);
        __syncthreads();
        if (SAVE == 2 && F16 && a.save_x0h) {                         // the encoded position as the weight-gradient products of layer 0 / the skip layer read it
            for (int item = tid; item < nm::kPeChunks * kTileM; item += kThreads) {
                const int c = item >> 7, row = item & (kTileM - 1);
                if (base + row < a.n) a.save_x0h[(base + row) * 8 + c] = lds[P_BASE + c * kChunkU4 + row];
            }
        }
        NM_TICK(0)
        if (a.stop_stage == -1) { dump_act<F16>(lds, true, 64, a, base, tid); __syncthreads(); continue; }

        // ---------------- stages 0..7: 256-wide ReLU layers (wave w = output block w, all 4 sample blocks)
        f32x16 acc[4];
        bool stopped = false;
#pragma unroll 1
        for (int st = 0; st <= 7; ++st) {
            const nm::StageShape sh = nm::stage_shape(st);
            init_bias<4>(acc, B);
            const int soff = wo(st, w);
            const int next = (st == 7 && a.sigma_only) ? (w < 4 ? so_s8a : so_s0) : wo(st + 1, w);
            if (sh.pe_steps)
                k_run<4, PREC>(acc, W, wsrc, voff, soff, sh.steps > sh.pe_steps ? soff + sh.pe_steps * nm::kStepBytes : next,
                               lds + P_BASE + g * kChunkU4 + s, sh.pe_steps);
            if (sh.steps > sh.pe_steps)
                k_run<4, PREC>(acc, W, wsrc, voff, soff + sh.pe_steps * nm::kStepBytes, next, lds + H_BASE + g * kChunkU4 + s,
                               sh.steps - sh.pe_steps);
            bias_prefetch(B, a.bias + nm::stage_b_off(st + 1) + 32 * w, g);     // next stage (st + 1 <= 8), block w
            NM_TICK(1)
            if (SAVE) {
                if (SAVE == 1) {
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) save_block(a.save_h + (int64_t)st * a.n * 256, 256, acc[mb], w, base + 32 * mb + s, acc2out(st), true);
                }
                if (a.save_bits) {                                     // one bit per saved activation: (value > 0).  A lane's 16 values MSB first
                    // (bits = 2 bits + carry: a compare and an add-with-carry each); the two lane halves of a sample share a word, half g in
                    // bits 16 g .. 16 g + 15: register r = 4 q + j of half g (feature 32 w + 8 q + 4 g + j) is bit 16 g + 15 - r
#pragma unroll
                    for (int mb = 0; mb < 4; ++mb) {
                        unsigned bits = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) bits = bits + bits + (acc[mb][r] > 0.f ? 1u : 0u);
                        bits <<= 16 * g;
                        bits |= (unsigned)__shfl_xor((int)bits, 32, 64);
                        const int64_t row = base + 32 * mb + s;
                        if (g == 0 && row < a.n) a.save_bits[((int64_t)st * a.n + row) * 8 + w] = bits;
                    }
                }
            }
            ActRegs<4> ar;
            convert_act<4, true, PREC>(acc, ar, acc2act(st));
            NM_TICK(3)
            __syncthreads();                                              // every wave has finished reading H (and P)
            NM_TICK(2)
            write_act<4, PREC>(ar, lds, w, 0, g, s);
            if (SAVE == 2 && F16) {                              // the 16-bit copy IS the operand's hi part: fp16(32 x), clamped, k-slot order
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const int64_t row = base + 32 * mb + s;
                    if (row < a.n) {
                        uint4* o = a.save_h16 + ((int64_t)st * a.n + row) * 32 + 4 * w + g;
                        o[0] = ar.hi[mb][0];
                        o[2] = ar.hi[mb][1];
                    }
                }
            }
            if (st == 5 && !a.sigma_only) fill_pe_any<F16>(lds, true, a, base, tid);   // P is free after the skip layer: direction PE -> P[0..3]
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            if (a.stop_stage == st) { dump_act<F16>(lds, false, 256, a, base, tid); stopped = true; break; }
        }
        if (stopped) {                                            // debug exit: restart the weight / bias pipelines
            __syncthreads();
            w_prefetch<PREC>(W, wsrc, voff, so_s0);
            bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
            continue;
        }

        // ---------------- density only (the coarse pass of a two-pass render: the reference composites its colours into a
        // frame it then discards, render_utils.py:139-141): the alpha block of stage 8 and nothing after it.  Same
        // instruction sequence as the alpha block below, so sigma is bit-identical to the full evaluation's.
        if (a.sigma_only) {
            if (w < 4) {
                const nm::StageShape sh = nm::stage_shape(8);
                f32x16 aacc[1];
                bias_prefetch(B, a.bias + nm::stage_b_off(8) + 32 * 8, g);
                init_bias<1>(aacc, B);
                k_run<1, PREC>(aacc, W, wsrc, voff, so_s8a, so_s0, lds + H_BASE + g * kChunkU4 + 32 * w + s, sh.steps);
                const int64_t i = base + 32 * w + s;
                if (g == 0 && i < a.n) {
                    const float os = acc2out(8);
                    reinterpret_cast<float4*>(a.out)[sample_record(a, i)] =
                        a.sigma_only == 2 ? make_float4(aacc[0][0] * os, aacc[0][1] * os, aacc[0][2] * os, aacc[0][3] * os * a.sigma_scale)   // vanilla.py:145
                                          : make_float4(0.f, 0.f, 0.f, aacc[0][0] * os * a.sigma_scale);
                }
            }
            bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
            NM_TICK(1)
            __syncthreads();                                      // H / P are rewritten by the next tile
            NM_TICK(5)
            continue;
        }

        // ---------------- stage 8: feature (linear, 256) + alpha block (waves 0..3, sample block w)
        float sigma = 0.f;
        if (SAVE == 2 && F16 && a.save_d0h) {                         // P[0..3] holds the direction encoding since the skip layer's epilogue
            for (int item = tid; item < nm::kPeChunks * kTileM; item += kThreads) {
                const int c = item >> 7, row = item & (kTileM - 1);
                uint4 v = make_uint4(0u, 0u, 0u, c == 7 ? 0x50000000u : 0u);      // slot 63 = fp16(32.0): the ones column (x 32)
                if (c < 4) {
                    v = lds[P_BASE + c * kChunkU4 + row];
                    // slots beyond the encoding's width still hold this tile's POSITION encoding (the network's weights there are zero; a product with
                    // these rows must see zeros)
                    const int keep = 3 + 6 * a.dir.nfreq - 8 * c;                  // leading halves of the chunk that belong to the encoding
                    unsigned wd[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                    for (int p = 0; p < 4; ++p) wd[p] &= (2 * p + 1 < keep ? 0xffffffffu : (2 * p < keep ? 0x0000ffffu : 0u));
                    v = make_uint4(wd[0], wd[1], wd[2], wd[3]);
                }
                if (base + row < a.n) a.save_d0h[(base + row) * 8 + c] = v;
            }
        }
        {
            const nm::StageShape sh = nm::stage_shape(8);
            init_bias<4>(acc, B);
            k_run<4, PREC>(acc, W, wsrc, voff, wo(8, w), w < 4 ? so_s8a : so_s9, lds + H_BASE + g * kChunkU4 + s, sh.steps);
            if (w < 4) {
                f32x16 aacc[1];
                bias_prefetch(B, a.bias + nm::stage_b_off(8) + 32 * 8, g);
                init_bias<1>(aacc, B);
                k_run<1, PREC>(aacc, W, wsrc, voff, so_s8a, so_s9, lds + H_BASE + g * kChunkU4 + 32 * w + s, sh.steps);
                sigma = aacc[0][0] * acc2out(8);                     // feature row 0 of the block: lanes 0..31 (g == 0)
            }
            bias_prefetch(B, a.bias + nm::stage_b_off(9) + 32 * (w & 3), g);
            NM_TICK(1)
            if (SAVE == 1 || (SAVE == 2 && a.save_h)) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb)
                    save_block(SAVE == 2 ? a.save_h : a.save_h + (int64_t)8 * a.n * 256, 256, acc[mb], w, base + 32 * mb + s, acc2out(8), false);
            }
            ActRegs<4> ar;
            convert_act<4, false, PREC>(acc, ar, acc2act(8));
            NM_TICK(3)
            __syncthreads();
            NM_TICK(2)
            write_act<4, PREC>(ar, lds, w, 0, g, s);
            if (SAVE == 2 && F16 && a.save_feat16) {                      // the feature layer's output as the views layer reads it: fp16(32 x), k-slot order
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
                    const int64_t row = base + 32 * mb + s;
                    if (row < a.n) {
                        uint4* o = a.save_feat16 + row * 32 + 4 * w + g;
                        o[0] = ar.hi[mb][0];
                        o[2] = ar.hi[mb][1];
                    }
                }
            }
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            if (a.stop_stage == 8) {
                dump_act<F16>(lds, false, 256, a, base, tid);
                __syncthreads();
                w_prefetch<PREC>(W, wsrc, voff, so_s0);
                bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
                continue;
            }
        }

        // ---------------- stage 9: views layer, K = feature(256) ++ d_pe(32), N = 128, ReLU
        {
            const nm::StageShape sh = nm::stage_shape(9);
            const int nb = w & 3, row0 = 64 * (w >> 2);
            const int hsteps = sh.steps - sh.pe_steps;
            f32x16 vacc[2];
            init_bias<2>(vacc, B);
            k_run<2, PREC>(vacc, W, wsrc, voff, so_s9, so_s9 + hsteps * nm::kStepBytes, lds + H_BASE + g * kChunkU4 + row0 + s, hsteps);
            k_run<2, PREC>(vacc, W, wsrc, voff, so_s9 + hsteps * nm::kStepBytes, w < 4 ? so_s10 : so_s0,
                           lds + P_BASE + g * kChunkU4 + row0 + s, sh.pe_steps);
            bias_prefetch(B, w < 4 ? a.bias + nm::stage_b_off(10) : a.bias + nm::stage_b_off(0) + 32 * w, g);
            NM_TICK(1)
            if (SAVE) {
#pragma unroll
                for (int mb = 0; mb < 2; ++mb) save_block(a.save_hv, 128, vacc[mb], nb, base + row0 + 32 * mb + s, acc2out(9), true);
                if (SAVE == 2 && a.save_hvbits) {                          // (the trunk's bit order: register r of lane half g = bit 16 g + 15 - r)
#pragma unroll
                    for (int mb = 0; mb < 2; ++mb) {
                        unsigned bits = 0;
#pragma unroll
                        for (int r = 0; r < 16; ++r) bits = bits + bits + (vacc[mb][r] > 0.f ? 1u : 0u);
                        bits <<= 16 * g;
                        bits |= (unsigned)__shfl_xor((int)bits, 32, 64);
                        const int64_t row = base + row0 + 32 * mb + s;
                        if (g == 0 && row < a.n) a.save_hvbits[row * 4 + nb] = bits;
                    }
                }
            }
            ActRegs<2> ar;
            convert_act<2, true, PREC>(vacc, ar, acc2act(9));
            NM_TICK(3)
            __syncthreads();
            NM_TICK(2)
            write_act<2, PREC>(ar, lds, nb, row0, g, s);
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            if (a.stop_stage == 9) {
                dump_act<F16>(lds, false, 128, a, base, tid);
                __syncthreads();
                w_prefetch<PREC>(W, wsrc, voff, so_s0);
                bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
                continue;
            }
        }

        // ---------------- stage 10: rgb (rows 0..2 of one 32-feature block), waves 0..3 take sample block w
        if (w < 4) {
            const nm::StageShape sh = nm::stage_shape(10);
            f32x16 racc[1];
            init_bias<1>(racc, B);
            k_run<1, PREC>(racc, W, wsrc, voff, so_s10, so_s0, lds + H_BASE + g * kChunkU4 + 32 * w + s, sh.steps);
            bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
            const int64_t i = base + 32 * w + s;
            if (g == 0 && i < a.n)                                // rows 0,1,2 = regs 0,1,2 of the g == 0 half
                reinterpret_cast<float4*>(a.out)[sample_record(a, i)] = make_float4(racc[0][0] * acc2out(10), racc[0][1] * acc2out(10), racc[0][2] * acc2out(10), sigma * a.sigma_scale);
        }
        NM_TICK(1)
        __syncthreads();                                          // H / P are rewritten by the next tile
        NM_TICK(5)
    }
    if (PROF && lane == 0) {
#pragma unroll
        for (int b = 0; b < 6; ++b) a.prof[((int64_t)blockIdx.x * 8 + w) * 8 + b] = pr[b];
    }
#undef NM_TICK
}

// =====================================================================================================================
// NM_PREC_I8X3: the hidden layers in 16-bit fixed point on v_mfma_i32_32x32x32_i8 (mlp_layout.h, DESIGN.md "K4-i8").
//
// Per sample row the 256-/128-wide hidden operand is X = rint(x / sx), sx = max|x| / 32639, stored in LDS as two balanced
// int8 limbs (X = 256*hi + lo); the weights are int16 limbs too, per output feature, with the per-feature steps folded
// into the next layer's columns on the host (mlp_host.hip pack_image8), so that
//     out[n] / unit[n] = sx * kappa * (65536*hi.hi + 256*(hi.lo + lo.hi) [+ lo.lo, dropped: <= 2^-16 of full scale]) + bias'[n]
// with EXACT int32 accumulation: three i8 MFMAs of K = 32 replace three bf16 MFMAs of K = 16 -- half the MFMA time and
// half the weight bytes.  The encodings keep the split-bf16 path (they need absolute precision): stage 0 is bf16 only,
// stages 5 / 9 add their PE part in f32 on top of the dequantised sum.  The row maxima need every wave's features:
// partial maxima go through a small LDS array around a barrier.  Activations take 64 KB of LDS instead of 128 KB; units,
// biases and kappa of all stages sit in LDS.
// =====================================================================================================================
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) short i16x2;

constexpr int H8_BASE = 0;                                   // [16 chunks][hi: 128 rows | lo: 128 rows][16 B]  (64 KB)
constexpr int S8_MAX = 16 * kChunkU4;                        // 4096: row-max partials [8][128] f32
constexpr int S8_SCALE = S8_MAX + 8 * kTileM / 4;            // 4352: row scales [128] f32
constexpr int S8_CONST = S8_SCALE + kTileM / 4;              // 4384: [units | biases | kappa] of all stages (mlp_host.hip)
constexpr int kConst8Floats = 2 * nm::kBiasFloats + 16;
static_assert(S8_CONST + kConst8Floats / 4 <= P_BASE && kConst8Floats % 4 == 0, "i8 scratch must fit below the PE buffer");

struct MlpArgs8 {
    MlpArgs a;
    const float* consts8;     // units (kBiasFloats), biases in those units (kBiasFloats), kappa (16)
    const uint4* wstream8;    // the limb / bf16 fragments as per-wave streams (mlp_layout.h wstream_*)
};

template <int MB>
__device__ __forceinline__ void zero8(i32x16 (&ah)[MB], i32x16 (&ac)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ah[mb][r] = 0; ac[mb][r] = 0; }
}

// accumulator init from the LDS bias table (stage 0: bf16 only)
template <int MB>
__device__ __forceinline__ void init_bias8(f32x16 (&f)[MB], const float* cst, int cblk, int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bs = *reinterpret_cast<const float4*>(cst + nm::kBiasFloats + cblk + 8 * q + 4 * g);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) { f[mb][4 * q] = bs.x; f[mb][4 * q + 1] = bs.y; f[mb][4 * q + 2] = bs.z; f[mb][4 * q + 3] = bs.w; }
    }
}

__device__ __forceinline__ void dump_act8(const uint4* lds, int width, const float* units, const MlpArgs& a, int64_t base, int tid,
                                          int nthreads = kThreads, int row0 = 0, int nrows = kTileM) {
    const signed char* hi = reinterpret_cast<const signed char*>(lds + H8_BASE);
    const signed char* lo = hi + kLoU4 * 16;
    const float* sscale = reinterpret_cast<const float*>(lds + S8_SCALE);
    for (int item = tid; item < nrows * width; item += nthreads) {
        const int row = row0 + item / width, n = item % width;
        if (base + row >= a.n) continue;
        const int off = (nm::feature_chunk8(n) * kChunkU4 + row) * 16 + nm::feature_elem8(n);
        a.dbg[(base + row) * width + n] = sscale[row] * (float)(256 * (int)hi[off] + (int)lo[off]) * units[n];
    }
}

// =====================================================================================================================
// NM_PREC_I8X3, wave-specialised: the two waves of every SIMD work half a stage out of phase.
//
// With all 8 waves running k-loop -> dequantise / row-max -> barrier -> quantise / store in lock step (the layout of
// nerf_mlp_kernel; measured 650 TFLOP/s) the MFMA pipe idles through every epilogue, ~45 % of the time: requantisation
// costs ~6 VALU instructions per accumulator and a barrier.  Here the tile is split into two 64-sample halves owned by
// wave groups A = waves 0..3 and B = waves 4..7 (one wave of each per SIMD); wave q of a group owns output features
// 64q..64q+63 (two 32-feature blocks) of its 64 samples (two 32-sample blocks).  Time is cut into slots, each
// `part 1 | barrier | part 2 | barrier`; a group alternates M slots (k-loop of block 2q | barrier | k-loop of block 2q+1)
// and E slots (dequantise + row-max | barrier | quantise + store), and group B runs the same sequence one slot later:
// while one wave of a SIMD issues MFMAs, the other does its epilogue on the VALU.  The barriers are workgroup-wide and
// both groups execute the same number of them (B two extra before its first slot, A two after its last), so the phase
// relation is fixed by construction.
//
// In an M slot a wave is alone on its SIMD's MFMA pipe, so only its own instruction order hides latency:
//   * its weights are one linear stream (mlp_layout.h wstream_*) prefetched kRing k-steps ahead through a register ring;
//   * the activation fragments of step t + 1 are requested before the MFMAs of step t;
//   * the hh / cross accumulators of a block are combined to one int32 (hh * 256 + cross) as soon as its k-loop ends,
//     so at most 96 accumulator registers are live and nothing spills (scratch would evict the weights from L2).
// Cost: both groups stream the whole weight image (2 x 1.2 MB per 128-sample tile = what bf16x3 streams).
// =====================================================================================================================
constexpr int kWPrio = 2;    // issue priority of a wave inside its k-loop (its SIMD partner is in an epilogue) ...
constexpr int kEPrio = 0;    // ... and outside it (epilogues, fills)
constexpr int kXD = 1;       // k-steps of activation fragments in flight
constexpr int kRing = 4;     // k-steps of weights in flight per wave (8 VGPRs each); runs are multiples of 4 steps
static_assert(kRing <= nm::kW8Pad, "the stream is padded for the ring's overrun");

struct WStep {
    v4u h, l;                                        // hi / lo limb (or bf16 hi / lo) fragment of one k-step
};
struct WRing {
    WStep s[kRing];                                  // the next kRing steps of the stream; slot = step index mod kRing
};
__device__ __forceinline__ void w_step_load(WStep& S, __amdgpu_buffer_rsrc_t wsrc, int voff, int soff) {
    S.h = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, soff, 0);
    S.l = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, soff + 1024, 0);
}

// NSTEPS k-steps of one feature block x MB sample blocks, fully unrolled.  PH = ring slot of the first step; `pos` is the
// stream offset of the step the ring loads next (kRing ahead of the step being consumed).  I8: limbs on the i8 MFMA into
// (ah, ac); otherwise split bf16 into f32 accumulators passed as ah (bit pattern) -- see the two wrappers below.
// PRE: the fragments of this run's first step were requested by the previous run (xp); POST: request the first step of the
// next run (at xnext) during this run's last step -- the two k-loops of an M slot read the same rows, so the second one's
// first LDS round trip is taken off the matrix pipe's critical path and out from behind the slot's middle barrier.
template <int MB>
struct XPre {
    uint4 h[MB], l[MB];
};
template <bool I8, int MB, int NSTEPS, int PH, bool PRE, bool POST>
__device__ __forceinline__ void w_run(i32x16 (&ah)[MB], i32x16 (&ac)[MB], f32x16 (&ff)[MB], WRing& R, __amdgpu_buffer_rsrc_t wsrc,
                                      int voff, int& pos, const uint4* xh, XPre<MB>& xp, const uint4* xnext) {
    // activation fragments kXD steps ahead, in a rotating set of kXD + 1 register groups (static indices: the loop is unrolled)
    uint4 xq[kXD + 1][2][MB];
#pragma unroll
    for (int d = 0; d < kXD && d < NSTEPS; ++d)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (PRE && d == 0) {
                xq[0][0][m] = xp.h[m];
                xq[0][1][m] = xp.l[m];
            } else {
                xq[d][0][m] = xh[d * (2 * kChunkU4) + m * 32];
                xq[d][1][m] = xh[d * (2 * kChunkU4) + kLoU4 + m * 32];
            }
        }
    __builtin_amdgcn_s_setprio(kWPrio);
#pragma unroll
    for (int t = 0; t < NSTEPS; ++t) {
        const int slot = (PH + t) % kRing;
        uint4 xhc[MB], xlc[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) { xhc[m] = xq[t % (kXD + 1)][0][m]; xlc[m] = xq[t % (kXD + 1)][1][m]; }
        if (t + kXD < NSTEPS) {
            const uint4* ph = xh + (t + kXD) * (2 * kChunkU4);
#pragma unroll
            for (int m = 0; m < MB; ++m) { xq[(t + kXD) % (kXD + 1)][0][m] = ph[m * 32]; xq[(t + kXD) % (kXD + 1)][1][m] = ph[kLoU4 + m * 32]; }
        }
        if (POST && t == NSTEPS - 1) {
#pragma unroll
            for (int m = 0; m < MB; ++m) { xp.h[m] = xnext[m * 32]; xp.l[m] = xnext[kLoU4 + m * 32]; }
        }
        const v4u wh = R.s[slot].h, wl = R.s[slot].l;
        __builtin_amdgcn_sched_barrier(0);
        if (I8) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
                ac[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xlc[m]), ac[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
                ac[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wl), __builtin_bit_cast(i32x4, xhc[m]), ac[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m)
                ah[m] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xhc[m]), ah[m], 0, 0, 0);
        } else {
#pragma unroll
            for (int m = 0; m < MB; ++m) ff[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), as_bf16x8(xlc[m]), ff[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m) ff[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl), as_bf16x8(xhc[m]), ff[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < MB; ++m) ff[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), as_bf16x8(xhc[m]), ff[m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        w_step_load(R.s[slot], wsrc, voff, pos);       // the slot just consumed <- the step kRing ahead
        pos += nm::kStepBytes;
    }
    __builtin_amdgcn_s_setprio(kEPrio);
}
template <int MB, int NSTEPS, int PH>
__device__ __forceinline__ void w_run8(i32x16 (&ah)[MB], i32x16 (&ac)[MB], WRing& R, __amdgpu_buffer_rsrc_t wsrc, int voff, int& pos,
                                       const uint4* xh) {
    f32x16 none[MB];
    XPre<MB> xp;
    w_run<true, MB, NSTEPS, PH, false, false>(ah, ac, none, R, wsrc, voff, pos, xh, xp, nullptr);
}
template <int MB, int NSTEPS, int PH>
__device__ __forceinline__ void w_runbf(f32x16 (&f)[MB], WRing& R, __amdgpu_buffer_rsrc_t wsrc, int voff, int& pos, const uint4* xh) {
    i32x16 none[MB];
    XPre<MB> xp;
    w_run<false, MB, NSTEPS, PH, false, false>(none, none, f, R, wsrc, voff, pos, xh, xp, nullptr);
}
// the two k-loops of an M slot (PH1 / PH2: ring slots of their first steps), the second one's first fragments requested by the first
template <int MB, int NSTEPS, int PH, bool PRE, bool POST>
__device__ __forceinline__ void w_run8x(i32x16 (&ah)[MB], i32x16 (&ac)[MB], WRing& R, __amdgpu_buffer_rsrc_t wsrc, int voff, int& pos,
                                        const uint4* xh, XPre<MB>& xp, const uint4* xnext) {
    f32x16 none[MB];
    w_run<true, MB, NSTEPS, PH, PRE, POST>(ah, ac, none, R, wsrc, voff, pos, xh, xp, xnext);
}
template <int MB, int NSTEPS, int PH, bool PRE, bool POST>
__device__ __forceinline__ void w_runbfx(f32x16 (&f)[MB], WRing& R, __amdgpu_buffer_rsrc_t wsrc, int voff, int& pos, const uint4* xh,
                                         XPre<MB>& xp, const uint4* xnext) {
    i32x16 none[MB];
    w_run<false, MB, NSTEPS, PH, PRE, POST>(none, none, f, R, wsrc, voff, pos, xh, xp, xnext);
}

// t = hh * 256 + cross (exact: |t| * 256 is the full 32-bit product sum)
template <int MB>
__device__ __forceinline__ void combine8(i32x16 (&t)[MB], const i32x16 (&ah)[MB], const i32x16 (&ac)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) t[mb][r] = (ah[mb][r] << 8) + ac[mb][r];
}
// f = t * sx256[row block] + bias'[feature], sx256 = 256 * sx * kappa_stage: the stage's output in its per-feature units.
// (scalar v_fma_f32 on purpose: packed f32 VALU costs ~20 extra cycles per instruction beside MFMAs, MI355X_MICROARCH.md)
template <int MB>
__device__ __forceinline__ void dequantw(f32x16 (&f)[MB], const i32x16 (&t)[MB], const float (&sx256)[MB], const float* cst, int cblk, int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 bs = *reinterpret_cast<const float4*>(cst + nm::kBiasFloats + cblk + 8 * q + 4 * g);
        const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int j = 0; j < 4; ++j) f[mb][4 * q + j] = fmaf((float)t[mb][4 * q + j], sx256[mb], bsv[j]);
    }
}

// per-row partial maximum over this wave's NB blocks -> smax[part][row]; the two lane halves (features +4) meet through
// v_permlane32_swap instead of an LDS round trip
template <int NB, int MB, bool RELU>
__device__ __forceinline__ void rowmaxw(const f32x16 (&f)[NB][MB], float* smax, int part, int row0, int g, int s) {
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    float m[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        m[mb] = 0.f;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) m[mb] = fmaxf(m[mb], RELU ? f[b][mb][r] : fabsf(f[b][mb][r]));
        const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m[mb]), __float_as_uint(m[mb]), false, false);
        m[mb] = fmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
    }
    if (g == 0) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) smax[part * kTileM + row0 + 32 * mb + s] = m[mb];
    }
}

// quantise the wave's NB blocks with the row scales (max over the 4 partials) and store the limbs.  All partial maxima
// are read up front: in an E slot this wave has no partner to cover an LDS round trip per block.
template <int NB, int MB, bool RELU>
__device__ __forceinline__ void quant_storew(const f32x16 (&f)[NB][MB], uint4* lds, const float* smax, float* sscale, int blk0, int row0,
                                             int g, int s, bool write_scale) {
    float part[MB][4];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int p = 0; p < 4; ++p) part[mb][p] = smax[p * kTileM + row0 + 32 * mb + s];
    float inv1[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        const float M = fmaxf(fmaxf(part[mb][0], part[mb][1]), fmaxf(part[mb][2], part[mb][3]));
        const float c = (float)nm::kFixedMax / 32767.f;        // cvt_pknorm maps [-1, 1] to rint(y * 32767)
        const float inv = M > 0.f ? c * __builtin_amdgcn_rcpf(M) : 0.f;
        inv1[mb] = inv;
        if (write_scale && g == 0) sscale[row0 + 32 * mb + s] = M > 0.f ? M * (1.f / (float)nm::kFixedMax) : 1.f;
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) {
            i16x2 P[8], Y[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float y0 = f[b][mb][2 * i] * inv1[mb], y1 = f[b][mb][2 * i + 1] * inv1[mb];
                if (RELU) {                                     // y <= 32639/32767 < 1: clamping to [0, 1] is the ReLU, and it folds
                    y0 = __builtin_amdgcn_fmed3f(y0, 0.f, 1.f);  // into the multiply's clamp modifier (no instruction of its own)
                    y1 = __builtin_amdgcn_fmed3f(y1, 0.f, 1.f);
                }
                const i16x2 p = __builtin_amdgcn_cvt_pknorm_i16(y0, y1);
                P[i] = p;
                Y[i] = p + (i16x2){128, 128};
            }
            unsigned lo[4], hi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                lo[k] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, P[2 * k + 1]), __builtin_bit_cast(unsigned, P[2 * k]), 0x06040200u);
                hi[k] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, Y[2 * k + 1]), __builtin_bit_cast(unsigned, Y[2 * k]), 0x07050301u);
            }
            const int idx = H8_BASE + (2 * (blk0 + b) + g) * kChunkU4 + row0 + 32 * mb + s;
            lds[idx] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
            lds[idx + kLoU4] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
        }
}

// PROF buckets (a.prof[(block*8 + wave)*8 + b]): 0 fill, 1 k-loops (incl. their middle barrier), 2 end barrier of an M slot,
// 3 epilogue part 1, 4 its middle barrier, 5 epilogue part 2, 6 end barrier of an E slot, 7 the rest
template <bool PROF>
__global__ __launch_bounds__(kThreads, 2) void nerf_mlp_i8w_kernel(const MlpArgs8 A) {
    unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = PROF ? __builtin_readcyclecounter() : 0;
#define NM_TICK(b)                                                   \
    if (PROF) {                                                      \
        const unsigned long long t_now = __builtin_readcyclecounter(); \
        pr[b] += t_now - t_prev;                                     \
        t_prev = t_now;                                              \
    }
    __shared__ uint4 lds[LDS_U4];
    const MlpArgs a = resolve_args(A.a);
    const int tid0 = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int G = w >> 2, wq = w & 3;                          // wave group (tile half) and this wave's feature quarter
    const int row0 = 64 * G;
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(A.wstream8), 0, (int)nm::kWeightBytes8w, 0x00020000);
    const int stream0 = (int)nm::wstream_off(wq);              // this wave's stream (wq is wave-uniform: scalar arithmetic)
    const int64_t ntiles = (a.n + kTileM - 1) / kTileM;
    float* smax = reinterpret_cast<float*>(lds + S8_MAX);
    float* sscale = reinterpret_cast<float*>(lds + S8_SCALE);
    float* cst = reinterpret_cast<float*>(lds + S8_CONST);
    constexpr int kS = nm::kStepBytes;
    constexpr int kKap = 2 * nm::kBiasFloats;                  // float offset of kappa[stage] in cst

    for (int i = tid0; i < nm::kPeChunks * kChunkU4; i += kThreads) lds[P_BASE + i] = make_uint4(0, 0, 0, 0);
    for (int i = tid0; i < kConst8Floats / 4; i += kThreads) lds[S8_CONST + i] = reinterpret_cast<const uint4*>(A.consts8)[i];
    __syncthreads();

    // thread index for the rarely executed paths, opaque so that their address arithmetic is not hoisted and spilled
    auto cold_gt = [tid0]() {
        int t = tid0;
        asm volatile("" : "+v"(t));
        return t & 255;
    };

    if (G == 1) { __syncthreads(); __syncthreads(); }          // group B runs one slot behind group A

#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileM;
        // Re-derive every lane-constant index from an opaque copy of the thread id once per tile: otherwise the compiler
        // hoists dozens of per-lane addresses out of this loop and spills them, and scratch competes with the weights for L2
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, gt = tid & 255;
        const int g = lane >> 5, s = lane & 31;
        const int voff = lane * 16;
        const uint4* xH = lds + H8_BASE + g * kChunkU4 + row0 + s;
        const uint4* xP = lds + P_BASE + g * kChunkU4 + row0 + s;

        // ---------------- slot F (shared with the previous tile's stage 10): position encoding of this group's rows;
        // the weight stream restarts behind it (the fill's f64 sin/cos needs the registers, the barriers cover the latency)
        NM_TICK(7)
        fill_pe_any(lds, false, a, base, gt, 256, row0, 6);
        WRing R;
        int pos = stream0;
#pragma unroll
        for (int i = 0; i < kRing; ++i, pos += kS) w_step_load(R.s[i], wsrc, voff, pos);
        NM_TICK(0)
        __syncthreads();
        __syncthreads();
        NM_TICK(7)
        if (a.stop_stage == -1) {                                  // (one empty slot: the dump must finish before the next fill)
            dump_act(lds, true, 64, a, base, cold_gt(), 256, row0, 64);
            __syncthreads();
            __syncthreads();
            continue;
        }

        // ---------------- stage 0: encodings only (split bf16)
        bool stopped = false;
        {
            f32x16 f[2][2];
            init_bias8<2>(f[0], cst, 64 * wq, g);
            init_bias8<2>(f[1], cst, 64 * wq + 32, g);
            XPre<2> xp;
            w_runbfx<2, 4, 0, false, true>(f[0], R, wsrc, voff, pos, xP, xp, xP);
            __syncthreads();
            w_runbfx<2, 4, 4 % kRing, true, false>(f[1], R, wsrc, voff, pos, xP, xp, nullptr);
            NM_TICK(1)
            __syncthreads();
            NM_TICK(2)
            rowmaxw<2, 2, true>(f, smax, wq, row0, g, s);
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            quant_storew<2, 2, true>(f, lds, smax, sscale, 2 * wq, row0, g, s, wq == 0);
            NM_TICK(5)
            __syncthreads();
            NM_TICK(6)
            if (a.stop_stage == 0) { dump_act8(lds, 256, cst, a, base, cold_gt(), 256, row0, 64); stopped = true; }
        }

        // ---------------- stages 1..7
#pragma unroll 1
        for (int st = 1; st <= 7 && !stopped; ++st) {
            const int cblk = 256 * st + 64 * wq;                    // = stage_b_off(st) + 64 wq
            f32x16 f[2][2];
            {
                i32x16 t[2][2];
                XPre<2> xp;
                {
                    i32x16 ah[2], ac[2];
                    zero8<2>(ah, ac);
                    w_run8x<2, 8, 0, false, true>(ah, ac, R, wsrc, voff, pos, xH, xp, xH);
                    combine8<2>(t[0], ah, ac);
                }
                __syncthreads();
                {
                    i32x16 ah[2], ac[2];
                    zero8<2>(ah, ac);
                    w_run8x<2, 8, 0, true, false>(ah, ac, R, wsrc, voff, pos, xH, xp, nullptr);
                    combine8<2>(t[1], ah, ac);
                }
                if (st != 5) {
                    NM_TICK(1)
                    __syncthreads();
                    NM_TICK(2)
                }
                const float k256 = 256.f * cst[kKap + st];
                const float sxin[2] = {sscale[row0 + s] * k256, sscale[row0 + 32 + s] * k256};
                dequantw<2>(f[0], t[0], sxin, cst, cblk, g);
                dequantw<2>(f[1], t[1], sxin, cst, cblk + 32, g);
            }
            if (st == 5) {                                          // skip layer: the position encoding on top, then this M slot ends
                w_runbf<2, 4, 0>(f[0], R, wsrc, voff, pos, xP);
                w_runbf<2, 4, 4 % kRing>(f[1], R, wsrc, voff, pos, xP);
                NM_TICK(1)
                __syncthreads();
                NM_TICK(2)
            }
            rowmaxw<2, 2, true>(f, smax, wq, row0, g, s);
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            quant_storew<2, 2, true>(f, lds, smax, sscale, 2 * wq, row0, g, s, wq == 0);
            NM_TICK(5)
            if (st == 5) fill_pe_any(lds, true, a, base, cold_gt(), 256, row0, 6);     // (opaque index: not hoisted out of the stage loop)
            NM_TICK(0)
            __syncthreads();
            NM_TICK(6)
            if (a.stop_stage == st) { dump_act8(lds, 256, cst + 256 * st, a, base, cold_gt(), 256, row0, 64); stopped = true; }
        }
        if (stopped) continue;

        // ---------------- stage 8: feature (linear) + alpha block (waves 0, 1 of the group: one 32-sample block each)
        float sigma = 0.f;
        {
            f32x16 f[2][2];
            {
                i32x16 t[2][2], ta[1];
                zero8<1>(ta, ta);                                  // (defined on every path: an undefined value would be carried -- and spilled -- around the tile loop)
                XPre<2> xp;
                {
                    i32x16 ah[2], ac[2];
                    zero8<2>(ah, ac);
                    w_run8x<2, 8, 0, false, true>(ah, ac, R, wsrc, voff, pos, xH, xp, xH);
                    combine8<2>(t[0], ah, ac);
                }
                __syncthreads();
                {
                    i32x16 ah[2], ac[2];
                    zero8<2>(ah, ac);
                    w_run8x<2, 8, 0, true, false>(ah, ac, R, wsrc, voff, pos, xH, xp, nullptr);
                    combine8<2>(t[1], ah, ac);
                }
                if (wq < 2) {
                    i32x16 ah[1], ac[1];
                    zero8<1>(ah, ac);
                    w_run8<1, 8, 0>(ah, ac, R, wsrc, voff, pos, xH + 32 * wq);
                    combine8<1>(ta, ah, ac);
                }
                NM_TICK(1)
                __syncthreads();
                NM_TICK(2)
                const float k256 = 256.f * cst[kKap + 8];
                const float sxin[2] = {sscale[row0 + s] * k256, sscale[row0 + 32 + s] * k256};
                dequantw<2>(f[0], t[0], sxin, cst, nm::stage_b_off(8) + 64 * wq, g);
                dequantw<2>(f[1], t[1], sxin, cst, nm::stage_b_off(8) + 64 * wq + 32, g);
                if (wq < 2) {
                    f32x16 fa[1];
                    const float sx1[1] = {sscale[row0 + 32 * wq + s] * k256};
                    dequantw<1>(fa, ta, sx1, cst, nm::stage_b_off(8) + 32 * 8, g);
                    sigma = fa[0][0] * cst[nm::stage_b_off(8) + 256];
                }
            }
            rowmaxw<2, 2, false>(f, smax, wq, row0, g, s);
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            quant_storew<2, 2, false>(f, lds, smax, sscale, 2 * wq, row0, g, s, wq == 0);
            NM_TICK(5)
            __syncthreads();
            NM_TICK(6)
            if (a.stop_stage == 8) {
                dump_act8(lds, 256, cst + nm::stage_b_off(8), a, base, cold_gt(), 256, row0, 64);
                continue;
            }
        }

        // ---------------- stage 9: views layer (one 32-feature block per wave), hidden part on i8 + direction encoding on bf16
        {
            f32x16 f[1][2];
            {
                i32x16 t[2];
                {
                    i32x16 ah[2], ac[2];
                    zero8<2>(ah, ac);
                    XPre<2> xp;
                    w_run8x<2, 4, 0, false, true>(ah, ac, R, wsrc, voff, pos, xH, xp, xH + 4 * (2 * kChunkU4));
                    __syncthreads();
                    w_run8x<2, 4, 4 % kRing, true, false>(ah, ac, R, wsrc, voff, pos, xH + 4 * (2 * kChunkU4), xp, nullptr);
                    combine8<2>(t, ah, ac);
                }
                const float k256 = 256.f * cst[kKap + 9];
                const float sxin[2] = {sscale[row0 + s] * k256, sscale[row0 + 32 + s] * k256};
                dequantw<2>(f[0], t, sxin, cst, nm::stage_b_off(9) + 32 * wq, g);
            }
            w_runbf<2, 4, 0>(f[0], R, wsrc, voff, pos, xP);          // 2 steps of direction encoding + 2 zero steps (chunks 4..7 of P
            NM_TICK(1)                                              //  hold finite leftovers of the position encoding: 0 * x = 0)
            __syncthreads();
            NM_TICK(2)
            rowmaxw<1, 2, true>(f, smax, wq, row0, g, s);
            NM_TICK(3)
            __syncthreads();
            NM_TICK(4)
            quant_storew<1, 2, true>(f, lds, smax, sscale, wq, row0, g, s, wq == 0);
            NM_TICK(5)
            __syncthreads();
            NM_TICK(6)
            if (a.stop_stage == 9) {
                dump_act8(lds, 128, cst + nm::stage_b_off(9), a, base, cold_gt(), 256, row0, 64);
                continue;
            }
        }

        // ---------------- stage 10: rgb (waves 0, 1: one 32-sample block each); shares the next tile's slot F
        if (wq < 2) {
            i32x16 ah[1], ac[1], t[1];
            f32x16 fr[1];
            zero8<1>(ah, ac);
            const float sx1[1] = {sscale[row0 + 32 * wq + s] * (256.f * cst[kKap + 10])};
            w_run8<1, 4, 4 % kRing>(ah, ac, R, wsrc, voff, pos, xH + 32 * wq);
            combine8<1>(t, ah, ac);
            dequantw<1>(fr, t, sx1, cst, nm::stage_b_off(10), g);
            int ls = s;
            asm volatile("" : "+v"(ls));
            const int64_t i = base + row0 + 32 * wq + ls;
            if (g == 0 && i < a.n)
                reinterpret_cast<float4*>(a.out)[sample_record(a, i)] = make_float4(fr[0][0] * cst[nm::stage_b_off(10)], fr[0][1] * cst[nm::stage_b_off(10) + 1],
                                                                  fr[0][2] * cst[nm::stage_b_off(10) + 2], sigma * a.sigma_scale);
        }
    }
    if (G == 0) { __syncthreads(); __syncthreads(); }
    NM_TICK(7)
    if (PROF && (tid0 & 63) == 0) {
#pragma unroll
        for (int b = 0; b < 8; ++b) a.prof[((int64_t)blockIdx.x * 8 + w) * 8 + b] = pr[b];
    }
#undef NM_TICK
}

}  // namespace

namespace nm {

int launch_mlp_mfma(const MlpLaunch& L, const float* pts, const float* dirs, const float* origin, const float* direction,
                    const float* z, int64_t n, int S, int in_mode, int precision, int stop_stage, float sigma_scale, float* out,
                    float* dbg, void* prof, hipStream_t stream, int sigma_only, const MlpChunk* chunk) {
    MlpArgs a;
    a.ray_idx = chunk ? chunk->ray_idx : nullptr;
    a.n_rays_dev = chunk ? chunk->n_rays_dev : nullptr;
    a.s0 = chunk ? chunk->s0 : 0;
    a.S_total = chunk ? chunk->S_total : S;
    a.wpack = reinterpret_cast<const uint4*>(precision == NM_PREC_FP16X3 ? L.wpack16 : L.wpack);
    a.bias = precision == NM_PREC_FP16X3 ? L.bias16 : L.bias;
    a.petab = L.petab;
    a.pts = pts; a.dirs = dirs; a.origin = origin; a.direction = direction; a.z = z;
    a.out = out; a.dbg = dbg; a.prof = reinterpret_cast<unsigned long long*>(prof); a.n = n; a.S = S; a.in_mode = in_mode; a.stop_stage = stop_stage; a.sigma_scale = sigma_scale;
    a.sigma_only = L.plain_head ? 2 : ((sigma_only && precision != NM_PREC_I8X3) ? 1 : 0);   // (the i8x3 kernel always evaluates the colour head)
    a.save_h = L.save_h; a.save_hv = L.save_hv; a.save_bits = L.save_bits; a.save_h16 = reinterpret_cast<uint4*>(L.save_h16);
    a.save_feat16 = reinterpret_cast<uint4*>(L.save_feat16); a.save_hvbits = L.save_hvbits;
    a.save_x0h = reinterpret_cast<uint4*>(L.save_x0h); a.save_d0h = reinterpret_cast<uint4*>(L.save_d0h);
    a.pos = PeSpec{L.pe_kind, L.pos_nfreq, L.pos_octaves};
    a.dir = PeSpec{L.pe_kind, L.dir_nfreq, L.dir_octaves};
    const int64_t ntiles = (n + kTileM - 1) / kTileM;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)(ntiles < cus ? ntiles : cus);          // one 160 KB workgroup per CU, grid-stride over tiles
    if (precision == NM_PREC_I8X3) {
        MlpArgs8 a8;
        a8.a = a;
        a8.consts8 = L.consts8;
        a8.wstream8 = reinterpret_cast<const uint4*>(L.wstream8);
        if (prof) hipLaunchKernelGGL(nerf_mlp_i8w_kernel<true>, dim3(grid), dim3(kThreads), 0, stream, a8);
        else hipLaunchKernelGGL(nerf_mlp_i8w_kernel<false>, dim3(grid), dim3(kThreads), 0, stream, a8);
        return check_launch("nerf_mlp_i8w_kernel");
    }
    if (L.save_h || L.save_h16) {                                 // the training forward: split fp16, the full head, activations kept
        if (L.save_h16) hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_FP16X3, false, 2>), dim3(grid), dim3(kThreads), 0, stream, a);
        else hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_FP16X3, false, 1>), dim3(grid), dim3(kThreads), 0, stream, a);
        return check_launch("nerf_mlp_kernel (save)");
    }
    if (prof && precision == NM_PREC_FP16X3)
        hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_FP16X3, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (prof)
        hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_BF16X3, true>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (precision == NM_PREC_BF16X3)
        hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_BF16X3, false>), dim3(grid), dim3(kThreads), 0, stream, a);
    else if (precision == NM_PREC_FP16X3)
        hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_FP16X3, false>), dim3(grid), dim3(kThreads), 0, stream, a);
    else
        hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_BF16, false>), dim3(grid), dim3(kThreads), 0, stream, a);
    return check_launch("nerf_mlp_kernel");
}

}  // namespace nm
