// NM_PREC_I8X3, activation-stationary: the same 16-bit fixed-point arithmetic as nerf_mlp_i8w_kernel (mlp.hip; DESIGN.md "K4-i8") with
// the roles of the two operands swapped.  There, a wave owns 64 FEATURES of 64 samples: the activations of the tile live in LDS (every wave
// reads all of them for every block it computes), the weights stream from L2 into every wave group separately (2 x 1.27 MB per 128
// samples), and requantisation needs the row maximum over all features = a partial-maximum exchange through LDS and two barriers per
// stage.  Here a wave owns 32 SAMPLES and ALL features:
//
//   * its input activations of a stage -- 256 features as two int8 limbs -- are 8 k-steps x {hi, lo} x 16 B per lane = 64 registers,
//     resident for the whole stage; the accumulator layout of v_mfma_i32_32x32x32_i8 gives a lane, for its sample, exactly the 16 k-slots
//     the next stage's B operand wants from it (mlp_layout.h slot_feature8), so the requantised outputs of block b ARE the next stage's
//     fragment of k-step b, in place -- activations never touch LDS, never cross lanes (one v_permlane32_swap for the row maximum);
//   * the row maximum is local to the wave: no exchange, no barrier; waves never wait for each other except for the weight ring;
//   * the weights are the A operands, one 1 KB fragment per limb and k-step, used for one MFMA triple.  All 8 waves of the workgroup want
//     the same fragments at about the same time, so the weight image -- re-cut on the host in exactly the order it is consumed
//     (mlp_host.hip pack_stream8s: 628 k-steps of 2 KB per 256-sample tile) -- is streamed ONCE per workgroup from L2 into an LDS ring
//     of whole ring blocks by LDS-DMA (global_load_lds_dwordx4: lane-linear, exactly the fragment format), two blocks ahead of the block
//     being multiplied; the only workgroup barrier is the one per ring block that hands a slot over.
//   Measurements, the variants tried and why the kernel is the way it is: profiles/r03_as_kernel_experiments.md, DESIGN.md "K4-i8s".
//
// Reference semantics: models/vanilla.py Embedder.forward (:82-92), NeRF.forward (:120-152), Joiner.forward (:162-166).
#include "mlp_i8as.h"
#include <stdio.h>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int kWaves = 8;                        // 8 waves x 32 samples = 256 samples per workgroup, one workgroup per CU
constexpr int kTile = kWaves * kRows;
// LDS: the encodings of each wave's 32 rows, wave-private: [wave][8 chunks][hi: 32 rows | lo: 32 rows][16 B] = 8 KB per wave
constexpr int kPWaveU4 = nm::kPeChunks * 2 * kRows;          // 512 uint4
constexpr int kPeU4 = kWaves * kPWaveU4;
// the weight ring: kSlots slots of one ring block (at most 10 k-steps of 2 KB; sized for 12)
constexpr int kStepU4 = nm::kStepBytes / 16;
constexpr int kSlotU4 = 12 * kStepU4;
constexpr int kSlots = 3;                                    // the block being multiplied + two being copied
constexpr int kBiasU4 = (nm::kBiasFloats + 16 + 3) / 4;      // the bias table and kappa, resident in LDS (a global load per block would sit in the same
                                                             // in-order VMEM queue as the copies and force them to land early)
// flat RING block index of a tile (what one barrier hands over): stage 0: 0-7 (4 steps) | 1-4: 8-39 | 5: 40-47 and its encoding part
// 48-51 (two output blocks x 4 steps each) | 6, 7: 52-67 | 8: 68-76 | 9: 77-80 (10 steps) | 10: 81 (4 steps)
constexpr int kTileBlocks = 82;
__host__ __device__ constexpr int block_steps(int i) {
    i = i >= kTileBlocks ? i - kTileBlocks : i;
    return i < 8 ? 4 : i < 77 ? 8 : i < 81 ? 10 : 4;
}
__host__ __device__ constexpr int block_pieces(int nsteps) { return (2 * nsteps + kWaves - 1) / kWaves; }   // 1 KB pieces per wave

// -DNM_AS_PROF: cycle buckets per wave (s_memtime = shader cycles): 0 wait at the top-of-block barrier (own copies + the other waves),
// 1 copy issue, 2 k-loop, 4 epilogue, 5 requantisation at the end of a stage, 6 encodings, 7 rest
#ifdef NM_AS_PROF
struct Prof {
    unsigned long long t, acc[8];
};
#define PROF_DECL Prof P; P.t = __builtin_amdgcn_s_memtime(); for (int i_ = 0; i_ < 8; ++i_) P.acc[i_] = 0;
#define PROF_TICK(b) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); P.acc[b] += n_ - P.t; P.t = n_; }
#define PROF_ARG , Prof& P
#define PROF_PASS , P
#else
#define PROF_DECL
#define PROF_TICK(b)
#define PROF_ARG
#define PROF_PASS
#endif
// ---- the weight ring.  Producer side: every wave copies its share (1 KB pieces i = w, w + 8, ..) of the block TWO ahead; consumer side:
// all waves read every fragment of the current block.  Hand-over, once per block: each wave waits until its own pieces of the block it is
// about to enter have landed (counted vmcnt: the pieces of the block after it stay in flight -- issue to landing is about 1 us, longer
// than a block), then the barrier makes everybody's pieces visible and proves that nobody still reads the slot that is refilled next.
// The only other VMEM operations of the kernel are the sample loads at the top of a tile and the 16-byte store at its end (the compiler
// waits vmcnt(0) for the former: two copies land early, once per tile).
struct Ring {
    const char* src;           // image + lane * 16 + w * 1024
    const uint4* rd;           // ring + lane
    unsigned lds0;             // LDS byte address of slot 0 + w * 1024
    int off;                   // image offset of the block to copy next
    int slot;                  // slot of the block to enter next
    int refill, np, nsteps2;   // the copy in progress: slot, pieces per wave, k-steps of the block
};
// this wave's piece j of the block at R.off -> slot (a 10-step block is padded to 3 pieces: the excess lands in the unused tail of the slot)
__device__ __forceinline__ void ring_piece(const Ring& R, int off, int slot, int j) {
    glds16(R.src + off + j * (kWaves * 1024), __builtin_amdgcn_readfirstlane(R.lds0 + slot * (kSlotU4 * 16) + j * (kWaves * 1024)));
}
__device__ __forceinline__ void ring_advance(Ring& R, int nsteps) {
    R.off += nsteps * nm::kStepBytes;
    if (R.off == (int)nm::kWeightBytes8) R.off = 0;
}
// enter flat block i of the tile; returns this lane's view of block i.  The copy of block i + 2 (into the slot block i - 1 has just given
// up) is issued from inside the k-loop (ring_copy after k-steps 0, 2, 4): the texture path takes one 1 KB piece at a time, and eight
// waves issuing theirs right after the barrier would all start their MFMAs late.
__device__ __forceinline__ const uint4* ring_enter(Ring& R, int i PROF_ARG) {
    PROF_TICK(4)
    const int np1 = block_pieces(block_steps(i + 1));
    if (np1 == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (np1 == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    PROF_TICK(0)
    const uint4* cur = R.rd + R.slot * kSlotU4;
    R.refill = R.slot == 0 ? kSlots - 1 : R.slot - 1;                  // the slot of block i - 1 = of block i + 2
    R.slot = R.slot == kSlots - 1 ? 0 : R.slot + 1;
    R.np = block_pieces(block_steps(i + 2));
    R.nsteps2 = block_steps(i + 2);
    return cur;
}
__device__ __forceinline__ void ring_copy(Ring& R, int j) {
    if (j < R.np) ring_piece(R, R.off, R.refill, j);
    if (j == R.np - 1) ring_advance(R, R.nsteps2);
}
struct W8 {
    uint4 h, l;
};

// NSTEPS limb k-steps of one output block: t = 256 * sum(hi.hi) + sum(hi.lo + lo.hi), exact, in two int32 accumulators (two dependency
// chains; every weight fragment read from LDS once).  Between the k-steps rides the dequantisation of the PREVIOUS block (PEND), two
// values per step: in lock-step with its SIMD partner a wave would otherwise do it while nobody uses the matrix pipe.
// PEND: fp = the previous block's outputs (written here, two per MFMA of the first pass), tp = its accumulators, bias_blk = its biases
// (this lane's half of every group of 8), m = the running row maximum
template <int NSTEPS, bool PEND, bool RELU = true>
__device__ __forceinline__ void k_i8_impl(i32x16& t, const X8& X, const uint4* ws, Ring& R, f32x16& fp, const i32x16& tp, lds_cfloat* bias_blk,
                                          float sx256, float& m) {
    i32x16 ah, ac;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = 0; ac[r] = 0; }
    W8 w[2];
    w[0].h = ws[0]; w[0].l = ws[64];
    w[1].h = ws[kStepU4]; w[1].l = ws[kStepU4 + 64];
#pragma unroll
    for (int s = 0; s < NSTEPS; ++s) {
        const uint4 wh = w[s & 1].h, wl = w[s & 1].l;
        ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(as_i32x4(wh), as_i32x4(X.l[s]), ac, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(as_i32x4(wh), as_i32x4(X.h[s]), ah, 0, 0, 0);      // (between the two links of the cross-term
        ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(as_i32x4(wl), as_i32x4(X.h[s]), ac, 0, 0, 0);      // chain: -2.5 % at steady state)
        if (s + 2 < NSTEPS) { w[s & 1].h = ws[(s + 2) * kStepU4]; w[s & 1].l = ws[(s + 2) * kStepU4 + 64]; }
        if (s == 0 || s == 2 || s == 4) ring_copy(R, s >> 1);
        if (PEND) {
            const int r = 2 * s;
            const float b0 = bias_blk[8 * (r >> 2) + (r & 3)], b1 = bias_blk[8 * (r >> 2) + (r & 3) + 1];
            const float f0 = fmaf((float)tp[r], sx256, b0), f1 = fmaf((float)tp[r + 1], sx256, b1);
            fp[r] = f0;
            fp[r + 1] = f1;
            m = RELU ? fmaxf(m, fmaxf(f0, f1)) : fmaxf(m, fmaxf(fabsf(f0), fabsf(f1)));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = (ah[r] << 8) + ac[r];
}
template <int NSTEPS>
__device__ __forceinline__ void k_i8(i32x16& t, const X8& X, const uint4* ws, Ring& R) {
    f32x16 nf;
    float nm_ = 0.f;
    k_i8_impl<NSTEPS, false>(t, X, ws, R, nf, t, nullptr, 0.f, nm_);
}
// NSTEPS split-bf16 k-steps over the wave's encoding rows (chunks c0 ..), accumulated into f
template <int NSTEPS, bool COPY = false>
__device__ __forceinline__ void k_bf(f32x16& f, const uint4* pw, int g, int s, const uint4* ws, Ring* R = nullptr, int j0 = 0) {
#pragma unroll
    for (int t = 0; t < NSTEPS; ++t) {
        if (COPY && (t == 1 || t == 3)) ring_copy(*R, j0 + (t >> 1));
        const uint4 wh = ws[t * kStepU4], wl = ws[t * kStepU4 + 64];
        const uint4 xh = pw[(2 * t + g) * (2 * kRows) + s], xl = pw[(2 * t + g) * (2 * kRows) + kRows + s];
        f = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wh), as_bf16x8(xl), f, 0, 0, 0);
        f = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wl), as_bf16x8(xh), f, 0, 0, 0);
        f = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(wh), as_bf16x8(xh), f, 0, 0, 0);
    }
}
// PLAIN: the use_viewdirs=False net (--specular_can no; models/vanilla.py:116-117, 145): stages 0..7, then ring block 68 holds output_linear's four rows
// (r, g, b, sigma) where the alpha row is otherwise, and the tile ends there.  The instantiation differs from the default one by ONE wave-uniform exit
// (and the direction encodings it does not compute) and nothing else: this kernel sits at exactly 256 registers, and every other way of telling the
// compiler about the shorter tile (69 as a constant, or as a register, in the ring's look-ahead) made it spill 900-1000 bytes per lane; so the ring
// stays the default one's, and the STREAM is cut to fit it (mlp_host.hip pack_stream8s): where the look-ahead expects blocks 69 and 70 (8 steps each)
// it finds the next tile's blocks 0 and 1 (4 steps each, padded to 8), and the exit points the ring at block 2.  The exit tests a.sigma_only == 2 (set by
// the launch) rather than PLAIN alone, so that the code after it stays in the instantiation: without it the allocation of the stage loop changes too.
// (HIP's second __launch_bounds__ argument is the minimum number of WAVES PER SIMD -- not CUDA's blocks per multiprocessor: 2 = the
// 8 waves of the ONE workgroup a CU holds, i.e. a 256-register budget per wave; the 147 KB of LDS allow no second workgroup anyway)
template <bool PLAIN>
__global__ __launch_bounds__(kWaves * 64, 2) void nerf_mlp_i8s_kernel(const Args8s A) {
    __shared__ uint4 lds[kPeU4 + kSlots * kSlotU4 + kBiasU4];
    const MlpArgs a = resolve_args(A.a);
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 5, s = lane & 31;
    uint4* pw = lds + w * kPWaveU4;
    Ring R;
    R.src = reinterpret_cast<const char*>(A.image8) + lane * 16 + w * 1024;
    R.rd = lds + kPeU4 + lane;
    R.lds0 = (unsigned)(uintptr_t)(lds + kPeU4) + w * 1024;
    R.off = 0;
    R.slot = 0;
    {                                                                                   // the bias table
        float* lb = reinterpret_cast<float*>(lds + kPeU4 + kSlots * kSlotU4);
        for (int i = tid; i < nm::kBiasFloats + 16; i += kWaves * 64) lb[i] = A.consts8[nm::kBiasFloats + i];
    }
    __syncthreads();
    ring_piece(R, 0, 0, 0);                                                             // blocks 0 and 1 of the first tile (one piece each)
    ring_piece(R, block_steps(0) * nm::kStepBytes, 1, 0);
    R.off = (block_steps(0) + block_steps(1)) * nm::kStepBytes;
    const float u_sigma = A.consts8[nm::stage_b_off(8) + 256];
    const float u_r = A.consts8[nm::stage_b_off(10)], u_g = A.consts8[nm::stage_b_off(10) + 1], u_b = A.consts8[nm::stage_b_off(10) + 2];
    const float* kappa = reinterpret_cast<const float*>(lds + kPeU4 + kSlots * kSlotU4) + nm::kBiasFloats;
    unsigned bias_lds = (unsigned)(uintptr_t)(lds + kPeU4 + kSlots * kSlotU4) + 16 * g;        // this lane's half of every group of 8:
    asm volatile("" : "+v"(bias_lds));                                                  // ONE address register, everything else an immediate
    lds_cfloat* bias = (lds_cfloat*)(uintptr_t)bias_lds;                                // (left alone, the compiler keeps one per block, hoisted
                                                                                        // out of the tile loop and spilled)
    const int64_t ntiles = (a.n + kTile - 1) / kTile;
    for (int i = lane; i < kPWaveU4; i += 64) pw[i] = make_uint4(0, 0, 0, 0);          // pad slots: finite once
    PROF_DECL

#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t row0 = tile * kTile + w * kRows;                                  // this wave's first sample (rows past n: clamped)
        PROF_TICK(7)
        fill_pe_wave(pw, false, a, row0, lane);
        PROF_TICK(6)
        X8 X;
        float sx;                                                                       // the row scale of X: x = sx * (256 hi + lo) * unit[feature]
        // ---------------- stage 0: encodings only (split bf16), ReLU
        {
            f32x16 f[8];
            float m = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint4* ws = ring_enter(R, b PROF_PASS);
                bias16(f[b], bias + nm::stage_b_off(0) + 32 * b);
                k_bf<4, true>(f[b], pw, g, s, ws, &R);
                PROF_TICK(2)
                m = max16<true>(m, f[b]);
            }
            PROF_TICK(4)
            const float M = row_max(m), inv = inv_of(M);
#pragma unroll
            for (int b = 0; b < 8; ++b) quant16<true>(f[b], inv, X.h[b], X.l[b]);
            sx = scale_of(M);
            PROF_TICK(5)
        }
        // ---------------- stages 1..7: 256 -> 256, ReLU; stage 5 adds the position encoding (four more ring blocks of two output blocks each)
#pragma unroll 1
        for (int st = 1; st <= 7; ++st) {
            const float sxin = sx * (256.f * kappa[st]);
            const int i0 = 8 * st + (st > 5 ? 4 : 0);
            f32x16 f[8];
            float m = 0.f;
            i32x16 tp;                                                                  // block b - 1, dequantised under block b's MFMAs
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const uint4* ws = ring_enter(R, i0 + b PROF_PASS);
                if (b == 0) {
                    k_i8<8>(tp, X, ws, R);
                } else {
                    i32x16 t;
                    k_i8_impl<8, true>(t, X, ws, R, f[b - 1], tp, bias + 256 * st + 32 * (b - 1), sxin, m);
                    tp = t;
                }
                PROF_TICK(2)
            }
            dequant16(f[7], tp, sxin, bias + 256 * st + 32 * 7);
            if (st == 5) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4* ws = ring_enter(R, 48 + u PROF_PASS);
                    k_bf<4, true>(f[2 * u], pw, g, s, ws, &R, 0);
                    k_bf<4, true>(f[2 * u + 1], pw, g, s, ws + 4 * kStepU4, &R, 2);
                    PROF_TICK(2)
                }
                m = 0.f;                                                                // (the running maximum was taken before the encodings)
#pragma unroll
                for (int b = 0; b < 7; ++b) m = max16<true>(m, f[b]);
            }
            m = max16<true>(m, f[7]);
            PROF_TICK(4)
            const float M = row_max(m), inv = inv_of(M);
#pragma unroll
            for (int b = 0; b < 8; ++b) quant16<true>(f[b], inv, X.h[b], X.l[b]);
            sx = scale_of(M);
            PROF_TICK(5)
            if (st == 5 && !(PLAIN && a.sigma_only == 2)) {
                PROF_TICK(7)
                fill_pe_wave(pw, true, a, row0, lane);                                  // the position encoding is done with: direction encoding
                PROF_TICK(6)
            }
        }
        if (PLAIN && a.sigma_only == 2) {
            // ---------------- the plain head: rows 0..3 of block 68 = output_linear's (r, g, b, sigma)
            i32x16 t;
            f32x16 fo;
            k_i8<8>(t, X, ring_enter(R, 68 PROF_PASS), R);
            PROF_TICK(2)
            dequant16(fo, t, sx * (256.f * kappa[8]), bias + nm::stage_b_off(8) + 256);
            const float* up = A.consts8 + nm::stage_b_off(8) + 256;                     // the four rows' units
            const int64_t i = row0 + s;
            if (g == 0 && i < a.n)
                reinterpret_cast<float4*>(a.out)[sample_record(a, i)] = make_float4(fo[0] * up[0], fo[1] * up[1], fo[2] * up[2], fo[3] * up[3] * a.sigma_scale);
            R.off = (block_steps(0) + block_steps(1)) * nm::kStepBytes;                 // blocks 0 and 1 of the next tile are in flight: block 2 is next
            continue;
        }
        // ---------------- stage 8: alpha (row 0 of its block; first in the stream) + feature (linear, 256)
        float sigma;
        {
            const float sxin = sx * (256.f * kappa[8]);
            {
                i32x16 t;
                f32x16 fa;
                k_i8<8>(t, X, ring_enter(R, 68 PROF_PASS), R);
                PROF_TICK(2)
                dequant16(fa, t, sxin, bias + nm::stage_b_off(8) + 256);
                sigma = fa[0] * u_sigma;
            }
            f32x16 f[8];
            float m = 0.f;
#pragma unroll
            for (int b = 0; b < 8; ++b) {                                               // (the riding dequantisation measured no gain here)
                i32x16 t;
                k_i8<8>(t, X, ring_enter(R, 69 + b PROF_PASS), R);
                PROF_TICK(2)
                dequant16(f[b], t, sxin, bias + nm::stage_b_off(8) + 32 * b);
                m = max16<false>(m, f[b]);
            }
            PROF_TICK(4)
            const float M = row_max(m), inv = inv_of(M);
#pragma unroll
            for (int b = 0; b < 8; ++b) quant16<false>(f[b], inv, X.h[b], X.l[b]);
            sx = scale_of(M);
            PROF_TICK(5)
        }
        // ---------------- stage 9: views layer, K = feature(256) ++ d_pe(32), N = 128, ReLU
        {
            const float sxin = sx * (256.f * kappa[9]);
            f32x16 f[4];
            float m = 0.f;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                i32x16 t;
                const uint4* ws = ring_enter(R, 77 + b PROF_PASS);
                k_i8<8>(t, X, ws, R);
                PROF_TICK(2)
                dequant16(f[b], t, sxin, bias + nm::stage_b_off(9) + 32 * b);
                k_bf<2>(f[b], pw, g, s, ws + 8 * kStepU4);
                m = max16<true>(m, f[b]);
            }
            PROF_TICK(4)
            const float M = row_max(m), inv = inv_of(M);
#pragma unroll
            for (int b = 0; b < 4; ++b) quant16<true>(f[b], inv, X.h[b], X.l[b]);
            sx = scale_of(M);
            PROF_TICK(5)
        }
        // ---------------- stage 10: rgb (rows 0..2 of one block), K = 128
        {
            i32x16 t;
            f32x16 fr;
            k_i8<4>(t, X, ring_enter(R, 81 PROF_PASS), R);
            PROF_TICK(2)
            dequant16(fr, t, sx * (256.f * kappa[10]), bias + nm::stage_b_off(10));
            const int64_t i = row0 + s;
            if (g == 0 && i < a.n)
                reinterpret_cast<float4*>(a.out)[sample_record(a, i)] =
                    make_float4(fr[0] * u_r, fr[1] * u_g, fr[2] * u_b,
                                sigma * a.sigma_scale);
        }
    }
#ifdef NM_AS_PROF
    PROF_TICK(7)
    if (lane == 0 && a.prof)
        for (int i = 0; i < 8; ++i) a.prof[((size_t)blockIdx.x * kWaves + w) * 8 + i] = P.acc[i];
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                    // the copy started for a tile that never comes
}

}  // namespace

namespace nm {

int launch_mlp_i8s(const MlpLaunch& L, const void* image8, const float* pts, const float* dirs, const float* origin, const float* direction,
                   const float* z, int64_t n, int S, int in_mode, float sigma_scale, float* out, hipStream_t stream, const MlpChunk* chunk) {
    Args8s A;
    MlpArgs& a = A.a;
    a.ray_idx = chunk ? chunk->ray_idx : nullptr;
    a.n_rays_dev = chunk ? chunk->n_rays_dev : nullptr;
    a.s0 = chunk ? chunk->s0 : 0;
    a.S_total = chunk ? chunk->S_total : S;
    a.wpack = nullptr; a.bias = nullptr;
    a.petab = L.petab;
    a.pts = pts; a.dirs = dirs; a.origin = origin; a.direction = direction; a.z = z;
    a.out = out; a.dbg = nullptr; a.prof = nullptr; a.n = n; a.S = S; a.in_mode = in_mode; a.stop_stage = -2; a.sigma_scale = sigma_scale;
    a.sigma_only = L.plain_head ? 2 : 0;                                                // (PLAIN's exit after block 68)
    a.save_h = nullptr; a.save_hv = nullptr; a.save_bits = nullptr; a.save_h16 = nullptr; a.save_feat16 = nullptr; a.save_hvbits = nullptr; a.save_x0h = nullptr; a.save_d0h = nullptr;
    a.pos = PeSpec{L.pe_kind, L.pos_nfreq, L.pos_octaves};
    a.dir = PeSpec{L.pe_kind, L.dir_nfreq, L.dir_octaves};
    A.consts8 = L.consts8;
    A.image8 = reinterpret_cast<const uint4*>(image8);
    const int64_t ntiles = (n + kTile - 1) / kTile;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    if (getenv("NEUMAN_I8S_DEBUG")) {
        int nb = -1;
        hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, nerf_mlp_i8s_kernel<false>, kWaves * 64, 0);
        fprintf(stderr, "nerf_mlp_i8s_kernel: occupancy %d blocks/CU (%s), grid %d, cus %d\n", nb, hipGetErrorString(e), grid, cus);
    }
#ifdef NM_AS_PROF
    static unsigned long long* d_prof = nullptr;
    const size_t nprof = (size_t)grid * kWaves * 8;
    if (!d_prof) (void)hipMalloc(&d_prof, (size_t)1024 * kWaves * 8 * 8);
    (void)hipMemsetAsync(d_prof, 0, nprof * 8, stream);
    a.prof = d_prof;
#endif
    if (L.plain_head) hipLaunchKernelGGL(nerf_mlp_i8s_kernel<true>, dim3(grid), dim3(kWaves * 64), 0, stream, A);
    else hipLaunchKernelGGL(nerf_mlp_i8s_kernel<false>, dim3(grid), dim3(kWaves * 64), 0, stream, A);
#ifdef NM_AS_PROF
    if (n > 1000000) {
        std::vector<unsigned long long> h(nprof);
        (void)hipStreamSynchronize(stream);
        (void)hipMemcpy(h.data(), d_prof, nprof * 8, hipMemcpyDeviceToHost);
        static const char* names[8] = {"top barrier", "copy issue", "k-loop", "-", "epilogue", "requantise", "encodings", "rest"};
        double acc[8] = {0}, tot = 0;
        for (size_t i = 0; i < nprof; ++i) acc[i & 7] += (double)h[i];
        for (int i = 0; i < 8; ++i) tot += acc[i];
        fprintf(stderr, "mean cycles per wave %.0f:", tot / (grid * (double)kWaves));
        for (int i = 0; i < 8; ++i) fprintf(stderr, "  %s %.1f%%", names[i], 100.0 * acc[i] / tot);
        fprintf(stderr, "\n");
    }
#endif
    return check_launch("nerf_mlp_i8s_kernel");
}

}  // namespace nm
