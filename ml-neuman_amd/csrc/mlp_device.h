// Device helpers of the MFMA kernels of the fused PE + MLP path (mlp.hip: the lock-step and the wave-specialised i8 kernels):
// argument block, sample addressing, encodings, the split-16-bit operand conversion, the MFMA step and the activation write --
// one definition, so that kernels that must agree bit for bit run the same instruction sequences.
#pragma once
#include "common.h"
#include "mlp_layout.h"
#include "mlp_launch.h"

namespace {

using nm::kTileM;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

// NM_PREC_FP16X3: the same three-product scheme on split fp16 (11 + 11 significand bits instead of 8 + 8: the dropped
// wl*xl term and the representation error are ~2^-22, float32 class).  fp16's narrow exponent range is handled by exact
// power-of-two scalings: the weights of stage s are stored as W * 2^k_s, k_s = 8 unless the stage's largest weight needs
// less (mlp_host.hip pack_image: |W| * 2^k_s <= 32000, so a weight's lo part stays a normal number down to |W| ~ 5e-4 for
// ordinary layers), activations and encodings as X * 2^5 (lo normal down to |X| ~ 4e-3, hi clamped at |X| = 2047);
// accumulators therefore carry Y * 2^(k_s + 5) (biases are pre-scaled) and the epilogue multiplies by 2^-k_s before the
// split (the per-stage factors sit behind the bias table).  Parts that fall below fp16's normal range lose at most
// 2^-25 * 2^-5 (activations) / 2^-25 * 2^-k_s (weights) absolutely, flushed or not.
constexpr bool is_split(int prec) { return prec == NM_PREC_BF16X3 || prec == NM_PREC_FP16X3; }
// internal (mlp_bwd.hip's 16-bit form): split-fp16 WEIGHTS against a SINGLE fp16 operand -- two MFMAs per k-step (wl x, wh x), half the operand
// reads; the operand is dZ * s already rounded to fp16 for the weight-gradient products, so nothing is dropped that the step keeps elsewhere
constexpr int kPrecF16W2 = 101;
constexpr bool w_is_split(int prec) { return is_split(prec) || prec == kPrecF16W2; }
constexpr float kF16ActScale = 32.f;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;

constexpr int kThreads = 512;
// LDS map, in uint4 (16 B) units.  Chunk c of an array is [hi: 128 rows][lo: 128 rows] (4 KB), so the lo
// half of any operand is a +2 KB immediate offset from its hi half (ds_read_b128 offset field is 16 bit).
constexpr int kChunkU4 = 2 * kTileM;                    // 256
constexpr int kLoU4 = kTileM;                           // 128
constexpr int H_BASE = 0;
constexpr int P_BASE = nm::kHChunks * kChunkU4;         // 8192
constexpr int LDS_U4 = P_BASE + nm::kPeChunks * kChunkU4;   // 10240 -> 163840 B = the whole CU
static_assert(LDS_U4 * 16 == 160 * 1024, "LDS plan must be exactly 160 KiB");

struct PeSpec {
    int kind;     // NM_PE_POSENC / NM_PE_ROTATE
    int nfreq;
    int octaves;  // 1: bands are consecutive powers of two -> octave recurrence (fill_pe_fast)
};

struct MlpArgs {
    const uint4* wpack;      // packed split-bf16 weight fragments (mlp_layout.h)
    const float* bias;       // kBiasFloats
    const float* petab;      // [0..95] position table, [96..191] direction table
    const float* pts;        // in_mode 0: [n,3]
    const float* dirs;       // in_mode 0: [n,3]
    const float* origin;     // in_mode 1: [R,3]
    const float* direction;  // in_mode 1: [R,3]
    const float* z;          // in_mode 1: [R,S]
    float* out;              // [n,4]
    float* dbg;              // debug dump or nullptr
    unsigned long long* prof;  // PROF instantiation only: [grid*8 waves][8] cycle buckets
    int64_t n;
    int S;
    int in_mode;             // 0: points / directions given; 1: rays + z [R,S]; 2: a chunk of S samples starting at s0 of the listed rays
    const int* ray_idx;      // in_mode 2: [n / S] global ray numbers (compacted list of live rays)
    const int* n_rays_dev;   // in_mode 2: the list's length lives on the device (no host sync between chunks); n = *n_rays_dev * S
    int s0, S_total;         // in_mode 2: z and out are [R, S_total] (x4); this launch covers samples s0 .. s0 + S - 1
    int stop_stage;          // -2 = run everything
    float sigma_scale;
    int sigma_only;          // 2: plain-head net (use_viewdirs=False): the 32-row block after layer 7 holds output_linear's 4 rows = the output;
                             // 1: only the density head is wanted (a pass whose colours the renderer discards): skip the
                             //    feature / views / rgb layers and write (0, 0, 0, sigma)
    PeSpec pos, dir;
    float* save_h;           // SAVE instantiation: [9][n][256] f32 outputs of stages 0..7 (after ReLU) and 8 (feature, linear)
    float* save_hv;          //                     [n][128] f32 output of stage 9 (after ReLU)
    unsigned* save_bits;     //                     nullable: [8][n][8] the signs of stages 0..7: word f >> 5 of (stage, sample); feature 32 w + 8 q + 4 g + j = bit 16 g + 15 - (4 q + j)
    uint4* save_h16;         //                     nullable: [8][n][32] the outputs of stages 0..7 as fp16 of 32 x value (the hi part the next layer's MFMA reads), k-slot
                             //                     order (chunk c, element e <-> feature slot_feature(c, e)) INSTEAD of their float32 copies; save_h = [n][256] feature only (nullable then)
    uint4* save_feat16;      //                     nullable (with save_h16): [n][32] the feature layer's output the same way (fp16 of 32 x value, k-slot order)
    unsigned* save_hvbits;   //                     nullable (with save_h16): [n][4] the signs of stage 9 (views layer): word nb of a sample, bits as save_bits
    uint4* save_x0h;         //                     nullable (with save_h16): [n][8] the position encoding as the kernel holds it: fp16 of 32 x value, natural order, slot 63 zero
    uint4* save_d0h;         //                     nullable (with save_h16): [n][8] the direction encoding likewise in slots 0..31, zeros after, slot 63 = 1 (x 32)
};

// ---- positional encoding feature p of a 3-vector (reference models/vanilla.py:60-92) ---------------
__device__ __forceinline__ float pe_feature(int p, float x0, float x1, float x2, PeSpec spec, const float* __restrict__ tab) {
    const int m = p - 3;
    float a = 0.f;
    bool is_cos = false;
    if (p >= 3 && m < 6 * spec.nfreq) {
        if (spec.kind == NM_PE_POSENC) {                      // [sin(f_b x) (3), cos(f_b x) (3)] per band, vanilla.py:73-76
            const int b = m / 6, r = m - 6 * b;
            const int dim = r >= 3 ? r - 3 : r;
            const float xv = dim == 0 ? x0 : (dim == 1 ? x1 : x2);
            a = xv * tab[b];
            is_cos = r >= 3;
        } else {                                              // rotate: [sin(x B^T) (3N), cos(x B^T) (3N)], vanilla.py:85-88
            const int n3 = 3 * spec.nfreq;
            is_cos = m >= n3;
            const float* b = tab + 3 * (is_cos ? m - n3 : m);
            a = fmaf(x2, b[2], fmaf(x1, b[1], x0 * b[0]));
        }
    }
    float sv, cv;
    sincosf(a, &sv, &cv);                                     // full-range reduction (arguments reach 2^9 * |x|)
    if (p < 3) return p == 0 ? x0 : (p == 1 ? x1 : x2);
    if (m >= 6 * spec.nfreq) return 0.f;                      // zero padding slots
    return is_cos ? cv : sv;
}

// sample i of the launch -> the 3-vector to encode (position or direction) and, for the stores, its record in `out`
__device__ __forceinline__ int64_t sample_record(const MlpArgs& a, int64_t i) {
    if (a.in_mode != 2) return i;
    const int64_t j = i / a.S;
    return (int64_t)a.ray_idx[j] * a.S_total + a.s0 + (i - j * a.S);
}
__device__ __forceinline__ void sample_input(const MlpArgs& a, int64_t i, bool is_dir, float& x0, float& x1, float& x2) {
    if (a.in_mode == 0) {
        const float* src = (is_dir ? a.dirs : a.pts) + i * 3;
        x0 = src[0]; x1 = src[1]; x2 = src[2];
        return;
    }
    int64_t r, zi;
    if (a.in_mode == 1) {
        r = i / a.S;
        zi = i;
    } else {
        const int64_t j = i / a.S;
        r = a.ray_idx[j];
        zi = r * a.S_total + a.s0 + (i - j * a.S);
    }
    const float* d = a.direction + r * 3;
    if (is_dir) {
        x0 = d[0]; x1 = d[1]; x2 = d[2];                        // ray_utils.py:132
    } else {
        const float zz = a.z[zi];
        const float* o = a.origin + r * 3;
        x0 = o[0] + d[0] * zz;                                  // ray_utils.py:131 (two roundings: built with -ffp-contract=off)
        x1 = o[1] + d[1] * zz;
        x2 = o[2] + d[2] * zz;
    }
}

// split 8 f32 into 16-bit hi and lo chunks (RNE both times; x - float(hi) is exact in f32).  F16: fp16 parts of
// v * scale (scale a power of two: exact), clamped below fp16's overflow so that a huge activation saturates instead of
// becoming inf - inf = NaN.
template <bool RELU, bool F16 = false>
__device__ __forceinline__ void split8(const float* v, uint4& hi, uint4& lo, float scale = 1.f) {
    unsigned h[4], l[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        f32x2 a = {v[2 * p], v[2 * p + 1]};
        if (F16) {
            a.x = __builtin_amdgcn_fmed3f(a.x * scale, RELU ? 0.f : -65504.f, 65504.f);
            a.y = __builtin_amdgcn_fmed3f(a.y * scale, RELU ? 0.f : -65504.f, 65504.f);
            const f16x2 hb = __builtin_convertvector(a, f16x2);
            const f32x2 hf = __builtin_convertvector(hb, f32x2);
            const f32x2 r = {a.x - hf.x, a.y - hf.y};
            const f16x2 lb = __builtin_convertvector(r, f16x2);
            h[p] = __builtin_bit_cast(unsigned, hb);
            l[p] = __builtin_bit_cast(unsigned, lb);
            continue;
        }
        if (RELU) {
            a.x = fmaxf(a.x, 0.f);
            a.y = fmaxf(a.y, 0.f);
        }
        const bf16x2 hb = __builtin_convertvector(a, bf16x2);
        const f32x2 hf = __builtin_convertvector(hb, f32x2);
        const f32x2 r = {a.x - hf.x, a.y - hf.y};              // two scalar v_sub_f32: v_pk_add_f32 is slow beside MFMAs on gfx950
        const bf16x2 lb = __builtin_convertvector(r, bf16x2);
        h[p] = __builtin_bit_cast(unsigned, hb);
        l[p] = __builtin_bit_cast(unsigned, lb);
    }
    hi = make_uint4(h[0], h[1], h[2], h[3]);
    lo = make_uint4(l[0], l[1], l[2], l[3]);
}

__device__ __forceinline__ bf16x8 as_bf16x8(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

// ---- one run of k-steps: acc[mb] += W(block) * X(rows row0 + 32*mb ..) -------------------------------
//   wsrc    : buffer descriptor of the weight image (SGPRs); voff = lane*16 is the only per-lane address
//   soff    : wave-uniform byte offset of (stage, block, first step of the run)
//   xh      : this lane's pointer into the activation array at (first chunk + g, hi half, row0 + lane&31)
//   nsteps  : even
typedef __attribute__((vector_size(16))) unsigned int v4u;
__device__ __forceinline__ bf16x8 ld_w(__amdgpu_buffer_rsrc_t wsrc, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, soff, 0));
}

// The wave's weight stream is one register-resident pipeline across runs, stages and tiles: W holds the fragments of
// the next two k-steps to be consumed.  The last iteration of a run does not prefetch past its own end but the first
// two steps of the NEXT run (next_soff), so the L2 latency of every run's head is hidden behind the epilogue /
// barriers in between instead of being exposed 13 times per tile.
struct WPre {
    bf16x8 h[2], l[2];
};
template <int PREC>
__device__ __forceinline__ void w_prefetch(WPre& W, __amdgpu_buffer_rsrc_t wsrc, int voff, int soff) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        W.h[u] = ld_w(wsrc, voff, soff + u * nm::kStepBytes);
        if (w_is_split(PREC)) W.l[u] = ld_w(wsrc, voff, soff + u * nm::kStepBytes + 1024);
    }
}

template <int MB, int PREC>
__device__ __forceinline__ void x_load(uint4 (&h)[MB], uint4 (&l)[MB], const uint4* ph) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        h[mb] = ph[mb * 32];
        if (is_split(PREC)) l[mb] = ph[kLoU4 + mb * 32];
    }
}
__device__ __forceinline__ f16x8 as_f16x8(uint4 v) { return __builtin_bit_cast(f16x8, v); }
template <int MB, int PREC>
__device__ __forceinline__ void mfma_step(f32x16 (&acc)[MB], bf16x8 wh, bf16x8 wl, const uint4 (&xh)[MB], const uint4 (&xl)[MB]) {
    if (PREC == kPrecF16W2) {
        const f16x8 fh = __builtin_bit_cast(f16x8, wh), fl = __builtin_bit_cast(f16x8, wl);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, as_f16x8(xh[mb]), acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(xh[mb]), acc[mb], 0, 0, 0);
        return;
    }
    if (PREC == NM_PREC_FP16X3) {                       // (the fragment registers hold fp16 bit patterns in this mode)
        const f16x8 fh = __builtin_bit_cast(f16x8, wh), fl = __builtin_bit_cast(f16x8, wl);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(xl[mb]), acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fl, as_f16x8(xh[mb]), acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fh, as_f16x8(xh[mb]), acc[mb], 0, 0, 0);
        return;
    }
    if (PREC == NM_PREC_BF16X3) {
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, as_bf16x8(xl[mb]), acc[mb], 0, 0, 0);
#pragma unroll
        for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, as_bf16x8(xh[mb]), acc[mb], 0, 0, 0);
    }
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, as_bf16x8(xh[mb]), acc[mb], 0, 0, 0);
}
template <int MB, int PREC>
__device__ __forceinline__ void k_run(f32x16 (&acc)[MB], WPre& W, __amdgpu_buffer_rsrc_t wsrc, int voff, int soff, int next_soff,
                                      const uint4* xh, int nsteps) {
    // Activation fragments of step t + 1 requested before the MFMAs of step t (two register sets): pays for the single-MFMA
    // NM_PREC_BF16 steps (+6 %), where an LDS round trip per step is exposed; with three MFMAs per step (bf16x3) the partner
    // wave of the SIMD already covers it and the extra registers cost more than they buy (-2.5 %, same GPU, A/B).
    if (PREC == NM_PREC_BF16) {
    uint4 xah[MB], xal[MB];
    x_load<MB, PREC>(xah, xal, xh);
#pragma unroll 1
    for (int t = 0; t < nsteps; t += 2) {
        const int pf = (t + 2 < nsteps) ? soff + (t + 2) * nm::kStepBytes : next_soff;   // wave-uniform
        WPre N;
        w_prefetch<PREC>(N, wsrc, voff, pf);
        uint4 xbh[MB], xbl[MB];
        x_load<MB, PREC>(xbh, xbl, xh + (t + 1) * (2 * kChunkU4));
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<MB, PREC>(acc, W.h[0], W.l[0], xah, xal);
        __builtin_amdgcn_sched_barrier(0);
        const int tn = t + 2 < nsteps ? t + 2 : t + 1;                                   // (last iteration: a harmless re-read)
        x_load<MB, PREC>(xah, xal, xh + tn * (2 * kChunkU4));
        __builtin_amdgcn_sched_barrier(0);
        mfma_step<MB, PREC>(acc, W.h[1], W.l[1], xbh, xbl);
        __builtin_amdgcn_sched_barrier(0);
        W = N;
    }
    return;
    }
#pragma unroll 1
    for (int t = 0; t < nsteps; t += 2) {
        const int pf = (t + 2 < nsteps) ? soff + (t + 2) * nm::kStepBytes : next_soff;   // wave-uniform
        WPre N;
        w_prefetch<PREC>(N, wsrc, voff, pf);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            uint4 bh[MB], bl[MB];
            x_load<MB, PREC>(bh, bl, xh + (t + u) * (2 * kChunkU4));
            mfma_step<MB, PREC>(acc, W.h[u], W.l[u], bh, bl);
        }
        W = N;
    }
}

// bias of this lane's 16 features (reg&3) + 8*(reg>>2) + 4*g of a 32-feature block: loaded early (before the previous
// stage's epilogue), used as the accumulators' initial value
struct BiasRegs {
    float4 q[4];
};
__device__ __forceinline__ void bias_prefetch(BiasRegs& B, const float* __restrict__ bias_blk, int g) {
#pragma unroll
    for (int q = 0; q < 4; ++q) B.q[q] = *reinterpret_cast<const float4*>(bias_blk + 8 * q + 4 * g);
}
template <int MB>
__device__ __forceinline__ void init_bias(f32x16 (&acc)[MB], const BiasRegs& B) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[mb][4 * q + 0] = B.q[q].x; acc[mb][4 * q + 1] = B.q[q].y;
            acc[mb][4 * q + 2] = B.q[q].z; acc[mb][4 * q + 3] = B.q[q].w;
        }
}

// write a wave's accumulators as the next layer's input: block blk, sample rows row0 + 32*mb + s
// The epilogue is split around the "all reads of H done" barrier: the VALU half (ReLU + hi/lo split) runs BEFORE it --
// the wave that finishes its k-loop first (the older wave of each SIMD wins MFMA arbitration) converts while its partner
// is still issuing MFMAs, on the otherwise idle VALU -- and only the ds_write_b128s remain after the barrier.
template <int MB>
struct ActRegs {
    uint4 hi[MB][2], lo[MB][2];
};
template <int MB, bool RELU, int PREC>
__device__ __forceinline__ void convert_act(const f32x16 (&acc)[MB], ActRegs<MB>& r, float acc2act = 1.f) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = acc[mb][8 * qp + e];
            split8<RELU, PREC == NM_PREC_FP16X3>(v, r.hi[mb][qp], r.lo[mb][qp], acc2act);
        }
}
template <int MB, int PREC>
__device__ __forceinline__ void write_act(const ActRegs<MB>& r, uint4* lds, int blk, int row0, int g, int s) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            const int idx = H_BASE + (4 * blk + 2 * qp + g) * kChunkU4 + row0 + 32 * mb + s;
            lds[idx] = r.hi[mb][qp];
            if (is_split(PREC)) lds[idx + kLoU4] = r.lo[mb][qp];
        }
}

// fill `nchunks` PE chunks for the tile: work item = (chunk, sample); 8 features -> one b128 write per array
// (`nthreads` threads numbered by tid cover rows row0 .. row0 + 2^rshift - 1: the whole tile, or one wave group's half)
template <bool F16 = false>
__device__ __forceinline__ void fill_pe(uint4* lds, int nchunks, bool is_dir, const MlpArgs& a, int64_t base, int tid,
                                        int nthreads = kThreads, int row0 = 0, int rshift = 7) {
    const PeSpec spec = is_dir ? a.dir : a.pos;
    const float* tab = a.petab + (is_dir ? 96 : 0);
    for (int item = tid; item < (nchunks << rshift); item += nthreads) {
        const int c = item >> rshift, row = row0 + (item & ((1 << rshift) - 1));
        int64_t i = base + row;
        if (i >= a.n) i = a.n - 1;                              // tail rows recompute the last sample (never stored)
        float x0, x1, x2;
        sample_input(a, i, is_dir, x0, x1, x2);
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = pe_feature(8 * c + e, x0, x1, x2, spec, tab);
        uint4 hi, lo;
        split8<false, F16>(v, hi, lo, kF16ActScale);
        lds[P_BASE + c * kChunkU4 + row] = hi;
        lds[P_BASE + c * kChunkU4 + kLoU4 + row] = lo;
    }
}

// Octave recurrence for the encodings (valid when the bands are consecutive powers of two -- the reference defaults,
// checked on the host): f32 scaling by 2^b is exact, so the argument of band b is exactly 2^b * a0, where a0 is the band-0
// argument (x_j for posenc, fmaf-chain(x, B[j]) for rotate).  One f64 sincos(a0) per (sample, component) and the double
// angle formulas in f64 (error doubles per octave from 1e-16: 1e-13 at band 9) give sin/cos(2^b a0) rounded to f32 --
// within an ulp of the reference's sinf(fl(x * f_b)) -- for ~1/6 of the instructions of 2N full-range sincosf calls.
// Work item = (component j, sample); each of the 2N values is one 2-byte LDS store per half.
// f64 sin/cos for the band-0 arguments (|a| up to a few scene units; valid to |a| ~ 1e9): two-term Cody-Waite reduction
// by pi/2 with fma, then the fdlibm minimax kernels on [-pi/4, pi/4].  Absolute error ~1e-16.  Written out instead of
// calling ocml's sincos(double) because that one carries a Payne-Hanek path with a private (scratch) array, and any
// scratch in this kernel competes with the 2.4 MB weight image for the XCD's 4 MB L2.
__device__ __forceinline__ void sincos_f64(double a, double& sn, double& cs) {
    const double fn = rint(a * 6.36619772367581382433e-01);
    double r = fma(-fn, 1.5707963267948966, a);
    r = fma(-fn, 6.123233995736766e-17, r);
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 +
                      z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = r + (z * r) * (-1.66666666666666324348e-01 + z * ps);
    const double pc = z * (4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
                      z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11)))));
    const double c = 1.0 - (0.5 * z - z * pc);
    const int q = (int)fn & 3;                              // a = r + q*pi/2 (mod 2 pi)
    sn = (q & 1) ? c : s;
    cs = (q & 1) ? s : c;
    if (q == 1 || q == 2) cs = -cs;
    if (q >= 2) sn = -sn;
}

template <bool F16 = false>
__device__ __forceinline__ void fill_pe_fast(uint4* lds, bool is_dir, const MlpArgs& a, int64_t base, int tid, int row0 = 0,
                                             int rshift = 7) {
    const PeSpec spec = is_dir ? a.dir : a.pos;
    const float* tab = a.petab + (is_dir ? 96 : 0);
    if (tid >= (3 << rshift)) return;
    const int j = tid >> rshift, row = row0 + (tid & ((1 << rshift) - 1));   // j is wave-uniform (rshift >= 6)
    int64_t i = base + row;
    if (i >= a.n) i = a.n - 1;
    float x0, x1, x2;
    sample_input(a, i, is_dir, x0, x1, x2);
    const float xj = j == 0 ? x0 : (j == 1 ? x1 : x2);
    float a0;
    if (spec.kind == NM_PE_POSENC) a0 = xj * tab[0];
    else a0 = fmaf(x2, tab[3 * j + 2], fmaf(x1, tab[3 * j + 1], x0 * tab[3 * j]));
    unsigned short* hi = reinterpret_cast<unsigned short*>(lds + P_BASE);
    unsigned short* lo = hi + kLoU4 * 8;
    auto put = [&](int p, float v) {                          // feature slot p of this row: chunk p>>3, element p&7
        const int off = ((p >> 3) * kChunkU4 + row) * 8 + (p & 7);
        if (F16) {
            const float sv = v * kF16ActScale;                    // |v| <= max(1, |x|): far below fp16's range after scaling
            const _Float16 hb = (_Float16)sv;
            const _Float16 lb = (_Float16)(sv - (float)hb);
            hi[off] = __builtin_bit_cast(unsigned short, hb);
            lo[off] = __builtin_bit_cast(unsigned short, lb);
            return;
        }
        const bf16x2 hb = __builtin_convertvector((f32x2){v, 0.f}, bf16x2);
        const f32x2 hf = __builtin_convertvector(hb, f32x2);
        const bf16x2 lb = __builtin_convertvector((f32x2){v - hf.x, 0.f}, bf16x2);
        hi[off] = (unsigned short)(__builtin_bit_cast(unsigned, hb) & 0xffffu);
        lo[off] = (unsigned short)(__builtin_bit_cast(unsigned, lb) & 0xffffu);
    };
    put(j, xj);                                               // include_input: features 0..2
    double sn, cs;
    sincos_f64((double)a0, sn, cs);
    const int n3 = 3 * spec.nfreq;
    for (int b = 0; b < spec.nfreq; ++b) {
        if (spec.kind == NM_PE_POSENC) {                      // [sin(f_b x)(3), cos(f_b x)(3)] per band, vanilla.py:73-76
            put(3 + 6 * b + j, (float)sn);
            put(3 + 6 * b + 3 + j, (float)cs);
        } else {                                              // [sin(x B^T)(3N), cos(x B^T)(3N)], vanilla.py:85-88
            put(3 + 3 * b + j, (float)sn);
            put(3 + n3 + 3 * b + j, (float)cs);
        }
        const double s2 = 2.0 * sn * cs, c2 = 1.0 - 2.0 * sn * sn;
        sn = s2;
        cs = c2;
    }
}

template <bool F16 = false>
__device__ __forceinline__ void fill_pe_any(uint4* lds, bool is_dir, const MlpArgs& a, int64_t base, int tid,
                                            int nthreads = kThreads, int row0 = 0, int rshift = 7) {
    if ((is_dir ? a.dir : a.pos).octaves) fill_pe_fast<F16>(lds, is_dir, a, base, tid, row0, rshift);
    else fill_pe<F16>(lds, is_dir ? 4 : nm::kPeChunks, is_dir, a, base, tid, nthreads, row0, rshift);
}

// debug: dump `width` features of the tile from the H (or P) arrays as f32 [n, width] in natural order
template <bool F16 = false>
__device__ __forceinline__ void dump_act(const uint4* lds, bool from_pe, int width, const MlpArgs& a, int64_t base, int tid,
                                         int nthreads = kThreads, int row0 = 0, int nrows = kTileM) {
    const unsigned short* hi = reinterpret_cast<const unsigned short*>(lds + (from_pe ? P_BASE : H_BASE));
    const unsigned short* lo = hi + kLoU4 * 8;
    for (int item = tid; item < nrows * width; item += nthreads) {
        const int row = row0 + item / width, n = item % width;
        if (base + row >= a.n) continue;
        const int c = from_pe ? (n >> 3) : nm::feature_chunk(n);
        const int e = from_pe ? (n & 7) : nm::feature_elem(n);
        const int off = (c * kChunkU4 + row) * 8 + e;
        if (F16) {
            a.dbg[(base + row) * width + n] = ((float)__builtin_bit_cast(_Float16, hi[off]) + (float)__builtin_bit_cast(_Float16, lo[off])) *
                                              (1.f / kF16ActScale);
            continue;
        }
        const float h = __uint_as_float((unsigned)hi[off] << 16);
        const float l = __uint_as_float((unsigned)lo[off] << 16);
        a.dbg[(base + row) * width + n] = h + l;
    }
}

// PROF: accumulate s_memtime deltas per wave into 6 buckets {pe, k-loops, wait before epilogue, epilogue, wait after
// epilogue, tail} (a.prof[(block*8 + wave)*8 + bucket]); a separate instantiation so the production kernel is untouched.
// in_mode 2 launches size themselves on the device: the live-ray count is the output of the compaction that ran just before
__device__ __forceinline__ MlpArgs resolve_args(MlpArgs a) {
    if (a.in_mode == 2 && a.n_rays_dev) a.n = (int64_t)(*a.n_rays_dev) * a.S;
    return a;
}

}  // namespace
