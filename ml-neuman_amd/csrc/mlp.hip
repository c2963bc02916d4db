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
        if (is_split(PREC)) W.l[u] = ld_w(wsrc, voff, soff + u * nm::kStepBytes + 1024);
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

template <int PREC, bool PROF, bool SAVE = false>
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
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) save_block(a.save_h + (int64_t)st * a.n * 256, 256, acc[mb], w, base + 32 * mb + s, acc2out(st), true);
            }
            ActRegs<4> ar;
            convert_act<4, true, PREC>(acc, ar, acc2act(st));
            NM_TICK(3)
            __syncthreads();                                              // every wave has finished reading H (and P)
            NM_TICK(2)
            write_act<4, PREC>(ar, lds, w, 0, g, s);
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
            if (SAVE) {
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) save_block(a.save_h + (int64_t)8 * a.n * 256, 256, acc[mb], w, base + 32 * mb + s, acc2out(8), false);
            }
            ActRegs<4> ar;
            convert_act<4, false, PREC>(acc, ar, acc2act(8));
            NM_TICK(3)
            __syncthreads();
            NM_TICK(2)
            write_act<4, PREC>(ar, lds, w, 0, g, s);
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
    a.save_h = L.save_h; a.save_hv = L.save_hv;
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
    if (L.save_h) {                                               // the training forward: split fp16, the full head, activations kept
        hipLaunchKernelGGL((nerf_mlp_kernel<NM_PREC_FP16X3, false, true>), dim3(grid), dim3(kThreads), 0, stream, a);
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
