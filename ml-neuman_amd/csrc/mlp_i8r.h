// NM_PREC_I8X3, resident weights (nerf_mlp_i8r_kernel): included into mlp.hip's anonymous namespace.
//
// nerf_mlp_i8w_kernel hides the requantisation by splitting the tile's SAMPLES between two wave groups that run half a stage
// out of phase; its price is that both groups stream the whole weight image (2 x 1.27 MB per 128-sample tile, 43 B/clk/CU at
// full matrix rate against ~56 B/clk/CU of L2 bandwidth): its k-loops wait on L2 21 % of the time.
//
// Here the two groups split the FEATURES (group A = waves 0..3 = features 0..127, group B = waves 4..7 = features 128..255; wave w
// owns the 32-feature block w) and both process BOTH 64-sample halves of the tile, one after the other:
//
//     slot:      0          1          2          3          4          5
//     group A:   M(H0,s)    E(H0,s)    M(H1,s)    E(H1,s)    M(H0,s+1)  E(H0,s+1) ...
//     group B:   ..         M(H0,s)    E(H0,s)    M(H1,s)    E(H1,s)    M(H0,s+1) ...      (one slot behind)
//
// so on every SIMD one wave issues MFMAs while its partner requantises, as before -- but a wave's 16 KB of weights for a stage are
// used for H0 and again, two slots later, for H1, and stay in 64 VGPRs in between: each fragment is fetched ONCE per CU and tile,
// and never inside a k-loop (the next stage's fragments replace the current ones step by step during M(H1), two slots ahead of
// their first use).  What makes the feature split possible:
//   * a layer's input is complete only after BOTH groups' E slots, hence the two interleaved halves (A's M(H0,s+1) comes two
//     slots after B's E(H0,s));
//   * the row scale of the requantisation is taken per GROUP (two scales per sample row: features 0..127, 128..255) -- a row
//     maximum over all 256 features would couple the groups inside a slot.  The consumer's k-loop runs k-steps 0..3 (group A's
//     features, int32 sum tA) and 4..7 (group B's, tB) separately and dequantises kappa * (sA * tA + sB * tB); finer scales than
//     one per row, so the error does not grow;
//   * group A finishes a half one slot before group B has read that half's previous activations: A's features (blocks 0..3) are
//     therefore double-buffered in LDS by stage parity (32 KB more: 96 + 32 KB of encodings + 22 KB of tables = 150 of the 160 KB);
//     B's are written after everybody has read them.  Row scales are double-buffered by stage parity as well.
// Stages 9 and 10 (128-wide, 1 block) and the alpha block have work for one group only: A takes H0, B takes H1.
//
// Per slot: part 1 | workgroup barrier | part 2 | workgroup barrier; both groups execute the same number of barriers (B two
// extra before its first slot, A two after its last), so the phase relation holds by construction.

// activations: chunks 0..7 = blocks 0..3 (group A's features) x 2 stage parities, chunks 16..23 = blocks 4..7 (group B's features)
constexpr int HA8_BASE = 0;                                  // [2 parities][8 chunks][hi: 128 rows | lo: 128 rows][16 B]  (64 KB)
constexpr int HB8_BASE = 16 * kChunkU4;                      // [8 chunks] ...                                            (32 KB)
constexpr int R8_MAX = 24 * kChunkU4;                        // 6144: row-max partials [2 groups][4 waves][128 rows] f32
constexpr int R8_SCALE = R8_MAX + 2 * 4 * kTileM / 4;        // row scales [2 parities][2 groups][128 rows] f32
constexpr int R8_CONST = R8_SCALE + 2 * 2 * kTileM / 4;      // [units | biases | kappa] of all stages
static_assert(R8_CONST + kConst8Floats / 4 <= P_BASE, "i8r scratch must fit below the PE buffer");

struct WRes {
    v4u h[8], l[8];                                          // the wave's block of the current stage: 8 k-steps x (hi | lo) fragments
};
struct Pending8 {                                            // a half's quantised output of one wave, not yet in LDS
    uint4 hi[2], lo[2];                                      // [mb]: the lane's own 16 features of the block, hi / lo limbs
    float scale[2];                                          // row scale of rows 32 mb + s (written by the group's wave 0)
};

__device__ __forceinline__ void wres_load(WRes& R, int t, __amdgpu_buffer_rsrc_t wsrc, int voff, int soff) {
    R.h[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, soff, 0);
    R.l[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, soff + 1024, 0);
}

// k-steps T0 .. T0+NT-1 of one feature block x ONE sample block on the i8 MFMA with the resident fragments -> t = hh * 256 + cross.
// (One sample block at a time: the fragments are in registers, so a second pass over them is free, and 32 accumulator registers are
// live instead of 64.)  NEXT: after the last use of step t its registers receive step t of the NEXT stage's block (byte offset
// next0 + t * kStepBytes).  The cross accumulator takes two of the three MFMAs of a step, separated by the hi.hi one.
template <int T0, int NT, bool NEXT>
__device__ __forceinline__ void r_run8(i32x16& t_out, WRes& R, const uint4* xh, __amdgpu_buffer_rsrc_t wsrc, int voff, int next0) {
    i32x16 ah, ac;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = 0; ac[r] = 0; }
    uint4 xq[2][2];                                          // [buffer][limb]: fragments one step ahead
    xq[0][0] = xh[T0 * (2 * kChunkU4)];
    xq[0][1] = xh[T0 * (2 * kChunkU4) + kLoU4];
    __builtin_amdgcn_s_setprio(kWPrio);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const uint4 xhc = xq[t & 1][0], xlc = xq[t & 1][1];
        if (t + 1 < NT) {
            xq[(t + 1) & 1][0] = xh[(T0 + t + 1) * (2 * kChunkU4)];
            xq[(t + 1) & 1][1] = xh[(T0 + t + 1) * (2 * kChunkU4) + kLoU4];
        }
        const v4u wh = R.h[T0 + t], wl = R.l[T0 + t];
        __builtin_amdgcn_sched_barrier(0);
        ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xlc), ac, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xhc), ah, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wl), __builtin_bit_cast(i32x4, xhc), ac, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (NEXT) wres_load(R, T0 + t, wsrc, voff, next0 + (T0 + t) * nm::kStepBytes);
    }
    __builtin_amdgcn_s_setprio(kEPrio);
#pragma unroll
    for (int r = 0; r < 16; ++r) t_out[r] = (ah[r] << 8) + ac[r];
}
// both sample blocks of a half as ONE pipeline of 2 * NT (sample block, k-step) items: the activation fragments of item i + 2 are
// requested before the MFMAs of item i (one wave alone on its SIMD's matrix pipe hides no LDS latency for free, and three MFMAs
// are only ~100 cycles), the accumulators are reset between the blocks; only the second block's pass may replace the fragments
template <int T0, int NT, bool NEXT>
__device__ __forceinline__ void r_run8x2(i32x16 (&t)[2], WRes& R, const uint4* xh, __amdgpu_buffer_rsrc_t wsrc, int voff, int next0) {
    constexpr int N = 2 * NT;
    auto xaddr = [&](int i) { return xh + (T0 + (i % NT)) * (2 * kChunkU4) + (i / NT) * 32; };
    uint4 xq[3][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { xq[i][0] = xaddr(i)[0]; xq[i][1] = xaddr(i)[kLoU4]; }
    i32x16 ah, ac;
#pragma unroll
    for (int r = 0; r < 16; ++r) { ah[r] = 0; ac[r] = 0; }
    __builtin_amdgcn_s_setprio(kWPrio);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const int k = T0 + (i % NT);
        const uint4 xhc = xq[i % 3][0], xlc = xq[i % 3][1];
        if (i + 2 < N) { xq[(i + 2) % 3][0] = xaddr(i + 2)[0]; xq[(i + 2) % 3][1] = xaddr(i + 2)[kLoU4]; }
        const v4u wh = R.h[k], wl = R.l[k];
        __builtin_amdgcn_sched_barrier(0);
#ifdef I8R_EXP_NO_MFMA
        ac[0] += (int)(wh[0] + xlc.x); ah[0] += (int)(wl[0] + xhc.x);
#else
        ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xlc), ac, 0, 0, 0);
        ah = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xhc), ah, 0, 0, 0);
        ac = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wl), __builtin_bit_cast(i32x4, xhc), ac, 0, 0, 0);
#endif
        __builtin_amdgcn_sched_barrier(0);
#ifndef I8R_EXP_NO_WLOAD
        if (NEXT && i >= NT) wres_load(R, k, wsrc, voff, next0 + k * nm::kStepBytes);
#endif
        if (i % NT == NT - 1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { t[i / NT][r] = (ah[r] << 8) + ac[r]; ah[r] = 0; ac[r] = 0; }
        }
    }
    __builtin_amdgcn_s_setprio(kEPrio);
}

// NT split-bf16 k-steps (16 k each) of one feature block x MB sample blocks from the encoding buffer: f += W_pe . PE
template <int MB, int NT>
__device__ __forceinline__ void r_runbf(f32x16 (&f)[MB], const v4u (&wh)[NT], const v4u (&wl)[NT], const uint4* xp) {
    __builtin_amdgcn_s_setprio(kWPrio);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        uint4 xhc[MB], xlc[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) { xhc[m] = xp[t * (2 * kChunkU4) + m * 32]; xlc[m] = xp[t * (2 * kChunkU4) + kLoU4 + m * 32]; }
#pragma unroll
        for (int m = 0; m < MB; ++m) f[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[t]), as_bf16x8(xlc[m]), f[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MB; ++m) f[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[t]), as_bf16x8(xhc[m]), f[m], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < MB; ++m) f[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh[t]), as_bf16x8(xhc[m]), f[m], 0, 0, 0);
    }
    __builtin_amdgcn_s_setprio(kEPrio);
}

// f += tA * sA + tB * sB  (scales already multiplied by 256 * kappa of the stage)
template <int MB>
__device__ __forceinline__ void dequant2(f32x16 (&f)[MB], const i32x16 (&tA)[MB], const i32x16 (&tB)[MB], const float (&sA)[MB], const float (&sB)[MB]) {
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int r = 0; r < 16; ++r) f[mb][r] = fmaf((float)tA[mb][r], sA[mb], fmaf((float)tB[mb][r], sB[mb], f[mb][r]));
}

template <int MB>
__device__ __forceinline__ void init_bias_r(f32x16 (&f)[MB], const float* cst, int cblk, int g) {
    init_bias8<MB>(f, cst, cblk, g);
}

// quantise one block x 2 sample blocks with the group's row maxima (4 partials) into limb registers
template <bool RELU>
__device__ __forceinline__ void quant_regs(const f32x16 (&f)[2], const float* smaxG, int row0, int s, Pending8& P) {
    float part[2][4];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb)
#pragma unroll
        for (int p = 0; p < 4; ++p) part[mb][p] = smaxG[p * kTileM + row0 + 32 * mb + s];
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const float M = fmaxf(fmaxf(part[mb][0], part[mb][1]), fmaxf(part[mb][2], part[mb][3]));
        const float c = (float)nm::kFixedMax / 32767.f;
        const float inv = M > 0.f ? c * __builtin_amdgcn_rcpf(M) : 0.f;
        P.scale[mb] = M > 0.f ? M * (1.f / (float)nm::kFixedMax) : 1.f;
        i16x2 Q[8], Y[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float y0 = f[mb][2 * i] * inv, y1 = f[mb][2 * i + 1] * inv;
            if (RELU) {
                y0 = __builtin_amdgcn_fmed3f(y0, 0.f, 1.f);
                y1 = __builtin_amdgcn_fmed3f(y1, 0.f, 1.f);
            }
            const i16x2 p = __builtin_amdgcn_cvt_pknorm_i16(y0, y1);
            Q[i] = p;
            Y[i] = p + (i16x2){128, 128};
        }
        unsigned lo[4], hi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            lo[k] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, Q[2 * k + 1]), __builtin_bit_cast(unsigned, Q[2 * k]), 0x06040200u);
            hi[k] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, Y[2 * k + 1]), __builtin_bit_cast(unsigned, Y[2 * k]), 0x07050301u);
        }
        P.hi[mb] = make_uint4(hi[0], hi[1], hi[2], hi[3]);
        P.lo[mb] = make_uint4(lo[0], lo[1], lo[2], lo[3]);
    }
}
// chunk0: first chunk (uint4 index / kChunkU4) of the wave's block in the buffer it writes
__device__ __forceinline__ void store_regs(const Pending8& P, uint4* lds, float* sscale_out, int chunk0, int row0, int g, int s, bool write_scale) {
#pragma unroll
    for (int mb = 0; mb < 2; ++mb) {
        const int idx = (chunk0 + g) * kChunkU4 + row0 + 32 * mb + s;
        lds[idx] = P.hi[mb];
        lds[idx + kLoU4] = P.lo[mb];
        if (write_scale && g == 0) sscale_out[row0 + 32 * mb + s] = P.scale[mb];
    }
}

__global__ __launch_bounds__(kThreads, 2) void nerf_mlp_i8r_kernel(const MlpArgs8 A) {
    __shared__ uint4 lds[LDS_U4];
    const MlpArgs a = resolve_args(A.a);
    const int tid0 = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid0 >> 6);
    const int G = w >> 2, wq = w & 3;                            // wave group (feature half) and the wave's rank in it
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(A.wimg8), 0, (int)(nm::kWeightBytes8 + nm::kWeightPadBytes), 0x00020000);
    const int64_t ntiles = (a.n + kTileM - 1) / kTileM;
    float* smaxG = reinterpret_cast<float*>(lds + R8_MAX) + G * 4 * kTileM;
    float* sscale = reinterpret_cast<float*>(lds + R8_SCALE);        // [parity][group][row]
    float* cst = reinterpret_cast<float*>(lds + R8_CONST);
    constexpr int kS = nm::kStepBytes;
    constexpr int kKap = 2 * nm::kBiasFloats;
    auto scales_of = [&](int st) { return sscale + (st & 1) * 2 * kTileM; };     // the scales of stage st's OUTPUT
    auto w_off = [](int st, int blk, int step) {                                 // byte offset of (stage, block, step) in the image
        const int per = (st == 0 ? 4 : st == 5 ? 12 : st == 9 ? 10 : st == 10 ? 4 : 8);                    // steps per block (mlp_layout.h stage_shape8)
        const int base = st <= 5 ? (st == 0 ? 0 : 32 + 64 * (st - 1)) : st <= 8 ? 384 + 64 * (st - 6) : st == 9 ? 584 : 624;   // steps before the stage
        return (base + blk * per + step) * kS;
    };

    for (int i = tid0; i < nm::kPeChunks * kChunkU4; i += kThreads) lds[P_BASE + i] = make_uint4(0, 0, 0, 0);
    for (int i = tid0; i < kConst8Floats / 4; i += kThreads) lds[R8_CONST + i] = reinterpret_cast<const uint4*>(A.consts8)[i];
    __syncthreads();
    {   // the first tile's position encoding: group A fills rows 0..63, group B rows 64..127
        const int64_t base0 = (int64_t)blockIdx.x * kTileM;
        if (base0 < a.n) fill_pe_any(lds, false, a, base0, tid0 & 255, 256, 64 * G, 6);
    }
    __syncthreads();
    if (G == 1) { __syncthreads(); __syncthreads(); }               // group B runs one slot behind group A

    // thread index for the rarely executed paths, opaque so that their address arithmetic is not hoisted and kept live
    auto cold_gt = [tid0]() {
        int t = tid0;
        asm volatile("" : "+v"(t));
        return t & 255;
    };
    WRes R;
    {
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int voff = (tid & 63) * 16;
#pragma unroll
        for (int t = 0; t < 4; ++t) wres_load(R, t, wsrc, voff, w_off(0, w, t));      // stage 0: four split-bf16 steps
    }

#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileM;
        const int64_t next_base = base + (int64_t)gridDim.x * kTileM;
        int tid = tid0;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63;
        const int g = lane >> 5, s = lane & 31;
        const int voff = lane * 16;
        float sigma = 0.f;

        // ================= stage 0: the position encoding only (split bf16 straight into f32), both halves
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
            const int row0 = 64 * h;
            const uint4* xP = lds + P_BASE + g * kChunkU4 + row0 + s;
            f32x16 fb[1][2];
            f32x16(&f)[2] = fb[0];
            init_bias8<2>(f, cst, 32 * w, g);
            {
                v4u wh[4] = {R.h[0], R.h[1], R.h[2], R.h[3]}, wl[4] = {R.l[0], R.l[1], R.l[2], R.l[3]};
                r_runbf<2, 4>(f, wh, wl, xP);
            }
            if (h == 1) {                                                            // stage 1's fragments, now that stage 0's are used up
#pragma unroll
                for (int t = 0; t < 8; ++t) wres_load(R, t, wsrc, voff, w_off(1, w, t));
            }
            __syncthreads();
            __syncthreads();
            rowmaxw<1, 2, true>(fb, smaxG, wq, row0, g, s);
            __syncthreads();
            {
                Pending8 pend;
                quant_regs<true>(f, smaxG, row0, s, pend);
                const int chunk0 = G == 0 ? (HA8_BASE / kChunkU4 + 2 * wq) : (HB8_BASE / kChunkU4 + 2 * wq);
                store_regs(pend, lds, scales_of(0) + G * kTileM, chunk0, row0, g, s, wq == 0);
            }
            __syncthreads();
        }
        // ================= stages 1 .. 8: both halves
#pragma unroll 1
        for (int st = 1; st <= 8; ++st) {
            const bool relu = st != 8;
            const int cblk = nm::stage_b_off(st) + 32 * w;                           // (stage_b_off(st) = 256 st for st <= 8)
            const int next_st = st + 1;
            // where the NEXT stage's block of this wave starts (stage 9: block w & 3, only the 8 i8 steps are resident)
            const int next0 = next_st <= 8 ? w_off(next_st, w, 0) : w_off(9, wq, 0);
#pragma unroll 1
            for (int h = 0; h < 2; ++h) {
                const int row0 = 64 * h;
                // input of stage st: group A's features from the parity buffer of stage st - 1, group B's from the single buffer; the
                // k-steps of r_run8 index chunks 2t, 2t+1 of the pointer they are given
                const uint4* xA = lds + HA8_BASE + (((st - 1) & 1) * 8 + g) * kChunkU4 + row0 + s;
                const uint4* xB = lds + HB8_BASE + g * kChunkU4 + row0 + s - 4 * (2 * kChunkU4);     // (k-steps 4..7 -> chunks 0..7 of the B buffer)
                const uint4* xP = lds + P_BASE + g * kChunkU4 + row0 + s;
                // ---------------- M slot
                i32x16 tA[2], tB[2];
                f32x16 fb[1][2];
                f32x16(&f)[2] = fb[0];
                {
                    if (h == 0) r_run8x2<0, 4, false>(tA, R, xA, wsrc, voff, 0);
                    else r_run8x2<0, 4, true>(tA, R, xA, wsrc, voff, next0);
                    __syncthreads();
                    if (h == 0) r_run8x2<4, 4, false>(tB, R, xB, wsrc, voff, 0);
                    else r_run8x2<4, 4, true>(tB, R, xB, wsrc, voff, next0);
#ifndef I8R_EXP_NO_ALPHA
                    if (st == 8 && wq < 2 && G == h)
#else
                    if (false)
#endif
                    {                                                                // alpha block: group A takes H0, group B H1; wave wq its sample block wq
                        const uint4* xaA = xA + 32 * wq;
                        const uint4* xaB = xB + 32 * wq;
                        const float k256 = 256.f * cst[kKap + 8];
                        const float* sin_ = scales_of(7);
                        const int row = row0 + 32 * wq + s;
                        float v = cst[nm::kBiasFloats + nm::stage_b_off(8) + 256];
                        v4u awh[4], awl[4];                                          // the alpha block's fragments: a ring four steps deep
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            awh[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(8, 8, t), 0);
                            awl[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(8, 8, t) + 1024, 0);
                        }
#pragma unroll
                        for (int half = 0; half < 2; ++half) {
                            i32x16 ah[1], ac[1];
                            zero8<1>(ah, ac);
#pragma unroll
                            for (int t = 4 * half; t < 4 * half + 4; ++t) {
                                const v4u wh = awh[t & 3], wl = awl[t & 3];
                                const uint4* xa = half == 0 ? xaA : xaB;
                                const uint4 xh_ = xa[t * (2 * kChunkU4)], xl_ = xa[t * (2 * kChunkU4) + kLoU4];
                                ac[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xl_), ac[0], 0, 0, 0);
                                ac[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wl), __builtin_bit_cast(i32x4, xh_), ac[0], 0, 0, 0);
                                ah[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wh), __builtin_bit_cast(i32x4, xh_), ah[0], 0, 0, 0);
                                if (half == 0) {
                                    awh[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(8, 8, t + 4), 0);
                                    awl[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(8, 8, t + 4) + 1024, 0);
                                }
                            }
                            // row 0 of the block (lanes with g == 0 hold it in register 0): hh * 256 + cross, times this half's row scale
                            v = fmaf((float)((ah[0][0] << 8) + ac[0][0]), sin_[half * kTileM + row] * k256, v);
                        }
                        sigma = v * cst[nm::stage_b_off(8) + 256];
                    }
                    __syncthreads();
                }
                // ---------------- E slot
#ifdef I8R_EXP_NO_E
                {
                    for (int m = 0; m < 2; ++m) for (int r = 0; r < 16; ++r) f[m][r] = (float)(tA[m][r] + tB[m][r]) * 1e-9f;
                }
                if (false)
#endif
                {
                    init_bias8<2>(f, cst, cblk, g);
                    const float k256 = 256.f * cst[kKap + st];
                    const float* sin_ = scales_of(st - 1);
                    const float sA[2] = {sin_[row0 + s] * k256, sin_[row0 + 32 + s] * k256};
                    const float sB[2] = {sin_[kTileM + row0 + s] * k256, sin_[kTileM + row0 + 32 + s] * k256};
                    dequant2<2>(f, tA, tB, sA, sB);
                    if (st == 5) {                                                   // skip layer: the position encoding's four bf16 steps on top
                        v4u wh[4], wl[4];
#pragma unroll
                        for (int t = 0; t < 4; ++t) {
                            wh[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(5, w, 8 + t), 0);
                            wl[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(5, w, 8 + t) + 1024, 0);
                        }
                        r_runbf<2, 4>(f, wh, wl, xP);
                    }
                }
#ifndef I8R_EXP_NO_E
                if (relu) rowmaxw<1, 2, true>(fb, smaxG, wq, row0, g, s);
                else rowmaxw<1, 2, false>(fb, smaxG, wq, row0, g, s);
#endif
                __syncthreads();
                {
                    Pending8 pend;
#ifdef I8R_EXP_NO_E
                    for (int m = 0; m < 2; ++m) { pend.hi[m] = make_uint4(__float_as_uint(f[m][0]), 1, 2, 3); pend.lo[m] = pend.hi[m]; pend.scale[m] = 1.f; }
#else
                    if (relu) quant_regs<true>(f, smaxG, row0, s, pend);
                    else quant_regs<false>(f, smaxG, row0, s, pend);
#endif
                    // group A writes its block into the parity buffer of THIS stage (group B may still be reading the other one);
                    // group B has seen every reader of its previous values pass (its own k-loop of this half was the last)
                    const int chunk0 = G == 0 ? (HA8_BASE / kChunkU4 + (st & 1) * 8 + 2 * wq) : (HB8_BASE / kChunkU4 + 2 * wq);
                    store_regs(pend, lds, scales_of(st) + G * kTileM, chunk0, row0, g, s, wq == 0);
                }
                // the encodings ride in E slots: direction encoding once the skip layer has read the position encoding of the half
                // (both groups: B's read of H0 is one slot after A's), next tile's position encoding after stage 9's read
                if (st == 6 && G == h) fill_pe_any(lds, true, a, base, cold_gt(), 256, row0, 6);
                __syncthreads();
            }
        }

        // ================= stages 9, 10: 128-wide -- group A finishes H0, group B finishes H1
        {
            const int row0 = 64 * G;
            const uint4* xA = lds + HA8_BASE + ((8 & 1) * 8 + g) * kChunkU4 + row0 + s;                 // stage 8's output (parity 0)
            const uint4* xB = lds + HB8_BASE + g * kChunkU4 + row0 + s - 4 * (2 * kChunkU4);
            const uint4* xP = lds + P_BASE + g * kChunkU4 + row0 + s;
            // stage 9's 128 features of this half go to the parity-1 buffer of group A's area (chunks 8..15), whichever group writes them
            const uint4* x9 = lds + HA8_BASE + (8 + g) * kChunkU4 + row0 + s;
            // ---------------- M slot of stage 9 (block wq): 8 resident i8 steps over the 256 features + 2 bf16 steps of the direction encoding
            i32x16 tA[2], tB[2];
            f32x16 fb[1][2];
            f32x16(&f)[2] = fb[0];
            v4u dwh[2], dwl[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                dwh[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(9, wq, 8 + t), 0);
                dwl[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(9, wq, 8 + t) + 1024, 0);
            }
            r_run8x2<0, 4, false>(tA, R, xA, wsrc, voff, 0);
            __syncthreads();
            r_run8x2<4, 4, false>(tB, R, xB, wsrc, voff, 0);
            init_bias8<2>(f, cst, nm::stage_b_off(9) + 32 * wq, g);
            r_runbf<2, 2>(f, dwh, dwl, xP);
            // stage 10's four steps (waves 0, 1 of the group) and the next tile's stage-0 steps into the free resident registers
            v4u rwh[4], rwl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                rwh[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(10, 0, t), 0);
                rwl[t] = __builtin_amdgcn_raw_buffer_load_b128(wsrc, voff, w_off(10, 0, t) + 1024, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) wres_load(R, t, wsrc, voff, w_off(0, w, t));
            __syncthreads();
            // ---------------- E slot of stage 9: one scale per row (the group owns all 128 features of its half)
            {
                const float k256 = 256.f * cst[kKap + 9];
                const float* sin_ = scales_of(8);
                const float sA[2] = {sin_[row0 + s] * k256, sin_[row0 + 32 + s] * k256};
                const float sB[2] = {sin_[kTileM + row0 + s] * k256, sin_[kTileM + row0 + 32 + s] * k256};
                dequant2<2>(f, tA, tB, sA, sB);
                rowmaxw<1, 2, true>(fb, smaxG, wq, row0, g, s);
            }
            __syncthreads();
            {
                Pending8 pend;
                quant_regs<true>(f, smaxG, row0, s, pend);
                store_regs(pend, lds, scales_of(9) + G * kTileM, HA8_BASE / kChunkU4 + 8 + 2 * wq, row0, g, s, wq == 0);
            }
            __syncthreads();
            // ---------------- stage 10 (waves 0, 1 of the group: one 32-sample block each) + the next tile's position encoding of this half
            if (wq < 2) {
                i32x16 ah[1], ac[1], t10[1];
                zero8<1>(ah, ac);
                const uint4* xr = x9 + 32 * wq;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const uint4 xh_ = xr[t * (2 * kChunkU4)], xl_ = xr[t * (2 * kChunkU4) + kLoU4];
                    ac[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, rwh[t]), __builtin_bit_cast(i32x4, xl_), ac[0], 0, 0, 0);
                    ac[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, rwl[t]), __builtin_bit_cast(i32x4, xh_), ac[0], 0, 0, 0);
                    ah[0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, rwh[t]), __builtin_bit_cast(i32x4, xh_), ah[0], 0, 0, 0);
                }
                combine8<1>(t10, ah, ac);
                const float sx = scales_of(9)[G * kTileM + row0 + 32 * wq + s] * (256.f * cst[kKap + 10]);
                const float* b10 = cst + nm::kBiasFloats + nm::stage_b_off(10);
                const float* u10 = cst + nm::stage_b_off(10);
                const int64_t i = base + row0 + 32 * wq + s;
                if (g == 0 && i < a.n)
                    reinterpret_cast<float4*>(a.out)[sample_record(a, i)] =
                        make_float4(fmaf((float)t10[0][0], sx, b10[0]) * u10[0], fmaf((float)t10[0][1], sx, b10[1]) * u10[1],
                                    fmaf((float)t10[0][2], sx, b10[2]) * u10[2], sigma * a.sigma_scale);
            }
            __syncthreads();
            if (next_base < a.n) fill_pe_any(lds, false, a, next_base, cold_gt(), 256, row0, 6);
            __syncthreads();
        }
    }
    if (G == 0) { __syncthreads(); __syncthreads(); }
}
