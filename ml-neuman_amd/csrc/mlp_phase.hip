// Density-only evaluation of the 8x256 NeRF MLP (the sampling pass of a two-pass render: reference utils/render_utils.py:139-141
// composites the coarse network's colours into a frame it then discards; only the compositing weights -- a function of sigma --
// place the importance samples) with the two waves of every SIMD a quarter of a stage OUT OF PHASE.  Same arithmetic, operand
// layout, weight image and instruction sequence per accumulator as nerf_mlp_kernel<NM_PREC_FP16X3> (mlp.hip, helpers shared
// through mlp_device.h): sigma is bit-identical to that kernel's, which tests/test_hip_mlp.py holds bit-identical to the full
// evaluation.  Replaces models/vanilla.py Embedder.forward (:82-92), NeRF.forward (:120-136) and ray_to_samples' points.
//
// Why: in the lock-step kernel all eight waves run k-loop -> convert -> barrier -> store -> barrier together, so every SIMD's
// matrix pipe idles through both epilogue halves and the barrier skew of every stage (measured: 190 ms per 81.92 M evaluations
// against 115-122 ms of pure MFMA time).  Splitting the SAMPLES between two out-of-phase workgroups doubles the weight stream and
// measured slower (profiles/r02_sigma64_experiment.md).  Here the FEATURES are split instead, inside one 128-sample tile:
//
//   group A = waves 0..3 owns output blocks 0..3 (features 0..127, LDS chunks 0..15 of the next layer's input),
//   group B = waves 4..7 owns blocks 4..7 (chunks 16..31); wave q and wave q + 4 share a SIMD.
//
// A stage's k-loop is cut where its input changes owner: half `a` reads the encodings and chunks 0..15 (written by A), half `b`
// chunks 16..31 (written by B).  Per stage four slots, a workgroup barrier after each:
//
//   slot 0   A: k-loop a (needs A's previous outputs)        B: converts and stores its previous outputs (chunks 16..31)
//   slot 1   A: k-loop b (needs B's, stored in slot 0)       B: idle (stages 6, 7: fills the NEXT tile's position encoding)
//   slot 2   A: converts its accumulators (VALU only)        B: k-loop a
//   slot 3   A: stores chunks 0..15 (B.a has read them)      B: k-loop b (accumulators kept in registers until slot 0)
//
// so on every SIMD exactly one wave issues MFMAs at any time, each weight fragment is still fetched once per tile, activations
// are single-buffered in place (the slot order above is what makes that hazard-free), and the alpha block follows as two more
// slots run by group A while B stores and waits.  A wave alone on its SIMD's matrix pipe has no partner to hide its latencies:
// weights come through a 4-step register ring that runs ahead across slots and stages, activation fragments are requested one
// k-step ahead.
#include "mlp_device.h"

namespace {

constexpr int kPRing = 4;                 // k-steps of weights in flight per wave (every run of k-steps is a multiple of this)
constexpr int PREC = NM_PREC_FP16X3;

struct PStep {
    bf16x8 h, l;                          // (fp16 bit patterns; mfma_step<.., NM_PREC_FP16X3> reinterprets them)
};

// The wave's weight stream: runs of k-steps, each contiguous in the lock-step image (one (stage, block) at a time), consumed in the
// order stages 0..7 of its block, then -- group A -- the alpha block of stage 8, then the next tile.  The ring always holds the next
// kPRing steps: the last kPRing steps of a run prefetch the head of the NEXT run, whatever slot or stage it belongs to.
__device__ __forceinline__ int p_stage_off(int st) {            // nm::stage_w_off(st) for a wave-uniform runtime stage
    int o = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) o += (i < st) ? nm::stage_shape(i).nblk * nm::stage_shape(i).steps * nm::kStepBytes : 0;
    return o;
}
__device__ __forceinline__ int p_stage_steps(int st) { return st == 0 ? 4 : (st == 5 ? 20 : 16); }
__device__ __forceinline__ int p_block_off(int st, int blk) { return p_stage_off(st) + (st == 8 ? 8 : blk) * p_stage_steps(st) * nm::kStepBytes; }

// nsteps (a multiple of kPRing) k-steps of one block x MB sample blocks; xh = this lane's pointer to the first step's hi fragment;
// cur = byte offset of the run's first step in the image, next = of the first step consumed after this run
template <int MB>
__device__ __forceinline__ void p_run(f32x16 (&acc)[MB], PStep (&R)[kPRing], __amdgpu_buffer_rsrc_t wsrc, int voff, int cur, int next,
                                      const uint4* xh, int nsteps) {
    uint4 xah[MB], xal[MB], xbh[MB], xbl[MB];
    x_load<MB, PREC>(xah, xal, xh);
#ifndef NM_PHASE_MPRIO
#define NM_PHASE_MPRIO 2
#endif
#ifndef NM_PHASE_EPRIO
#define NM_PHASE_EPRIO 0
#endif
    __builtin_amdgcn_s_setprio(NM_PHASE_MPRIO);
#pragma unroll 1
    for (int t = 0; t < nsteps; t += kPRing) {
        const int pf = t + kPRing < nsteps ? cur + (t + kPRing) * nm::kStepBytes : next;      // wave-uniform
#pragma unroll
        for (int u = 0; u < kPRing; ++u) {
            const int tn = t + u + 1 < nsteps ? t + u + 1 : t + u;          // (last step: a harmless re-read)
            const uint4* nx = xh + tn * (2 * kChunkU4);
            // (the compiler sinks each fragment load towards its MFMA; pinning the order with sched_barriers -- all eight loads of
            //  step t + 1 ahead of the MFMAs of step t -- measured 2 % slower, profiles/r03_sigma_phase_experiment.md)
#ifdef NM_PHASE_NO_X                                                         // probe: activation fragments read once per run
            (void)nx;
            mfma_step<MB, PREC>(acc, R[u].h, R[u].l, xah, xal);
#else
            if (u & 1) {
                x_load<MB, PREC>(xah, xal, nx);
                mfma_step<MB, PREC>(acc, R[u].h, R[u].l, xbh, xbl);
            } else {
                x_load<MB, PREC>(xbh, xbl, nx);
                mfma_step<MB, PREC>(acc, R[u].h, R[u].l, xah, xal);
            }
#endif
#ifndef NM_PHASE_NO_W                                                        // probe: the ring is never refilled
            R[u].h = ld_w(wsrc, voff, pf + u * nm::kStepBytes);              // the slot just consumed <- the step kPRing ahead
            R[u].l = ld_w(wsrc, voff, pf + u * nm::kStepBytes + 1024);
#endif
        }
    }
    __builtin_amdgcn_s_setprio(NM_PHASE_EPRIO);
}

// ReLU + hi / lo split of the accumulators into split-fp16 activations with 7 VALU instructions per pair of values instead of the 10
// of split8<true, true> (mlp_device.h) -- same values bit for bit: the power-of-two scale commutes with the clamp, v_fma_mixlo/hi_f16
// round a * 2^-k to fp16 once (= v_mul_f32, exact, then v_cvt_pk_f16_f32), and v_fma_mix_f32 forms a * 2^-k - hi exactly, reading hi
// as the fp16 it is (no conversion back).  Fewer instructions matter here beyond their own issue time: every VALU instruction of the
// converting wave costs the wave that is inside its k-loop on the same SIMD ~4 cycles of matrix issue (measured: 320 instructions per
// stage and wave -> k-loops at 2/3 speed).
__device__ __forceinline__ void split_pair_f16mix(float a0, float a1, float scale, float hi_max, unsigned& hi, unsigned& lo) {
    a0 = __builtin_amdgcn_fmed3f(a0, 0.f, hi_max);
    a1 = __builtin_amdgcn_fmed3f(a1, 0.f, hi_max);
    unsigned h;
    float r0, r1;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(a0), "s"(scale));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(a1), "s"(scale));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(a0), "s"(scale), "v"(h));
    asm("v_fma_mix_f32 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(a1), "s"(scale), "v"(h));
    hi = h;
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){r0, r1}, f16x2));
}
__device__ __forceinline__ void convert_act_f16mix(const f32x16 (&acc)[4], ActRegs<4>& r, float scale) {
    const float hi_max = 65504.f / scale;                             // (scale = 2^-k: exact)
#pragma unroll
    for (int mb = 0; mb < 4; ++mb)
#pragma unroll
        for (int qp = 0; qp < 2; ++qp) {
            unsigned h[4], l[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) split_pair_f16mix(acc[mb][8 * qp + 2 * p], acc[mb][8 * qp + 2 * p + 1], scale, hi_max, h[p], l[p]);
            r.hi[mb][qp] = make_uint4(h[0], h[1], h[2], h[3]);
            r.lo[mb][qp] = make_uint4(l[0], l[1], l[2], l[3]);
        }
}

// ---- probe builds (tools/build_variant.py --src mlp_phase.hip -D...; never part of the product library) -----------------------
#ifdef NM_PHASE_PROF      // per-wave cycle buckets instead of sigma, see tools/phase_profile.py
#define NM_TICK(b) { const unsigned long long t_now = __builtin_readcyclecounter(); pr[b] += t_now - t_prev; t_prev = t_now; }
#else
#define NM_TICK(b)
#endif
#define NM_SLOT(bwork, bwait) { NM_TICK(bwork) __syncthreads(); NM_TICK(bwait) }
#ifdef NM_PHASE_NO_E      // no conversion, no stores -- the accumulators are kept alive so that the k-loops stay (results garbage)
#define NM_WRITE_ACT(ar)
#define NM_CONVERT(acc, ar, sc) { for (int mb_ = 0; mb_ < 4; ++mb_) asm volatile("" :: "v"(acc[mb_])); (void)ar; }
#else
#define NM_WRITE_ACT(ar) write_act<4, PREC>(ar, lds, w, 0, g, s);
#define NM_CONVERT(acc, ar, sc) convert_act_f16mix(acc, ar, sc);
#endif

__global__ __launch_bounds__(kThreads, 2) void nerf_sigma_phase_kernel(const MlpArgs a_in) {
    const MlpArgs a = resolve_args(a_in);
#ifdef NM_PHASE_PROF
    unsigned long long pr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long t_prev = __builtin_readcyclecounter();
#endif
    __shared__ uint4 lds[LDS_U4];
    const float* f16tab = a.bias + nm::kBiasFloats;                   // per-stage 2^-k (-> activations) and 2^-(k+5) (-> outputs), mlp.hip
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int G = w >> 2, q = w & 3;                                  // group (feature half) and SIMD
    const int g = lane >> 5, s = lane & 31;
    const bool alpha = G == 0;                                        // group A also runs the alpha block
    const __amdgpu_buffer_rsrc_t wsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint4*>(a.wpack), 0, (int)(nm::kWeightBytes + nm::kWeightPadBytes), 0x00020000);
    const int voff = lane * 16;
    const int64_t ntiles = (a.n + kTileM - 1) / kTileM;
    const uint4* xP = lds + P_BASE + g * kChunkU4 + s;
    const uint4* xHa = lds + H_BASE + g * kChunkU4 + s;               // chunks 0..15: group A's outputs
    const uint4* xHb = xHa + 16 * kChunkU4;                           // chunks 16..31: group B's

    // pad slots of the encodings (63) are never written by the octave path: give them a finite value once; first tile's encoding
    for (int i = tid; i < nm::kPeChunks * kChunkU4; i += kThreads) lds[P_BASE + i] = make_uint4(0, 0, 0, 0);
    __syncthreads();
    if ((int64_t)blockIdx.x < ntiles) fill_pe_any<true>(lds, false, a, (int64_t)blockIdx.x * kTileM, tid);
    PStep R[kPRing];
    const int wo0 = p_block_off(0, w);
#pragma unroll
    for (int i = 0; i < kPRing; ++i) {
        R[i].h = ld_w(wsrc, voff, wo0 + i * nm::kStepBytes);
        R[i].l = ld_w(wsrc, voff, wo0 + i * nm::kStepBytes + 1024);
    }
    BiasRegs B;
    bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
    __syncthreads();

#pragma unroll 1
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int64_t base = tile * kTileM;
        const int64_t next_base = (tile + gridDim.x) * kTileM;
        const bool has_next = tile + gridDim.x < ntiles;
        f32x16 acc[4];
#pragma unroll 1
        for (int st = 0; st <= 7; ++st) {
            const int pe = (st == 0 || st == 5) ? 4 : 0, hs = st == 0 ? 0 : 8;
            constexpr int kS = nm::kStepBytes;
            const int wo = p_block_off(st, w);                                            // this stage's run: [pe | a-half | b-half]
            const int wn = st < 7 ? p_block_off(st + 1, w) : (alpha ? p_block_off(8, 8) : wo0);   // what follows it
            if (G == 0) {
                init_bias<4>(acc, B);
                if (pe) p_run<4>(acc, R, wsrc, voff, wo, hs ? wo + pe * kS : wn, xP, pe);
                if (hs) p_run<4>(acc, R, wsrc, voff, wo + pe * kS, wo + (pe + hs) * kS, xHa, hs);
                NM_SLOT(0, 1)                                         // ---- slot 0
                if (hs) p_run<4>(acc, R, wsrc, voff, wo + (pe + hs) * kS, wn, xHb, hs);
                NM_SLOT(0, 1)                                         // ---- slot 1
                ActRegs<4> ar;
                NM_CONVERT(acc, ar, f16tab[st])
                NM_SLOT(2, 3)                                         // ---- slot 2
                NM_WRITE_ACT(ar)
                bias_prefetch(B, a.bias + (st < 7 ? nm::stage_b_off(st + 1) + 32 * w : nm::stage_b_off(8) + 32 * 8), g);   // (not across a k-loop: registers)
                NM_SLOT(4, 5)                                         // ---- slot 3
            } else {
                if (st > 0) {                                         // stage st - 1's outputs, under A's k-loop
                    ActRegs<4> ar;
                    NM_CONVERT(acc, ar, f16tab[st - 1])
                    NM_WRITE_ACT(ar)
                }
                NM_SLOT(2, 3)                                         // ---- slot 0
                if (has_next && st >= 6) {
                    int t_ = tid;                                     // opaque: the fill's per-lane addresses are not hoisted out of the
                    asm volatile("" : "+v"(t_));                      // stage / tile loops (they would be carried through every k-loop)
                    fill_pe_any<true>(lds, false, a, next_base, t_ & 255, 256, 64 * (st - 6), 6);
                }
                bias_prefetch(B, a.bias + nm::stage_b_off(st) + 32 * w, g);
                NM_SLOT(6, 5)                                         // ---- slot 1
                init_bias<4>(acc, B);
                if (pe) p_run<4>(acc, R, wsrc, voff, wo, hs ? wo + pe * kS : wn, xP, pe);
                if (hs) p_run<4>(acc, R, wsrc, voff, wo + pe * kS, wo + (pe + hs) * kS, xHa, hs);
                NM_SLOT(0, 1)                                         // ---- slot 2
                if (hs) p_run<4>(acc, R, wsrc, voff, wo + (pe + hs) * kS, wn, xHb, hs);
                NM_SLOT(0, 1)                                         // ---- slot 3
            }
        }
        // ---- the alpha block of stage 8 (vanilla.py:135): group A, wave q takes sample rows 32 q .. 32 q + 31; B stores stage 7
        if (G == 0) {
            f32x16 aacc[1];
            init_bias<1>(aacc, B);
            const int wa = p_block_off(8, 8);
            p_run<1>(aacc, R, wsrc, voff, wa, wa + 8 * nm::kStepBytes, xHa + 32 * q, 8);
            NM_SLOT(7, 1)                                             // ---- slot 32
            p_run<1>(aacc, R, wsrc, voff, wa + 8 * nm::kStepBytes, wo0, xHb + 32 * q, 8);
            const int64_t i = base + 32 * q + s;
#ifndef NM_PHASE_PROF
            if (g == 0 && i < a.n)
                reinterpret_cast<float4*>(a.out)[sample_record(a, i)] =
                    make_float4(0.f, 0.f, 0.f, aacc[0][0] * f16tab[nm::kStages + 8] * a.sigma_scale);
#else
            if (i < 0) a.out[0] = aacc[0][0];
#endif
            bias_prefetch(B, a.bias + nm::stage_b_off(0) + 32 * w, g);
            NM_SLOT(7, 1)                                             // ---- slot 33
        } else {
            ActRegs<4> ar;
            NM_CONVERT(acc, ar, f16tab[7])
            NM_WRITE_ACT(ar)
            NM_SLOT(2, 3)                                             // ---- slot 32
            NM_SLOT(6, 5)                                             // ---- slot 33
        }
    }
#ifdef NM_PHASE_PROF
    // buckets: 0 k-loops, 1 barrier after a k-loop, 2 convert (+ store for B), 3 barrier after it, 4 A's store, 5 barrier after a store / idle slot,
    // 6 idle slot work (next tile's encoding), 7 alpha block
    if (lane == 0)
        for (int b = 0; b < 8; ++b) a.out[((int64_t)blockIdx.x * 8 + w) * 8 + b] = (float)pr[b];
#endif
}

}  // namespace

namespace nm {

// The density-only launch of the split-fp16 arithmetic (precision NM_PREC_FP16X3, sigma_only == 1, no debug stop, the usual head):
// same arguments as launch_mlp_mfma, which routes such launches here.
int launch_mlp_sigma_phase(const MlpLaunch& L, const float* pts, const float* dirs, const float* origin, const float* direction,
                           const float* z, int64_t n, int S, int in_mode, float sigma_scale, float* out, hipStream_t stream,
                           const MlpChunk* chunk) {
    MlpArgs a;
    a.ray_idx = chunk ? chunk->ray_idx : nullptr;
    a.n_rays_dev = chunk ? chunk->n_rays_dev : nullptr;
    a.s0 = chunk ? chunk->s0 : 0;
    a.S_total = chunk ? chunk->S_total : S;
    a.wpack = reinterpret_cast<const uint4*>(L.wpack16);
    a.bias = L.bias16;
    a.petab = L.petab;
    a.pts = pts; a.dirs = dirs; a.origin = origin; a.direction = direction; a.z = z;
    a.out = out; a.dbg = nullptr; a.prof = nullptr; a.n = n; a.S = S; a.in_mode = in_mode; a.stop_stage = -2; a.sigma_scale = sigma_scale;
    a.sigma_only = 1;
    a.save_h = nullptr; a.save_hv = nullptr;
    a.pos = PeSpec{L.pe_kind, L.pos_nfreq, L.pos_octaves};
    a.dir = PeSpec{L.pe_kind, L.dir_nfreq, L.dir_octaves};
    const int64_t ntiles = (n + kTileM - 1) / kTileM;
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) cus = v;
    }
    const int grid = (int)(ntiles < cus ? ntiles : cus);
    hipLaunchKernelGGL(nerf_sigma_phase_kernel, dim3(grid), dim3(kThreads), 0, stream, a);
    return check_launch("nerf_sigma_phase_kernel");
}

}  // namespace nm
