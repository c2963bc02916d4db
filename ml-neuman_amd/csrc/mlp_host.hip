// Host side of the MLP entry points: weight packing (split bf16, MFMA fragment order), the opaque
// handle, and the dispatch of nm_mlp_forward* onto the MFMA kernel (mlp.hip) or the exact-f32
// validation kernel (mlp_ref.hip).  Also the library-level basics (version, errors, device count).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "common.h"
#include "mlp_layout.h"
#include "mlp_launch.h"

namespace nm {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int check_hip(hipError_t e, const char* what) {
    if (e == hipSuccess) return NM_OK;
    set_error("%s: %s", what, hipGetErrorString(e));
    return NM_ERR_HIP;
}
int check_launch(const char* what) { return check_hip(hipGetLastError(), what); }

// ---- bf16 helpers (round to nearest even) -------------------------------------------------------
static inline uint16_t f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static inline float bf16_to_f32(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// ---- fp16 helpers (round to nearest even, subnormals kept, overflow saturates to the largest finite value) -----------------
static inline uint16_t f32_to_f16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u >= 0x7F800000u) return (uint16_t)(sign | 0x7BFFu);            // inf / nan never occur in a weight image: saturate
    if (u >= 0x477FF000u) return (uint16_t)(sign | 0x7BFFu);            // >= 65520 rounds past the largest fp16: saturate
    if (u < 0x38800000u) {                                              // below 2^-14: subnormal fp16 (or zero)
        if (u < 0x33000000u) return (uint16_t)sign;                     // < 2^-25: rounds to zero
        const int e = (int)(u >> 23);                                   // biased f32 exponent, 102..112
        uint32_t m = (u & 0x7FFFFFu) | 0x800000u;                       // 24-bit significand
        const int shift = 126 - e;                                      // 14..24: value = m * 2^(e-150); fp16 subnormal unit 2^-24
        const uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
        uint32_t r = q + ((rem > half || (rem == half && (q & 1))) ? 1u : 0u);
        return (uint16_t)(sign | r);
    }
    uint32_t v = u - 0x38000000u;                                       // rebias 127 -> 15
    const uint32_t rem = v & 0x1FFFu;
    v >>= 13;
    if (rem > 0x1000u || (rem == 0x1000u && (v & 1))) ++v;
    return (uint16_t)(sign | v);
}
static inline float f16_to_f32(uint16_t h) {
    const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    const int e = (h >> 10) & 31;
    const uint32_t m = h & 0x3FFu;
    float f;
    if (e == 0) {
        f = (float)m * 5.9604644775390625e-08f;                         // m * 2^-24
    } else {
        const uint32_t u = ((uint32_t)(e + 112) << 23) | (m << 13);
        memcpy(&f, &u, 4);
    }
    uint32_t u;
    memcpy(&u, &f, 4);
    u |= sign;
    memcpy(&f, &u, 4);
    return f;
}

// (indices into host_params: mlp_launch.h P_*)

static int validate_desc(const nm_mlp_desc* d) {
    NM_REQUIRE(d, "nm_mlp: null descriptor");
    NM_REQUIRE(d->depth == 8 && d->width == 256 && d->skip == 4,
               "nm_mlp: only the reference default net (depth 8, width 256, skip 4) is implemented, got %d/%d/%d", d->depth,
               d->width, d->skip);
    NM_REQUIRE(d->pe_kind == NM_PE_POSENC || d->pe_kind == NM_PE_ROTATE, "nm_mlp: bad pe_kind %d", d->pe_kind);
    NM_REQUIRE(d->pos_n_freqs >= 1 && d->pos_n_freqs <= 10, "nm_mlp: pos_n_freqs %d outside 1..10", d->pos_n_freqs);
    NM_REQUIRE(d->dir_n_freqs >= 1 && d->dir_n_freqs <= 4, "nm_mlp: dir_n_freqs %d outside 1..4", d->dir_n_freqs);
    NM_REQUIRE(d->plain_head == 0 || d->plain_head == 1, "nm_mlp: plain_head must be 0 or 1, got %d", d->plain_head);
    return NM_OK;
}

// value of the weight that multiplies k-slot (chunk cc of the stage's chunk sequence, element e) for output feature n
static __host__ __device__ float stage_weight(const nm_mlp_desc* d, const float* const* P, int st, int n, int cc, int e) {
    const int kpe = 3 + 6 * d->pos_n_freqs, kdpe = 3 + 6 * d->dir_n_freqs;
    switch (st) {
        case 0: {
            const int p = 8 * cc + e;
            return p < kpe ? P[P_PTS_W][(int64_t)n * kpe + p] : 0.f;
        }
        case 5: {
            const int K = kpe + 256;
            if (cc < 8) {
                const int p = 8 * cc + e;
                return p < kpe ? P[P_PTS_W + 10][(int64_t)n * K + p] : 0.f;
            }
            return P[P_PTS_W + 10][(int64_t)n * K + kpe + slot_feature(cc - 8, e)];
        }
        case 8:
            if (d->plain_head) return (n >= 256 && n < 260) ? P[P_OUT_W][(int64_t)(n - 256) * 256 + slot_feature(cc, e)] : 0.f;
            if (n < 256) return P[P_FEAT_W][(int64_t)n * 256 + slot_feature(cc, e)];
            return n == 256 ? P[P_ALPHA_W][slot_feature(cc, e)] : 0.f;
        case 9: {
            if (d->plain_head) return 0.f;
            const int K = 256 + kdpe;
            if (cc < 32) return P[P_VIEWS_W][(int64_t)n * K + slot_feature(cc, e)];
            const int p = 8 * (cc - 32) + e;
            return p < kdpe ? P[P_VIEWS_W][(int64_t)n * K + 256 + p] : 0.f;
        }
        case 10:
            if (d->plain_head) return 0.f;
            return n < 3 ? P[P_RGB_W][(int64_t)n * 128 + slot_feature(cc, e)] : 0.f;
        default:
            return P[P_PTS_W + 2 * st][(int64_t)n * 256 + slot_feature(cc, e)];
    }
}

// f16 = false: split bf16 of W (NM_PREC_BF16X3 / NM_PREC_BF16); true (NM_PREC_FP16X3): split fp16 of W * 2^k_s per stage,
// k_s = min(8, floor(log2(32000 / max|W_s|))), biases * 2^(k_s + 5), and behind the bias table the per-stage factors
// [2^-k_s (11)] [2^-(k_s+5) (11)] the kernel undoes the scalings with (csrc/mlp.hip).  Same fragment layout.
constexpr int kF16TabFloats = 2 * kStages + 2;
static inline int64_t image_bytes() { return kWeightBytes + kWeightPadBytes + ((int64_t)kBiasFloats + kF16TabFloats) * 4; }
static void pack_image(const nm_mlp_desc* d, const float* const* P, uint8_t* img, bool f16 = false) {
    memset(img, 0, (size_t)image_bytes());
    float wscale[kStages];
    for (int st = 0; st < kStages; ++st) {
        wscale[st] = 1.f;
        if (!f16) continue;
        const StageShape sh = stage_shape(st);
        float mx = 0.f;
        for (int n = 0; n < sh.nblk * 32; ++n)
            for (int cc = 0; cc < 2 * sh.steps; ++cc)
                for (int e = 0; e < 8; ++e) mx = fmaxf(mx, fabsf(stage_weight(d, P, st, n, cc, e)));
        int k = 8;
        while (k > -40 && mx * ldexpf(1.f, k) > 32000.f) --k;
        wscale[st] = ldexpf(1.f, k);
    }
    for (int st = 0; st < kStages; ++st) {
        const StageShape sh = stage_shape(st);
        for (int nb = 0; nb < sh.nblk; ++nb)
            for (int t = 0; t < sh.steps; ++t) {
                uint16_t* hi = reinterpret_cast<uint16_t*>(img + frag_off(st, nb, t));
                uint16_t* lo = hi + 64 * 8;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float wv = stage_weight(d, P, st, 32 * nb + (lane & 31), 2 * t + (lane >> 5), j);
                        if (f16) {
                            const float ws = wv * wscale[st];
                            const uint16_t h = f32_to_f16(ws);
                            hi[lane * 8 + j] = h;
                            lo[lane * 8 + j] = f32_to_f16(ws - f16_to_f32(h));
                            continue;
                        }
                        const uint16_t h = f32_to_bf16(wv);
                        hi[lane * 8 + j] = h;
                        lo[lane * 8 + j] = f32_to_bf16(wv - bf16_to_f32(h));
                    }
            }
    }
    float* bias = reinterpret_cast<float*>(img + kWeightBytes + kWeightPadBytes);
    for (int st = 0; st < kStages; ++st) {
        float* b = bias + stage_b_off(st);
        if (st <= 7) memcpy(b, P[P_PTS_W + 2 * st + 1], 256 * 4);
        else if (d->plain_head) { if (st == 8) memcpy(b + 256, P[P_OUT_B], 4 * 4); }
        else if (st == 8) { memcpy(b, P[P_FEAT_B], 256 * 4); b[256] = P[P_ALPHA_B][0]; }
        else if (st == 9) memcpy(b, P[P_VIEWS_B], 128 * 4);
        else memcpy(b, P[P_RGB_B], 3 * 4);
    }
    if (f16) {
        for (int st = 0; st < kStages; ++st) {
            const StageShape sh = stage_shape(st);
            for (int i = 0; i < sh.nblk * 32; ++i) bias[stage_b_off(st) + i] *= wscale[st] * 32.f;   // accumulators carry Y * 2^(k_s + 5)
            bias[kBiasFloats + st] = 1.f / wscale[st];
            bias[kBiasFloats + kStages + st] = 1.f / (wscale[st] * 32.f);
        }
    }
}

// ---- the NM_PREC_FP16X3 image rebuilt ON THE DEVICE from device-resident parameters: what pack_image(..., f16 = true) writes, for
// a training loop whose weights change every iteration (a host repack would cost a 2.4 MB download, a CPU pass and an upload per
// network and step).  Same scale rule, same rounding (RNE to fp16 twice), same layout: the forward through a refreshed handle is
// bit-identical to the forward through a handle created from the same values (tests/test_hip_train.py).
// (round 6: kScaleSlices workgroups per stage instead of one -- 45 -> a few microseconds in front of every training forward; the largest magnitude is
//  the same number in whatever order it is found, so the image is bit-identical.  work = [kStages] running maxima as uint bits | [kStages] arrival counters,
//  zeroed by the caller; the last slice of a stage to arrive turns its maximum into the scale.)
constexpr int kScaleSlices = 16;
__global__ __launch_bounds__(1024) void f16_stage_scale_kernel(nm_mlp_desc d, DevParams P, unsigned* __restrict__ work, float* __restrict__ wscale) {
    const int st = blockIdx.x;
    const StageShape sh = stage_shape(st);
    const int per_row = 2 * sh.steps * 8, total = sh.nblk * 32 * per_row;
    float mx = 0.f;
    for (int i = blockIdx.y * 1024 + threadIdx.x; i < total; i += 1024 * kScaleSlices) {
        const int n = i / per_row, r = i - n * per_row;
        mx = fmaxf(mx, fabsf(stage_weight(&d, P.p, st, n, r >> 3, r & 7)));
    }
    __shared__ float part[1024];
    part[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (threadIdx.x < o) part[threadIdx.x] = fmaxf(part[threadIdx.x], part[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(&work[st], __float_as_uint(part[0]));                  // (magnitudes: non-negative floats order like their bit patterns; a NaN weight never raised mx)
        __threadfence();
        if (atomicAdd(&work[kStages + st], 1u) == kScaleSlices - 1) {
            const float all = __uint_as_float(atomicMax(&work[st], 0u));
            int k = 8;
            while (k > -40 && all * ldexpf(1.f, k) > 32000.f) --k;
            wscale[st] = ldexpf(1.f, k);
        }
    }
}
__global__ __launch_bounds__(256) void f16_pack_kernel(nm_mlp_desc d, DevParams P, const float* __restrict__ wscale, uint8_t* __restrict__ img) {
    const int gid = blockIdx.x * 256 + threadIdx.x;                   // (k-step of the image, lane)
    const int step = gid >> 6, lane = gid & 63;
    if (step >= (int)(kWeightBytes / kStepBytes)) return;
    int st = 0, first = 0;
    for (; st < kStages; ++st) {
        const int nsteps = stage_shape(st).nblk * stage_shape(st).steps;
        if (step < first + nsteps) break;
        first += nsteps;
    }
    const int steps = stage_shape(st).steps, nb = (step - first) / steps, t = (step - first) - nb * steps;
    const float sc = wscale[st];
    unsigned short h[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float ws = stage_weight(&d, P.p, st, 32 * nb + (lane & 31), 2 * t + (lane >> 5), j) * sc;
        const _Float16 hb = (_Float16)ws;
        const _Float16 lb = (_Float16)(ws - (float)hb);
        h[j] = __builtin_bit_cast(unsigned short, hb);
        l[j] = __builtin_bit_cast(unsigned short, lb);
    }
    uint4* hi = reinterpret_cast<uint4*>(img + (int64_t)step * kStepBytes) + lane;
    *hi = make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    hi[64] = make_uint4(l[0] | ((unsigned)l[1] << 16), l[2] | ((unsigned)l[3] << 16), l[4] | ((unsigned)l[5] << 16), l[6] | ((unsigned)l[7] << 16));
}
__global__ __launch_bounds__(256) void f16_bias_kernel(DevParams P, const float* __restrict__ wscale, float* __restrict__ bias, int plain_head = 0) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= kBiasFloats + kF16TabFloats) return;
    if (i >= kBiasFloats) {                                           // [2^-k (11)] [2^-(k + 5) (11)] [2 spare]
        const int q = i - kBiasFloats;
        bias[i] = q < kStages ? 1.f / wscale[q] : (q < 2 * kStages ? 1.f / (wscale[q - kStages] * 32.f) : 0.f);
        return;
    }
    int st = 0;
    while (st + 1 < kStages && i >= stage_b_off(st + 1)) ++st;
    const int r = i - stage_b_off(st);
    float v = 0.f;
    if (st <= 7) v = P.p[P_PTS_W + 2 * st + 1][r];
    else if (plain_head) v = (st == 8 && r >= 256 && r < 260) ? P.p[P_OUT_B][r - 256] : 0.f;       // output_linear's four rows sit where the alpha block is
    else if (st == 8) v = r < 256 ? P.p[P_FEAT_B][r] : (r == 256 ? P.p[P_ALPHA_B][0] : 0.f);
    else if (st == 9) v = r < 128 ? P.p[P_VIEWS_B][r] : 0.f;
    else v = r < 3 ? P.p[P_RGB_B][r] : 0.f;
    bias[i] = v * wscale[st] * 32.f;
}

// ---- NM_PREC_I8X3 image: [fragments (i8 limb steps, then bf16 PE steps) | pad | units | biases | kappa] ----------------
// The hidden state of stage L is kept in per-feature units u_L[n] (true value = stored value * u_L[n]) chosen so that the
// epilogue needs no per-feature multiplication: the hidden-part weights of output feature n are folded with the units of
// their inputs, w_eff[n][f] = W[n][f] * u_in[f], quantised to int16 with step q[n] = max_f |w_eff| / 32639 and stored as
// balanced int8 limbs in MFMA A-operand order; then sx * sum_f Wq[n][f] * X[f] = out[n] / q[n] directly.  To keep the units
// O(1) through the layers every stage carries one scalar kappa_L = max_n q[n] (folded into the row scale in the kernel):
// u_L[n] = q[n] / kappa_L.  Biases and the rows of the encoding (split-bf16) parts are divided by u_L[n] (ReLU commutes
// with positive scaling).  Stage 0 has u = 1.  sigma (stage 8 row 256) and rgb (stage 10 rows 0..2) are multiplied by
// their unit at the output.  `units` holds u_L[n] for every stage in the bias table's layout.
constexpr int kKappaFloats = 16;
static inline int64_t image8_bytes() { return kWeightBytes8 + kWeightPadBytes + (2 * (int64_t)kBiasFloats + kKappaFloats) * 4; }

// hidden-part weights of output feature n (the i8 operand): row pointer, first hidden column
static const float* hidden_row(const nm_mlp_desc* d, const float* const* P, int st, int n, int* col0) {
    const int kpe = 3 + 6 * d->pos_n_freqs;
    *col0 = 0;
    switch (st) {
        case 5: *col0 = kpe; return P[P_PTS_W + 10] + (int64_t)n * (kpe + 256);
        case 8:
            if (d->plain_head) return (n >= 256 && n < 260) ? P[P_OUT_W] + (int64_t)(n - 256) * 256 : nullptr;   // output_linear's rows where the alpha row is
            if (n < 256) return P[P_FEAT_W] + (int64_t)n * 256;
            return n == 256 ? P[P_ALPHA_W] : nullptr;
        case 9: if (d->plain_head) return nullptr;
            return P[P_VIEWS_W] + (int64_t)n * (256 + 3 + 6 * d->dir_n_freqs);
        case 10: return (n < 3 && !d->plain_head) ? P[P_RGB_W] + (int64_t)n * 128 : nullptr;
        default: return P[P_PTS_W + 2 * st] + (int64_t)n * 256;
    }
}

static void pack_image8(const nm_mlp_desc* d, const float* const* P, uint8_t* img) {
    memset(img, 0, (size_t)image8_bytes());
    float* units = reinterpret_cast<float*>(img + kWeightBytes8 + kWeightPadBytes);
    float* bias = units + kBiasFloats;
    float* kappa = bias + kBiasFloats;
    std::vector<float> weff(256);
    for (int st = 0; st < kStages; ++st) {
        const StageShape8 sh = stage_shape8(st);
        const int nh = sh.i8steps * 32;                       // hidden input width of the i8 part
        const int nrows = sh.nblk * 32;
        float* u = units + stage_b_off(st);
        float* b = bias + stage_b_off(st);
        if (st <= 7) memcpy(b, P[P_PTS_W + 2 * st + 1], 256 * 4);
        else if (d->plain_head) { if (st == 8) memcpy(b + 256, P[P_OUT_B], 4 * 4); }      // (the plain head: stages 9, 10 and the feature rows stay zero)
        else if (st == 8) { memcpy(b, P[P_FEAT_B], 256 * 4); b[256] = P[P_ALPHA_B][0]; }
        else if (st == 9) memcpy(b, P[P_VIEWS_B], 128 * 4);
        else memcpy(b, P[P_RGB_B], 3 * 4);
        kappa[st] = 1.f;
        for (int n = 0; n < nrows; ++n) u[n] = 1.f;
        if (sh.i8steps) {
            // units of this stage's hidden input: the previous stage's (stage 9 reads the 256 feature rows of stage 8)
            const float* uin = units + stage_b_off(st - 1);
            std::vector<float> q(nrows, 0.f);
            float kap = 0.f;
            for (int n = 0; n < nrows; ++n) {
                int col0 = 0;
                const float* row = hidden_row(d, P, st, n, &col0);
                float mx = 0.f;
                if (row) for (int f = 0; f < nh; ++f) mx = fmaxf(mx, fabsf(row[col0 + f] * uin[f]));
                q[n] = mx / (float)kFixedMax;
                kap = fmaxf(kap, q[n]);
            }
            if (!(kap > 0.f)) kap = 1.f;
            kappa[st] = kap;
            for (int n = 0; n < nrows; ++n) {
                if (!(q[n] > 0.f)) q[n] = kap;                // all-zero (or padding) row: any unit works
                u[n] = q[n] / kap;
            }
            for (int nb = 0; nb < sh.nblk; ++nb)
                for (int r = 0; r < 32; ++r) {
                    const int n = 32 * nb + r;
                    int col0 = 0;
                    const float* row = hidden_row(d, P, st, n, &col0);
                    for (int t = 0; t < sh.i8steps; ++t) {
                        int8_t* hi = reinterpret_cast<int8_t*>(img + frag_off8(st, nb, t));
                        int8_t* lo = hi + 1024;
                        for (int g = 0; g < 2; ++g)
                            for (int e = 0; e < 16; ++e) {
                                const int f = slot_feature8(2 * t + g, e);
                                const float w = row ? row[col0 + f] * uin[f] : 0.f;
                                int v = (int)lrintf(w / q[n]);
                                if (v > kFixedMax) v = kFixedMax;
                                if (v < -kFixedMax) v = -kFixedMax;
                                const int l = ((v + 128) & 255) - 128;
                                const int h = (v - l) >> 8;
                                hi[(g * 32 + r) * 16 + e] = (int8_t)h;
                                lo[(g * 32 + r) * 16 + e] = (int8_t)l;
                            }
                    }
                }
        }
        for (int n = 0; n < nrows; ++n) b[n] /= u[n];
        // ---- split-bf16 encoding steps: the bf16 image's values for these k-slots, in this stage's output units
        for (int nb = 0; nb < sh.nblk; ++nb)
            for (int t = 0; t < sh.bfsteps; ++t) {
                uint16_t* hi = reinterpret_cast<uint16_t*>(img + frag_off8(st, nb, sh.i8steps + t));
                uint16_t* lo = hi + 64 * 8;
                // chunk index of PE step t inside the bf16 stage's chunk sequence: stage 0/5 put the PE first, stage 9 last
                const int cc0 = st == 9 ? 32 : 0;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = 32 * nb + (lane & 31);
                        const float wv = stage_weight(d, P, st, n, cc0 + 2 * t + (lane >> 5), j) / u[n];
                        const uint16_t h = f32_to_bf16(wv);
                        hi[lane * 8 + j] = h;
                        lo[lane * 8 + j] = f32_to_bf16(wv - bf16_to_f32(h));
                    }
            }
    }
}

// workgroup stream image of the activation-stationary kernel (mlp_i8s.hip): the k-steps of the block image in the order that kernel
// consumes them.  As the block image, except: stage 5's four encoding steps per block come AFTER its eight i8 blocks, as four units of
// two blocks (so that every hidden stage is eight identical i8 blocks -- they are added on top of the dequantised sums, as everywhere);
// the alpha block of stage 8 comes BEFORE the eight feature blocks (all of the stage's outputs are held in registers until the row
// maximum is known; the alpha block must not be the one that is multiplied while they are all live).  Same size as the block image.
// plain (the use_viewdirs=False net): the tile ends with the alpha block, which holds output_linear's four rows; the kernel's ring then looks two
// blocks ahead as if the feature blocks followed, so the next tile's first two blocks stand there, padded to the feature blocks' 8 steps.
static void pack_stream8s(const uint8_t* img8, uint8_t* out, bool plain = false) {
    memset(out, 0, (size_t)(kWeightBytes8 + kWeightPadBytes));
    uint8_t* dst = out;
    auto put = [&](int st, int nb, int t0, int n) {
        for (int t = 0; t < n; ++t, dst += kStepBytes) memcpy(dst, img8 + frag_off8(st, nb, t0 + t), (size_t)kStepBytes);
    };
    for (int nb = 0; nb < 8; ++nb) put(0, nb, 0, 4);
    for (int st = 1; st <= 7; ++st) {
        for (int nb = 0; nb < 8; ++nb) put(st, nb, 0, 8);
        if (st == 5)
            for (int nb = 0; nb < 8; ++nb) put(5, nb, 8, 4);
    }
    put(8, 8, 0, 8);
    if (plain) {
        for (int nb = 0; nb < 2; ++nb) { put(0, nb, 0, 4); dst += 4 * kStepBytes; }
        if (dst - out != kPlainStreamBytes8) abort();
        return;
    }
    for (int nb = 0; nb < 8; ++nb) put(8, nb, 0, 8);
    for (int nb = 0; nb < 4; ++nb) put(9, nb, 0, 10);
    put(10, 0, 0, 4);
    if (dst - out != kWeightBytes8) abort();
}

// ---- the density-only fp16x3 activation-stationary kernel's stream (mlp_f16t.hip): the k-steps of stages 0..7 and of the alpha block of the
// fp16 image (frag_off) in the order the kernel consumes them -- output blocks in PAIRS (two accumulator chains), a ring unit = 8 steps of
// block b followed by the same 8 steps of block b + 1 (stage 0: its 4 steps; stage 5: its 4 encoding steps first, as a unit of their own).
// Built on the device from the image (so that nm_mlp_refresh_f16 can rebuild it): step i of the stream <- image byte offset tab[i].
constexpr int kSigmaSteps = 8 * 4 + 6 * 8 * 16 + 8 * 20 + 16;            // 976
// 1 KB pieces (a k-step's hi or lo half of one output block) in the order nerf_sigma_f16t_kernel consumes them: first the parts that go through
// its LDS ring -- per ring unit (a pair of output blocks x 4 or 8 k-steps) and k-step: block A hi, A lo, block B hi, B lo minus the last `ndir`
// of them; the alpha row's 16 k-steps as two units of (hi, lo) -- then, for every pair k-step in the same order, the `ndir` parts that are loaded
// straight into registers (tools/gen_f16t.py NDIR)
static void sigma_stream_table(std::vector<int>& tab, int ndir) {
    tab.clear();
    std::vector<int> direct;
    auto put = [&](int st, int b, int t0, int n) {
        for (int t = 0; t < n; ++t)
            for (int k = 0; k < 4; ++k) {
                const int off = (int)frag_off(st, b + (k >> 1), t0 + t) + 1024 * (k & 1);
                (k < 4 - ndir ? tab : direct).push_back(off);
            }
    };
    for (int st = 0; st <= 7; ++st) {
        const StageShape sh = stage_shape(st);
        for (int b = 0; b < 8; b += 2) {
            if (sh.pe_steps) put(st, b, 0, sh.pe_steps);
            for (int t0 = sh.pe_steps; t0 < sh.steps; t0 += 8) put(st, b, t0, 8);
        }
    }
    for (int t = 0; t < 16; ++t) { tab.push_back((int)frag_off(8, 8, t)); tab.push_back((int)frag_off(8, 8, t) + 1024); }
    tab.insert(tab.end(), direct.begin(), direct.end());
    if ((int)tab.size() != 2 * kSigmaSteps) abort();
}
__global__ void sigma_stream_kernel(const uint4* __restrict__ image, const int* __restrict__ tab, int npieces, uint4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;                       // 16 bytes each
    if (i >= npieces * 64) return;
    out[i] = image[tab[i >> 6] / 16 + (i & 63)];
}

// per-wave stream image (mlp_layout.h wstream_*): the steps of the NM_PREC_I8X3 image in each wave's consumption order
static void pack_stream8(const uint8_t* img8, uint8_t* out) {
    memset(out, 0, (size_t)kWeightBytes8w);
    for (int q = 0; q < 4; ++q) {
        uint8_t* dst = out + wstream_off(q);
        auto put = [&](int st, int nb, int t0, int n) {
            for (int t = 0; t < n; ++t, dst += kStepBytes) memcpy(dst, img8 + frag_off8(st, nb, t0 + t), (size_t)kStepBytes);
        };
        put(0, 2 * q, 0, 4); put(0, 2 * q + 1, 0, 4);
        for (int st = 1; st <= 8; ++st) {
            put(st, 2 * q, 0, 8); put(st, 2 * q + 1, 0, 8);
            if (st == 5) { put(5, 2 * q, 8, 4); put(5, 2 * q + 1, 8, 4); }
        }
        if (q < 2) put(8, 8, 0, 8);
        put(9, q, 0, 8); put(9, q, 8, 2);
        dst += 2 * kStepBytes;
        if (q < 2) put(10, 0, 0, 4);
        dst += kW8Pad * kStepBytes;
        if (dst != out + wstream_off(q) + (int64_t)wstream_steps(q) * kStepBytes) abort();
    }
}

}  // namespace nm

struct nm_mlp_s {
    nm_mlp_desc desc;
    uint8_t* d_image;      // weight fragments | pad | bias
    uint8_t* d_image16;    // NM_PREC_FP16X3: the same layout, split fp16 of W * 2^8 | pad | bias * 2^13
    float* d_consts8;      // NM_PREC_I8X3: units | biases | kappa (the tail of the nm_mlp_pack_i8 image)
    uint8_t* d_stream8;    // NM_PREC_I8X3: the image's fragments as per-wave streams (nerf_mlp_i8w_kernel)
    uint8_t* d_image8;     // NM_PREC_I8X3: the workgroup stream of nerf_mlp_i8s_kernel (pack_stream8s), fragments + prefetch pad
    uint8_t* d_stream16t;  // NM_PREC_FP16X3, density only: the stream of nerf_sigma_f16t_kernel (sigma_stream_kernel over d_image16)
    int* d_sigma_tab;      //   its piece table
    uint8_t* d_bwd_image;  // the transposed hidden weights of the backward-data chain (mlp_bwd.hip), repacked from live parameters per call
    int sigma_ndir;        //   parts of a pair k-step that bypass the LDS ring: the value the kernel body was generated for (mlp_f16t.hip sigma_f16t_ndir)
    float* d_petab;        // 192 floats
    float* d_ref;          // transposed f32 weights | natural biases (NM_PREC_FP32 path)
    float* d_wscale16;     // nm_mlp_refresh_f16: the per-stage weight scales of the fp16 image (device scratch)
    int ref_off[12], ref_boff[12];
    int pos_octaves, dir_octaves;
};

extern "C" {

int nm_version(void) { return NM_ABI_VERSION; }
const char* nm_last_error(void) { return nm::g_err; }
int nm_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int64_t nm_mlp_pack_bytes(const nm_mlp_desc* desc) {
    if (nm::validate_desc(desc) != NM_OK) return -1;
    return nm::image_bytes();
}

int nm_mlp_pack(const nm_mlp_desc* desc, const float* const* host_params, void* host_out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_out, "nm_mlp_pack: null pointer");
    for (int i = 0; i < (desc->plain_head ? 18 : 24); ++i) NM_REQUIRE(host_params[i], "nm_mlp_pack: host_params[%d] is null", i);
    nm::pack_image(desc, host_params, static_cast<uint8_t*>(host_out));
    return NM_OK;
}

int nm_mlp_pack_f16(const nm_mlp_desc* desc, const float* const* host_params, void* host_out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_out, "nm_mlp_pack_f16: null pointer");
    for (int i = 0; i < (desc->plain_head ? 18 : 24); ++i) NM_REQUIRE(host_params[i], "nm_mlp_pack_f16: host_params[%d] is null", i);
    nm::pack_image(desc, host_params, static_cast<uint8_t*>(host_out), true);
    return NM_OK;
}

int64_t nm_mlp_pack_i8_bytes(const nm_mlp_desc* desc) {
    if (nm::validate_desc(desc) != NM_OK) return -1;
    return nm::image8_bytes();
}

int nm_mlp_pack_i8(const nm_mlp_desc* desc, const float* const* host_params, void* host_out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_out, "nm_mlp_pack_i8: null pointer");
    for (int i = 0; i < (desc->plain_head ? 18 : 24); ++i) NM_REQUIRE(host_params[i], "nm_mlp_pack_i8: host_params[%d] is null", i);
    nm::pack_image8(desc, host_params, static_cast<uint8_t*>(host_out));
    return NM_OK;
}

int64_t nm_mlp_pack_i8s_bytes(const nm_mlp_desc* desc) {
    if (nm::validate_desc(desc) != NM_OK) return -1;
    return nm::kWeightBytes8 + nm::kWeightPadBytes;
}

int nm_mlp_pack_i8s(const nm_mlp_desc* desc, const float* const* host_params, void* host_out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_out, "nm_mlp_pack_i8s: null pointer");
    for (int i = 0; i < (desc->plain_head ? 18 : 24); ++i) NM_REQUIRE(host_params[i], "nm_mlp_pack_i8s: host_params[%d] is null", i);
    std::vector<uint8_t> img8((size_t)nm::image8_bytes());
    nm::pack_image8(desc, host_params, img8.data());
    nm::pack_stream8s(img8.data(), static_cast<uint8_t*>(host_out), desc->plain_head != 0);
    return NM_OK;
}

int nm_mlp_create(const nm_mlp_desc* desc, const float* const* host_params, const float* host_pos_tab,
                  const float* host_dir_tab, nm_mlp_t* out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_pos_tab && host_dir_tab && out, "nm_mlp_create: null pointer");
    const int plain = desc->plain_head;
    for (int i = 0; i < (plain ? 18 : 24); ++i) NM_REQUIRE(host_params[i], "nm_mlp_create: host_params[%d] is null", i);
    const int64_t bytes = nm_mlp_pack_bytes(desc);
    std::vector<uint8_t> img((size_t)bytes);
    nm::pack_image(desc, host_params, img.data());
    std::vector<uint8_t> img16((size_t)bytes);
    nm::pack_image(desc, host_params, img16.data(), true);
    std::vector<uint8_t> img8((size_t)nm::image8_bytes());
    std::vector<uint8_t> str8((size_t)nm::kWeightBytes8w);
    std::vector<uint8_t> str8s((size_t)(nm::kWeightBytes8 + nm::kWeightPadBytes));
    nm::pack_image8(desc, host_params, img8.data());            // (the plain-head net: output_linear's rows in the alpha block, stages 9 / 10 zero;
    nm::pack_stream8(img8.data(), str8.data());                 //  only the activation-stationary kernel has its form)
    nm::pack_stream8s(img8.data(), str8s.data(), plain != 0);

    // reference-layout image for the exact-f32 kernel
    const int kpe = 3 + 6 * desc->pos_n_freqs, kdpe = 3 + 6 * desc->dir_n_freqs;
    // (plain head: layer 10 is output_linear [4,256]; layers 8, 9, 11 are empty)
    const int K[12] = {kpe, 256, 256, 256, 256, kpe + 256, 256, 256, plain ? 0 : 256 + kdpe, plain ? 0 : 256, 256, plain ? 0 : 128};
    const int N[12] = {256, 256, 256, 256, 256, 256, 256, 256, plain ? 0 : 128, plain ? 0 : 256, plain ? 4 : 1, plain ? 0 : 3};
    const int src[12] = {0, 2, 4, 6, 8, 10, 12, 14, 16, 18, plain ? (int)nm::P_OUT_W : 20, 22};
    nm_mlp_s* m = new nm_mlp_s();
    m->desc = *desc;
    int woff = 0;
    for (int l = 0; l < 12; ++l) { m->ref_off[l] = woff; woff += K[l] * N[l]; }
    int boff = woff;
    for (int l = 0; l < 12; ++l) { m->ref_boff[l] = boff; boff += N[l]; }
    std::vector<float> ref((size_t)boff);
    for (int l = 0; l < 12; ++l) {
        if (!K[l]) continue;
        const float* W = host_params[src[l]];
        for (int k = 0; k < K[l]; ++k)
            for (int n = 0; n < N[l]; ++n) ref[m->ref_off[l] + (size_t)k * N[l] + n] = W[(size_t)n * K[l] + k];
        memcpy(&ref[m->ref_boff[l]], host_params[src[l] + 1], (size_t)N[l] * 4);
    }
    float tab[192];
    memset(tab, 0, sizeof(tab));
    const int npos = desc->pe_kind == NM_PE_POSENC ? desc->pos_n_freqs : 9 * desc->pos_n_freqs;
    const int ndir = desc->pe_kind == NM_PE_POSENC ? desc->dir_n_freqs : 9 * desc->dir_n_freqs;
    memcpy(tab, host_pos_tab, (size_t)npos * 4);
    memcpy(tab + 96, host_dir_tab, (size_t)ndir * 4);
    // octave structure: every band table entry is exactly twice the previous band's (true for the reference defaults
    // 2**linspace(0, N-1, N); vanilla.py:46-51, 67-68) -> the kernel may use the double-angle recurrence
    auto octaves = [&](const float* t, int nfreq) {
        const int per = desc->pe_kind == NM_PE_POSENC ? 1 : 9;
        for (int b = 0; b + 1 < nfreq; ++b)
            for (int k = 0; k < per; ++k)
                if (t[(b + 1) * per + k] != 2.f * t[b * per + k]) return 0;
        return 1;
    };
    m->pos_octaves = octaves(tab, desc->pos_n_freqs);
    m->dir_octaves = octaves(tab + 96, desc->dir_n_freqs);

    m->d_image = nullptr; m->d_image16 = nullptr; m->d_consts8 = nullptr; m->d_stream8 = nullptr; m->d_image8 = nullptr; m->d_stream16t = nullptr; m->d_sigma_tab = nullptr; m->d_bwd_image = nullptr; m->d_petab = nullptr; m->d_ref = nullptr; m->d_wscale16 = nullptr;
    int rc = nm::check_hip(hipMalloc(&m->d_image, (size_t)bytes), "nm_mlp_create: hipMalloc(image)");
    const size_t consts_off = (size_t)(nm::kWeightBytes8 + nm::kWeightPadBytes), consts_bytes = img8.size() - consts_off;
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_consts8, consts_bytes), "nm_mlp_create: hipMalloc(consts8)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_consts8, img8.data() + consts_off, consts_bytes, hipMemcpyHostToDevice), "nm_mlp_create: upload consts8");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_stream8, str8.size()), "nm_mlp_create: hipMalloc(stream8)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_stream8, str8.data(), str8.size(), hipMemcpyHostToDevice), "nm_mlp_create: upload stream8");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_image8, consts_off), "nm_mlp_create: hipMalloc(image8)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_image8, str8s.data(), consts_off, hipMemcpyHostToDevice), "nm_mlp_create: upload image8");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_petab, sizeof(tab)), "nm_mlp_create: hipMalloc(petab)");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_ref, ref.size() * 4), "nm_mlp_create: hipMalloc(ref)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_image, img.data(), (size_t)bytes, hipMemcpyHostToDevice), "nm_mlp_create: upload image");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_image16, (size_t)bytes), "nm_mlp_create: hipMalloc(image16)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_image16, img16.data(), (size_t)bytes, hipMemcpyHostToDevice), "nm_mlp_create: upload image16");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_petab, tab, sizeof(tab), hipMemcpyHostToDevice), "nm_mlp_create: upload petab");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_ref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice), "nm_mlp_create: upload ref");
    if (!rc && !plain) {
        std::vector<int> stab;
        m->sigma_ndir = nm::sigma_f16t_ndir();                                          // what the generated kernel body was emitted for
        nm::sigma_stream_table(stab, m->sigma_ndir);
        rc = nm::check_hip(hipMalloc(&m->d_sigma_tab, stab.size() * sizeof(int)), "nm_mlp_create: hipMalloc(sigma table)");
        if (!rc) rc = nm::check_hip(hipMemcpy(m->d_sigma_tab, stab.data(), stab.size() * sizeof(int), hipMemcpyHostToDevice), "nm_mlp_create: upload sigma table");
        if (!rc) rc = nm::check_hip(hipMalloc(&m->d_stream16t, (size_t)nm::kSigmaSteps * nm::kStepBytes), "nm_mlp_create: hipMalloc(stream16t)");
        if (!rc) {
            const int n16 = nm::kSigmaSteps * (nm::kStepBytes / 16);
            hipLaunchKernelGGL(nm::sigma_stream_kernel, dim3((n16 + 255) / 256), dim3(256), 0, 0, reinterpret_cast<const uint4*>(m->d_image16), m->d_sigma_tab,
                               2 * nm::kSigmaSteps, reinterpret_cast<uint4*>(m->d_stream16t));
            rc = nm::check_launch("sigma_stream_kernel");
            if (!rc) rc = nm::check_hip(hipDeviceSynchronize(), "nm_mlp_create: sigma stream");
        }
    }
    if (rc) { nm_mlp_destroy(m); return rc; }
    *out = m;
    return NM_OK;
}

int nm_mlp_refresh_f16(nm_mlp_t m, const float* const* dev_params, nm_stream_t stream) {
    NM_REQUIRE(m && dev_params, "nm_mlp_refresh_f16: null pointer");
    nm::DevParams P;
    const int need = m->desc.plain_head ? 18 : 24;                        // the plain-head net: 16 trunk tensors + output_linear's weight [4][256] and bias [4]
    for (int i = 0; i < 24; ++i) {
        NM_REQUIRE(i >= need || dev_params[i], "nm_mlp_refresh_f16: dev_params[%d] is null", i);
        P.p[i] = i < need ? dev_params[i] : nullptr;
    }
    if (!m->d_wscale16)
        if (int rc = nm::check_hip(hipMalloc(&m->d_wscale16, 3 * nm::kStages * sizeof(float)), "nm_mlp_refresh_f16: hipMalloc")) return rc;     // scales | maxima | counters
    hipStream_t st = nm::as_stream(stream);
    float* bias = reinterpret_cast<float*>(m->d_image16 + nm::kWeightBytes + nm::kWeightPadBytes);
    unsigned* scale_work = reinterpret_cast<unsigned*>(m->d_wscale16 + nm::kStages);
    if (int rc = nm::check_hip(hipMemsetAsync(scale_work, 0, 2 * nm::kStages * sizeof(unsigned), st), "nm_mlp_refresh_f16: clear")) return rc;
    hipLaunchKernelGGL(nm::f16_stage_scale_kernel, dim3(nm::kStages, nm::kScaleSlices), dim3(1024), 0, st, m->desc, P, scale_work, m->d_wscale16);
    const int threads = (int)(nm::kWeightBytes / nm::kStepBytes) * 64;
    hipLaunchKernelGGL(nm::f16_pack_kernel, dim3((threads + 255) / 256), dim3(256), 0, st, m->desc, P, m->d_wscale16, m->d_image16);
    hipLaunchKernelGGL(nm::f16_bias_kernel, dim3((nm::kBiasFloats + nm::kF16TabFloats + 255) / 256), dim3(256), 0, st, P, m->d_wscale16, bias, m->desc.plain_head);
    if (m->d_stream16t) {                                                  // the density-only stream follows the image
        const int n16 = nm::kSigmaSteps * (nm::kStepBytes / 16);
        hipLaunchKernelGGL(nm::sigma_stream_kernel, dim3((n16 + 255) / 256), dim3(256), 0, st, reinterpret_cast<const uint4*>(m->d_image16), m->d_sigma_tab,
                           2 * nm::kSigmaSteps, reinterpret_cast<uint4*>(m->d_stream16t));
    }
    return nm::check_launch("nm_mlp_refresh_f16");
}

int nm_mlp_destroy(nm_mlp_t m) {
    if (!m) return NM_OK;
    if (m->d_image) (void)hipFree(m->d_image);
    if (m->d_image16) (void)hipFree(m->d_image16);
    if (m->d_consts8) (void)hipFree(m->d_consts8);
    if (m->d_stream8) (void)hipFree(m->d_stream8);
    if (m->d_image8) (void)hipFree(m->d_image8);
    if (m->d_stream16t) (void)hipFree(m->d_stream16t);
    if (m->d_sigma_tab) (void)hipFree(m->d_sigma_tab);
    if (m->d_bwd_image) (void)hipFree(m->d_bwd_image);
    if (m->d_petab) (void)hipFree(m->d_petab);
    if (m->d_ref) (void)hipFree(m->d_ref);
    if (m->d_wscale16) (void)hipFree(m->d_wscale16);
    delete m;
    return NM_OK;
}

static int mlp_dispatch(nm_mlp_t m, const float* pts, const float* dirs, const float* origin, const float* direction,
                        const float* z, int64_t n, int S, int in_mode, int precision, int stop_stage, float sigma_scale,
                        float* out, float* dbg, nm_stream_t stream, void* prof = nullptr, int sigma_only = 0,
                        const nm::MlpChunk* chunk = nullptr) {
    NM_REQUIRE(m, "nm_mlp_forward: null handle");
    NM_REQUIRE(n >= 0, "nm_mlp_forward: negative n");
    NM_REQUIRE(precision == NM_PREC_FP32 || precision == NM_PREC_BF16X3 || precision == NM_PREC_BF16 || precision == NM_PREC_I8X3 ||
                   precision == NM_PREC_FP16X3,
               "nm_mlp_forward: bad precision %d", precision);
    if (n == 0) return NM_OK;
    NM_REQUIRE(!(chunk && precision == NM_PREC_FP32), "nm_mlp_forward_ray_chunk: the exact-f32 validation kernel has no chunked form");
    NM_REQUIRE(!(m->desc.plain_head && precision == NM_PREC_I8X3 && (stop_stage != -2 || dbg || prof || sigma_only)),
               "nm_mlp_forward: the plain-head (use_viewdirs=False) net's i8x3 form is the whole-network launch only (no stage-by-stage / density-only form)");
    NM_REQUIRE(!(m->desc.plain_head && stop_stage > 7), "nm_mlp_forward_debug: the plain-head net has stages -1..7 only");
    if (precision == NM_PREC_FP32) {
        nm::RefLaunch L;
        L.plain_head = m->desc.plain_head;
        L.wt = m->d_ref; L.bias = m->d_ref; L.petab = m->d_petab;
        for (int i = 0; i < 12; ++i) { L.off[i] = m->ref_off[i]; L.boff[i] = m->ref_boff[i]; }
        L.pe_kind = m->desc.pe_kind; L.pos_nfreq = m->desc.pos_n_freqs; L.dir_nfreq = m->desc.dir_n_freqs;
        return nm::launch_mlp_ref(L, pts, dirs, origin, direction, z, n, S, in_mode, stop_stage, sigma_scale, out, dbg,
                                  nm::as_stream(stream));
    }
    nm::MlpLaunch L;
    L.wpack = m->d_image;
    L.bias = reinterpret_cast<const float*>(m->d_image + nm::kWeightBytes + nm::kWeightPadBytes);
    L.wpack16 = m->d_image16;
    L.bias16 = reinterpret_cast<const float*>(m->d_image16 + nm::kWeightBytes + nm::kWeightPadBytes);
    L.petab = m->d_petab;
    L.pe_kind = m->desc.pe_kind; L.pos_nfreq = m->desc.pos_n_freqs; L.dir_nfreq = m->desc.dir_n_freqs;
    L.pos_octaves = m->pos_octaves; L.dir_octaves = m->dir_octaves;
    L.plain_head = m->desc.plain_head;
    L.wstream8 = m->d_stream8;
    L.consts8 = m->d_consts8;
    // NM_PREC_I8X3, whole network, all four outputs: the activation-stationary kernel (mlp_i8s.hip) -- bit-identical to the wave-specialised
    // one of mlp.hip, which keeps the stage-by-stage / profiling / density-only forms (and everything under NEUMAN_I8_KERNEL=w or =r)
    static const bool i8_as = [] { const char* e = getenv("NEUMAN_I8_KERNEL"); return !e || !strcmp(e, "as"); }();
    // NM_PREC_FP16X3, density only (the sampling pass of a two-pass render): the activation-stationary kernel of mlp_f16t.hip -- bit-identical
    // sigma; NEUMAN_SIGMA_KERNEL=w keeps nerf_mlp_kernel for it
    const char* sk = getenv("NEUMAN_SIGMA_KERNEL");              // (read per call: the parity tests switch it inside one process)
    const bool sigma_t = !(sk && !strcmp(sk, "w")) && m->d_stream16t;
    if (sigma_t && precision == NM_PREC_FP16X3 && sigma_only == 1 && stop_stage == -2 && !dbg && !prof && !m->desc.plain_head)
        return nm::launch_sigma_f16t(L, m->d_stream16t, m->sigma_ndir, pts, dirs, origin, direction, z, n, S, in_mode, sigma_scale, out, nm::as_stream(stream), chunk, nullptr, -1);
    if ((i8_as || m->desc.plain_head) && precision == NM_PREC_I8X3 && stop_stage == -2 && !dbg && !prof && !sigma_only)
        return nm::launch_mlp_i8s(L, m->d_image8, pts, dirs, origin, direction, z, n, S, in_mode, sigma_scale, out, nm::as_stream(stream), chunk);
    return nm::launch_mlp_mfma(L, pts, dirs, origin, direction, z, n, S, in_mode, precision, stop_stage, sigma_scale, out, dbg,
                               prof, nm::as_stream(stream), sigma_only, chunk);
}

int nm_mlp_forward(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision, float sigma_scale,
                   float* out, nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (pts && dirs && out), "nm_mlp_forward: null pointer");
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_forward: out must be 16-byte aligned");
    return mlp_dispatch(mlp, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, precision, -2, sigma_scale, out, nullptr, stream);
}

int nm_mlp_forward_save(nm_mlp_t m, const float* pts, const float* dirs, int64_t n, float* save_h, float* save_hv, float* out,
                        nm_stream_t stream) {
    return nm_mlp_forward_save_bits(m, pts, dirs, n, save_h, save_hv, nullptr, out, stream);
}

static int forward_save_impl(nm_mlp_t m, const float* pts, const float* dirs, int64_t n, float* save_h, float* save_hv, uint32_t* save_bits,
                             void* save_h16, float* out, nm_stream_t stream, void* save_feat16 = nullptr, uint32_t* save_hvbits = nullptr,
                             void* save_x0h = nullptr, void* save_d0h = nullptr) {
    NM_REQUIRE(m, "nm_mlp_forward_save: null handle");
    NM_REQUIRE(n >= 0, "nm_mlp_forward_save: negative n");
    if (n == 0) return NM_OK;
    const bool plain = m->desc.plain_head != 0;                           // output_linear straight off layer 7: the trunk's copies only, 16-bit form only
    NM_REQUIRE(!plain || (save_h16 && !save_h && !save_feat16 && !save_hvbits && !save_d0h), "nm_mlp_forward_save: the plain-head net keeps save_h16 / save_bits / save_x0h only");
    NM_REQUIRE(pts && (dirs || plain) && (plain || ((save_h || (save_h16 && save_feat16)) && save_hv)) && out, "nm_mlp_forward_save: null pointer");
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(out) | reinterpret_cast<uintptr_t>(save_h) | reinterpret_cast<uintptr_t>(save_hv) | reinterpret_cast<uintptr_t>(save_h16) |
                 reinterpret_cast<uintptr_t>(save_feat16)) & 15) == 0, "nm_mlp_forward_save: outputs must be 16-byte aligned");
    nm::MlpLaunch L;
    L.wpack = m->d_image;
    L.bias = reinterpret_cast<const float*>(m->d_image + nm::kWeightBytes + nm::kWeightPadBytes);
    L.wpack16 = m->d_image16;
    L.bias16 = reinterpret_cast<const float*>(m->d_image16 + nm::kWeightBytes + nm::kWeightPadBytes);
    L.petab = m->d_petab;
    L.pe_kind = m->desc.pe_kind; L.pos_nfreq = m->desc.pos_n_freqs; L.dir_nfreq = m->desc.dir_n_freqs;
    L.pos_octaves = m->pos_octaves; L.dir_octaves = m->dir_octaves;
    L.plain_head = plain ? 1 : 0;
    L.wstream8 = m->d_stream8;
    L.consts8 = m->d_consts8;
    L.save_h = save_h; L.save_hv = save_hv; L.save_bits = save_bits; L.save_h16 = save_h16; L.save_feat16 = save_feat16; L.save_hvbits = save_hvbits; L.save_x0h = save_x0h; L.save_d0h = save_d0h;
    return nm::launch_mlp_mfma(L, pts, dirs ? dirs : pts, nullptr, nullptr, nullptr, n, 1, 0, NM_PREC_FP16X3, -2, 1.f, out, nullptr, nullptr,
                               nm::as_stream(stream), 0, nullptr);
}

int nm_mlp_forward_save_bits(nm_mlp_t m, const float* pts, const float* dirs, int64_t n, float* save_h, float* save_hv, uint32_t* save_bits,
                             float* out, nm_stream_t stream) {
    return forward_save_impl(m, pts, dirs, n, save_h, save_hv, save_bits, nullptr, out, stream);
}

int nm_mlp_forward_save16(nm_mlp_t m, const float* pts, const float* dirs, int64_t n, uint16_t* save_h16, float* save_feat, uint16_t* save_feat16,
                          float* save_hv, uint32_t* save_bits, uint32_t* save_hvbits, uint16_t* save_x0h, uint16_t* save_d0h, float* out, nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (save_h16 && save_bits && (save_feat || save_feat16 || (m && m->desc.plain_head))), "nm_mlp_forward_save16: null pointer");
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(save_x0h) | reinterpret_cast<uintptr_t>(save_d0h)) & 15) == 0, "nm_mlp_forward_save16: outputs must be 16-byte aligned");
    return forward_save_impl(m, pts, dirs, n, save_feat, save_hv, save_bits, save_h16, out, stream, save_feat16, save_hvbits, save_x0h, save_d0h);
}

int64_t nm_mlp_backward_chain_workspace_floats(int64_t n) { return ((n + nm::kTileM - 1) / nm::kTileM) * 9 * 256; }

int nm_mlp_backward_chain(nm_mlp_t m, const float* const* dev_params, const float* dz_top, const float* d_feat, const float* d_raw, const float* acts,
                          const uint32_t* relu_bits, int64_t n, float* dz_out, float* bias_grads, float* workspace, int64_t workspace_floats,
                          nm_stream_t stream) {
    NM_REQUIRE(m && dev_params, "nm_mlp_backward_chain: null pointer");
    NM_REQUIRE(n >= 0, "nm_mlp_backward_chain: negative n");
    if (n == 0) return NM_OK;
    NM_REQUIRE((dz_top || (d_feat && d_raw)) && (acts || relu_bits) && dz_out && bias_grads && workspace, "nm_mlp_backward_chain: null pointer");
    NM_REQUIRE(!d_feat || !m->desc.plain_head, "nm_mlp_backward_chain: the plain-head net has no feature layer (pass dz_top)");
    NM_REQUIRE(workspace_floats >= nm_mlp_backward_chain_workspace_floats(n), "nm_mlp_backward_chain: workspace of %lld floats, %lld needed",
               (long long)workspace_floats, (long long)nm_mlp_backward_chain_workspace_floats(n));
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(dz_top) | reinterpret_cast<uintptr_t>(d_feat) | reinterpret_cast<uintptr_t>(d_raw) | reinterpret_cast<uintptr_t>(acts) |
                 reinterpret_cast<uintptr_t>(dz_out) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0, "nm_mlp_backward_chain: buffers must be 16-byte aligned");
    nm::DevParams P;
    const int need = d_feat ? 24 : 16;
    for (int i = 0; i < 24; ++i) {
        NM_REQUIRE(i >= need || dev_params[i], "nm_mlp_backward_chain: dev_params[%d] is null", i);
        P.p[i] = i < need ? dev_params[i] : nullptr;
    }
    if (!m->d_bwd_image)
        if (int rc = nm::check_hip(hipMalloc(&m->d_bwd_image, (size_t)nm::mlp_bwd_image_bytes()), "nm_mlp_backward_chain: hipMalloc")) return rc;
    return nm::launch_mlp_bwd(P, 3 + 6 * m->desc.pos_n_freqs, m->d_bwd_image, dz_top, d_feat, d_raw, acts, relu_bits, n, dz_out, workspace, bias_grads,
                              nm::as_stream(stream));
}

int nm_mlp_backward_chain16(nm_mlp_t m, const float* const* dev_params, const float* d_feat, const float* d_raw, const uint32_t* relu_bits, int64_t n,
                            const float* amax, uint16_t* dz16, uint16_t* dfeat16, float* dz32_layer5, float* dz32_layer0, float* bias_grads,
                            float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    NM_REQUIRE(m && dev_params, "nm_mlp_backward_chain16: null pointer");
    NM_REQUIRE(n >= 0, "nm_mlp_backward_chain16: negative n");
    if (n == 0) return NM_OK;
    NM_REQUIRE(d_feat && d_raw && relu_bits && amax && dz16 && bias_grads && workspace, "nm_mlp_backward_chain16: null pointer");
    NM_REQUIRE(!m->desc.plain_head, "nm_mlp_backward_chain16: the plain-head net has no feature layer");
    NM_REQUIRE(workspace_floats >= nm_mlp_backward_chain_workspace_floats(n), "nm_mlp_backward_chain16: workspace of %lld floats, %lld needed",
               (long long)workspace_floats, (long long)nm_mlp_backward_chain_workspace_floats(n));
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(d_feat) | reinterpret_cast<uintptr_t>(d_raw) | reinterpret_cast<uintptr_t>(dz16) | reinterpret_cast<uintptr_t>(dfeat16) |
                 reinterpret_cast<uintptr_t>(dz32_layer5) | reinterpret_cast<uintptr_t>(dz32_layer0) | reinterpret_cast<uintptr_t>(workspace)) & 15) == 0,
               "nm_mlp_backward_chain16: buffers must be 16-byte aligned");
    nm::DevParams P;
    for (int i = 0; i < 24; ++i) {
        NM_REQUIRE(dev_params[i], "nm_mlp_backward_chain16: dev_params[%d] is null", i);
        P.p[i] = dev_params[i];
    }
    if (!m->d_bwd_image)
        if (int rc = nm::check_hip(hipMalloc(&m->d_bwd_image, (size_t)nm::mlp_bwd_image_bytes()), "nm_mlp_backward_chain16: hipMalloc")) return rc;
    nm::Bwd16 h;
    h.dz16 = dz16; h.dfeat16 = dfeat16; h.amax = amax;
    for (int i = 0; i < 8; ++i) h.dz32[i] = nullptr;
    h.dz32[5] = dz32_layer5; h.dz32[0] = dz32_layer0;
    return nm::launch_mlp_bwd(P, 3 + 6 * m->desc.pos_n_freqs, m->d_bwd_image, nullptr, d_feat, d_raw, nullptr, relu_bits, n, nullptr, workspace, bias_grads,
                              nm::as_stream(stream), &h);
}

int nm_mlp_backward_net16(nm_mlp_t m, const float* const* dev_params, const float* d_raw, const float* d_feat_add, const uint32_t* relu_bits,
                          const uint32_t* hv_bits, int64_t n, const float* amax, uint16_t* dz16, uint16_t* dfeat16, uint16_t* dhv16, float* dz32_layer5, float* dz32_layer0, float* dhv32,
                          float* bias_grads, float* workspace, int64_t workspace_floats, nm_stream_t stream) {
    NM_REQUIRE(m && dev_params, "nm_mlp_backward_net16: null pointer");
    NM_REQUIRE(n >= 0, "nm_mlp_backward_net16: negative n");
    if (n == 0) return NM_OK;
    NM_REQUIRE(d_raw && relu_bits && hv_bits && amax && dz16 && dfeat16 && dhv16 && bias_grads && workspace, "nm_mlp_backward_net16: null pointer");
    NM_REQUIRE(!m->desc.plain_head, "nm_mlp_backward_net16: the plain-head net has no views layer");
    NM_REQUIRE(workspace_floats >= nm_mlp_backward_chain_workspace_floats(n), "nm_mlp_backward_net16: workspace of %lld floats, %lld needed",
               (long long)workspace_floats, (long long)nm_mlp_backward_chain_workspace_floats(n));
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(d_raw) | reinterpret_cast<uintptr_t>(d_feat_add) | reinterpret_cast<uintptr_t>(dz16) | reinterpret_cast<uintptr_t>(dfeat16) |
                 reinterpret_cast<uintptr_t>(dhv16) | reinterpret_cast<uintptr_t>(dz32_layer5) | reinterpret_cast<uintptr_t>(dz32_layer0) | reinterpret_cast<uintptr_t>(dhv32) |
                 reinterpret_cast<uintptr_t>(workspace)) & 15) == 0, "nm_mlp_backward_net16: buffers must be 16-byte aligned");
    nm::DevParams P;
    for (int i = 0; i < 24; ++i) {
        NM_REQUIRE(dev_params[i], "nm_mlp_backward_net16: dev_params[%d] is null", i);
        P.p[i] = dev_params[i];
    }
    NM_REQUIRE((reinterpret_cast<uintptr_t>(P.p[nm::P_RGB_W]) & 15) == 0, "nm_mlp_backward_net16: rgb_linear.weight must be 16-byte aligned");
    if (!m->d_bwd_image)
        if (int rc = nm::check_hip(hipMalloc(&m->d_bwd_image, (size_t)nm::mlp_bwd_image_bytes()), "nm_mlp_backward_net16: hipMalloc")) return rc;
    nm::Bwd16 h;
    h.dz16 = dz16; h.dfeat16 = dfeat16; h.amax = amax;
    for (int i = 0; i < 8; ++i) h.dz32[i] = nullptr;
    h.dz32[5] = dz32_layer5; h.dz32[0] = dz32_layer0;
    h.hvbits = hv_bits; h.dhv16 = dhv16; h.dhv32 = dhv32; h.kdir = 3 + 6 * m->desc.dir_n_freqs;
    return nm::launch_mlp_bwd(P, 3 + 6 * m->desc.pos_n_freqs, m->d_bwd_image, nullptr, d_feat_add, d_raw, nullptr, relu_bits, n, nullptr, workspace, bias_grads,
                              nm::as_stream(stream), &h);
}

int nm_mlp_backward_plain16(nm_mlp_t m, const float* const* dev_params, const float* d_out, const uint32_t* relu_bits, int64_t n, const float* amax,
                            uint16_t* dz16, float* dz32_layer5, float* dz32_layer0, float* bias_grads, float* workspace, int64_t workspace_floats,
                            nm_stream_t stream) {
    NM_REQUIRE(m && dev_params, "nm_mlp_backward_plain16: null pointer");
    NM_REQUIRE(n >= 0, "nm_mlp_backward_plain16: negative n");
    if (n == 0) return NM_OK;
    NM_REQUIRE(d_out && relu_bits && amax && dz16 && bias_grads && workspace, "nm_mlp_backward_plain16: null pointer");
    NM_REQUIRE(m->desc.plain_head, "nm_mlp_backward_plain16: the net has the view-dependent head (nm_mlp_backward_net16)");
    NM_REQUIRE(workspace_floats >= nm_mlp_backward_chain_workspace_floats(n), "nm_mlp_backward_plain16: workspace of %lld floats, %lld needed",
               (long long)workspace_floats, (long long)nm_mlp_backward_chain_workspace_floats(n));
    NM_REQUIRE(((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(dz16) | reinterpret_cast<uintptr_t>(dz32_layer5) | reinterpret_cast<uintptr_t>(dz32_layer0) |
                 reinterpret_cast<uintptr_t>(workspace)) & 15) == 0, "nm_mlp_backward_plain16: buffers must be 16-byte aligned");
    nm::DevParams P;
    for (int i = 0; i < 24; ++i) {
        NM_REQUIRE(i >= 18 || dev_params[i], "nm_mlp_backward_plain16: dev_params[%d] is null", i);
        P.p[i] = i < 18 ? dev_params[i] : nullptr;
    }
    if (!m->d_bwd_image)
        if (int rc = nm::check_hip(hipMalloc(&m->d_bwd_image, (size_t)nm::mlp_bwd_image_bytes()), "nm_mlp_backward_plain16: hipMalloc")) return rc;
    nm::Bwd16 h;
    h.dz16 = dz16; h.dfeat16 = nullptr; h.amax = amax;
    for (int i = 0; i < 8; ++i) h.dz32[i] = nullptr;
    h.dz32[5] = dz32_layer5; h.dz32[0] = dz32_layer0;
    h.plain = 1;
    return nm::launch_mlp_bwd(P, 3 + 6 * m->desc.pos_n_freqs, m->d_bwd_image, nullptr, nullptr, d_out, nullptr, relu_bits, n, nullptr, workspace, bias_grads,
                              nm::as_stream(stream), &h);
}

int nm_mlp_forward_rays(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int64_t R, int S,
                        int precision, float sigma_scale, float* out, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (origin && direction && z_vals && out), "nm_mlp_forward_rays: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1, "nm_mlp_forward_rays: bad sizes");
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_forward_rays: out must be 16-byte aligned");
    return mlp_dispatch(mlp, nullptr, nullptr, origin, direction, z_vals, R * (int64_t)S, S, 1, precision, -2, sigma_scale, out,
                        nullptr, stream);
}

int nm_mlp_sigma_rays(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int64_t R, int S,
                      int precision, float sigma_scale, float* out, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (origin && direction && z_vals && out), "nm_mlp_sigma_rays: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1, "nm_mlp_sigma_rays: bad sizes");
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_sigma_rays: out must be 16-byte aligned");
    return mlp_dispatch(mlp, nullptr, nullptr, origin, direction, z_vals, R * (int64_t)S, S, 1, precision, -2, sigma_scale, out,
                        nullptr, stream, nullptr, 1);
}

int nm_mlp_forward_ray_chunk(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int S_total,
                             const int32_t* ray_idx, const int32_t* n_rays_dev, int64_t n_rays, int s0, int S, int precision,
                             float sigma_scale, float* out, nm_stream_t stream) {
    NM_REQUIRE(n_rays == 0 || (origin && direction && z_vals && ray_idx && out), "nm_mlp_forward_ray_chunk: null pointer");
    NM_REQUIRE(n_rays >= 0 && S >= 1 && s0 >= 0 && s0 + S <= S_total, "nm_mlp_forward_ray_chunk: bad sizes (s0=%d S=%d S_total=%d)", s0, S, S_total);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_forward_ray_chunk: out must be 16-byte aligned");
    nm::MlpChunk c{ray_idx, n_rays_dev, s0, S_total};
    return mlp_dispatch(mlp, nullptr, nullptr, origin, direction, z_vals, n_rays * (int64_t)S, S, 2, precision, -2, sigma_scale, out,
                        nullptr, stream, nullptr, 0, &c);
}

int nm_mlp_sigma_ray_chunk(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int S_total,
                           const int32_t* ray_idx, const int32_t* n_rays_dev, int64_t n_rays, int s0, int S, int precision,
                           float sigma_scale, float* out, nm_stream_t stream) {
    NM_REQUIRE(n_rays == 0 || (origin && direction && z_vals && ray_idx && out), "nm_mlp_sigma_ray_chunk: null pointer");
    NM_REQUIRE(n_rays >= 0 && S >= 1 && s0 >= 0 && s0 + S <= S_total, "nm_mlp_sigma_ray_chunk: bad sizes (s0=%d S=%d S_total=%d)", s0, S, S_total);
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_sigma_ray_chunk: out must be 16-byte aligned");
    nm::MlpChunk c{ray_idx, n_rays_dev, s0, S_total};
    return mlp_dispatch(mlp, nullptr, nullptr, origin, direction, z_vals, n_rays * (int64_t)S, S, 2, precision, -2, sigma_scale, out,
                        nullptr, stream, nullptr, 1, &c);
}

int nm_mlp_forward_profile(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision, float* out,
                           uint64_t* cycles, nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (pts && dirs && out && cycles), "nm_mlp_forward_profile: null pointer");
    NM_REQUIRE(precision == NM_PREC_BF16X3 || precision == NM_PREC_I8X3 || precision == NM_PREC_FP16X3,
               "nm_mlp_forward_profile: precision %d has no profiling build",
               precision);
    return mlp_dispatch(mlp, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, precision, -2, 1.f, out, nullptr, stream, cycles);
}

int nm_mlp_sigma_f16t_debug(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int stage, float* state, float* out, nm_stream_t stream) {
    NM_REQUIRE(mlp && pts && dirs && state && out && n > 0, "nm_mlp_sigma_f16t_debug: null pointer");
    NM_REQUIRE(stage >= 0 && stage % 100 <= 7 && !mlp->desc.plain_head, "nm_mlp_sigma_f16t_debug: stage %d outside 0..7 (+ 100 x tile round)", stage);
    nm::MlpLaunch L;
    L.petab = mlp->d_petab;
    L.pe_kind = mlp->desc.pe_kind; L.pos_nfreq = mlp->desc.pos_n_freqs; L.dir_nfreq = mlp->desc.dir_n_freqs;
    L.pos_octaves = mlp->pos_octaves; L.dir_octaves = mlp->dir_octaves;
    L.plain_head = 0;
    L.bias16 = reinterpret_cast<const float*>(mlp->d_image16 + nm::kWeightBytes + nm::kWeightPadBytes);
    return nm::launch_sigma_f16t(L, mlp->d_stream16t, mlp->sigma_ndir, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, 1.f, out, nm::as_stream(stream), nullptr, state, stage);
}

int nm_mlp_forward_debug(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision, int stage,
                         float* hidden, nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (pts && dirs && hidden), "nm_mlp_forward_debug: null pointer");
    NM_REQUIRE(stage >= -1 && stage <= 9, "nm_mlp_forward_debug: stage %d outside -1..9", stage);
    return mlp_dispatch(mlp, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, precision, stage, 1.f, nullptr, hidden, stream);
}

}  // extern "C"
