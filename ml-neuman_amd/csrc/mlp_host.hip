// Host side of the MLP entry points: weight packing (split bf16, MFMA fragment order), the opaque
// handle, and the dispatch of nm_mlp_forward* onto the MFMA kernel (mlp.hip) or the exact-f32
// validation kernel (mlp_ref.hip).  Also the library-level basics (version, errors, device count).
#include <stdarg.h>
#include <stdio.h>
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

// indices into host_params (reference state_dict order, include/neuman_hip.h)
enum { P_PTS_W = 0, P_VIEWS_W = 16, P_VIEWS_B = 17, P_FEAT_W = 18, P_FEAT_B = 19, P_ALPHA_W = 20, P_ALPHA_B = 21, P_RGB_W = 22, P_RGB_B = 23 };

static int validate_desc(const nm_mlp_desc* d) {
    NM_REQUIRE(d, "nm_mlp: null descriptor");
    NM_REQUIRE(d->depth == 8 && d->width == 256 && d->skip == 4,
               "nm_mlp: only the reference default net (depth 8, width 256, skip 4) is implemented, got %d/%d/%d", d->depth,
               d->width, d->skip);
    NM_REQUIRE(d->pe_kind == NM_PE_POSENC || d->pe_kind == NM_PE_ROTATE, "nm_mlp: bad pe_kind %d", d->pe_kind);
    NM_REQUIRE(d->pos_n_freqs >= 1 && d->pos_n_freqs <= 10, "nm_mlp: pos_n_freqs %d outside 1..10", d->pos_n_freqs);
    NM_REQUIRE(d->dir_n_freqs >= 1 && d->dir_n_freqs <= 4, "nm_mlp: dir_n_freqs %d outside 1..4", d->dir_n_freqs);
    return NM_OK;
}

// value of the weight that multiplies k-slot (chunk cc of the stage's chunk sequence, element e) for output feature n
static float stage_weight(const nm_mlp_desc* d, const float* const* P, int st, int n, int cc, int e) {
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
            if (n < 256) return P[P_FEAT_W][(int64_t)n * 256 + slot_feature(cc, e)];
            return n == 256 ? P[P_ALPHA_W][slot_feature(cc, e)] : 0.f;
        case 9: {
            const int K = 256 + kdpe;
            if (cc < 32) return P[P_VIEWS_W][(int64_t)n * K + slot_feature(cc, e)];
            const int p = 8 * (cc - 32) + e;
            return p < kdpe ? P[P_VIEWS_W][(int64_t)n * K + 256 + p] : 0.f;
        }
        case 10:
            return n < 3 ? P[P_RGB_W][(int64_t)n * 128 + slot_feature(cc, e)] : 0.f;
        default:
            return P[P_PTS_W + 2 * st][(int64_t)n * 256 + slot_feature(cc, e)];
    }
}

static void pack_image(const nm_mlp_desc* d, const float* const* P, uint8_t* img) {
    memset(img, 0, (size_t)(kWeightBytes + kWeightPadBytes + (int64_t)kBiasFloats * 4));
    for (int st = 0; st < kStages; ++st) {
        const StageShape sh = stage_shape(st);
        for (int nb = 0; nb < sh.nblk; ++nb)
            for (int t = 0; t < sh.steps; ++t) {
                uint16_t* hi = reinterpret_cast<uint16_t*>(img + frag_off(st, nb, t));
                uint16_t* lo = hi + 64 * 8;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float wv = stage_weight(d, P, st, 32 * nb + (lane & 31), 2 * t + (lane >> 5), j);
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
        else if (st == 8) { memcpy(b, P[P_FEAT_B], 256 * 4); b[256] = P[P_ALPHA_B][0]; }
        else if (st == 9) memcpy(b, P[P_VIEWS_B], 128 * 4);
        else memcpy(b, P[P_RGB_B], 3 * 4);
    }
}

// ---- NM_PREC_I8X3 image: [fragments (i8 limb steps, then bf16 PE steps) | pad | weight scales | biases] ---------------
// Per output feature n of a stage the hidden-part weights are quantised to int16 with scale sw[n] = max|w| / 32639 and
// stored as balanced int8 limbs in MFMA A-operand order; PE-part weights keep the split-bf16 fragments of pack_image.
static inline int64_t image8_bytes() { return kWeightBytes8 + kWeightPadBytes + 2 * (int64_t)kBiasFloats * 4; }

// hidden-part weight of output feature n for hidden input feature f (the i8 operand), and the number of hidden features
static const float* hidden_row(const nm_mlp_desc* d, const float* const* P, int st, int n, int* stride, int* col0) {
    const int kpe = 3 + 6 * d->pos_n_freqs;
    switch (st) {
        case 5: *stride = kpe + 256; *col0 = kpe; return P[P_PTS_W + 10] + (int64_t)n * (kpe + 256);
        case 8:
            *stride = 256; *col0 = 0;
            if (n < 256) return P[P_FEAT_W] + (int64_t)n * 256;
            return n == 256 ? P[P_ALPHA_W] : nullptr;
        case 9: { const int K = 256 + 3 + 6 * d->dir_n_freqs; *stride = K; *col0 = 0; return P[P_VIEWS_W] + (int64_t)n * K; }
        case 10: *stride = 128; *col0 = 0; return n < 3 ? P[P_RGB_W] + (int64_t)n * 128 : nullptr;
        default: *stride = 256; *col0 = 0; return P[P_PTS_W + 2 * st] + (int64_t)n * 256;
    }
}

static void pack_image8(const nm_mlp_desc* d, const float* const* P, uint8_t* img) {
    memset(img, 0, (size_t)image8_bytes());
    float* scales = reinterpret_cast<float*>(img + kWeightBytes8 + kWeightPadBytes);
    float* bias = scales + kBiasFloats;
    for (int st = 0; st < kStages; ++st) {
        const StageShape8 sh = stage_shape8(st);
        const int nh = sh.i8steps * 32;                       // hidden input width of the i8 part
        for (int nb = 0; nb < sh.nblk; ++nb) {
            // ---- i8 limb steps
            for (int r = 0; r < 32 && sh.i8steps; ++r) {
                const int n = 32 * nb + r;
                int stride = 0, col0 = 0;
                const float* row = hidden_row(d, P, st, n, &stride, &col0);
                float mx = 0.f;
                if (row) for (int f = 0; f < nh; ++f) mx = fmaxf(mx, fabsf(row[col0 + f]));
                const float sw = mx > 0.f ? mx / (float)kFixedMax : 1.f;
                scales[stage_b_off(st) + n] = sw;
                for (int t = 0; t < sh.i8steps; ++t) {
                    int8_t* hi = reinterpret_cast<int8_t*>(img + frag_off8(st, nb, t));
                    int8_t* lo = hi + 1024;
                    for (int g = 0; g < 2; ++g)
                        for (int e = 0; e < 16; ++e) {
                            const int f = slot_feature8(2 * t + g, e);
                            const float w = row ? row[col0 + f] : 0.f;
                            int q = (int)lrintf(w / sw);
                            if (q > kFixedMax) q = kFixedMax;
                            if (q < -kFixedMax) q = -kFixedMax;
                            const int l = ((q + 128) & 255) - 128;
                            const int h = (q - l) >> 8;
                            hi[(g * 32 + r) * 16 + e] = (int8_t)h;
                            lo[(g * 32 + r) * 16 + e] = (int8_t)l;
                        }
                }
            }
            // ---- split-bf16 PE steps: same values as the bf16 image's PE steps of this stage
            for (int t = 0; t < sh.bfsteps; ++t) {
                uint16_t* hi = reinterpret_cast<uint16_t*>(img + frag_off8(st, nb, sh.i8steps + t));
                uint16_t* lo = hi + 64 * 8;
                // chunk index of PE step t inside the bf16 stage's chunk sequence: stage 0/5 put the PE first, stage 9 last
                const int cc0 = st == 9 ? 32 : 0;
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const float wv = stage_weight(d, P, st, 32 * nb + (lane & 31), cc0 + 2 * t + (lane >> 5), j);
                        const uint16_t h = f32_to_bf16(wv);
                        hi[lane * 8 + j] = h;
                        lo[lane * 8 + j] = f32_to_bf16(wv - bf16_to_f32(h));
                    }
            }
        }
        float* b = bias + stage_b_off(st);
        if (st <= 7) memcpy(b, P[P_PTS_W + 2 * st + 1], 256 * 4);
        else if (st == 8) { memcpy(b, P[P_FEAT_B], 256 * 4); b[256] = P[P_ALPHA_B][0]; }
        else if (st == 9) memcpy(b, P[P_VIEWS_B], 128 * 4);
        else memcpy(b, P[P_RGB_B], 3 * 4);
        if (!sh.i8steps) for (int n = 0; n < sh.nblk * 32; ++n) scales[stage_b_off(st) + n] = 1.f;
    }
}

}  // namespace nm

struct nm_mlp_s {
    nm_mlp_desc desc;
    uint8_t* d_image;      // weight fragments | pad | bias
    uint8_t* d_image8;     // NM_PREC_I8X3: limb fragments | pad | weight scales | bias
    float* d_petab;        // 192 floats
    float* d_ref;          // transposed f32 weights | natural biases (NM_PREC_FP32 path)
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
    return nm::kWeightBytes + nm::kWeightPadBytes + (int64_t)nm::kBiasFloats * 4;
}

int nm_mlp_pack(const nm_mlp_desc* desc, const float* const* host_params, void* host_out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_out, "nm_mlp_pack: null pointer");
    for (int i = 0; i < 24; ++i) NM_REQUIRE(host_params[i], "nm_mlp_pack: host_params[%d] is null", i);
    nm::pack_image(desc, host_params, static_cast<uint8_t*>(host_out));
    return NM_OK;
}

int64_t nm_mlp_pack_i8_bytes(const nm_mlp_desc* desc) {
    if (nm::validate_desc(desc) != NM_OK) return -1;
    return nm::image8_bytes();
}

int nm_mlp_pack_i8(const nm_mlp_desc* desc, const float* const* host_params, void* host_out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_out, "nm_mlp_pack_i8: null pointer");
    for (int i = 0; i < 24; ++i) NM_REQUIRE(host_params[i], "nm_mlp_pack_i8: host_params[%d] is null", i);
    nm::pack_image8(desc, host_params, static_cast<uint8_t*>(host_out));
    return NM_OK;
}

int nm_mlp_create(const nm_mlp_desc* desc, const float* const* host_params, const float* host_pos_tab,
                  const float* host_dir_tab, nm_mlp_t* out) {
    if (int e = nm::validate_desc(desc)) return e;
    NM_REQUIRE(host_params && host_pos_tab && host_dir_tab && out, "nm_mlp_create: null pointer");
    for (int i = 0; i < 24; ++i) NM_REQUIRE(host_params[i], "nm_mlp_create: host_params[%d] is null", i);
    const int64_t bytes = nm_mlp_pack_bytes(desc);
    std::vector<uint8_t> img((size_t)bytes);
    nm::pack_image(desc, host_params, img.data());
    std::vector<uint8_t> img8((size_t)nm::image8_bytes());
    nm::pack_image8(desc, host_params, img8.data());

    // reference-layout image for the exact-f32 kernel
    const int kpe = 3 + 6 * desc->pos_n_freqs, kdpe = 3 + 6 * desc->dir_n_freqs;
    const int K[12] = {kpe, 256, 256, 256, 256, kpe + 256, 256, 256, 256 + kdpe, 256, 256, 128};
    const int N[12] = {256, 256, 256, 256, 256, 256, 256, 256, 128, 256, 1, 3};
    nm_mlp_s* m = new nm_mlp_s();
    m->desc = *desc;
    int woff = 0;
    for (int l = 0; l < 12; ++l) { m->ref_off[l] = woff; woff += K[l] * N[l]; }
    int boff = woff;
    for (int l = 0; l < 12; ++l) { m->ref_boff[l] = boff; boff += N[l]; }
    std::vector<float> ref((size_t)boff);
    for (int l = 0; l < 12; ++l) {
        const float* W = host_params[2 * l];
        for (int k = 0; k < K[l]; ++k)
            for (int n = 0; n < N[l]; ++n) ref[m->ref_off[l] + (size_t)k * N[l] + n] = W[(size_t)n * K[l] + k];
        memcpy(&ref[m->ref_boff[l]], host_params[2 * l + 1], (size_t)N[l] * 4);
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

    m->d_image = nullptr; m->d_image8 = nullptr; m->d_petab = nullptr; m->d_ref = nullptr;
    int rc = nm::check_hip(hipMalloc(&m->d_image, (size_t)bytes), "nm_mlp_create: hipMalloc(image)");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_image8, img8.size()), "nm_mlp_create: hipMalloc(image8)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_image8, img8.data(), img8.size(), hipMemcpyHostToDevice), "nm_mlp_create: upload image8");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_petab, sizeof(tab)), "nm_mlp_create: hipMalloc(petab)");
    if (!rc) rc = nm::check_hip(hipMalloc(&m->d_ref, ref.size() * 4), "nm_mlp_create: hipMalloc(ref)");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_image, img.data(), (size_t)bytes, hipMemcpyHostToDevice), "nm_mlp_create: upload image");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_petab, tab, sizeof(tab), hipMemcpyHostToDevice), "nm_mlp_create: upload petab");
    if (!rc) rc = nm::check_hip(hipMemcpy(m->d_ref, ref.data(), ref.size() * 4, hipMemcpyHostToDevice), "nm_mlp_create: upload ref");
    if (rc) { nm_mlp_destroy(m); return rc; }
    *out = m;
    return NM_OK;
}

int nm_mlp_destroy(nm_mlp_t m) {
    if (!m) return NM_OK;
    if (m->d_image) (void)hipFree(m->d_image);
    if (m->d_image8) (void)hipFree(m->d_image8);
    if (m->d_petab) (void)hipFree(m->d_petab);
    if (m->d_ref) (void)hipFree(m->d_ref);
    delete m;
    return NM_OK;
}

static int mlp_dispatch(nm_mlp_t m, const float* pts, const float* dirs, const float* origin, const float* direction,
                        const float* z, int64_t n, int S, int in_mode, int precision, int stop_stage, float sigma_scale,
                        float* out, float* dbg, nm_stream_t stream, void* prof = nullptr) {
    NM_REQUIRE(m, "nm_mlp_forward: null handle");
    NM_REQUIRE(n >= 0, "nm_mlp_forward: negative n");
    NM_REQUIRE(precision == NM_PREC_FP32 || precision == NM_PREC_BF16X3 || precision == NM_PREC_BF16 || precision == NM_PREC_I8X3,
               "nm_mlp_forward: bad precision %d", precision);
    if (n == 0) return NM_OK;
    if (precision == NM_PREC_FP32) {
        nm::RefLaunch L;
        L.wt = m->d_ref; L.bias = m->d_ref; L.petab = m->d_petab;
        for (int i = 0; i < 12; ++i) { L.off[i] = m->ref_off[i]; L.boff[i] = m->ref_boff[i]; }
        L.pe_kind = m->desc.pe_kind; L.pos_nfreq = m->desc.pos_n_freqs; L.dir_nfreq = m->desc.dir_n_freqs;
        return nm::launch_mlp_ref(L, pts, dirs, origin, direction, z, n, S, in_mode, stop_stage, sigma_scale, out, dbg,
                                  nm::as_stream(stream));
    }
    nm::MlpLaunch L;
    L.wpack = m->d_image;
    L.bias = reinterpret_cast<const float*>(m->d_image + nm::kWeightBytes + nm::kWeightPadBytes);
    L.petab = m->d_petab;
    L.pe_kind = m->desc.pe_kind; L.pos_nfreq = m->desc.pos_n_freqs; L.dir_nfreq = m->desc.dir_n_freqs;
    L.pos_octaves = m->pos_octaves; L.dir_octaves = m->dir_octaves;
    L.wpack8 = m->d_image8;
    L.scales8 = reinterpret_cast<const float*>(m->d_image8 + nm::kWeightBytes8 + nm::kWeightPadBytes);
    L.bias8 = L.scales8 + nm::kBiasFloats;
    return nm::launch_mlp_mfma(L, pts, dirs, origin, direction, z, n, S, in_mode, precision, stop_stage, sigma_scale, out, dbg,
                               prof, nm::as_stream(stream));
}

int nm_mlp_forward(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision, float sigma_scale,
                   float* out, nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (pts && dirs && out), "nm_mlp_forward: null pointer");
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_forward: out must be 16-byte aligned");
    return mlp_dispatch(mlp, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, precision, -2, sigma_scale, out, nullptr, stream);
}

int nm_mlp_forward_rays(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int64_t R, int S,
                        int precision, float sigma_scale, float* out, nm_stream_t stream) {
    NM_REQUIRE(R == 0 || (origin && direction && z_vals && out), "nm_mlp_forward_rays: null pointer");
    NM_REQUIRE(R >= 0 && S >= 1, "nm_mlp_forward_rays: bad sizes");
    NM_REQUIRE((reinterpret_cast<uintptr_t>(out) & 15) == 0, "nm_mlp_forward_rays: out must be 16-byte aligned");
    return mlp_dispatch(mlp, nullptr, nullptr, origin, direction, z_vals, R * (int64_t)S, S, 1, precision, -2, sigma_scale, out,
                        nullptr, stream);
}

int nm_mlp_forward_profile(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, float* out, uint64_t* cycles,
                           nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (pts && dirs && out && cycles), "nm_mlp_forward_profile: null pointer");
    return mlp_dispatch(mlp, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, NM_PREC_BF16X3, -2, 1.f, out, nullptr, stream, cycles);
}

int nm_mlp_forward_debug(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision, int stage,
                         float* hidden, nm_stream_t stream) {
    NM_REQUIRE(n == 0 || (pts && dirs && hidden), "nm_mlp_forward_debug: null pointer");
    NM_REQUIRE(stage >= -1 && stage <= 9, "nm_mlp_forward_debug: stage %d outside -1..9", stage);
    return mlp_dispatch(mlp, pts, dirs, nullptr, nullptr, nullptr, n, 1, 0, precision, stage, 1.f, nullptr, hidden, stream);
}

}  // extern "C"
