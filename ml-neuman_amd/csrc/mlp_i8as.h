// Shared device helpers of the activation-stationary i8x3 kernels (mlp_i8s.hip: 8 waves x 32 samples; round 4's mlp_i8t.hip, removed in round 5: 4 waves x two
// 32-sample sub-tiles): encodings of a wave's rows, the resident activation fragments, LDS-DMA, de- and requantisation -- one definition,
// so that the kernels run the same arithmetic instruction for instruction (they must agree bit for bit).
#pragma once
#include "mlp_device.h"

namespace {

typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;
typedef __attribute__((ext_vector_type(2))) short i16x2;
typedef __attribute__((address_space(3))) const float lds_cfloat;      // (an LDS address is 32 bits and fits a ds_read's base register)
typedef __attribute__((ext_vector_type(4))) float vf4;
typedef __attribute__((address_space(3))) const vf4 lds_cvf4;

constexpr int kRows = 32;                                    // samples of one MFMA column block = of one (sub-)tile of a wave
struct Args8s {
    MlpArgs a;
    const float* consts8;      // units (kBiasFloats) | biases in those units (kBiasFloats) | kappa (16)
    const uint4* image8;       // the stream (mlp_host.hip pack_stream8s): [ring block][step][hi | lo][64 lanes][16 B] in block_steps() order
};

__device__ __forceinline__ i32x4 as_i32x4(uint4 v) { return __builtin_bit_cast(i32x4, v); }

// ---- encodings of this wave's 32 rows (octave recurrence, mlp_device.h fill_pe_fast, wave-private layout)
//   F16: the NM_PREC_FP16X3 form (fp16 parts of 32 v; the general path clamps like split8<false, true>, the octave path has |v| <= max(1, |x|))
template <bool F16 = false>
__device__ __forceinline__ void fill_pe_wave(uint4* pw, bool is_dir, const MlpArgs& a, int64_t base_row, int lane) {
    const PeSpec spec = is_dir ? a.dir : a.pos;
    const float* tab = a.petab + (is_dir ? 96 : 0);
    unsigned short* hi = reinterpret_cast<unsigned short*>(pw);
    auto put = [&](int row, int p, float v, bool clamp = false) {
        const int off = ((p >> 3) * (2 * kRows) + row) * 8 + (p & 7);
        if (F16) {
            float sv = v * kF16ActScale;
            if (clamp) sv = __builtin_amdgcn_fmed3f(sv, -65504.f, 65504.f);
            const _Float16 hb = (_Float16)sv;
            const _Float16 lb = (_Float16)(sv - (float)hb);
            hi[off] = __builtin_bit_cast(unsigned short, hb);
            hi[off + kRows * 8] = __builtin_bit_cast(unsigned short, lb);
            return;
        }
        const bf16x2 hb = __builtin_convertvector((f32x2){v, 0.f}, bf16x2);
        const f32x2 hf = __builtin_convertvector(hb, f32x2);
        const bf16x2 lb = __builtin_convertvector((f32x2){v - hf.x, 0.f}, bf16x2);
        hi[off] = (unsigned short)(__builtin_bit_cast(unsigned, hb) & 0xffffu);
        hi[off + kRows * 8] = (unsigned short)(__builtin_bit_cast(unsigned, lb) & 0xffffu);
    };
    if (spec.octaves) {
#pragma unroll 1
        for (int round = 0; round < 2; ++round) {
            const int j = 2 * round + (lane >> 5), row = lane & 31;
            if (j > 2) break;
            int64_t i = base_row + row;
            if (i >= a.n) i = a.n - 1;
            float x0, x1, x2;
            sample_input(a, i, is_dir, x0, x1, x2);
            const float xj = j == 0 ? x0 : (j == 1 ? x1 : x2);
            const float a0 = spec.kind == NM_PE_POSENC ? xj * tab[0] : fmaf(x2, tab[3 * j + 2], fmaf(x1, tab[3 * j + 1], x0 * tab[3 * j]));
            put(row, j, xj);
            double sn, cs;
            sincos_f64((double)a0, sn, cs);
            const int n3 = 3 * spec.nfreq;
            for (int b = 0; b < spec.nfreq; ++b) {
                if (spec.kind == NM_PE_POSENC) { put(row, 3 + 6 * b + j, (float)sn); put(row, 3 + 6 * b + 3 + j, (float)cs); }
                else { put(row, 3 + 3 * b + j, (float)sn); put(row, 3 + n3 + 3 * b + j, (float)cs); }
                const double s2 = 2.0 * sn * cs, c2 = 1.0 - 2.0 * sn * sn;
                sn = s2; cs = c2;
            }
        }
    } else {
        const int nchunks = is_dir ? 4 : nm::kPeChunks;
        for (int item = lane; item < nchunks * kRows; item += 64) {
            const int c = item >> 5, row = item & 31;
            int64_t i = base_row + row;
            if (i >= a.n) i = a.n - 1;
            float x0, x1, x2;
            sample_input(a, i, is_dir, x0, x1, x2);
#pragma unroll
            for (int e = 0; e < 8; ++e) put(row, 8 * c + e, pe_feature(8 * c + e, x0, x1, x2, spec, tab), true);
        }
    }
}

struct X8 {
    uint4 h[8], l[8];          // the wave's activations: k-step t = feature block t of the producing stage, hi / lo limbs
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// (bias_blk: the block's 32 biases + 4 * g, this lane's half of every group of 8)
__device__ __forceinline__ void bias16(f32x16& f, lds_cfloat* bias_blk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const vf4 bs = *(lds_cvf4*)(bias_blk + 8 * q);
        f[4 * q] = bs.x; f[4 * q + 1] = bs.y; f[4 * q + 2] = bs.z; f[4 * q + 3] = bs.w;
    }
}
__device__ __forceinline__ void dequant16(f32x16& f, const i32x16& t, float sx256, lds_cfloat* bias_blk) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const vf4 bs = *(lds_cvf4*)(bias_blk + 8 * q);
        const float bsv[4] = {bs.x, bs.y, bs.z, bs.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) f[4 * q + j] = fmaf((float)t[4 * q + j], sx256, bsv[j]);
    }
}
template <bool RELU>
__device__ __forceinline__ float max16(float m, const f32x16& f) {
#pragma unroll
    for (int r = 0; r < 16; ++r) m = fmaxf(m, RELU ? f[r] : fabsf(f[r]));
    return m;
}
__device__ __forceinline__ float row_max(float m) {                  // the two lane halves of a sample hold different features
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    const u32x2 sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    return fmaxf(__uint_as_float(sw.x), __uint_as_float(sw.y));
}
// one block's 16 outputs of this lane -> the next stage's k-step fragment (two balanced int8 limbs); nerf_mlp_i8w_kernel's quant_storew
template <bool RELU>
__device__ __forceinline__ void quant16(const f32x16& f, float inv, uint4& xh, uint4& xl) {
    i16x2 P[8], Y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float y0 = f[2 * i] * inv, y1 = f[2 * i + 1] * inv;
        if (RELU) {
            y0 = __builtin_amdgcn_fmed3f(y0, 0.f, 1.f);
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
    xh = make_uint4(hi[0], hi[1], hi[2], hi[3]);
    xl = make_uint4(lo[0], lo[1], lo[2], lo[3]);
}
__device__ __forceinline__ float inv_of(float M) { return M > 0.f ? ((float)nm::kFixedMax / 32767.f) * __builtin_amdgcn_rcpf(M) : 0.f; }
__device__ __forceinline__ float scale_of(float M) { return M > 0.f ? M * (1.f / (float)nm::kFixedMax) : 1.f; }


}  // namespace
