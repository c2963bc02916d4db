// Layout contract between the host packer (mlp_pack.cpp), the MFMA kernel (mlp.hip) and the numpy
// emulator in tests/test_mlp_pack.py.  Pure constexpr, host + device.
//
// The net (reference models/vanilla.py:95-152, options/options.py:52-71): 8 x 256 ReLU layers with
// cat([x_pe, h]) after layer 4, then alpha (256->1) and feature (256->256, linear), views
// (cat([feature, d_pe]) 283 -> 128, ReLU), rgb (128 -> 3).  It is executed as 11 GEMM "stages":
//
//   stage 0      K = x_pe(64 = 63 + zero pad)             N = 256   relu
//   stage 1-4    K = h(256)                               N = 256   relu
//   stage 5      K = x_pe(64) ++ h(256)                   N = 256   relu      (skip, vanilla.py:130-131)
//   stage 6-7    K = h(256)                               N = 256   relu
//   stage 8      K = h(256)                               N = 256 feature (linear) + 32-wide block whose
//                                                             row 0 is alpha (vanilla.py:135-136)
//   stage 9      K = feature(256) ++ d_pe(32 = 27 + pad)  N = 128   relu      (vanilla.py:137-141)
//   stage 10     K = h(128)                               N = 32 (rows 0..2 = rgb, vanilla.py:143)
//
// GEMM orientation: D[feature][sample] = W[feature][k] * X[k][sample] on v_mfma_f32_32x32x16_bf16,
// A operand = weights (32 output features x 16 k), B operand = activations (16 k x 32 samples).
//
// Activations live in LDS as 16-byte "chunks": chunk c of a sample holds 8 consecutive k-slots as
// bf16; a k-step (16 k) reads chunks 2t (lanes 0-31) and 2t+1 (lanes 32-63).  The hi and lo halves of
// the split-bf16 value are separate arrays with identical indexing.
//
// Which feature sits in k-slot (chunk c, element e) is fixed by what the epilogue can write with one
// ds_write_b128: a lane of the 32x32 accumulator tile holds, for its sample, features
// (reg&3) + 8*(reg>>2) + 4*(lane>>5); registers 8*qp .. 8*qp+7 (qp = 0,1) become one chunk.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define NM_HD __host__ __device__
#else
#define NM_HD
#endif

namespace nm {

constexpr int kTileM = 128;      // samples per workgroup tile
constexpr int kStages = 11;
constexpr int kHChunks = 32;     // 256 features
constexpr int kPeChunks = 8;     // 64 position-PE slots (dir PE uses the first 4)
constexpr int kStepBytes = 2048; // one k-step of one 32-feature block: [hi|lo][64 lanes][8 bf16]

struct StageShape {
    int nblk;    // 32-feature output blocks (incl. the alpha block of stage 8)
    int steps;   // k-steps (16 k each)
    int pe_steps;  // of which read the PE buffer: before h for stage 0/5, after h for stage 9
};

NM_HD constexpr StageShape stage_shape(int s) {
    return s == 0 ? StageShape{8, 4, 4}
         : s == 5 ? StageShape{8, 20, 4}
         : s == 8 ? StageShape{9, 16, 0}
         : s == 9 ? StageShape{4, 18, 2}
         : s == 10 ? StageShape{1, 8, 0}
                   : StageShape{8, 16, 0};
}

// byte offset of stage s, block nb, k-step t inside the weight image
NM_HD constexpr int64_t stage_w_off(int s) {
    int64_t o = 0;
    for (int i = 0; i < s; ++i) o += (int64_t)stage_shape(i).nblk * stage_shape(i).steps * kStepBytes;
    return o;
}
NM_HD constexpr int64_t frag_off(int s, int nb, int t) {
    return stage_w_off(s) + ((int64_t)nb * stage_shape(s).steps + t) * kStepBytes;
}
// float offset of stage s inside the bias image (32 floats per block, natural feature order)
NM_HD constexpr int stage_b_off(int s) {
    int o = 0;
    for (int i = 0; i < s; ++i) o += stage_shape(i).nblk * 32;
    return o;
}
constexpr int64_t kWeightBytes = stage_w_off(kStages);
constexpr int64_t kWeightPadBytes = 4 * kStepBytes;   // the k-loop prefetches 2 steps past the end
constexpr int kBiasFloats = stage_b_off(kStages);

// feature held by k-slot (chunk c, element e) of a 256- or 128-wide activation written by the epilogue
NM_HD constexpr int slot_feature(int c, int e) {
    // c = 4*blk + 2*qp + h ;  feature = 32*blk + 8*(2*qp + (e>>2)) + 4*h + (e&3)
    return 32 * (c >> 2) + 8 * (2 * ((c >> 1) & 1) + (e >> 2)) + 4 * (c & 1) + (e & 3);
}
// inverse: chunk and element of feature n
NM_HD constexpr int feature_chunk(int n) { return 4 * (n >> 5) + 2 * ((n >> 4) & 1) + ((n >> 2) & 1); }
NM_HD constexpr int feature_elem(int n) { return 4 * ((n >> 3) & 1) + (n & 3); }

// ---- NM_PREC_I8X3: 16-bit fixed-point limbs on v_mfma_i32_32x32x32_i8 ------------------------------------------
// The 256-/128-wide hidden operands are per-row scaled int16, stored as two balanced int8 limbs (X = 256*hi + lo,
// hi, lo in [-128,127], |X| <= 32639); weights likewise per output feature.  A k-step covers 32 k (one 16-byte chunk of
// 16 int8 per lane half), so k-step t of a stage reads exactly the 32 features of block t.  The position / direction
// encodings keep the split-bf16 path (they need absolute, not row-relative, precision): stage 0 is bf16 only, stages 5
// and 9 run their hidden part on i8 first and then accumulate the PE part in f32 on top of the dequantised sum.
struct StageShape8 {
    int nblk;     // 32-feature output blocks
    int i8steps;  // 32-k steps over the hidden input (i8 limbs)
    int bfsteps;  // 16-k steps over the PE buffer (split bf16), after the i8 steps in the stream
};
NM_HD constexpr StageShape8 stage_shape8(int s) {
    return s == 0 ? StageShape8{8, 0, 4}
         : s == 5 ? StageShape8{8, 8, 4}
         : s == 8 ? StageShape8{9, 8, 0}
         : s == 9 ? StageShape8{4, 8, 2}
         : s == 10 ? StageShape8{1, 4, 0}
                   : StageShape8{8, 8, 0};
}
NM_HD constexpr int64_t stage_w_off8(int s) {
    int64_t o = 0;
    for (int i = 0; i < s; ++i) o += (int64_t)stage_shape8(i).nblk * (stage_shape8(i).i8steps + stage_shape8(i).bfsteps) * kStepBytes;
    return o;
}
NM_HD constexpr int64_t frag_off8(int s, int nb, int step) {   // step counts i8 steps first, then bf steps
    return stage_w_off8(s) + ((int64_t)nb * (stage_shape8(s).i8steps + stage_shape8(s).bfsteps) + step) * kStepBytes;
}
constexpr int64_t kWeightBytes8 = stage_w_off8(kStages);
// the activation-stationary kernel's stream of the plain-head net (mlp_i8s.hip PLAIN): stages 0..7, the block of output_linear's four rows, and -- where the
// ring's look-ahead expects the 8-step blocks 69 and 70 -- the next tile's blocks 0 and 1 (4 steps each) padded to 8 steps
constexpr int64_t kPlainStreamBytes8 = (int64_t)(8 * 4 + 7 * 8 * 8 + 8 * 4 + 8 + 2 * 8) * kStepBytes;
// after the fragments (+ the same prefetch pad): per-feature weight scales, then biases, both laid out like stage_b_off()
constexpr int kFixedMax = 32639;   // 127*256 + 127: largest magnitude whose balanced limbs fit int8
// feature held by k-slot (16-slot chunk c, element e) of an i8 activation: c = 2*blk + g, e = reg index of the accumulator
NM_HD constexpr int slot_feature8(int c, int e) { return 32 * (c >> 1) + (e & 3) + 8 * (e >> 2) + 4 * (c & 1); }
NM_HD constexpr int feature_chunk8(int n) { return 2 * (n >> 5) + ((n >> 2) & 1); }
NM_HD constexpr int feature_elem8(int n) { return (n & 3) + 4 * ((n >> 3) & 3); }

// ---- stream image of the wave-specialised i8x3 kernel (nerf_mlp_i8w_kernel) -----------------------------------------------
// Wave q (0..3) of a group owns feature blocks 2q, 2q+1 of the 256-wide stages, block q of stage 9 and -- q < 2 only -- one
// 32-sample half of the alpha block and of stage 10.  Its k-steps are stored in exactly the order it consumes them, so the
// whole weight traffic of a wave is ONE linear stream (base + i * kStepBytes) that a deep register ring can prefetch with
// no address logic.  Order (steps of kStepBytes, fragment format as above):
//   stage 0: blk 2q bf 0..3, blk 2q+1 bf 0..3 | stages 1-4,6,7: blk 2q i8 0..7, blk 2q+1 i8 0..7
//   stage 5: blk 2q i8, blk 2q+1 i8, blk 2q bf 0..3, blk 2q+1 bf 0..3 | stage 8: blk 2q i8, blk 2q+1 i8, [q < 2: alpha blk i8 0..7]
//   stage 9: blk q i8 0..7, bf 0..1, two zero steps | [q < 2: stage 10 blk 0 i8 0..3] | kW8Pad zero steps (prefetch overrun)
constexpr int kW8Pad = 8;
NM_HD constexpr int wstream_steps(int q) { return 8 + 6 * 16 + 24 + 16 + (q < 2 ? 8 : 0) + 12 + (q < 2 ? 4 : 0) + kW8Pad; }
NM_HD constexpr int64_t wstream_off(int q) {
    int64_t o = 0;
    for (int i = 0; i < q; ++i) o += (int64_t)wstream_steps(i) * kStepBytes;
    return o;
}
constexpr int64_t kWeightBytes8w = wstream_off(4);

}  // namespace nm
