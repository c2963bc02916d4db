// Launch descriptors shared by mlp_host.hip (dispatch) and the two kernel files.
#pragma once
#include "common.h"

namespace nm {

// indices into the 24 parameter arrays of a net (reference state_dict order, include/neuman_hip.h)
enum { P_OUT_W = 16, P_OUT_B = 17 };      // plain head (use_viewdirs=False): output_linear follows the 16 pts_linears tensors
enum { P_PTS_W = 0, P_VIEWS_W = 16, P_VIEWS_B = 17, P_FEAT_W = 18, P_FEAT_B = 19, P_ALPHA_W = 20, P_ALPHA_B = 21, P_RGB_W = 22, P_RGB_B = 23 };
struct DevParams {                        // the same arrays on the device (a net under training: the live parameters)
    const float* p[24];
};

struct MlpLaunch {
    const void* wpack; const float* bias; const float* petab;
    int pe_kind, pos_nfreq, dir_nfreq;
    int pos_octaves, dir_octaves;   // encodings whose bands are consecutive powers of two (octave recurrence allowed)
    int plain_head;                 // use_viewdirs=False net: output_linear instead of the alpha / feature / views / rgb heads
    const void* wpack16; const float* bias16;     // NM_PREC_FP16X3: the same fragment image as split fp16 of W * 2^8, biases * 2^13
    const void* wstream8; const float* consts8;   // NM_PREC_I8X3: per-wave fragment streams; units | biases | kappa (mlp_layout.h)
    float* save_h = nullptr;                      // nm_mlp_forward_save (NM_PREC_FP16X3): [9][n][256] post-activation outputs of layers 0..7, then feature
    float* save_hv = nullptr;                     //   and [n][128] of the views layer: what a training step's backward pass reads
    unsigned* save_bits = nullptr;                //   nullable: [8][n][8] one bit per trunk activation (> 0): what nm_mlp_backward_chain masks with
    void* save_h16 = nullptr;                     //   nullable: [8][n][256] fp16 trunk activations (x 32, k-slot order) instead of float32; save_h is then [n][256]: feature only (nullable)
    void* save_feat16 = nullptr;                  //   nullable: [n][256] fp16 feature output likewise;  save_hvbits nullable: [n][4] signs of the views layer
    unsigned* save_hvbits = nullptr;
    void* save_x0h = nullptr; void* save_d0h = nullptr;   //   nullable: [n][64] fp16 encodings of position / direction (x 32; the direction's slot 63 holds 1)
};
struct RefLaunch {
    const float* wt; const float* bias; int off[12]; int boff[12]; const float* petab;
    int pe_kind, pos_nfreq, dir_nfreq;
    int plain_head;
};

// in_mode 2: the launch covers samples s0 .. s0 + S - 1 of the rays listed in ray_idx (n = upper bound of listed rays * S; the
// actual list length is read from *n_rays_dev on the device when that is non-null); z and out are [R, S_total] (x4)
struct MlpChunk {
    const int* ray_idx; const int* n_rays_dev; int s0, S_total;
};

int launch_mlp_mfma(const MlpLaunch& L, const float* pts, const float* dirs, const float* origin, const float* direction,
                    const float* z, int64_t n, int S, int in_mode, int precision, int stop_stage, float sigma_scale, float* out,
                    float* dbg, void* prof, hipStream_t stream, int sigma_only = 0, const MlpChunk* chunk = nullptr);
// NM_PREC_I8X3, activation-stationary schedule (mlp_i8s.hip); image8 = the block image of mlp_layout.h frag_off8 on the device
int launch_mlp_i8s(const MlpLaunch& L, const void* image8, const float* pts, const float* dirs, const float* origin, const float* direction,
                   const float* z, int64_t n, int S, int in_mode, float sigma_scale, float* out, hipStream_t stream, const MlpChunk* chunk);
// NM_PREC_FP16X3 density only, activation-stationary (mlp_f16t.hip); stream16t = sigma_stream_kernel's re-cut of the fp16 image.  dbg (nullable): the
// activations after stage dbg_stage of the first tile (+ 100 x round) of workgroup 0 as float32 [128 samples][256] in k-slot order
int sigma_f16t_ndir();       // the NDIR tools/gen_f16t.py emitted mlp_f16t_body.h for: the stream is cut for exactly that
int launch_sigma_f16t(const MlpLaunch& L, const void* stream16t, int stream_ndir, const float* pts, const float* dirs, const float* origin, const float* direction,
                      const float* z, int64_t n, int S, int in_mode, float sigma_scale, float* out, hipStream_t stream, const MlpChunk* chunk,
                      float* dbg, int dbg_stage);
// the backward-data chain of the 8 x 256 trunk (mlp_bwd.hip): packs the transposed weights from the live parameters into `image`
// (mlp_bwd_image_bytes()).  d_feat == nullptr: from dz_top = dZ_7, NS = 7 stages -> dz_out [7][n][256] = dZ_6 .. dZ_0; d_feat != nullptr: the
// first stage forms dZ_7 from d_feat and d_raw's sigma column itself, NS = 8 -> dz_out [8][n][256] = dZ_7 .. dZ_0.  colsum [tiles][NS][256]
// scratch, gb [NS][256] = the column sums (bias gradients)
int64_t mlp_bwd_image_bytes();
// h != nullptr: the 16-bit form -- dz16 [NS][n][256] fp16 of dZ * nm_dz_scale(*amax) in k-slot order instead of dz_out, dfeat16 (nullable) d_feat the
// same way, dz32[layer] (nullable each) float32 copies of single layers
struct Bwd16 {
    void* dz16; void* dfeat16; const float* amax; float* dz32[8];
    // hvbits != nullptr: the whole backward pass from d_raw (d_feat unused): the views layer's adjoint is formed in the kernel; dhv16 [n][128] fp16
    // (x scale, k-slot order), dhv32 (nullable) its float32 copy; the bias-gradient block gets a 9th row: feature_linear's; kdir = encoded view width
    const unsigned* hvbits = nullptr; void* dhv16 = nullptr; float* dhv32 = nullptr; int kdir = 0;
    // plain: the plain-head net (output_linear [4][256] straight off layer 7, no views layer): d_raw = d_out [n][4], dZ_7 = (d_out W_out) * (H_7 > 0)
    // is the first stage; 8 stages, bias_grads [8][256] = layers 7 .. 0
    int plain = 0;
};
int launch_mlp_bwd(const DevParams& P, int kpe, uint8_t* image, const float* dz_top, const float* d_feat, const float* d_raw, const float* acts,
                   const unsigned* relu_bits, int64_t n, float* dz_out, float* colsum, float* gb, hipStream_t stream, const Bwd16* h = nullptr);
int launch_mlp_ref(const RefLaunch& L, const float* pts, const float* dirs, const float* origin, const float* direction,
                   const float* z, int64_t n, int S, int in_mode, int stop_stage, float sigma_scale, float* out, float* dbg,
                   hipStream_t stream);

}  // namespace nm
