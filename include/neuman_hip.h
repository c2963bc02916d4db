/*
 * neuman_hip.h -- C ABI of libneuman_hip.so: the NeuMan ray-march hot path on MI355X (gfx950).
 *
 * The reference (apple/ml-neuman) has no FFI seam: its hot path is the Python function level of
 * utils/ray_utils.py, utils/render_utils.py and models/vanilla.py.  This header IS the seam a
 * maintainer binds underneath those functions (ctypes stub: INTEGRATION.md).  Each entry point names
 * the reference lines it replaces.
 *
 * Conventions
 *   - every function returns 0 (NM_OK) or a negative NM_ERR_* code; nm_last_error() returns a
 *     thread-local human readable message for the last failure on the calling thread;
 *   - every pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr()) unless the
 *     parameter name starts with host_;  all tensors are dense, row-major, float32 / int32;
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); kernels
 *     are enqueued on it and the call returns without synchronising;
 *   - no hidden allocation except inside the opaque nm_mlp_t handle; no global mutable state;
 *   - there is NO CPU fallback: without a HIP device the compute entry points fail with NM_ERR_HIP.
 */
#ifndef NEUMAN_HIP_H
#define NEUMAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NM_OK 0
#define NM_ERR_ARG (-1)         /* bad argument (null pointer, size, unsupported shape) */
#define NM_ERR_HIP (-2)         /* HIP runtime error (message carries hipGetErrorString) */
#define NM_ERR_UNSUPPORTED (-3) /* valid request this build does not implement */

#define NM_ABI_VERSION 1

typedef void* nm_stream_t;

/* precision of the MLP contraction (nm_mlp_forward*) */
#define NM_PREC_FP32 0   /* exact-f32 FMA chains on the vector ALU: slow validation path            */
#define NM_PREC_BF16X3 1 /* split-bf16 (hi+lo) x3 MFMA, f32 accumulate: the parity-grade default     */
#define NM_PREC_BF16 2   /* single bf16 MFMA, f32 accumulate: fast, NOT parity grade     */
#define NM_PREC_I8X3 3   /* hidden layers as per-row-scaled int16 = two int8 limbs on the i8 MFMA (hh + hl + lh, exact
                            int32 accumulate), encodings on split bf16: composited colours within 2e-5 of f32 on identical
                            samples -- parity grade for passes that are only composited (the host's default policy uses it
                            there), NOT for a pass whose weights place importance samples */
#define NM_PREC_FP16X3 4 /* split-fp16 (hi+lo, 11+11 significand bits) x3 MFMA, f32 accumulate, exact power-of-two operand
                            scalings: float32-class results (sigma within ~2e-6 of an f64 evaluation, as an f32 sgemm is)
                            at the bf16x3 price -- the arithmetic of a pass whose weights place importance samples      */

/* positional-encoding kinds (reference models/vanilla.py:44-79) */
#define NM_PE_POSENC 0 /* [x, sin(f0 x), cos(f0 x), sin(f1 x), ...]   vanilla.py:60-79,92 */
#define NM_PE_ROTATE 1 /* [x, sin(x B^T), cos(x B^T)]                 vanilla.py:44-58,83-89 */

int nm_version(void);
const char* nm_last_error(void);
/* number of visible HIP devices (0 when none): lets a host fail loudly before any compute call */
int nm_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * a4  ray_to_samples -- reference utils/ray_utils.py:96-135
 *   z = near*(1-t) + far*t  (or the lindisp form, :113-114); optional stratified jitter with a
 *   caller-drawn, already clipped t_rand[R,S] (:116-129); pts = o + d*z (:131); dirs = d repeated.
 *   t_vals[S] is the caller's torch.linspace(0,1,S) so both sides consume identical t.
 *   pts [R,S,3] and dirs [R,S,3] are optional (NULL = do not materialise).
 * ------------------------------------------------------------------------------------------- */
int nm_ray_to_samples(const float* origin, const float* direction, const float* near, const float* far,
                      int64_t R, int S, const float* t_vals, int lindisp, const float* t_rand,
                      float* pts, float* dirs, float* z_vals, nm_stream_t stream);

/* pts[i,s,:] = origin[i,:] + direction[i,:] * z[i,s]; dirs likewise (ray_utils.py:153-156). */
int nm_z_to_points(const float* origin, const float* direction, const float* z_vals, int64_t R, int S,
                   float* pts, float* dirs, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a5  raw2outputs -- reference utils/render_utils.py:69-105
 *   raw [R,S,4] = (r,g,b,sigma), z [R,S], rays_d [R,3]; noise [R,S] optional (the caller's
 *   torch.randn * raw_noise_std, :91-93).  Outputs: rgb [R,3], disp [R], acc [R], weights [R,S]
 *   (optional), depth [R].  One wavefront per ray, transmittance by a wave prefix product.
 * ------------------------------------------------------------------------------------------- */
int nm_composite(const float* raw, const float* z_vals, const float* rays_d, int64_t R, int S, int white_bkg,
                 const float* noise, float* rgb, float* disp, float* acc, float* weights, float* depth,
                 nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a6  sample_pdf (det=True) -- reference utils/ray_utils.py:164-194
 *   bins [R,B], weights [R,B-1], u [N] (caller's torch.linspace(0,1,N)) -> samples [R,N]
 * a7  ray_to_importance_samples -- reference utils/ray_utils.py:138-160
 *   z [R,S], weights [R,S] (as returned by raw2outputs) -> z_out [R,S+N] sorted (including_old)
 *   or [R,N] (not including_old).  Mid-points, weights[1:-1] slicing and the sort are fused.
 * ------------------------------------------------------------------------------------------- */
int nm_sample_pdf(const float* bins, const float* weights, int64_t R, int B, const float* u, int N,
                  float* samples, nm_stream_t stream);
int nm_importance_z(const float* z_vals, const float* weights, int64_t R, int S, const float* u, int N,
                    int including_old, float* z_out, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a2  geometry_guided_near_far -- reference utils/ray_utils.py:197-233
 *   near = min_v(z0 - dz), far = max_v(z0 + dz), dz = sqrt(tau^2 - (|v-o|^2 - z0^2)), NaN -> +-inf.
 * a3  hit-ray compaction -- reference utils/render_utils.py:199-212 (boolean-mask indexing)
 *   hit_idx receives the indices of rays with near < far in ascending order, miss_idx (optional)
 *   the others; counts[0] = n_hit, counts[1] = n_miss (device int32[2]).  Wave ballot + prefix sum.
 *   workspace: device int32, at least nm_compact_workspace_ints(R) entries.
 * ------------------------------------------------------------------------------------------- */
int nm_near_far(const float* origin, const float* direction, int64_t R, const float* verts, int V, double geo_threshold,
                float* near, float* far, nm_stream_t stream);
int64_t nm_compact_workspace_ints(int64_t R);
int nm_compact_hits(const float* near, const float* far, int64_t R, int32_t* hit_idx, int32_t* miss_idx,
                    int32_t* counts, int32_t* workspace, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a8/a9/a10  Joiner.forward = Embedder x2 + NeRF.forward -- reference models/vanilla.py:82-92,120-152,162-166
 *
 *   The net is the reference's default: depth 8, width 256, skip after layer 4, use_viewdirs
 *   (options/options.py:52-57), pos PE -> 63 features, dir PE -> 27 features.
 *   host_params: 24 HOST pointers to float32 arrays in the reference state_dict order
 *     nerf.pts_linears.{0..7}.{weight,bias}, nerf.views_linears.0.{weight,bias},
 *     nerf.feature_linear.{weight,bias}, nerf.alpha_linear.{weight,bias}, nerf.rgb_linear.{weight,bias}
 *   (weights are [out,in] row-major as torch.nn.Linear stores them).
 *   pe tables: posenc -> host_pos_tab[n_freqs] = the f32 frequency bands; rotate ->
 *   host_pos_tab[3*n_freqs*3] = Embedder.bvals ([3N,3] row-major).
 *   nm_mlp_create packs the weights into MFMA fragment order (split bf16 hi/lo) and uploads them;
 *   the handle owns that device memory.  A handle is immutable: rebuild it after the weights change.
 * ------------------------------------------------------------------------------------------- */
typedef struct nm_mlp_s* nm_mlp_t;

typedef struct nm_mlp_desc {
    int32_t depth;        /* 8 */
    int32_t width;        /* 256 */
    int32_t skip;         /* 4: cat([x_pe, h]) after pts_linears[4] */
    int32_t pe_kind;      /* NM_PE_POSENC / NM_PE_ROTATE, used for both inputs (vanilla.py:216,225) */
    int32_t pos_n_freqs;  /* 10 */
    int32_t dir_n_freqs;  /* 4 */
    int32_t plain_head;   /* 0: the use_viewdirs=True net (alpha / feature / views / rgb heads, vanilla.py:112-115, 133-144);
                             1: use_viewdirs=False (`--specular_can no`, models/human_nerf.py:28): one output_linear 256 -> 4 =
                             (r, g, b, sigma) on the eighth layer (vanilla.py:116-117, 145), view directions ignored.  host_params
                             then holds the 16 pts_linears tensors followed by output_linear.weight [4,256] and .bias [4] (the other
                             entries are not read).  NM_PREC_I8X3: the whole-network launch only (nerf_mlp_i8s_kernel<true>: output_linear's rows
                             in the alpha block, the tile ends there); no stage-by-stage or density-only i8 form. */
} nm_mlp_desc;

int64_t nm_mlp_pack_bytes(const nm_mlp_desc* desc);
/* host-only: write the packed weight image (what nm_mlp_create uploads) into host_out. */
int nm_mlp_pack(const nm_mlp_desc* desc, const float* const* host_params, void* host_out);
/* host-only: the NM_PREC_FP16X3 image -- same size and fragment layout, split fp16 of W * 2^8, biases * 2^13 (the exact
 * power-of-two scalings that keep both parts of every operand inside fp16's normal range; csrc/mlp.hip). */
int nm_mlp_pack_f16(const nm_mlp_desc* desc, const float* const* host_params, void* host_out);
/* host-only: the NM_PREC_I8X3 image by stage and block (limb fragments | pad | per-feature units | biases in those
 * units | one scalar per stage); nm_mlp_create uploads its fragments re-ordered into per-wave streams. */
int64_t nm_mlp_pack_i8_bytes(const nm_mlp_desc* desc);
int nm_mlp_pack_i8(const nm_mlp_desc* desc, const float* const* host_params, void* host_out);
/* host-only: the limb fragments of that image as the activation-stationary kernel streams them (what nm_mlp_create uploads for it):
 * k-steps of 2 KB in consumption order -- stage 0; stages 1-7, stage 5's four encoding steps per block behind its eight i8 blocks;
 * stage 8 with the alpha block first; stage 9; stage 10 -- followed by 8 KB of zeros (the ring copies whole 1 KB pieces: a 10-step block is rounded up). */
int64_t nm_mlp_pack_i8s_bytes(const nm_mlp_desc* desc);
int nm_mlp_pack_i8s(const nm_mlp_desc* desc, const float* const* host_params, void* host_out);
int nm_mlp_create(const nm_mlp_desc* desc, const float* const* host_params, const float* host_pos_tab,
                  const float* host_dir_tab, nm_mlp_t* out);
int nm_mlp_destroy(nm_mlp_t mlp);

/* out[i,:] = NeRF(PE(pts[i]), PE(dirs[i])) * (1,1,1,sigma_scale);  pts/dirs [n,3], out [n,4].
 * sigma_scale carries `out[..., -1] *= interval_comp` (render_utils.py:229); pass 1.0 otherwise. */
int nm_mlp_forward(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision,
                   float sigma_scale, float* out, nm_stream_t stream);
/* The forward of a TRAINING step (reference trainers/vanilla_nerf_trainer.py:66, 79: `self.coarse_net(pts, dirs)` with grad):
 * nm_mlp_forward in split-fp16 x3 (float32 class) that also keeps what the backward pass reads -- save_h [9][n][256] = the outputs of
 * pts_linears 0..7 after their ReLU, then feature_linear's (models/vanilla.py:125-138); save_hv [n][128] = views_linears[0]'s after its
 * ReLU (:140-142) -- written from the kernel's epilogues, so a tile's activations never return from HBM between layers.
 * nm_mlp_refresh_f16 rewrites the handle's split-fp16 weight image from DEVICE-resident parameters (24 device pointers, reference
 * state_dict order) in three small kernels: what nm_mlp_create packs on the host, bit for bit, for weights an optimiser changes every
 * iteration.  The plain-head net: 18 tensors (the trunk's 16, output_linear's weight [4][256] and bias [4]). */
int nm_mlp_refresh_f16(nm_mlp_t mlp, const float* const* dev_params, nm_stream_t stream);
int nm_mlp_forward_save(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, float* save_h, float* save_hv,
                        float* out, nm_stream_t stream);
/* ... and, with save_bits [8][n][8] != NULL, one bit per trunk activation (> 0): word f >> 5 of (layer, sample) holds output f = 32 w + 8 q + 4 g + j
 * of pts_linears[layer] at bit 16 g + 15 - (4 q + j) (the producing kernel's register order) -- all the backward-data chain needs of them (1/32 of the bytes). */
int nm_mlp_forward_save_bits(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, float* save_h, float* save_hv,
                             uint32_t* save_bits, float* out, nm_stream_t stream);
/* The backward-data chain of the trunk in a TRAINING step (autograd's adjoint of models/vanilla.py:126-134 inside
 * trainers/vanilla_nerf_trainer.py:45-96 / human_nerf_trainer.py:382-446 `loss.backward()`), one kernel with a 128-sample tile's dZ kept on
 * chip between the layers (split-bf16 x3 products, as nm_gemm_bf16x3):
 *     dZ_{i-1} = (dZ_i W_i[:, hidden columns]) * (H_{i-1} > 0),   bias gradient of layer i-1 = column sums of dZ_{i-1},   i = 7 .. 1.
 * Two forms: from dz_top [n][256] = dZ_7 (d_feat = d_raw = NULL): dz_out [7][n][256] = dZ_6 .. dZ_0, bias_grads [7][256];
 * or from d_feat [n][256] (gradient of feature_linear's output) and d_raw [n][4] (column 3: d sigma), the first stage forming
 *     dZ_7 = (d_feat W_feature + d sigma w_alpha) * (H_7 > 0)                                             (models/vanilla.py:133-134)
 * itself: dz_out [8][n][256] = dZ_7 .. dZ_0, bias_grads [8][256].  acts = nm_mlp_forward_save's save_h, relu_bits = nm_mlp_forward_save_bits'
 * save_bits (either may be NULL; the bits are read when given); dev_params = the 24 DEVICE pointers of nm_mlp_refresh_f16 (the weights are
 * repacked, transposed, from their live values in the call); workspace of nm_mlp_backward_chain_workspace_floats(n) floats.  The weight
 * gradients are the caller's products of dz_out with the saved activations (nm_gemm_*). */
int64_t nm_mlp_backward_chain_workspace_floats(int64_t n);
int nm_mlp_backward_chain(nm_mlp_t mlp, const float* const* dev_params, const float* dz_top, const float* d_feat, const float* d_raw,
                          const float* acts, const uint32_t* relu_bits, int64_t n, float* dz_out, float* bias_grads, float* workspace,
                          int64_t workspace_floats, nm_stream_t stream);
/* The 16-bit form of the two calls above (round 5): what a training step keeps between its forward and backward pass -- the reference's autograd
 * keeps every layer's float32 output (models/vanilla.py:126-141 under trainers/vanilla_nerf_trainer.py:45-96) -- is stored as fp16 wherever it is only
 * an operand of a weight-gradient product (a sum over all samples: the 2^-12 roundings of independent samples average out):
 *   nm_mlp_forward_save16: save_h16 [8][n][256] = fp16 of 32 x (output of pts_linears[l], after ReLU) -- the very hi part the next layer's MFMA reads --
 *     in K-SLOT ORDER: 16-byte chunk c, element e of a row holds feature 32 (c >> 2) + 8 (2 ((c >> 1) & 1) + (e >> 2)) + 4 (c & 1) + (e & 3)
 *     (mlp_layout.h slot_feature: the order the producing lanes hold them); feature_linear's output as save_feat [n][256] float32 and / or save_feat16
 *     [n][256] fp16 (x 32, k-slot order) (one of them required); save_hv [n][128] float32; save_bits [8][n][8] required; save_hvbits (nullable) [n][4]:
 *     the signs of the views layer's output, word f >> 5, bit order as save_bits; save_x0h / save_d0h (nullable) [n][64] fp16: the encoded position /
 *     direction exactly as the kernel holds them (32 x value, natural order, zero beyond the encoding; the direction's column 63 holds 1 -- what
 *     nm_pe_encode16 would give, without the extra launches).
 *   nm_mlp_backward_chain16: from d_feat / d_raw as nm_mlp_backward_chain's second form; dz16 [8][n][256] = fp16 of s x dZ_7 .. dZ_0 in k-slot order,
 *     dfeat16 (nullable) [n][256] = s x d_feat likewise, s = the power of two that puts *amax into [2, 4) (amax: a device scalar >= the largest magnitude
 *     entering the chain, nm_absmax; values beyond fp16's range saturate); dz32_layer5 / dz32_layer0 (nullable): float32 copies, natural order, of the
 *     two layers whose input-gradient products still want one; bias_grads [8][256] from the unrounded values.
 *   nm_mlp_backward_net16: the WHOLE backward-data pass of the net from d_raw [n][4] in one kernel -- autograd's adjoint of models/vanilla.py:133-145 too:
 *         d_hv = (d_rgb W_rgb) * (hv > 0)      d_feat = d_hv W_views[:, :256]      dZ_7 = (d_feat W_feature + d sigma w_alpha) * (H_7 > 0)  ...
 *     with the tile's d_hv and d_feat kept on chip.  amax >= max |d_raw| (the scale has 2^13 of headroom for growth through the layers).  Outputs as
 *     above plus dhv16 [n][128] (fp16 of s x d_hv, k-slot order of a 128-wide row: the views layer's weight-gradient operand), dhv32 (nullable) [n][128]
 *     float32 natural order (for the view-direction gradient); bias_grads [9][256]: rows 0..7 = layers 7..0, row 8 = feature_linear's.
 *     d_feat_add (nullable) [n][256] float32: a gradient that reaches feature_linear's output from elsewhere -- a second evaluation of the views head on the
 *     same features with other view directions (human_nerf_trainer.py:280-290 asks the net about the SAME points twice) -- added to the kernel's d_feat.
 *   Workspace of nm_mlp_backward_chain_workspace_floats(n) floats each.  The consumers: nm_wgrad16, nm_wgrad_alpha16 below (section "training"). */
int nm_mlp_forward_save16(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, uint16_t* save_h16, float* save_feat, uint16_t* save_feat16,
                          float* save_hv, uint32_t* save_bits, uint32_t* save_hvbits, uint16_t* save_x0h, uint16_t* save_d0h, float* out,
                          nm_stream_t stream);
int nm_mlp_backward_chain16(nm_mlp_t mlp, const float* const* dev_params, const float* d_feat, const float* d_raw, const uint32_t* relu_bits,
                            int64_t n, const float* amax, uint16_t* dz16, uint16_t* dfeat16, float* dz32_layer5, float* dz32_layer0,
                            float* bias_grads, float* workspace, int64_t workspace_floats, nm_stream_t stream);
/* ... of the plain-head net (use_viewdirs=False: output_linear [4][256] straight off layer 7; nm_mlp_forward_save16 then keeps save_h16 / save_bits /
 * save_x0h only and dev_params are 18 tensors: the trunk's 16, output_linear's weight padded to [4][256] and bias [4]): dZ_7 = (d_out W_out) * (H_7 > 0)
 * from d_out [n][4], then the trunk; dz16 [8][n][256], bias_grads [8][256] = layers 7 .. 0.  What the offset nets of the human trainer run on
 * (models/vanilla.py:169-205, 3 outputs: a zero fourth row) once their constant time input is folded into the biases (neuman_hip/train.py). */
int nm_mlp_backward_plain16(nm_mlp_t mlp, const float* const* dev_params, const float* d_out, const uint32_t* relu_bits, int64_t n, const float* amax,
                            uint16_t* dz16, float* dz32_layer5, float* dz32_layer0, float* bias_grads, float* workspace, int64_t workspace_floats,
                            nm_stream_t stream);
int nm_mlp_backward_net16(nm_mlp_t mlp, const float* const* dev_params, const float* d_raw, const float* d_feat_add, const uint32_t* relu_bits,
                          const uint32_t* hv_bits, int64_t n, const float* amax, uint16_t* dz16, uint16_t* dfeat16, uint16_t* dhv16, float* dz32_layer5, float* dz32_layer0,
                          float* dhv32, float* bias_grads, float* workspace, int64_t workspace_floats, nm_stream_t stream);
/* Same with ray_to_samples' point construction fused: sample (r,s) is at origin[r] + direction[r]*z[r,s]
 * with view direction direction[r] (ray_utils.py:131-132); out [R,S,4]. */
int nm_mlp_forward_rays(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals,
                        int64_t R, int S, int precision, float sigma_scale, float* out, nm_stream_t stream);
/* Density only, for a pass whose colours the caller discards -- the coarse pass of a two-pass render: the reference
 * composites it (render_utils.py:139) and keeps nothing but the weights that place the importance samples (:141), which
 * depend on sigma alone.  Same arguments as nm_mlp_forward_rays; out [R,S,4] receives (0, 0, 0, sigma * sigma_scale) with
 * sigma BIT-IDENTICAL to nm_mlp_forward_rays' (same instruction sequence for the alpha row); feature_linear,
 * views_linears and rgb_linear -- 17 % of the MACs -- are not evaluated.  NM_PREC_BF16X3 / NM_PREC_BF16; the other
 * precisions evaluate everything and return the colours too. */
int nm_mlp_sigma_rays(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals,
                      int64_t R, int S, int precision, float sigma_scale, float* out, nm_stream_t stream);
/* Front-to-back marching with early ray termination (the north star's "wavefront ballot / prefix-sum for early termination
 * and sample compaction ... MLP over the compacted (rays x samples) batch"; the reference evaluates every sample,
 * utils/render_utils.py:139-151).  One chunk of the pass: samples s0 .. s0+S-1 of the rays listed in ray_idx [n_rays]
 * (int32, compacted list of live rays; when n_rays_dev is non-null the list's length is read from the device and n_rays
 * is only its upper bound, so no host synchronisation separates the chunks).  origin / direction [R,3], z_vals
 * [R,S_total], out [R,S_total,4]: only the listed rays' records of this chunk are written (pre-zeroed records of samples
 * never evaluated composite with weight exactly 0).  NM_PREC_FP32 is not available in this form. */
int nm_mlp_forward_ray_chunk(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int S_total,
                             const int32_t* ray_idx, const int32_t* n_rays_dev, int64_t n_rays, int s0, int S, int precision,
                             float sigma_scale, float* out, nm_stream_t stream);
/* The same chunk, density only (nm_mlp_sigma_rays' arithmetic: sigma bit-identical, colours 0) -- the march of a COARSE pass, whose
 * colours the reference composites and discards (render_utils.py:139-141). */
int nm_mlp_sigma_ray_chunk(nm_mlp_t mlp, const float* origin, const float* direction, const float* z_vals, int S_total,
                           const int32_t* ray_idx, const int32_t* n_rays_dev, int64_t n_rays, int s0, int S, int precision,
                           float sigma_scale, float* out, nm_stream_t stream);
/* T[r] *= prod_{i in chunk} (1 - alpha_i + 1e-10) for the listed rays (ray_idx nullable = rays 0..n_rays-1): the
 * transmittance factors of raw2outputs (render_utils.py:85-95) over samples s0 .. s0+S-1 of raw [R,S_total,4]; rays whose T
 * falls below the caller's epsilon are dropped by nm_compact_hits(eps, T). */
int nm_transmittance_chunk(const float* raw, const float* z_vals, const float* rays_d, const int32_t* ray_idx, const int32_t* n_rays_dev,
                           int64_t n_rays, int s0, int S, int S_total, float* T, nm_stream_t stream);
/* The same with the samples' z INTERVALS given (dz [R,S_total], before the multiplication by |rays_d|): a list that will be merged
 * with other lists before it is composited (render_utils.py:330-345, 441-456) -- the interval behind a sample then ends at its
 * successor in the merged order, and the transmittance the early-termination cut is decided on is the merged list's. */
int nm_transmittance_chunk_dz(const float* raw, const float* dz, const float* rays_d, const int32_t* ray_idx, const int32_t* n_rays_dev,
                              int64_t n_rays, int s0, int S, int S_total, float* T, nm_stream_t stream);
/* Debug: the density-only NM_PREC_FP16X3 activation-stationary kernel (csrc/mlp_f16t.hip) on n points (out [n,4] = (0, 0, 0, sigma)), plus the
 * activations of its first tile after `stage` (0..7): state [128 samples][256] float32 in natural feature order (hi + lo parts, unscaled). */
int nm_mlp_sigma_f16t_debug(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int stage, float* state, float* out, nm_stream_t stream);
/* Debug: stop after `stage` and write that stage's activations as f32 [n, width_of_stage]:
 *   -1 -> position PE (64 wide, col 63 = 0);  0..7 -> relu(pts_linears[i]) (256);
 *    8 -> feature_linear output (256);  9 -> relu(views_linears[0]) (128). */
int nm_mlp_forward_debug(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision,
                         int stage, float* hidden, nm_stream_t stream);
/* Profiling build of the NM_PREC_BF16X3 or NM_PREC_I8X3 kernel: same result in `out`, plus per-wave s_memtime
 * totals in cycles[(workgroup*8 + wave)*8 + bucket]; cycles must hold 64 * min(#CUs, ceil(n/128)) uint64.
 *   bf16x3 buckets: {0 PE, 1 k-loops, 2 wait before epilogue, 3 epilogue, 4 wait after epilogue, 5 tile tail}
 *   i8x3 buckets:   {0 PE fill, 1 k-loops, 2 end barrier of an M slot, 3 epilogue part 1, 4 its middle barrier,
 *                    5 epilogue part 2, 6 end barrier of an E slot, 7 rest}  (csrc/mlp.hip) */
int nm_mlp_forward_profile(nm_mlp_t mlp, const float* pts, const float* dirs, int64_t n, int precision, float* out,
                           uint64_t* cycles, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a11  warp_samples_to_canonical -- reference utils/ray_utils.py:48-66
 *   closest point on the posed mesh (replaces igl.point_mesh_squared_distance, :53), barycentrics
 *   (:55), blended per-vertex transform T (f64, [>=V,4,4]), 4x4 inverse, canonical point, and
 *   finite-difference canonical directions along each ray (:62-64).
 *
 *   nm_mesh_create builds, once per posed mesh (per frame and actor), an exact search structure on the
 *   device: triangle records in Morton order and an implicit 4-ary tree of child boxes over them
 *   (NM_SEARCH_TREE).  verts [V,3] f32 and faces [F,3] int32 are DEVICE pointers and are copied into the
 *   handle; F <= 2^24.  NM_SEARCH_ALL makes the query loop over every triangle instead (diagnostics and
 *   tests: both modes return bit-identical results, ties between equidistant triangles going to the lowest
 *   face id).  The call synchronises the stream once (non-finite vertices are an error).
 *   nm_mesh_update: the SAME topology with moved vertices (verts [V,3], device) -- what a training iteration has (the posed SMPL mesh of
 *   human_nerf_trainer.py:262-271 after every optimiser step): the tree, and on the next nm_signed_distance the normals, are rebuilt into the handle's
 *   buffers; no allocation, no read-back, no synchronisation (a non-finite vertex is NOT reported here: the queries then fall back to the
 *   all-triangles loop point by point).
 *   nm_warp_to_canonical: pts [R,S,3] f32, T [*,16] f64 -> can_pts, can_dirs [R,S,3] f32,
 *   closest [R,S,3] f32 (optional).
 * ------------------------------------------------------------------------------------------- */
#define NM_SEARCH_TREE 0
#define NM_SEARCH_ALL 1
#define NM_SEARCH_TREE_WIDE 2 /* the tree search with the stack layout of meshes beyond 65,536 nodes (tests) */
typedef struct nm_mesh_s* nm_mesh_t;
int nm_mesh_create(const float* verts, int V, const int32_t* faces, int F, int search, nm_mesh_t* out,
                   nm_stream_t stream);
int nm_mesh_update(nm_mesh_t mesh, const float* verts, nm_stream_t stream);
int nm_mesh_destroy(nm_mesh_t mesh);
/* tree levels, node count and device bytes of a built mesh (diagnostics) */
int nm_mesh_info(nm_mesh_t mesh, int32_t* levels, int64_t* nodes, int64_t* bytes);
int nm_warp_to_canonical(nm_mesh_t mesh, const float* pts, int64_t R, int S, const double* T, float* can_pts,
                         float* can_dirs, float* closest, nm_stream_t stream);
/* igl.signed_distance(P, V, F) -- reference utils/ray_utils.py:70, trainers/human_nerf_trainer.py:310, 326: for pts [N,3]
 * the distance to the mesh signed by the angle-weighted pseudonormal of the closest feature (negative inside a closed,
 * outward-oriented mesh), the closest face (the caller's id; lowest id on ties) and the closest point.  The pseudonormal
 * table is built by the first call on a mesh (synchronises once). */
int nm_signed_distance(nm_mesh_t mesh, const float* pts, int64_t N, float* sdist, int32_t* face, float* closest,
                       nm_stream_t stream);

/* The differentiable warp of the human trainer (utils/ray_utils.py:85-93 + trainers/human_nerf_trainer.py:262-266), per sample i of N:
 *   can[i] = inv( sum_k bary[i][k] T[tri[i][k]] ) [pts[i]; 1]     T [V,4,4] f32, tri [N,3] int32 vertex ids, bary / pts / can [N,3] f32
 * forward and backward as one kernel each.  backward: g_can [N,3] -> g_T [V,4,4] (cleared here, then float atomics: the summation
 * order over a vertex's samples is not fixed) and g_bary [N,3]; the points carry no gradient (the trainer detaches them). */
int nm_warp_apply_forward(const float* T, const int32_t* tri, const float* bary, const float* pts, int64_t N, float* can, nm_stream_t stream);
int nm_warp_apply_backward(const float* T, const int32_t* tri, const float* bary, const float* pts, const float* g_can, int64_t N, int64_t V,
                           float* g_T, float* g_bary, nm_stream_t stream);
/* The barycentric coordinates the warp blends with (utils/ray_utils.py:72-84), per sample i of N, of the closest point closest[i] (a constant:
 * the reference gets it from igl as numpy) in the triangle tri[i] of verts [V,3]:  bary [N,3] = (u, v, 1 - u - v), the reference's float32
 * cross / dot / divide sequence -- and their adjoint, g_bary [N,3] -> g_verts [V,3] (cleared here, then float atomics), which is what carries
 * the loss to the SMPL parameters through the posed vertices (trainers/human_nerf_trainer.py:241-278). */
int nm_bary_forward(const float* verts, const int32_t* tri, const float* closest, int64_t N, float* bary, nm_stream_t stream);
int nm_bary_backward(const float* verts, const int32_t* tri, const float* closest, const float* g_bary, int64_t N, int64_t V, float* g_verts,
                     nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a12  SMPL linear blend skinning, batched over frames -- reference models/smpl.py:266-360 (lbs),
 *   :407-438 (batch_rodrigues), :454-505 (batch_rigid_transform), :109-216 (SMPL.verts_transformations /
 *   forward); callers data_io/neuman_helper.py:288-330 (read_smpls) and models/human_nerf.py:92-122
 *   (HumanNeRF.vertex_forward).
 *
 *   nm_smpl_create copies the model to the device.  HOST pointers, float32 as the reference registers them
 *   (smpl.py:74-107): v_template [V,3], shapedirs [V,3,NB], j_regressor [J,V], parents [J] (parents[0]
 *   ignored, parents[j] < j), lbs_weights [V,J], da_pose [J*3] (the canonical pose: neuman_helper.py:294-299).
 *   J <= 64, NB <= 32.
 *   nm_smpl_frames, per frame b of B (DEVICE pointers): poses [B,J*3] f32, betas [B,NB] f32,
 *   alignments [B,4,4] f64 = the matrix whose TRANSPOSE the reference applies (read_smpls' temp_alignment,
 *   vertex_forward's alignments[idx]).  Rows 0..V-1 are vertices, rows V..V+J-1 the joints
 *   (concat_joints=True):
 *     T_out      [B,V+J,4,4] f64  T_da2scene = S(scale) align^T T_t2pose inv(T_t2da)
 *     world_out  [B,V+J,3]   f32  T_da2scene [da-pose point; 1]   (world_verts | joints_3d)
 *     static_out [B,V+J,3]   f32  the da-pose points              (static_vert | static_joints_3d)
 *   precise = 1: read_smpls' arithmetic (float32 up to T_da2pose, float64 after); precise = 0:
 *   vertex_forward's (float32 throughout; T_out then holds float32 values).
 *   The first call with a larger B than any before allocates workspace (synchronises).
 * ------------------------------------------------------------------------------------------- */
typedef struct nm_smpl_s* nm_smpl_t;
int nm_smpl_create(const float* v_template, const float* shapedirs, const float* j_regressor, const int32_t* parents,
                   const float* lbs_weights, const float* da_pose, int V, int J, int NB, nm_smpl_t* out);
int nm_smpl_destroy(nm_smpl_t smpl);
/* The differentiable form of ONE frame -- HumanNeRF.vertex_forward under autograd in the human trainer (models/human_nerf.py:92-122
 * over models/smpl.py:266-360; SURVEY 8f-1: gradients of the loss with respect to the frame's pose, shape and alignment), float32
 * throughout like the reference.  DEVICE pointers: pose [J*3], beta [NB] f32, alignment [4,4] f64 (its TRANSPOSE is applied),
 * da_pose [J*3] f32 or null (= the handle's).  workspace: nm_smpl_vertex_workspace_floats() floats.
 *   forward:  world_out [V,3], T_out [V,4,4] f32 (T_da2scene of the vertices; two launches)
 *   backward: upstream g_world [V,3] and / or g_T [V,4,4] (null = zero) -> g_pose [J*3], g_beta [NB], g_align [4,4] (gradient with
 *             respect to `alignment` as passed).  Six launches, every reduction in a fixed order (no float atomics): run to run
 *             bit-identical.  Nothing is kept between the two calls: backward recomputes the joints. */
int64_t nm_smpl_vertex_workspace_floats(nm_smpl_t smpl);
int nm_smpl_vertex_forward(nm_smpl_t smpl, const float* pose, const float* beta, const double* alignment, double scale, const float* da_pose,
                           float* workspace, float* world_out, float* T_out, nm_stream_t stream);
int nm_smpl_vertex_backward(nm_smpl_t smpl, const float* pose, const float* beta, const double* alignment, double scale, const float* da_pose,
                            const float* g_world, const float* g_T, float* workspace, float* g_pose, float* g_beta, float* g_align,
                            nm_stream_t stream);
int nm_smpl_frames(nm_smpl_t smpl, const float* poses, const float* betas, const double* alignments, int B, double scale,
                   int precise, double* T_out, float* world_out, float* static_out, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused per-ray passes (nm_render_rays_bkg / nm_render_rays_human / merge + composite): ONE call per pass of the
 * reference's renderers -- utils/render_utils.py:131-151 / 287-297 (two-pass background), :213-229 / 320-329 (human pass of
 * compacted hit rays), :330-345 / 441-456 (merge + composite) -- the pass's kernels enqueued back to back on `stream`: no host
 * synchronisation inside, no hidden allocation (intermediates in the caller's `workspace`, sized by the *_workspace_floats
 * queries), the same kernels as the step-by-step entry points and therefore the same bits.
 *
 *   nm_render_rays_bkg   origin / direction [R,3], near / far [R], t_vals [S] and u [N] = the caller's torch.linspace(0,1,.);
 *       fine == NULL and N == 0: one pass (S samples); else coarse density pass -> compositing weights -> importance samples ->
 *       fine pass on S + N samples.  raw_out [R,S+N,4] and z_out [R,S+N] always; rgb [R,3] / depth [R] / acc [R] when rgb != NULL
 *       (hybrid renderers composite later, after merging).  precision_*: NM_PREC_* of each pass.
 *   nm_render_rays_human   already compacted hit rays with their per-ray near / far; mesh == NULL (and T == NULL): canonical
 *       render (camera-ray directions); else ray_to_samples -> nm_warp_to_canonical(mesh, T) -> network on the warped points and
 *       finite-difference directions.  sigma_scale = interval_comp (:229).
 *   nm_merge_composite   nm_merge_sorted + nm_composite of the merged list (disp discarded).
 * ------------------------------------------------------------------------------------------- */
int64_t nm_render_rays_bkg_workspace_floats(int64_t R, int S, int N);
int nm_render_rays_bkg(nm_mlp_t coarse, nm_mlp_t fine, const float* origin, const float* direction, const float* near, const float* far,
                       int64_t R, int S, int N, const float* t_vals, const float* u, int white_bkg, int precision_coarse, int precision_fine,
                       float* workspace, float* raw_out, float* z_out, float* rgb, float* depth, float* acc, nm_stream_t stream);
int64_t nm_render_rays_human_workspace_floats(int64_t R, int S, int posed);
int nm_render_rays_human(nm_mlp_t human, nm_mesh_t mesh, const double* T, const float* origin, const float* direction, const float* near,
                         const float* far, int64_t R, int S, const float* t_vals, int white_bkg, float sigma_scale, int precision,
                         float* workspace, float* raw_out, float* z_out, float* rgb, float* depth, float* acc, nm_stream_t stream);
/* render_hybrid_nerf's per-batch body (utils/render_utils.py:287-353; SURVEY 8b nm_render_rays_hybrid) as one call: two-pass background
 * of every ray (scalar bkg_near / bkg_far) and its composite, near / far against the posed body `verts` [V,3] (geo_threshold), compaction
 * of the hit rays (ONE host read: their count), human pass of the hit rays through `mesh` / `T`, merged composite and the human-only
 * accumulation scattered back -> rgb [R,3], depth [R], acc [R] (0 where the body is missed).  t_vals [S], u [N], t_vals_human
 * [S_human] = the caller's torch.linspace(0, 1, .).  Same kernels, same bits as the separate calls. */
int64_t nm_render_rays_hybrid_workspace_floats(int64_t R, int S, int N, int S_human);
int nm_render_rays_hybrid(nm_mlp_t coarse, nm_mlp_t fine, nm_mlp_t human, nm_mesh_t mesh, const double* T, const float* verts, int V,
                          double geo_threshold, const float* origin, const float* direction, int64_t R, float bkg_near, float bkg_far, int S, int N,
                          int S_human, const float* t_vals, const float* u, const float* t_vals_human, int white_bkg, int precision_coarse,
                          int precision_fine, int precision_human, float* workspace, float* rgb, float* depth, float* acc, nm_stream_t stream);
/* ONE KERNEL for the merge + composite tail of the hybrid renderers (utils/render_utils.py:330-345, 441-456): k <= 4 sorted lists per
 * ray (z[l] [.,S[l]], raw[l] [.,S[l],4]; rows[l] nullable: list l's arrays are indexed by rows[l][ray] -- e.g. the background arrays of
 * ALL rays read in place for the hit rays) -> the merged order (ties: the earlier list first, exactly what nm_merge_sorted applied list
 * by list gives) -> raw2outputs' sums.  The merged list lives in LDS only; bit-identical to nm_merge_sorted (+ ...) + nm_composite.
 * z / raw / rows / S are HOST arrays of k entries. */
int nm_merge_composite_lists(int k, const float* const* z, const float* const* raw, const int32_t* const* rows, const int* S, int64_t R,
                             const float* rays_d, int white_bkg, float* rgb, float* depth, float* acc, nm_stream_t stream);
/* ONE KERNEL for the coarse tail of a two-pass render (utils/render_utils.py:139-147; ray_utils.py:138-194): the compositing weights of
 * raw [R,S,4] (raw2outputs), their inverse-CDF samples at u [N] and the sorted merge with z_vals -> z_out [R,S+N]; sigma is read once,
 * the weights are written only when weights_out [R,S] != NULL.  Bit-identical to nm_composite + nm_importance_z. */
int nm_importance_from_raw(const float* raw, const float* z_vals, const float* rays_d, int64_t R, int S, const float* u, int N, float* z_out,
                           float* weights_out, nm_stream_t stream);
/* The intervals the samples of k <= 4 sorted lists per ray (z[l] [R,S[l]]) will be composited with ONCE MERGED: for every sample the distance
 * to its successor in the stable merged order (ties: the earlier list first -- nm_merge_sorted's order, the reference's sort(cat(...)),
 * utils/render_utils.py:330-337, 441-448), 1e10 for the last sample of the merged list (:86) -> dz[l] [R,S[l]].  What an early-termination
 * cut is decided on before the lists are merged.  z / S / dz are HOST arrays of k entries. */
int nm_merged_intervals(int k, const float* const* z, const int* S, int64_t R, float* const* dz, nm_stream_t stream);
int64_t nm_merge_composite_workspace_floats(int64_t R, int Sa, int Sb);
int nm_merge_composite(const float* za, const float* rawa, int Sa, const float* zb, const float* rawb, int Sb, int64_t R, const float* rays_d,
                       int white_bkg, float* workspace, float* rgb, float* depth, float* acc, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a13  sorted merge of sample lists -- reference utils/render_utils.py:330-337, 441-448
 *   Two lists per ray, each already sorted in z: (za [R,Sa], rawa [R,Sa,4]) and (zb, rawb).
 *   Writes z_out [R,Sa+Sb] sorted and raw_out [R,Sa+Sb,4] gathered in the same order
 *   (== torch.sort(cat(z)) + the three-index gather).  Apply repeatedly for k lists.
 * ------------------------------------------------------------------------------------------- */
int nm_merge_sorted(const float* za, const float* rawa, int Sa, const float* zb, const float* rawb, int Sb,
                    int64_t R, float* z_out, float* raw_out, nm_stream_t stream);

/* row gather / scatter of per-ray records (boolean-mask indexing of render_utils.py:206-212, 231-233):
 *   dst[i, :] = src[idx[i], :]  for i < n (gather)   |   dst[idx[i], :] = src[i, :] (scatter)
 * n is read from the device (counts pointer) so the host never synchronises on the hit count. */
int nm_gather_rows(const float* src, const int32_t* idx, const int32_t* n_dev, int64_t n_max, int width,
                   float* dst, nm_stream_t stream);
int nm_scatter_rows(const float* src, const int32_t* idx, const int32_t* n_dev, int64_t n_max, int width,
                    float* dst, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * a1  shot_rays / shot_all_rays on the device -- reference utils/ray_utils.py:23-38 via
 *     geometry/pcd_projector.py:85-120 (pcd_2d_to_pcd_3d) and :209-227 (integer pixel grid, no +0.5)
 *   pixel (x, y) at depth 1 -> K^-1 [x,y,1] -> camera-to-world (4x4) -> / w, all in f64, then
 *     mode 0 (shot_all_rays, :32-38): dir = (p - centre) / |p - centre| in f64, cast to f32
 *     mode 1 (shot_rays, :23-29):     p is cast to f32 first (:25), then the same (numpy promotes to an f64 centre)
 *     mode 2 (shot_rays, f32 pose):   as the reference's own CameraPose (f32 matrix from f32 quaternions,
 *                                     cameras/camera_pose.py:29-45) makes numpy do it: subtraction, norm and division in f32
 *   xy: device int32 [n,2] pixel coordinates, or NULL = the full grid of `width` columns in
 *   row-major order (render_utils.py:185).  inv_intrinsic (3x3) and cam2world (4x4) are HOST f64
 *   row-major matrices (25 by-value parameters, like the reference's numpy arrays); the centre is
 *   cam2world[:3,3] (cameras/camera_pose.py:94-95).  origin [n,3] (centre repeated), direction [n,3].
 * ------------------------------------------------------------------------------------------- */
int nm_shot_rays(const int32_t* xy, int64_t n, int width, int mode, const double* inv_intrinsic,
                 const double* cam2world, float* origin, float* direction, nm_stream_t stream);

/* The same for a training batch whose rays come from many captures at once -- reference
 *   datasets/background_rays.py:47-101 and datasets/human_rays.py:133-209 call shot_rays once per
 *   capture inside a host loop; here ray i uses camera cam_id[i] of a DEVICE table cams [n_cams][25]
 *   of f64 (K^-1 row-major, then cam2world row-major), so a batch is one launch whatever the number
 *   of captures it touches.  mode 1 / 2 as above (there is no full-grid form).  cam_id out of range
 *   is clamped.  xy device int32 [n,2]; origin, direction [n,3].                                  */
int nm_shot_rays_cams(const int32_t* xy, const int32_t* cam_id, int64_t n, int mode, const double* cams,
                      int n_cams, float* origin, float* direction, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * frame egress -- reference render_test_views.py:83-88, render_360.py:77-81 (imageio.imsave of the
 *   renderer's f32 [H,W,3] frame) and render_test_views.py:35 (PSNR of the uint8 frames).
 *   imageio is a conda dependency (environment.yml:22, unpinned), not vendored: nm_frame_to_uint8
 *   restates its published float -> uint8 rule (v2 image_as_uint: clip to [0,1], x*255 + 0.499999999,
 *   truncate).  nm_ssd_u8 returns the exact integer sum of squared differences of two uint8 arrays;
 *   PSNR = 10 log10(255^2 * n / ssd) is skimage.metrics.peak_signal_noise_ratio (environment.yml:30)
 *   for uint8 inputs.  ssd: device uint64[1].
 * ------------------------------------------------------------------------------------------- */
int nm_frame_to_uint8(const float* src, int64_t n, uint8_t* dst, nm_stream_t stream);
int nm_ssd_u8(const uint8_t* a, const uint8_t* b, int64_t n, uint64_t* ssd, nm_stream_t stream);
/* skimage.metrics.structural_similarity(pred, gt, multichannel=True) as render_test_views.py:33 calls it, for uint8 [H,W,C]
 * images: 7x7 uniform window, K1 0.01, K2 0.03, data range 255, sample covariance, mean over the image cropped by 3 pixels
 * and over channels.  ssim: device double[1]; workspace: device double[NM_SSIM_WORKSPACE_DOUBLES]. */
#define NM_SSIM_WORKSPACE_DOUBLES 4096
int nm_ssim_u8(const uint8_t* a, const uint8_t* b, int H, int W, int C, double* ssim, double* workspace, nm_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * SURVEY 8f-1 (first slice): device primitives of a training step of the background NeRF --
 *   reference trainers/vanilla_nerf_trainer.py:45-96 (forward with grad) + torch autograd's backward
 *   of models/vanilla.py:120-152 and utils/render_utils.py:69-105.  The layer loop is host code
 *   (neuman_hip/train.py), as the reference's is Python.
 *
 *   nm_gemm_f32: C[M,N] = op(A) op(B) in float32 on the f32 MFMA (true f32 products and accumulation);
 *   nm_gemm_bf16x3 below is the faster, slightly less exact form (NEUMAN_TRAIN_GEMM=bf16x3).
 *     a_kmajor = 0: A is an [M,K] row-major array (lda);  1: A is stored [K,M] (its transpose is multiplied)
 *     b_kmajor = 1: B is a [K,N] row-major array (ldb);   0: B is stored [N,K] (nn.Linear's weight layout)
 *     forward      Z  = A  W^T      (0, 0)        backward-data     dA = dZ W      (0, 1)
 *     backward-weights  dW = dZ^T A  (1, 1): split over K with a deterministic second pass when M*N is
 *     small and K large; needs nm_gemm_workspace_floats(M,N,K) floats of workspace and allows no flag
 *     but ACCUMULATE.
 *     flags: ACCUMULATE  C += ...;  BIAS  + bias[col];  RELU  max(.,0);  MASK  zero where mask[row,col] <= 0
 *     (applied in that order).  M, N, K, lda, ldb multiples of 4 (pad with zeros), A and B 16-byte aligned.
 *   nm_pe_encode: models/vanilla.py:60-92 stand-alone: x [n,dims] (dims 3, or 4 = point + time for the offset net,
 *     vanilla.py:180-188) -> out [n,ld], dims (1 + 2 n_freqs) features then
 *     zeros; table = the n_freqs bands (posenc) or the [3 n_freqs, 3] projection (rotate), device f32.
 *   nm_composite_backward: d loss / d raw [R,S,4] through raw2outputs given the gradients of rgb_map [R,3],
 *     acc_map [R], depth_map [R], weights [R,S] (each nullable = zero); disp_map's gradient is not supported.
 * ------------------------------------------------------------------------------------------- */
#define NM_GEMM_ACCUMULATE 1
#define NM_GEMM_BIAS 2
#define NM_GEMM_RELU 4
#define NM_GEMM_MASK 8
#define NM_GEMM_COLSUM 16   /* also write the column sums of the stored output, per band of 64 rows, to workspace [ceil(M/64)][N]
                              (deterministic; the bias gradient of the layer below is the sum of the bands: nm_colsum on them).  Not
                              with a split-K product; needs the 16-byte aligned output layout (ldc % 4 == 0) */
int64_t nm_gemm_workspace_floats(int M, int N, int K);
int nm_gemm_f32(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                int ldc, const float* bias, const float* mask, int ldmask, int flags, float* workspace,
                int64_t workspace_floats, nm_stream_t stream);
/* the same product with each float32 operand split into bf16 hi + lo and hi*hi + hi*lo + lo*hi accumulated in float32 on the
 * bf16 MFMA (2^-17 relative per product; what the rendering kernels call bf16x3).  Same arguments, same rules. */
int nm_gemm_bf16x3(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc, const float* bias, const float* mask, int ldmask, int flags, float* workspace,
                   int64_t workspace_floats, nm_stream_t stream);
/* and split into fp16 hi + lo on the fp16 MFMA (11 + 11 significand bits: float32 class, what the rendering kernels call fp16x3).
 * Operands are taken as they are (no scaling): parts below fp16's normal range lose at most 2^-25 absolutely, magnitudes beyond
 * 65504 saturate -- meant for the forward products (activations and weights of O(1)); gradients, whose magnitudes are unbounded
 * below, belong on nm_gemm_bf16x3 or nm_gemm_f32.  Same arguments, same rules. */
int nm_gemm_fp16x3(int a_kmajor, int b_kmajor, int M, int N, int K, const float* A, int lda, const float* B, int ldb, float* C,
                   int ldc, const float* bias, const float* mask, int ldmask, int flags, float* workspace,
                   int64_t workspace_floats, nm_stream_t stream);
int nm_pe_encode(const float* x, int64_t n, int dims, int kind, int n_freqs, const float* table, float* out, int ld,
                 nm_stream_t stream);
/* nm_pe_encode with the output as fp16 of 32 x value (an operand of nm_wgrad16; ld a multiple of 8, out 16-byte aligned).  ones_col >= 0 (a padding
 * column of the row): that column holds 1 (x 32) -- the product with it is the column sum of the other operand, i.e. the layer's bias gradient */
int nm_pe_encode16(const float* x, int64_t n, int dims, int kind, int n_freqs, const float* table, uint16_t* out, int ld, int ones_col,
                   nm_stream_t stream);
/* *out_max = max(*out_max, max |x[i]|), i < count: device scalar, updated atomically (zero it first); x 16-byte aligned */
int nm_absmax(const float* x, int64_t count, float* out_max, nm_stream_t stream);
/* Backward-weights from 16-bit operands, nprod <= 8 products of one n in ONE launch (+ one deterministic reduction):
 *     dW_p [p_cols][q_cols] = (1 / (32 s)) sum_n dz16_p[n][:]^T act16_p[n][:]          (s from *amax as in nm_mlp_backward_chain16)
 * dz16_p [n][p_cols] fp16 in k-slot order, p_cols = 256 (a trunk layer, feature_linear) or 128 (the views layer: dhv16); q_cols == 256: act16_p
 * [n][256] fp16 in k-slot order (nm_mlp_forward_save16's save_h16[l] / save_feat16); q_cols <= 64: act16_p [n][64] fp16 in natural order
 * (nm_pe_encode16, zero beyond the encoding).  dW_p in the natural [out][in] order of nn.Linear, row stride ldw[p] (so the hidden columns of the skip
 * layer land inside its [256][319] gradient).  One fp16 MFMA per product, 1 KB per sample and 256 x 256 product.  dz16 / act16 / dW / ldw: HOST arrays
 * of nprod entries. */
int64_t nm_wgrad16_workspace_floats(int nprod, int64_t n, int p_cols, int q_cols);
int nm_wgrad16(int nprod, int p_cols, int q_cols, const uint16_t* const* dz16, const uint16_t* const* act16, float* const* dW, const int* ldw,
               int64_t n, const float* amax, float* workspace, int64_t workspace_floats, nm_stream_t stream);
/* The 4-row heads of a step in one pass over d_raw [n][4], save_h16[7] and save_hv [n][128]: out[0..255] = alpha_linear's weight gradient
 * sum_n d_raw[n][3] H7[n][:], out[256..639] = rgb_linear's [3][128] = sum_n d_raw[n][k] hv[n][:], out[640..643] = the column sums of d_raw (rgb_linear's and
 * alpha_linear's bias gradients); *amax (nullable, a zeroed device scalar) = max |d_raw|: the scale source of nm_mlp_backward_net16 from the same read */
int64_t nm_wgrad_heads16_workspace_floats(int64_t n);
int nm_wgrad_heads16(const float* d_raw, const uint16_t* h16_7, const float* hv, int64_t n, float* out644, float* amax, float* workspace,
                     int64_t workspace_floats, nm_stream_t stream);
/* ... and of the plain-head net: out[0..1023] = output_linear's [4][256] = sum_n d_out[n][k] H7[n][:], out[1024..1027] = the column sums of d_out; amax as above */
int64_t nm_wgrad_out16_workspace_floats(int64_t n);
int nm_wgrad_out16(const float* d_out, const uint16_t* h16_7, int64_t n, float* out1028, float* amax, float* workspace, int64_t workspace_floats,
                   nm_stream_t stream);
/* alpha_linear's weight gradient out[256] = sum_n d_raw[n][3] H7[n][:] from save_h16[7] (x 32, k-slot order) */
int64_t nm_wgrad_alpha16_workspace_floats(int64_t n);
int nm_wgrad_alpha16(const float* d_raw, const uint16_t* h16, int64_t n, float* out, float* workspace, int64_t workspace_floats,
                     nm_stream_t stream);
/* out[W] = column sums of X [n,W] (row stride ld): the bias gradients; bands of 256 rows summed in order (deterministic) */
int64_t nm_colsum_workspace_floats(int64_t n, int W);
int nm_colsum(const float* X, int64_t n, int W, int ld, float* out, float* workspace, int64_t workspace_floats, nm_stream_t stream);
/* adjoint of nm_pe_encode: g [n,ld] = gradient of the encoded features -> dx [n,3] */
int nm_pe_backward(const float* x, int64_t n, int dims, int kind, int n_freqs, const float* table, const float* g, int ld,
                   float* dx, nm_stream_t stream);
int nm_composite_backward(const float* raw, const float* z_vals, const float* rays_d, int64_t R, int S, int white_bkg,
                          const float* g_rgb, const float* g_acc, const float* g_depth, const float* g_weights,
                          float* d_raw, nm_stream_t stream);

/* --------------------------------------------------------------------------------------------
 * The scalar regularisers of the human trainer's loss (trainers/human_nerf_trainer.py:280-380) as single passes: the VALUE of a term and its
 * GRADIENT with respect to the network outputs it reads, in one kernel + one deterministic finalisation (csrc/loss.hip) -- torch's elementwise
 * algebra takes 10-25 launches per term and as many again in autograd's backward pass.  raw arrays are [n][4] = (r, g, b, sigma) rows, 16-byte
 * aligned; `loss` is one device float; workspace: nm_loss_workspace_doubles() doubles.
 *   nm_loss_bimodal: mean_i(-log(e^-|y_i| + e^-|1 - y_i|) + offset), y = clamp(x, 0, 1) when clamp01 (:368-379 with HARD_SURFACE_OFFSET); dx [n]
 *   nm_loss_pair_mse: scale * mse between two raw outputs: mode 0 over sigmoid(rgb) (the colour-range term, :280-290), mode 1 over tanh(relu(sigma))
 *                     (the symmetry term, :292-304); da_raw / db_raw [n][4]
 *   nm_loss_shape: the SMPL shape prior (:305-343): w_smpl * mean_{dist_h < 0}((1 - occ(pred))^2) + w_dummy * (mean_{dist_d < 0}((1 - occ(dummy))^2)
 *                  + mean_{dist_d > 0}(|occ(dummy) * (|dist_d| * outside_factor)^exponent|)), occ = 1 - exp(-relu(sigma)), empty selections count as 1;
 *                  nd = 0: the first term alone; norm3: three device floats of scratch */
int64_t nm_loss_workspace_doubles(void);
int nm_loss_bimodal(const float* x, int64_t n, int clamp01, float offset, float* loss, float* dx, double* workspace, nm_stream_t stream);
int nm_loss_pair_mse(int mode, const float* a_raw, const float* b_raw, int64_t n, float scale, float* loss, float* da_raw, float* db_raw,
                     double* workspace, nm_stream_t stream);
int nm_loss_shape(const float* pred_raw, const float* dist_h, int64_t nh, const float* dummy_raw, const float* dist_d, int64_t nd, float w_smpl,
                  float w_dummy, float outside_factor, float exponent, float* loss, float* d_pred_raw, float* d_dummy_raw, double* workspace,
                  float* norm3, nm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* NEUMAN_HIP_H */
