"""ctypes binding of libneuman_hip.so (the C ABI declared in include/neuman_hip.h).

Thin by design: torch provides device memory and the stream, every compute call goes through the
shared library.  There is no CPU fallback -- a missing library or a non-CUDA tensor raises.
"""
import ctypes
import os

import torch  # imported first so that torch's bundled libamdhip64.so.7 is the HIP runtime the library binds to

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NEUMAN_HIP_LIB") or os.path.join(_HERE, "..", "lib", "libneuman_hip.so")   # (the override serves tools/ A/B builds)

c_f32p = ctypes.c_void_p
c_i32p = ctypes.c_void_p
c_stream = ctypes.c_void_p
i64 = ctypes.c_int64
i32 = ctypes.c_int

NM_PREC_FP32, NM_PREC_BF16X3, NM_PREC_BF16, NM_PREC_I8X3, NM_PREC_FP16X3 = 0, 1, 2, 3, 4
NM_PE_POSENC, NM_PE_ROTATE = 0, 1
PRECISIONS = {"fp32": NM_PREC_FP32, "bf16x3": NM_PREC_BF16X3, "bf16": NM_PREC_BF16, "i8x3": NM_PREC_I8X3, "fp16x3": NM_PREC_FP16X3}


class MlpDesc(ctypes.Structure):
    _fields_ = [("depth", ctypes.c_int32), ("width", ctypes.c_int32), ("skip", ctypes.c_int32),
                ("pe_kind", ctypes.c_int32), ("pos_n_freqs", ctypes.c_int32), ("dir_n_freqs", ctypes.c_int32),
                ("plain_head", ctypes.c_int32)]


# name -> (restype, argtypes); mirrors include/neuman_hip.h one to one (tests/test_abi.py checks the set)
SIGNATURES = {
    "nm_version": (i32, []),
    "nm_last_error": (ctypes.c_char_p, []),
    "nm_device_count": (i32, []),
    "nm_ray_to_samples": (i32, [c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, c_f32p, i32, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_z_to_points": (i32, [c_f32p, c_f32p, c_f32p, i64, i32, c_f32p, c_f32p, c_stream]),
    "nm_composite": (i32, [c_f32p, c_f32p, c_f32p, i64, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_sample_pdf": (i32, [c_f32p, c_f32p, i64, i32, c_f32p, i32, c_f32p, c_stream]),
    "nm_importance_z": (i32, [c_f32p, c_f32p, i64, i32, c_f32p, i32, i32, c_f32p, c_stream]),
    "nm_near_far": (i32, [c_f32p, c_f32p, i64, c_f32p, i32, ctypes.c_double, c_f32p, c_f32p, c_stream]),
    "nm_compact_workspace_ints": (i64, [i64]),
    "nm_compact_hits": (i32, [c_f32p, c_f32p, i64, c_i32p, c_i32p, c_i32p, c_i32p, c_stream]),
    "nm_mlp_pack_bytes": (i64, [ctypes.POINTER(MlpDesc)]),
    "nm_mlp_pack_i8s_bytes": (i64, [ctypes.POINTER(MlpDesc)]),
    "nm_mlp_pack_i8s": (i32, [ctypes.POINTER(MlpDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "nm_mlp_pack": (i32, [ctypes.POINTER(MlpDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "nm_mlp_pack_f16": (i32, [ctypes.POINTER(MlpDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "nm_mlp_pack_i8_bytes": (i64, [ctypes.POINTER(MlpDesc)]),
    "nm_mlp_pack_i8": (i32, [ctypes.POINTER(MlpDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p]),
    "nm_mlp_create": (i32, [ctypes.POINTER(MlpDesc), ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_void_p,
                            ctypes.POINTER(ctypes.c_void_p)]),
    "nm_mlp_destroy": (i32, [ctypes.c_void_p]),
    "nm_mlp_forward": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, i32, ctypes.c_float, c_f32p, c_stream]),
    "nm_mlp_refresh_f16": (i32, [ctypes.c_void_p, ctypes.c_void_p, c_stream]),
    "nm_mlp_forward_save": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_mlp_backward_chain_workspace_floats": (i64, [i64]),
    "nm_mlp_forward_save_bits": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, c_f32p, c_f32p, ctypes.c_void_p, c_f32p, c_stream]),
    "nm_mlp_backward_chain": (i32, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, i64, c_f32p, c_f32p, c_f32p, i64, c_stream]),
    "nm_mlp_forward_save16": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, ctypes.c_void_p, c_f32p, ctypes.c_void_p, c_f32p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_stream]),
    "nm_mlp_backward_plain16": (i32, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, ctypes.c_void_p, i64, c_f32p, ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, i64, c_stream]),
    "nm_mlp_backward_net16": (i32, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_void_p, i64, c_f32p, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, i64, c_stream]),
    "nm_mlp_backward_chain16": (i32, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, i64, c_f32p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p,
                                      c_f32p, c_f32p, i64, c_stream]),
    "nm_mlp_forward_rays": (i32, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, i64, i32, i32, ctypes.c_float, c_f32p, c_stream]),
    "nm_mlp_sigma_rays": (i32, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, i64, i32, i32, ctypes.c_float, c_f32p, c_stream]),
    "nm_mlp_forward_ray_chunk": (i32, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, i32, c_i32p, c_i32p, i64, i32, i32, i32, ctypes.c_float, c_f32p,
                                       c_stream]),
    "nm_mlp_sigma_ray_chunk": (i32, [ctypes.c_void_p, c_f32p, c_f32p, c_f32p, i32, c_i32p, c_i32p, i64, i32, i32, i32, ctypes.c_float, c_f32p,
                                     c_stream]),
    "nm_transmittance_chunk": (i32, [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, i64, i32, i32, i32, c_f32p, c_stream]),
    "nm_transmittance_chunk_dz": (i32, [c_f32p, c_f32p, c_f32p, c_i32p, c_i32p, i64, i32, i32, i32, c_f32p, c_stream]),
    "nm_mlp_sigma_f16t_debug": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, i32, c_f32p, c_f32p, c_stream]),
    "nm_mlp_forward_debug": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, i32, i32, c_f32p, c_stream]),
    "nm_mlp_forward_profile": (i32, [ctypes.c_void_p, c_f32p, c_f32p, i64, i32, c_f32p, ctypes.c_void_p, c_stream]),
    "nm_mesh_create": (i32, [c_f32p, i32, c_i32p, i32, i32, ctypes.POINTER(ctypes.c_void_p), c_stream]),
    "nm_mesh_update": (i32, [ctypes.c_void_p, c_f32p, c_stream]),
    "nm_mesh_destroy": (i32, [ctypes.c_void_p]),
    "nm_mesh_info": (i32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
    "nm_warp_to_canonical": (i32, [ctypes.c_void_p, c_f32p, i64, i32, ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_warp_apply_forward": (i32, [c_f32p, c_i32p, c_f32p, c_f32p, i64, c_f32p, c_stream]),
    "nm_warp_apply_backward": (i32, [c_f32p, c_i32p, c_f32p, c_f32p, c_f32p, i64, i64, c_f32p, c_f32p, c_stream]),
    "nm_bary_forward": (i32, [c_f32p, c_i32p, c_f32p, i64, c_f32p, c_stream]),
    "nm_bary_backward": (i32, [c_f32p, c_i32p, c_f32p, c_f32p, i64, i64, c_f32p, c_stream]),
    "nm_smpl_create": (i32, [c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p, i32, i32, i32, ctypes.POINTER(ctypes.c_void_p)]),
    "nm_smpl_destroy": (i32, [ctypes.c_void_p]),
    "nm_smpl_vertex_workspace_floats": (i64, [ctypes.c_void_p]),
    "nm_smpl_vertex_forward": (i32, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_double, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_smpl_vertex_backward": (i32, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, ctypes.c_double, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                      c_f32p, c_f32p, c_stream]),
    "nm_smpl_frames": (i32, [ctypes.c_void_p, c_f32p, c_f32p, ctypes.c_void_p, i32, ctypes.c_double, i32, ctypes.c_void_p, c_f32p, c_f32p,
                             c_stream]),
    "nm_gemm_workspace_floats": (i64, [i32, i32, i32]),
    "nm_gemm_f32": (i32, [i32, i32, i32, i32, i32, c_f32p, i32, c_f32p, i32, c_f32p, i32, c_f32p, c_f32p, i32, i32, c_f32p, i64, c_stream]),
    "nm_gemm_bf16x3": (i32, [i32, i32, i32, i32, i32, c_f32p, i32, c_f32p, i32, c_f32p, i32, c_f32p, c_f32p, i32, i32, c_f32p, i64, c_stream]),
    "nm_gemm_fp16x3": (i32, [i32, i32, i32, i32, i32, c_f32p, i32, c_f32p, i32, c_f32p, i32, c_f32p, c_f32p, i32, i32, c_f32p, i64, c_stream]),
    "nm_pe_encode": (i32, [c_f32p, i64, i32, i32, i32, c_f32p, c_f32p, i32, c_stream]),
    "nm_pe_encode16": (i32, [c_f32p, i64, i32, i32, i32, c_f32p, ctypes.c_void_p, i32, i32, c_stream]),
    "nm_absmax": (i32, [c_f32p, i64, c_f32p, c_stream]),
    "nm_wgrad16_workspace_floats": (i64, [i32, i64, i32, i32]),
    "nm_wgrad16": (i32, [i32, i32, i32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), i64,
                         c_f32p, c_f32p, i64, c_stream]),
    "nm_wgrad_heads16_workspace_floats": (i64, [i64]),
    "nm_wgrad_heads16": (i32, [c_f32p, ctypes.c_void_p, c_f32p, i64, c_f32p, c_f32p, c_f32p, i64, c_stream]),
    "nm_wgrad_out16_workspace_floats": (i64, [i64]),
    "nm_wgrad_out16": (i32, [c_f32p, ctypes.c_void_p, i64, c_f32p, c_f32p, c_f32p, i64, c_stream]),
    "nm_wgrad_alpha16_workspace_floats": (i64, [i64]),
    "nm_wgrad_alpha16": (i32, [c_f32p, ctypes.c_void_p, i64, c_f32p, c_f32p, i64, c_stream]),
    "nm_colsum_workspace_floats": (i64, [i64, i32]),
    "nm_colsum": (i32, [c_f32p, i64, i32, i32, c_f32p, c_f32p, i64, c_stream]),
    "nm_pe_backward": (i32, [c_f32p, i64, i32, i32, i32, c_f32p, c_f32p, i32, c_f32p, c_stream]),
    "nm_composite_backward": (i32, [c_f32p, c_f32p, c_f32p, i64, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_signed_distance": (i32, [ctypes.c_void_p, c_f32p, i64, c_f32p, c_i32p, c_f32p, c_stream]),
    "nm_ssim_u8": (i32, [ctypes.c_void_p, ctypes.c_void_p, i32, i32, i32, ctypes.c_void_p, ctypes.c_void_p, c_stream]),
    "nm_merge_sorted": (i32, [c_f32p, c_f32p, i32, c_f32p, c_f32p, i32, i64, c_f32p, c_f32p, c_stream]),
    "nm_gather_rows": (i32, [c_f32p, c_i32p, c_i32p, i64, i32, c_f32p, c_stream]),
    "nm_scatter_rows": (i32, [c_f32p, c_i32p, c_i32p, i64, i32, c_f32p, c_stream]),
    "nm_shot_rays": (i32, [c_i32p, i64, i32, i32, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double), c_f32p, c_f32p,
                           c_stream]),
    "nm_shot_rays_cams": (i32, [c_i32p, c_i32p, i64, i32, ctypes.c_void_p, i32, c_f32p, c_f32p, c_stream]),
    "nm_render_rays_bkg_workspace_floats": (i64, [i64, i32, i32]),
    "nm_render_rays_bkg": (i32, [ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, i32, c_f32p, c_f32p, i32, i32, i32, c_f32p, c_f32p,
                                 c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_render_rays_human_workspace_floats": (i64, [i64, i32, i32]),
    "nm_render_rays_human": (i32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, c_f32p, c_f32p, c_f32p, i64, i32, c_f32p, i32, ctypes.c_float, i32,
                                   c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_render_rays_hybrid_workspace_floats": (i64, [i64, i32, i32, i32]),
    "nm_render_rays_hybrid": (i32, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, c_f32p, i32, ctypes.c_double, c_f32p, c_f32p,
                                    i64, ctypes.c_float, ctypes.c_float, i32, i32, i32, c_f32p, c_f32p, c_f32p, i32, i32, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p,
                                    c_stream]),
    "nm_merge_composite_lists": (i32, [i32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p),
                                       ctypes.POINTER(ctypes.c_int), i64, c_f32p, i32, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_merged_intervals": (i32, [i32, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), i64, ctypes.POINTER(ctypes.c_void_p), c_stream]),
    "nm_importance_from_raw": (i32, [c_f32p, c_f32p, c_f32p, i64, i32, c_f32p, i32, c_f32p, c_f32p, c_stream]),
    "nm_merge_composite_workspace_floats": (i64, [i64, i32, i32]),
    "nm_merge_composite": (i32, [c_f32p, c_f32p, i32, c_f32p, c_f32p, i32, i64, c_f32p, i32, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "nm_loss_workspace_doubles": (i64, []),
    "nm_loss_bimodal": (i32, [c_f32p, i64, i32, ctypes.c_float, c_f32p, c_f32p, ctypes.c_void_p, c_stream]),
    "nm_loss_pair_mse": (i32, [i32, c_f32p, c_f32p, i64, ctypes.c_float, c_f32p, c_f32p, c_f32p, ctypes.c_void_p, c_stream]),
    "nm_loss_shape": (i32, [c_f32p, c_f32p, i64, c_f32p, c_f32p, i64, ctypes.c_float, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_f32p, c_f32p, c_f32p,
                            ctypes.c_void_p, c_f32p, c_stream]),
    "nm_frame_to_uint8": (i32, [c_f32p, i64, ctypes.c_void_p, c_stream]),
    "nm_ssd_u8": (i32, [ctypes.c_void_p, ctypes.c_void_p, i64, ctypes.c_void_p, c_stream]),
}

_lib = None


class NeumanHipError(RuntimeError):
    pass


def lib():
    """Load the shared library once.  Raises (loudly) if it has not been built."""
    global _lib
    if _lib is None:
        path = os.path.abspath(LIB_PATH)
        if not os.path.exists(path):
            raise NeumanHipError(f"{path} is missing: run `python ml-neuman_amd/build.py` (there is no CPU fallback)")
        handle = ctypes.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        raise NeumanHipError(f"{what} failed (code {rc}): {lib().nm_last_error().decode(errors='replace')}")


def stream_ptr():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_ptr(t, dtype=torch.float32, name="tensor"):
    """Raw device pointer of a dense CUDA tensor; refuses anything else (no silent host path)."""
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NeumanHipError(f"{name} must be a CUDA (HIP) tensor: libneuman_hip has no CPU path")
    if t.dtype != dtype:
        raise NeumanHipError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise NeumanHipError(f"{name} must be contiguous")
    return ctypes.c_void_p(t.data_ptr())


def require_gpu():
    if not torch.cuda.is_available() or lib().nm_device_count() < 1:
        raise NeumanHipError("no HIP device visible: the NeuMan HIP path cannot run (there is no CPU fallback)")
