"""neuman_hip -- MI355X-native NeuMan ray-march hot path behind the reference's own Python interface.

    import neuman_hip
    neuman_hip.install()        # inside the reference tree: rebinds utils.ray_utils / utils.render_utils /
                                # models.vanilla so train-free scripts (render_360.py, render_test_views.py,
                                # render_reposing.py, render_gathering.py) run on libneuman_hip.so unchanged.

Sub-modules mirror the reference's: ``ray_utils`` (utils/ray_utils.py), ``render_utils`` (utils/render_utils.py:69-461),
``vanilla`` (models/vanilla.py), ``smpl`` (models/smpl.py skinning + data_io/neuman_helper.py:read_smpls, batched on the
device); ``parallel`` adds the ray-tile sharding for 1/2/4/8 GPUs; ``synthetic`` the asset-free workloads.  Around the path
(imported on demand): ``data_io`` + ``scene_content`` (COLMAP / split / checkpoint / per-frame file readers), ``ray_batches``
(datasets/*.py: training batches drawn on the device), ``train`` + ``bkg_trainer`` + ``human_trainer`` (trainers/*.py),
``lpips``.  Nothing in this package evaluates the hot path on the CPU.
"""
from . import _lib, parallel, ray_utils, render_utils, smpl, synthetic, vanilla  # noqa: F401
from ._lib import NeumanHipError  # noqa: F401

__all__ = ["ray_utils", "render_utils", "vanilla", "smpl", "parallel", "synthetic", "install", "NeumanHipError"]

_RAY_FNS = ["shot_ray", "shot_rays", "shot_all_rays", "to_homogeneous", "ray_to_samples", "ray_to_importance_samples",
            "sample_pdf", "geometry_guided_near_far", "geometry_guided_near_far_torch", "geometry_guided_near_far_np",
            "warp_samples_to_canonical", "warp_samples_to_canonical_diff",
            "shot_all_rays_dev", "shot_rays_dev"]                      # additions: a1 on the device (CUDA tensors out)
_RENDER_FNS = ["raw2outputs", "render_vanilla", "render_smpl_nerf", "render_hybrid_nerf", "render_hybrid_nerf_multi_persons",
               "frame_to_uint8", "psnr_uint8", "ssim_uint8", "save_png"]                        # additions: the egress of render_test_views.py:83-92, on the device
_MODEL_CLASSES = ["Embedder", "NeRF", "Joiner", "build_nerf", "OffsetNet", "build_offset_net"]


def install(ref_ray_utils=None, ref_render_utils=None, ref_vanilla=None):
    """Rebind the hot-path names of the (already imported) reference modules to the HIP implementations.

    Call it from the reference checkout before building `HumanNeRF`:
        from utils import ray_utils, render_utils; from models import vanilla
        neuman_hip.install(ray_utils, render_utils, vanilla)
    With no arguments the three modules are imported by their reference names.
    """
    import importlib
    ru = ref_ray_utils or importlib.import_module("utils.ray_utils")
    rr = ref_render_utils or importlib.import_module("utils.render_utils")
    mv = ref_vanilla or importlib.import_module("models.vanilla")
    for n in _RAY_FNS:
        setattr(ru, n, getattr(ray_utils, n))
    for n in _RENDER_FNS:
        setattr(rr, n, getattr(render_utils, n))
    for n in _MODEL_CLASSES:
        setattr(mv, n, getattr(vanilla, n))
    return ru, rr, mv
