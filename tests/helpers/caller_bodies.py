"""The loop bodies of the reference's render scripts, written ONCE against a namespace of modules, so that the very same driver code runs
  * in the build container on the REFERENCE's own modules (tests/golden/make_golden_callers.py -> tests/golden/callers.npz), and
  * on the GPU box on stand-in `utils.render_utils` / `utils.ray_utils` / `models.vanilla` modules that neuman_hip.install() has filled
    (tests/test_hip_install_callers.py) -- the reference tree does not exist there (a Python reference cannot travel), so what install()
    rebinds there are empty modules registered under the reference's names; every call below resolves through them.

What the scripts do around these bodies -- read a scene from disk, load a checkpoint, write PNGs -- needs assets that do not exist offline
and is outside the hot path (f3 / f4).  M: namespace with `render_utils` (the module the scripts import as `from utils import
render_utils`); `make_cap(i)`: the capture of frame i (the reference's ResizedPinholeCapture there, the duck-typed capture here)."""
import numpy as np


def canonical_360(M, net, make_cap, n_frames, static_vert, faces, opt, can_bone_mean):
    """render_360.py:52-76 (main_canonical_360's loop): every pose of the 360 path through render_smpl_nerf(render_can=True) -> [n, H, W, 3]"""
    frames = []
    for i in range(n_frames):
        can_cap = make_cap(i)
        out = M.render_utils.render_smpl_nerf(
            net, can_cap, static_vert, faces, Ts=None, rays_per_batch=opt.rays_per_batch, samples_per_ray=opt.samples_per_ray,
            render_can=True, return_mask=False, return_depth=False, interval_comp=opt.geo_threshold / can_bone_mean)
        frames.append(np.asarray(out))
    return np.stack(frames)


def test_views(M, net, make_cap, frame_ids, verts, faces, Ts, opt):
    """render_test_views.py:69-82 (main's loop): every test view through render_hybrid_nerf -> [n, H, W, 3]"""
    frames = []
    for i in frame_ids:
        cap = make_cap(i)
        out = M.render_utils.render_hybrid_nerf(
            net, cap, verts[i], faces, Ts[i], rays_per_batch=opt.rays_per_batch, samples_per_ray=opt.samples_per_ray,
            geo_threshold=opt.geo_threshold, return_depth=False)
        frames.append(np.asarray(out))
    return np.stack(frames)


# the synthetic stand-in for what the scripts read from a scene directory (both sides build it from these definitions)
W360, H360, N360, S360 = 48, 40, 3, 64
WTV, HTV, STV = 40, 32, 64
TV_FRAMES = (0, 1)


def scene_inputs():
    from neuman_hip import synthetic
    verts_c, faces = synthetic.capsule_mesh()
    posed, T = synthetic.twist_transforms(verts_c)
    posed2, T2 = synthetic.twist_transforms(verts_c, twist=0.5, shift=(-0.04, 0.03, 0.02))
    return {'static_vert': verts_c, 'faces': faces, 'verts': [posed, posed2], 'Ts': [T, T2]}
