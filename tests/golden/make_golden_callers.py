"""Frames made by the REFERENCE through the call sequences of its own render scripts (build container only):

    python tests/golden/make_golden_callers.py   ->  tests/golden/callers.npz

tests/helpers/caller_bodies.py holds the loop bodies of render_360.py:52-76 and render_test_views.py:69-82 written against a namespace of
modules; here that namespace is the reference's own `utils.render_utils` (imported unmodified, igl = tests/golden/igl_shim.py, the other
absent wheels stubbed), the net is the reference's own HumanNeRF(opt) (models/human_nerf.py:21-31) with the synthetic weights loaded, the
captures are the reference's ResizedPinholeCapture / BasePinholeCapture and the 360-degree path is the reference's default_360_path.
The GPU test runs the same bodies through neuman_hip.install() and compares with what is stored here."""
import argparse
import contextlib
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_posed as MP  # noqa: E402  (igl shim + stubs + the reference's modules)
from make_golden_posed import R_render, R_ray, synthetic, PinholeCamera, CameraPose, BasePinholeCapture  # noqa: E402

sys.path.insert(0, os.path.join(MP.ROOT, "tests", "helpers"))
import caller_bodies as CB  # noqa: E402
from cameras.captures import ResizedPinholeCapture  # noqa: E402  (reference)
from models import human_nerf as R_human_nerf  # noqa: E402
from options import options  # noqa: E402
from utils.constant import CANONICAL_ZOOM_FACTOR, CANONICAL_CAMERA_DIST  # noqa: E402


def parse_opt(extra):
    parser = argparse.ArgumentParser()
    for f in (options.set_general_option, options.set_nerf_option, options.set_pe_option, options.set_render_option):
        f(parser)
    parser.add_argument('--offset_scale', type=float, default=1.0)
    parser.add_argument('--num_offset_nets', type=int, default=1)
    parser.add_argument('--offset_scale_type', type=str, default='linear')
    parser.add_argument('--out_dir', type=str, default='./out')
    parser.add_argument('--load_background', type=str, default='none')
    parser.add_argument('--load_can', type=str, default='none')
    parser.add_argument('--posenc', type=str, default='posenc')
    return parser.parse_args(['--use_cuda', 'no', '--can_posenc', 'rotate'] + extra)


@contextlib.contextmanager
def numpy2_array_copy_shim():
    """geometry/transformations.py:1845 calls numpy.array(..., copy=False), which numpy 2 refuses when a copy is needed:
    for the duration of default_360_path, copy=False means numpy.asarray"""
    orig = np.array

    def array(*a, **k):
        if k.get('copy', True) is False:
            k.pop('copy')
            return np.asarray(*a, **k)
        return orig(*a, **k)
    np.array = array
    try:
        yield
    finally:
        np.array = orig


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    out = {}
    inp = CB.scene_inputs()
    M = type('M', (), {'render_utils': R_render})
    # ---- render_360.py main_canonical_360
    opt = parse_opt(['--rays_per_batch', '1024', '--samples_per_ray', str(CB.S360)])
    opt.geo_threshold = 0.2
    with contextlib.redirect_stdout(io.StringIO()):
        net = R_human_nerf.HumanNeRF(opt)
    for sub, seed, mp in ((net.coarse_bkg_net, 0, 'posenc'), (net.fine_bkg_net, 1, 'posenc'), (net.coarse_human_net, 2, 'rotate')):
        sub.load_state_dict(synthetic.make_joiner(seed, mp).state_dict(), strict=True)
    net = net.eval()
    # (utils.smpl_verts_to_center_and_up needs the licensed SMPL joint regressor: the capsule's centre and axis stand in)
    center, up = inp['static_vert'].mean(0).astype(np.float64), np.array([0.0, 1.0, 0.0])
    with numpy2_array_copy_shim():
        poses = R_render.default_360_path(center, up, CANONICAL_CAMERA_DIST, CB.N360)
    out['c360_c2w'] = np.stack([p.camera_to_world for p in poses])
    out['c360_center_up'] = np.stack([center, up])
    base = PinholeCamera(CB.W360 * 4, CB.H360 * 4, CANONICAL_ZOOM_FACTOR * CB.W360 * 4, CANONICAL_ZOOM_FACTOR * CB.W360 * 4, CB.W360 * 2.0, CB.H360 * 2.0)

    def cap360(i):
        return ResizedPinholeCapture(base, poses[i], tgt_size=(CB.H360, CB.W360))
    c0 = cap360(0)
    out['c360_K'] = c0.intrinsic_matrix
    can_bone_mean = 0.25
    with contextlib.redirect_stdout(io.StringIO()):
        frames = CB.canonical_360(M, net, cap360, CB.N360, inp['static_vert'], inp['faces'], opt, can_bone_mean)
    out['c360_frames'], out['c360_can_bone_mean'] = frames.astype(np.float32), np.array(can_bone_mean)
    print('canonical_360', frames.shape, frames.dtype, 'hit fraction', float((frames.min(-1) < 1).mean()))
    # ---- render_test_views.py main
    opt = parse_opt(['--rays_per_batch', '512', '--samples_per_ray', str(CB.STV)])
    opt.geo_threshold = 0.2
    caps = []
    for k, th in enumerate((20., -35.)):
        cap = BasePinholeCapture(PinholeCamera(CB.WTV, CB.HTV, 100.0, 100.0, CB.WTV / 2, CB.HTV / 2), CameraPose.from_camera_to_world(synthetic.spherical_c2w(th, -10., 3.0)))
        cap.near, cap.far = {'bkg': 0.5}, {'bkg': 4.0}
        caps.append(cap)
    out['tv_c2w'] = np.stack([c.cam_pose.camera_to_world for c in caps])
    with contextlib.redirect_stdout(io.StringIO()):
        frames = CB.test_views(M, net, lambda i: caps[i], CB.TV_FRAMES, inp['verts'], inp['faces'], inp['Ts'], opt)
    out['tv_frames'] = frames.astype(np.float32)
    print('test_views', frames.shape, frames.dtype)
    np.savez_compressed(os.path.join(HERE, 'callers.npz'), **out)
    print('callers.npz', os.path.getsize(os.path.join(HERE, 'callers.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
