"""Golden vectors for the two network variants beside the default one, generated from the REFERENCE ITSELF (build container only):

  * the plain head, use_viewdirs=False (`--specular_can no`, models/vanilla.py:116-117, 145): Joiner outputs on random points for both
    encodings, and a canonical-human frame through render_smpl_nerf;
  * the time-conditioned ablation net, `--ablate_nerft` (raw_pos_dim = 4; ray_utils.py:133-134, render_utils.py:134-148): a two-pass
    render_vanilla frame.

    python tests/golden/make_golden_heads.py   ->  tests/golden/heads.npz
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

from utils import render_utils as R_render  # noqa: E402  (reference)
from models import vanilla as R_vanilla  # noqa: E402
from cameras.pinhole_camera import PinholeCamera  # noqa: E402
from cameras.camera_pose import CameraPose  # noqa: E402
from cameras.captures import BasePinholeCapture  # noqa: E402

from neuman_hip import synthetic  # noqa: E402  (ours: workload definitions only)


def ref_net(seed, **opt_over):
    """the reference's own Joiner with the synthetic-dense preset applied to whichever head it has (synthetic.densify)"""
    opt = synthetic.default_opt(**opt_over)
    torch.manual_seed(seed)
    net, _ = R_vanilla.build_nerf(opt)
    synthetic.densify(net)
    if getattr(opt, 'posenc', 'posenc') == 'rotate':
        net.pos_pe.bvals = net.pos_pe.bvals.cpu()
        net.dir_pe.bvals = net.dir_pe.bvals.cpu()
    return net.eval()


def checksum(net):
    """weights are NOT stored: the tests rebuild the same nets with neuman_hip.synthetic.make_variant_joiner (same seed, same
    construction order as the reference's build_nerf) and compare this checksum"""
    sd = net.state_dict()
    return np.array([float(sum(v.abs().sum(dtype=torch.float64) for v in sd.values())), float(sd['nerf.pts_linears.0.weight'][0, 0]),
                     float(sd['nerf.pts_linears.7.bias'][5])])


def main():
    rng = np.random.default_rng(4321)
    out = {}
    pts = rng.uniform(-1.5, 1.5, size=(257, 3)).astype(np.float32)
    dirs = rng.normal(size=(257, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out['pts'], out['dirs'] = pts, dirs
    # ---- plain head, both encodings
    for mapping in ('posenc', 'rotate'):
        net = ref_net(5, posenc=mapping, use_viewdirs=False)
        assert hasattr(net.nerf, 'output_linear') and not hasattr(net.nerf, 'alpha_linear')
        with torch.no_grad():
            out[f'plain_{mapping}_out'] = net(torch.from_numpy(pts), torch.from_numpy(dirs)).numpy()
        out[f'plain_{mapping}_checksum'] = checksum(net)
    # ---- plain head through render_smpl_nerf (canonical render, the path `--specular_can no` takes in render_360.py)
    net = ref_net(5, posenc='rotate', use_viewdirs=False)
    cap = BasePinholeCapture(PinholeCamera(40, 40, 50., 50., 20., 20.), CameraPose.from_camera_to_world(synthetic.spherical_c2w(40., 0., 3.0)))
    holder = mock.MagicMock()
    holder.coarse_human_net = net
    holder.parameters = net.parameters
    verts = synthetic.human_vertex_cloud(0)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        rgb, depth, acc = R_render.render_smpl_nerf(holder, cap, verts, None, None, rays_per_batch=4096, samples_per_ray=24, render_can=True,
                                                    geo_threshold=0.2, return_depth=True, return_mask=True, interval_comp=0.8)
    out['plain_c3_c2w'], out['plain_c3_rgb'], out['plain_c3_depth'], out['plain_c3_acc'] = cap.cam_pose.camera_to_world, rgb, depth, acc
    # ---- time-conditioned nets: 4-D position encoding
    coarse, fine = ref_net(6, raw_pos_dim=4), ref_net(7, raw_pos_dim=4)
    assert coarse.pos_pe.out_dim == 84
    cap = BasePinholeCapture(PinholeCamera(24, 18, 30., 30., 12., 9.), CameraPose.from_camera_to_world(np.eye(4)))
    cap.near, cap.far = {'bkg': 0.0}, {'bkg': 3.14}
    cap.frame_id = {'frame_id': 7, 'total_frames': 20}
    with contextlib.redirect_stdout(io.StringIO()):
        rgb, depth = R_render.render_vanilla(coarse, cap, fine, rays_per_batch=256, samples_per_ray=16, importance_samples_per_ray=16,
                                             return_depth=True, ablate_nerft=True)
        rgb1 = R_render.render_vanilla(coarse, cap, None, rays_per_batch=256, samples_per_ray=16, ablate_nerft=True)
    out['nerft_rgb'], out['nerft_depth'], out['nerft_coarse_only_rgb'] = rgb, depth, rgb1
    out['nerft_coarse_checksum'], out['nerft_fine_checksum'] = checksum(coarse), checksum(fine)
    t4 = np.concatenate([pts, rng.uniform(0, 1, size=(257, 1)).astype(np.float32)], 1)
    with torch.no_grad():
        out['nerft_pts4'], out['nerft_out'] = t4, coarse(torch.from_numpy(t4), torch.from_numpy(dirs)).numpy()
    np.savez_compressed(os.path.join(HERE, "heads.npz"), **out)
    print("wrote heads.npz:", {k: v.shape for k, v in out.items() if '/' not in k})


if __name__ == "__main__":
    main()
