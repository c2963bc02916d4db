"""Generate the golden vectors under tests/golden/ from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference (apple/ml-neuman) is imported unmodified; the third-party wheels that are absent offline and
that none of the exercised functions touch are stubbed.  `igl` is one of them, so
warp_samples_to_canonical cannot be executed -> no golden for it (oracle/warp.py is "parity unpinned").

Weights come from neuman_hip.synthetic.make_joiner(seed) and are loaded into the reference's own modules with
load_state_dict(strict=True) -- which also pins state_dict compatibility; the fixture stores a checksum so the
tests can tell when a torch upgrade changes nn.Linear's default init.
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

from utils import ray_utils as R_ray, render_utils as R_render  # noqa: E402  (reference)
from models import vanilla as R_vanilla  # noqa: E402
from cameras.pinhole_camera import PinholeCamera  # noqa: E402
from cameras.camera_pose import CameraPose  # noqa: E402
from cameras.captures import BasePinholeCapture  # noqa: E402

from neuman_hip import synthetic  # noqa: E402  (ours: workload definitions only)


def weight_checksum(sd):
    return np.array([float(sum(np.abs(v).sum(dtype=np.float64) for v in sd.values())),
                     float(sd['nerf.pts_linears.0.weight'][0, 0]), float(sd['nerf.rgb_linear.weight'][2, 5])])


def ref_net(seed, mapping):
    ours = synthetic.make_joiner(seed, mapping)
    opt = synthetic.default_opt(posenc=mapping)
    net, _ = R_vanilla.build_nerf(opt)
    net.load_state_dict(ours.state_dict(), strict=True)
    if mapping == 'rotate':
        net.pos_pe.bvals = net.pos_pe.bvals.cpu()
        net.dir_pe.bvals = net.dir_pe.bvals.cpu()
    return net.eval(), synthetic.state_numpy(ours)


def ref_cap(w, h, c2w, near=0.0, far=3.14):
    cap = BasePinholeCapture(PinholeCamera(w, h, 1.25 * w, 1.25 * w, w / 2, h / 2), CameraPose.from_camera_to_world(c2w))
    cap.near, cap.far = {'bkg': near}, {'bkg': far}
    return cap


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def main():
    rng = np.random.default_rng(1234)
    out = {}

    # ---------------------------------------------------------------- rays
    c2w = synthetic.spherical_c2w(30., -20., 3.0)
    cap = ref_cap(16, 12, c2w)
    out['cam_c2w'] = cap.cam_pose.camera_to_world          # what the reference really uses (f32 quaternion round trip)
    out['cam_K'] = cap.intrinsic_matrix
    coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
    o, d = R_ray.shot_rays(cap, coords)
    out['shot_rays_o'], out['shot_rays_d'] = o, d
    o, d = R_ray.shot_all_rays(cap)
    out['shot_all_o'], out['shot_all_d'] = o, d

    # ---------------------------------------------------------------- ray_to_samples
    Rn, S = 37, 32
    ro = rng.normal(size=(Rn, 3)).astype(np.float32)
    rd = rng.normal(size=(Rn, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    near = rng.uniform(0.1, 1.0, size=(Rn, 1)).astype(np.float32)
    far = (near + rng.uniform(0.5, 3.0, size=(Rn, 1))).astype(np.float32)
    batch = {'origin': t(ro), 'direction': t(rd), 'near': t(near), 'far': t(far)}
    out.update(rs_o=ro, rs_d=rd, rs_near=near, rs_far=far)
    for tag, kw in [('lin', {}), ('disp', {'lindisp': True})]:
        p, dd, z = R_ray.ray_to_samples(batch, S, **kw)
        out[f'rs_{tag}_pts'], out[f'rs_{tag}_dirs'], out[f'rs_{tag}_z'] = p.numpy(), dd.numpy(), z.numpy()
    torch.manual_seed(5)
    t_rand = torch.clip(torch.rand((Rn, S)), min=0.01, max=0.99).numpy()
    torch.manual_seed(5)
    p, dd, z = R_ray.ray_to_samples(batch, S, perturb=1.0)
    out['rs_perturb_trand'], out['rs_perturb_pts'], out['rs_perturb_z'] = t_rand, p.numpy(), z.numpy()
    out['linspace32'] = torch.linspace(0., 1., steps=S).numpy()

    # ---------------------------------------------------------------- sample_pdf / importance
    bins = np.sort(rng.uniform(0.2, 3.0, size=(Rn, 31)).astype(np.float32), axis=1)
    w = (rng.uniform(size=(Rn, 30)) ** 4).astype(np.float32)
    w[3] = 0.0                                              # all-zero weights row (pdf = uniform via the 1e-5 floor)
    out['pdf_bins'], out['pdf_w'] = bins, w
    out['pdf_samples'] = R_ray.sample_pdf(t(bins), t(w), 16, det=True).numpy()
    zc = R_ray.ray_to_samples(batch, S)[2]
    wc = (rng.uniform(size=(Rn, S)) ** 6).astype(np.float32)
    wc[5] = 0.0
    out['imp_w'] = wc
    p, dd, z = R_ray.ray_to_importance_samples(batch, zc, t(wc), 24)
    out['imp_pts'], out['imp_dirs'], out['imp_z'] = p.numpy(), dd.numpy(), z.numpy()
    _, _, z = R_ray.ray_to_importance_samples(batch, zc, t(wc), 24, including_old=False)
    out['imp_z_new_only'] = z.numpy()

    # ---------------------------------------------------------------- near/far
    verts = synthetic.human_vertex_cloud(3, 200)
    cam_o = np.array([0., 0., -3.], np.float32)
    tgt = rng.uniform(-1, 1, size=(64, 3)).astype(np.float32) * np.array([0.6, 1.2, 0.3], np.float32)
    nd = tgt - cam_o
    nd /= np.linalg.norm(nd, axis=1, keepdims=True)
    no = np.repeat(cam_o[None], 64, 0)
    n_t, f_t = R_ray.geometry_guided_near_far(t(no), t(nd), t(verts), 0.2)
    n_n, f_n = R_ray.geometry_guided_near_far(no, nd, verts, 0.2)
    out.update(nf_o=no, nf_d=nd, nf_verts=verts, nf_near_torch=n_t.numpy(), nf_far_torch=f_t.numpy(), nf_near_np=n_n, nf_far_np=f_n)

    # ---------------------------------------------------------------- raw2outputs
    raw = (rng.normal(size=(29, S, 4)) * np.array([1, 1, 1, 5])).astype(np.float32)
    zz = np.sort(rng.uniform(0.0, 3.14, size=(29, S)).astype(np.float32), axis=1)
    dd = rng.normal(size=(29, 3)).astype(np.float32)         # deliberately non-unit: dists are scaled by |d| (:88)
    out.update(c_raw=raw, c_z=zz, c_d=dd)
    for tag, wb in [('white', True), ('black', False)]:
        rgb, disp, acc, wts, depth = R_render.raw2outputs(t(raw), t(zz), t(dd), white_bkg=wb)
        out[f'c_{tag}_rgb'], out[f'c_{tag}_disp'], out[f'c_{tag}_acc'] = rgb.numpy(), disp.numpy(), acc.numpy()
        out[f'c_{tag}_w'], out[f'c_{tag}_depth'] = wts.numpy(), depth.numpy()
    np.savez_compressed(os.path.join(HERE, 'ray_ops.npz'), **out)

    # ---------------------------------------------------------------- PE + MLP
    out = {}
    pts = rng.uniform(-1.5, 1.5, size=(300, 3)).astype(np.float32)
    dirs = rng.normal(size=(300, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out['pts'], out['dirs'] = pts, dirs
    with torch.no_grad():
        for seed, mapping in [(0, 'posenc'), (2, 'rotate')]:
            net, sd = ref_net(seed, mapping)
            out[f'{mapping}_checksum'] = weight_checksum(sd)
            out[f'{mapping}_pos_pe'] = net.pos_pe(t(pts)).numpy()
            out[f'{mapping}_dir_pe'] = net.dir_pe(t(dirs)).numpy()
            out[f'{mapping}_out'] = net(t(pts), t(dirs)).numpy()
    np.savez_compressed(os.path.join(HERE, 'mlp.npz'), **out)

    # ---------------------------------------------------------------- whole frames
    out = {}
    coarse, sd0 = ref_net(0, 'posenc')
    fine, sd1 = ref_net(1, 'posenc')
    out['checksum_seed0'], out['checksum_seed1'] = weight_checksum(sd0), weight_checksum(sd1)
    cap = ref_cap(64, 64, np.eye(4))
    out['c1_c2w'] = cap.cam_pose.camera_to_world
    rgb, depth = R_render.render_vanilla(coarse, cap, fine, rays_per_batch=2048, samples_per_ray=32,
                                         importance_samples_per_ray=32, return_depth=True)
    out['c1_rgb'], out['c1_depth'] = rgb, depth
    rgb = R_render.render_vanilla(coarse, cap, None, rays_per_batch=4096, samples_per_ray=32)
    out['c1_coarse_only_rgb'] = rgb

    human, sd2 = ref_net(2, 'rotate')
    out['checksum_seed2'] = weight_checksum(sd2)

    class Wrap(torch.nn.Module):
        def __init__(self, n):
            super().__init__()
            self.coarse_human_net = n
    cap = ref_cap(48, 48, synthetic.spherical_c2w(40., 0., 3.0))
    cap.pinhole_cam.fx = cap.pinhole_cam.fy = 1.6 * 48
    out['c3_c2w'] = cap.cam_pose.camera_to_world
    out['c3_fx'] = np.array(cap.pinhole_cam.fx)
    hv = synthetic.human_vertex_cloud(0)
    rgb, depth, acc = R_render.render_smpl_nerf(Wrap(human), cap, hv, None, None, rays_per_batch=1024, samples_per_ray=32,
                                                render_can=True, geo_threshold=0.2, return_depth=True, return_mask=True,
                                                interval_comp=0.7)
    out['c3_rgb'], out['c3_depth'], out['c3_acc'] = rgb, depth, acc
    np.savez_compressed(os.path.join(HERE, 'render.npz'), **out)
    for f in ('ray_ops.npz', 'mlp.npz', 'render.npz'):
        print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
