"""Golden vectors for SURVEY 8f-1 (training step of the background NeRF), from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_train.py

What runs is the reference's own training loss, unmodified: trainers/vanilla_nerf_trainer.py:NeRFTrainer.loss_func (:45-96)
-- ray_to_samples, coarse net, raw2outputs, MSE, empty-space penalty, ray_to_importance_samples, fine net, raw2outputs,
MSE -- on a stand-in `self` holding exactly the attributes the method reads, followed by torch autograd's backward.
Weights come from neuman_hip.synthetic.make_joiner(seed) loaded into the reference's modules.

Parameter gradients are 2 x 595,844 floats; the fixture keeps, per parameter tensor, its sum, its sum of magnitudes, a fixed
random projection and three full rows -- enough to catch any wrong element -- plus everything small in full (sample
positions, raw outputs, maps, losses, d loss / d raw).
"""
import os
import sys
import types
from unittest import mock

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

from models import vanilla as R_vanilla  # noqa: E402  (reference)
from trainers import vanilla_nerf_trainer as R_trainer  # noqa: E402
from utils import render_utils as R_render  # noqa: E402
from neuman_hip import synthetic  # noqa: E402  (ours: workload definitions only)

R, S, NI = 40, 24, 16


def ref_net(seed):
    ours = synthetic.make_joiner(seed)
    net, _ = R_vanilla.build_nerf(synthetic.default_opt())
    net.load_state_dict(ours.state_dict(), strict=True)
    return net.train()


def grad_summary(name, g, out, prefix):
    g = g.detach().numpy().astype(np.float64)
    g2 = g.reshape(g.shape[0], -1)
    proj = np.random.default_rng(sum(map(ord, name))).normal(size=g2.size)      # a seed both sides can rebuild
    out[f'{prefix}/{name}/stats'] = np.array([g2.sum(), np.abs(g2).sum(), float(g2.reshape(-1) @ proj)])
    rows = sorted({0, g2.shape[0] // 2, g2.shape[0] - 1})
    out[f'{prefix}/{name}/rows'] = g2[rows].astype(np.float32)


def main():
    rng = np.random.default_rng(77)
    out = {}
    ro = (rng.normal(size=(R, 3)) * 0.3).astype(np.float32)
    rd = rng.normal(size=(R, 3)).astype(np.float32)
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    near = rng.uniform(0.1, 0.6, size=(R, 1)).astype(np.float32)
    far = (near + rng.uniform(1.0, 2.5, size=(R, 1))).astype(np.float32)
    color = rng.uniform(size=(R, 3)).astype(np.float32)
    depth = rng.uniform(0.8, 2.0, size=(R,)).astype(np.float32)
    out.update(origin=ro, direction=rd, near=near, far=far, color=color, depth=depth)
    for tag, white, penal in [('white', True, 0.0), ('black_penalty', False, 0.1)]:
        coarse, fine = ref_net(0), ref_net(1)
        opt = types.SimpleNamespace(ablate_nerft=False, samples_per_ray=S, importance_samples_per_ray=NI, perturb=0.0, raw_noise_std=0.0,
                                    white_bkg=white, margin=0.9)
        fake = types.SimpleNamespace(opt=opt, coarse_net=coarse, fine_net=fine, penalize_empty_space=penal, empty_space_loss_fn=F.mse_loss)
        captured = {}
        real_r2o = R_render.raw2outputs

        def spy(raw, z, d, **kw):                     # record what the loss saw, keep grads of the raw outputs
            raw.retain_grad()
            res = real_r2o(raw, z, d, **kw)
            captured.setdefault('calls', []).append((raw, z, d, res))
            return res
        R_trainer.render_utils.raw2outputs = spy
        batch = {k: torch.from_numpy(v)[None] for k, v in dict(origin=ro, direction=rd, near=near, far=far, color=color, depth=depth).items()}
        losses = R_trainer.NeRFTrainer.loss_func(fake, batch, 'cpu')
        R_trainer.render_utils.raw2outputs = real_r2o
        sum(losses).backward()
        out[f'{tag}/losses'] = np.array([float(x.detach()) for x in losses])
        for name, net, (raw, z, d, res) in zip(('coarse', 'fine'), (coarse, fine), captured['calls']):
            p = f'{tag}/{name}'
            out[f'{p}/z'] = z.detach().numpy()
            out[f'{p}/raw'] = raw.detach().numpy()
            out[f'{p}/d_raw'] = raw.grad.numpy()
            out[f'{p}/rgb_map'], out[f'{p}/acc_map'], out[f'{p}/weights'], out[f'{p}/depth_map'] = (res[0].detach().numpy(), res[2].detach().numpy(),
                                                                                             res[3].detach().numpy(), res[4].detach().numpy())
            for n, prm in net.named_parameters():
                grad_summary(n, prm.grad, out, p)
    # raw2outputs backward on its own, every output driven: d(sum of c_k * output_k) / d raw
    raw = torch.from_numpy((rng.normal(size=(29, S, 4)) * np.array([1, 1, 1, 5])).astype(np.float32)).requires_grad_(True)
    zz = torch.from_numpy(np.sort(rng.uniform(0.0, 3.14, size=(29, S)).astype(np.float32), axis=1))
    dd = torch.from_numpy(rng.normal(size=(29, 3)).astype(np.float32))
    g_rgb, g_acc, g_depth, g_w = (rng.normal(size=(29, 3)).astype(np.float32), rng.normal(size=(29,)).astype(np.float32),
                                  rng.normal(size=(29,)).astype(np.float32), rng.normal(size=(29, S)).astype(np.float32))
    out.update({'c/raw': raw.detach().numpy(), 'c/z': zz.numpy(), 'c/d': dd.numpy(), 'c/g_rgb': g_rgb, 'c/g_acc': g_acc, 'c/g_depth': g_depth, 'c/g_w': g_w})
    for tag, wb in [('white', True), ('black', False)]:
        raw.grad = None
        rgb, disp, acc, wts, dep = R_render.raw2outputs(raw, zz, dd, white_bkg=wb)
        ((rgb * torch.from_numpy(g_rgb)).sum() + (acc * torch.from_numpy(g_acc)).sum() + (dep * torch.from_numpy(g_depth)).sum() +
         (wts * torch.from_numpy(g_w)).sum()).backward()
        out[f'c/{tag}/d_raw'] = raw.grad.numpy().copy()
    # gradients with respect to the sample positions and view directions (what pose / offset optimisation differentiates):
    # d (sum of g . net(pts, dirs)) / d pts, d dirs, for both encodings
    for mapping in ('posenc', 'rotate'):
        ours = synthetic.make_joiner(2, mapping)
        net, _ = R_vanilla.build_nerf(synthetic.default_opt(posenc=mapping))
        net.load_state_dict(ours.state_dict(), strict=True)
        if mapping == 'rotate':
            net.pos_pe.bvals = net.pos_pe.bvals.cpu()
            net.dir_pe.bvals = net.dir_pe.bvals.cpu()
        pts = torch.from_numpy(rng.uniform(-1.2, 1.2, size=(96, 3)).astype(np.float32)).requires_grad_(True)
        dirs = rng.normal(size=(96, 3)).astype(np.float32)
        dirs = torch.from_numpy(dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).requires_grad_(True)
        gout = torch.from_numpy(rng.normal(size=(96, 4)).astype(np.float32))
        o = net(pts, dirs)
        (o * gout).sum().backward()
        out.update({f'in/{mapping}/pts': pts.detach().numpy(), f'in/{mapping}/dirs': dirs.detach().numpy(), f'in/{mapping}/g_out': gout.numpy(),
                    f'in/{mapping}/out': o.detach().numpy(), f'in/{mapping}/d_pts': pts.grad.numpy(), f'in/{mapping}/d_dirs': dirs.grad.numpy()})
    # the offset net (vanilla.py:169-205, human_nerf_trainer.py:259-261): space-time input, both scale types
    from neuman_hip import vanilla as H_vanilla
    for scale_type in ('linear', 'tanh'):
        opt = synthetic.default_opt(offset_scale=0.7, offset_scale_type=scale_type)
        torch.manual_seed(11)
        ours = H_vanilla.build_offset_net(opt)
        ref = R_vanilla.build_offset_net(opt)
        ref.load_state_dict(ours.state_dict(), strict=True)
        x = torch.from_numpy(np.concatenate([rng.uniform(-1, 1, size=(77, 3)), rng.uniform(0, 1, size=(77, 1))], 1).astype(np.float32)).requires_grad_(True)
        gout = torch.from_numpy(rng.normal(size=(77, 3)).astype(np.float32))
        o = ref(x)
        (o * gout).sum().backward()
        p = f'off/{scale_type}'
        out.update({f'{p}/x': x.detach().numpy(), f'{p}/g_out': gout.numpy(), f'{p}/out': o.detach().numpy(), f'{p}/d_x': x.grad.numpy()})
        for n, prm in ref.named_parameters():
            grad_summary(n, prm.grad, out, p)
    np.savez_compressed(os.path.join(HERE, 'train.npz'), **out)
    print({k: out[k] for k in out if k.endswith('losses')})
    print(len(out), "arrays,", os.path.getsize(os.path.join(HERE, 'train.npz')), "bytes")


if __name__ == '__main__':
    main()
