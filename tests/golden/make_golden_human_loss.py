"""Goldens for SURVEY 8f-1's human trainer, produced by the reference's OWN loss code and autograd (build container only):

    python tests/golden/make_golden_human_loss.py     ->  tests/golden/human_loss.npz

Executed unmodified: trainers/human_nerf_trainer.py HumanNeRFTrainer.loss_func (:382-446) with everything it calls --
_eval_bkg_samples (:180-239), _eval_human_samples (:241-278), the colour-range, symmetry, SMPL-shape and sparsity regularisers
(:280-380) -- utils/ray_utils.py warp_samples_to_canonical_diff (:69-93), models/human_nerf.py HumanNeRF.vertex_forward (:92-122),
models/smpl.py, models/vanilla.py (the NeRF and offset networks), utils/render_utils.py raw2outputs, then `sum(loss_dict).backward()`
through the reference's torch graph.  `igl` is tests/golden/igl_shim.py (signed_distance: closest point + pseudonormal sign from
oracle/warp.py, igl 2.2.1's return conventions); the other absent wheels are stubbed.  Two constructors are bypassed because they need
assets that do not exist offline, none of their code being on the path: HumanNeRFTrainer.__init__ (datasets, tensorboard, lpips) --
the object is made with __new__ and given exactly the attributes loss_func reads -- and HumanNeRF.__init__ (the licensed SMPL file,
checkpoints) -- a subclass whose __init__ builds the same sub-modules with the reference's own builders around the synthetic body
model written in SMPL's file layout (as tests/golden/make_golden_smpl.py does).

The scene is the one tests/test_hip_human_trainer.py builds (same seeds): 256 rays of a 48 x 48 camera at an SMPL-sized synthetic
body, 24 + 24 background and 24 human samples per ray, all seven loss weights on except LPIPS (no weights offline).  The random draws
the loss makes (dummy directions, dummy points, canonical camera and pixels) are recorded so that the device implementation can
replay them.
"""
import os
import pickle
import random
import sys
import tempfile
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, HERE)
import igl_shim  # noqa: E402

sys.modules["igl"] = igl_shim
for m in ["open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

from models import smpl as R_smpl, human_nerf as R_hn, vanilla as R_vanilla  # noqa: E402  (reference)
from trainers import human_nerf_trainer as R_tr  # noqa: E402
from utils import ray_utils as R_ray  # noqa: E402
from cameras.pinhole_camera import PinholeCamera  # noqa: E402
from cameras.camera_pose import CameraPose  # noqa: E402
from cameras.captures import BasePinholeCapture  # noqa: E402

from neuman_hip import synthetic, vanilla as our_vanilla  # noqa: E402  (ours: workload definitions and initial weights only)

FULL = '--full' in sys.argv     # the trainer's real sizes: 128 + 128 background and 128 human samples (the merged 384-sample list of
#                                 human_nerf_trainer.py:415-428) on 512 rays -> human_loss_full.npz; default: 24 + 24 / 24 on 256 rays -> human_loss.npz
N_RAYS = 512 if FULL else 256
OUT_NAME = 'human_loss_full.npz' if FULL else 'human_loss.npz'
OPT = dict(samples_per_ray=128 if FULL else 24, importance_samples_per_ray=128 if FULL else 24, perturb=0.0, white_bkg=True, penalize_smpl_alpha=1.0,
           penalize_symmetric_alpha=0.1, penalize_dummy=1.0, penalize_hard_surface=0.1, penalize_color_range=0.1, penalize_mask=0.01,
           penalize_lpips=0.0, penalize_sharp_edge=0.1, penalize_outside_factor=2.0, dist_exponent=2.0)
INTERVAL_COMP = 0.8
CAN_ANGLES = (0., 90., 200.)


class Body(R_hn.HumanNeRF):
    """models/human_nerf.py HumanNeRF with its sub-modules built by the reference's own builders; vertex_forward is inherited"""

    def __init__(self, smpl_dir):
        torch.nn.Module.__init__(self)
        nets = {}
        for name, seed, mapping in (("coarse_bkg_net", 0, "posenc"), ("fine_bkg_net", 1, "posenc"), ("coarse_human_net", 2, "rotate")):
            ours = synthetic.make_joiner(seed, mapping)
            net, _ = R_vanilla.build_nerf(synthetic.default_opt(posenc=mapping))
            net.load_state_dict(ours.state_dict(), strict=True)
            if mapping == 'rotate':
                net.pos_pe.bvals, net.dir_pe.bvals = net.pos_pe.bvals.cpu(), net.dir_pe.bvals.cpu()
            nets[name] = net
        self.coarse_bkg_net, self.fine_bkg_net, self.coarse_human_net = nets["coarse_bkg_net"].eval(), nets["fine_bkg_net"].eval(), nets["coarse_human_net"].train()
        oopt = synthetic.default_opt(offset_scale=0.05, offset_scale_type='linear')
        torch.manual_seed(3)
        ours = our_vanilla.build_offset_net(oopt)
        ref = R_vanilla.build_offset_net(oopt)
        ref.load_state_dict(ours.state_dict(), strict=True)
        self.offset_nets = torch.nn.ModuleList([ref]).train()
        pose, betas, align = synthetic.smpl_like_frames(3, 0)
        al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
        al[:, :3, :3] = np.eye(3)[None] * 1.0
        al[:, 3, :3] = 0.0
        self.poses = torch.nn.Parameter(torch.from_numpy(pose * 0.3).float())
        self.betas = torch.nn.Parameter(torch.from_numpy(betas * 0.3).float())
        self.alignments = torch.nn.Parameter(torch.from_numpy(al).float())
        self.scale = 1.0
        self.body_model = R_smpl.SMPL(smpl_dir, gender='neutral', device=torch.device('cpu'))
        da = torch.zeros(24, 3)
        da[1], da[2] = torch.tensor([0, 0, 1.0]), torch.tensor([0, 0, -1.0])
        self.da_smpl = torch.nn.Parameter(da.reshape(1, -1), requires_grad=False)


class Packed:
    def __init__(self, verts, faces):
        self.v, self.f = torch.as_tensor(verts), torch.as_tensor(faces)

    def verts_packed(self):
        return self.v

    def faces_packed(self):
        return self.f


def ref_cap(w, h, fx, c2w, near=None, far=None):
    cap = BasePinholeCapture(PinholeCamera(w, h, fx, fx, w / 2, h / 2), CameraPose.from_camera_to_world(c2w))
    if near is not None:
        cap.near, cap.far = {'bkg': near}, {'bkg': far}
    return cap


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    model = synthetic.smpl_like_model(0)
    faces = model['f'].astype(np.int64)
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
            pickle.dump(model, f, protocol=2)
        net = Body(tmp)
    # the canonical (da-pose) body of frame 1: what scene.captures[i].can_mesh holds (utils.add_pytorch3d_cache, static_vert)
    with torch.no_grad():
        world, _ = net.vertex_forward(1)
        can_verts = net.body_model(return_tensor=True, return_joints=False, poses=net.da_smpl, betas=net.betas[1][None], transl=None)
    can_verts = can_verts.numpy().reshape(-1, 3).astype(np.float32)
    out['can_verts'] = can_verts
    out['world_verts_sample'] = world[0].numpy()[::97]
    cap = ref_cap(48, 48, 110., synthetic.spherical_c2w(15., -5., 3.0), 0.5, 5.0)
    out['cam_c2w'] = cap.cam_pose.camera_to_world
    coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
    coords = coords[np.random.default_rng(1).choice(len(coords), N_RAYS, replace=False)]
    o, d = R_ray.shot_rays(cap, coords)
    o, d = torch.from_numpy(o).float(), torch.from_numpy(d).float()
    near, far = R_ray.geometry_guided_near_far(o, d, world[0], 0.2)
    hit = near < far
    near = torch.where(hit, near, torch.full_like(near, 2.0))
    far = torch.where(hit, far, torch.full_like(far, 3.0))
    color = torch.rand((N_RAYS, 3), generator=torch.Generator().manual_seed(5))
    batch = {'origin': o, 'direction': d, 'bkg_near': torch.full((N_RAYS, 1), 0.5), 'bkg_far': torch.full((N_RAYS, 1), 5.0), 'human_near': near[:, None].contiguous(),
             'human_far': far[:, None].contiguous(), 'is_hit': hit, 'is_bkg': (~hit).long(), 'color': color, 'cur_view_f': torch.tensor(0.35),
             'cap_id': torch.tensor(1), 'patch_counter': torch.tensor(0)}
    out.update({'batch_' + k: v.numpy() for k, v in batch.items()})
    can_caps = [ref_cap(32, 32, 40., synthetic.spherical_c2w(a, 0., 3.0)) for a in CAN_ANGLES]
    out['can_c2w'] = np.stack([c.cam_pose.camera_to_world for c in can_caps])

    tr = R_tr.HumanNeRFTrainer.__new__(R_tr.HumanNeRFTrainer)
    tr.net, tr.opt = net, types.SimpleNamespace(**OPT)
    for k in ('penalize_smpl_alpha', 'penalize_symmetric_alpha', 'penalize_dummy', 'penalize_hard_surface', 'penalize_color_range', 'penalize_mask',
              'penalize_lpips', 'penalize_sharp_edge'):
        setattr(tr, k, OPT[k])
    tr.penalize_outside = 0.0
    tr.interval_comp = INTERVAL_COMP
    tr.can_caps = can_caps
    capture = types.SimpleNamespace(can_mesh=Packed(can_verts, faces), posed_mesh_cpu=Packed(world[0].numpy(), faces))
    tr.val_dataset = types.SimpleNamespace(scene=types.SimpleNamespace(captures=[capture] * 3))

    # ---- record the loss's random draws
    rec = {}
    real_randn, real_rand, real_choice, real_randint = torch.randn, torch.rand, random.choice, np.random.randint

    def randn(*a, **k):
        r = real_randn(*a, **k)
        rec.setdefault('randn', []).append(r.numpy().copy())
        return r

    def rand(*a, **k):
        r = real_rand(*a, **k)
        rec.setdefault('rand', []).append(r.numpy().copy())
        return r

    def choice(seq):
        r = real_choice(seq)
        rec.setdefault('choice', []).append([i for i, x in enumerate(seq) if x is r][0])
        return r

    def randint(*a, **k):
        r = real_randint(*a, **k)
        rec.setdefault('randint', []).append(np.asarray(r).copy())
        return r
    torch.manual_seed(11)
    random.seed(4)
    np.random.seed(4)
    torch.randn, torch.rand, random.choice, np.random.randint = randn, rand, choice, randint
    try:
        loss_dict, rgb_map = tr.loss_func({k: v[None] for k, v in batch.items()}, return_rgb=True)
    finally:
        torch.randn, torch.rand, random.choice, np.random.randint = real_randn, real_rand, real_choice, real_randint
    assert len(rec['randn']) == 1 and len(rec['rand']) == 1 and len(rec['choice']) == 2 and len(rec['randint']) == 1, {k: len(v) for k, v in rec.items()}
    out['dummy_dirs_randn'] = rec['randn'][0]                 # :283, before normalisation
    out['dummy_pts_rand'] = rec['rand'][0]                    # :323, before (x - 0.5) * 3
    out['offset_net_choice'], out['can_cap_choice'] = np.array(rec['choice'][0]), np.array(rec['choice'][1])
    out['can_pixel_choice'] = rec['randint'][0]               # :351: indices into np.argwhere(np.ones(shape))
    for k, v in loss_dict.items():
        out['loss_' + k] = np.array(float(v.detach()))
        print(f"{k:18s} {float(v.detach()):.6e}")
    out['fine_rgb_map'] = rgb_map.detach().numpy()
    total = sum(loss_dict.values())
    total.backward()
    grads = {"human.pts_linears.0.weight": net.coarse_human_net.nerf.pts_linears[0].weight, "human.pts_linears.7.weight": net.coarse_human_net.nerf.pts_linears[7].weight,
             "human.alpha_linear.weight": net.coarse_human_net.nerf.alpha_linear.weight, "human.views_linears.0.weight": net.coarse_human_net.nerf.views_linears[0].weight,
             "human.rgb_linear.weight": net.coarse_human_net.nerf.rgb_linear.weight, "offset.pts_linears.0.weight": net.offset_nets[0].nerf.pts_linears[0].weight,
             "offset.output_linear.weight": net.offset_nets[0].nerf.output_linear.weight, "poses": net.poses, "betas": net.betas, "alignments": net.alignments}
    for k, p in grads.items():
        assert p.grad is not None, k
        out['grad_' + k] = p.grad.numpy().copy()
        print(f"grad {k:32s} |g|_inf {float(p.grad.abs().max()):.3e}")
    assert all(p.grad is None for p in net.coarse_bkg_net.parameters())

    # ---- how well defined are these gradients in float32?  The same reference code once more, with the poses moved by 1e-6 (a float32
    # epsilon of the posed vertices): the networks' gradients barely move, the gradients of the SMPL parameters move by 5-16 % -- they
    # run through d(barycentric) / d(vertex) ~ 1 / edge length of whichever face each sample's foot lands on and a network with 512
    # rad / unit encodings.  Stored as the floor an independent float32 implementation can be held to.
    g0 = {k: p.grad.clone() for k, p in grads.items()}
    for p_ in net.parameters():
        p_.grad = None
    with torch.no_grad():
        net.poses.add_(torch.randn(net.poses.shape, generator=torch.Generator().manual_seed(9)) * 1e-6)
        world2, _ = net.vertex_forward(1)
    capture.posed_mesh_cpu = Packed(world2[0].numpy(), faces)
    torch.manual_seed(11)
    random.seed(4)
    np.random.seed(4)
    ld2 = tr.loss_func({k: v[None] for k, v in batch.items()})
    sum(ld2.values()).backward()
    for k, p_ in grads.items():
        dev = float((p_.grad - g0[k]).abs().max() / g0[k].abs().max())
        out['grad_floor_' + k] = np.array(dev)
        print(f"grad floor {k:32s} {dev:.3e}   (poses + 1e-6)")
    out['opt_samples'] = np.array([OPT['samples_per_ray'], OPT['importance_samples_per_ray']])
    np.savez_compressed(os.path.join(HERE, OUT_NAME), **out)
    print(OUT_NAME, os.path.getsize(os.path.join(HERE, OUT_NAME)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
