"""Gradients of the REFERENCE's HumanNeRF.vertex_forward (models/human_nerf.py:92-122, autograd through models/smpl.py lbs) with
respect to pose, betas and alignment on the synthetic SMPL-like body -> tests/golden/smpl_grad.npz.  Build container only."""
import os
import pickle
import sys
import tempfile
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, "/root/reference")
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

from models import smpl as R_smpl, human_nerf as R_hn  # noqa: E402  (reference)
from neuman_hip import synthetic  # noqa: E402


def main():
    model = synthetic.smpl_like_model(0)
    pose, betas, align = synthetic.smpl_like_frames(3, 0)
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
            pickle.dump(model, f, protocol=2)
        body = R_smpl.SMPL(tmp, gender='neutral', device=torch.device('cpu'))
    da = np.zeros((24, 3), np.float32)
    da[1], da[2] = (0, 0, 1.0), (0, 0, -1.0)
    al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
    P = torch.tensor(pose, requires_grad=True)
    B = torch.tensor(betas, requires_grad=True)
    A = torch.tensor(al, requires_grad=True)
    fake = types.SimpleNamespace(poses=P, betas=B, body_model=body, da_smpl=torch.from_numpy(da.reshape(1, 72)), alignments=A, scale=1.37)
    rng = np.random.default_rng(77)
    g_w = rng.normal(size=(6890, 3)).astype(np.float32)
    g_T = (rng.normal(size=(6890, 4, 4)) * 0.3).astype(np.float32)
    wv, T = R_hn.HumanNeRF.vertex_forward(fake, 1)
    ((wv[0] * torch.from_numpy(g_w)).sum() + (T[0] * torch.from_numpy(g_T)).sum()).backward()
    out = {'d_pose': P.grad[1].numpy(), 'd_betas': B.grad[1].numpy(), 'd_align': A.grad[1].numpy(), 'world_sum': wv.detach().numpy().sum(dtype=np.float64),
           'loss': float(((wv[0] * torch.from_numpy(g_w)).sum() + (T[0] * torch.from_numpy(g_T)).sum()).detach())}
    np.savez_compressed(os.path.join(HERE, 'smpl_grad.npz'), **out)
    print({k: (v.shape if hasattr(v, 'shape') else v) for k, v in out.items()})


if __name__ == '__main__':
    main()
