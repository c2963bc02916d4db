"""Differentiable skinning (neuman_hip.smpl.SMPLDiff: HumanNeRF.vertex_forward with autograd) against the reference's
own forward values (tests/golden/smpl.npz) and the reference's own autograd gradients (tests/golden/smpl_grad.npz).  Tensor
algebra through torch: runs on the CPU here and on the HIP device in the -m gpu suite."""
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))


def setup(device):
    from neuman_hip import smpl, synthetic
    model = synthetic.smpl_like_model(0)
    pose, betas, align = synthetic.smpl_like_frames(3, 0)
    al = np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1) for i in range(3)]).astype(np.float32)
    body = smpl.SMPLDiff(model, device)
    P = torch.tensor(pose[1:2], device=device, requires_grad=True)
    B = torch.tensor(betas[1:2], device=device, requires_grad=True)
    A = torch.tensor(al[1], device=device, requires_grad=True)
    return body, P, B, A


def check(device):
    G = dict(np.load(os.path.join(HERE, "golden", "smpl.npz")))
    D = dict(np.load(os.path.join(HERE, "golden", "smpl_grad.npz")))
    body, P, B, A = setup(device)
    wv, T = body.vertex_forward(P, B, A, 1.37)
    rows = G['rows'][G['rows'] < 6890]
    np.testing.assert_allclose(wv[0].detach().cpu().numpy()[rows], G['vf_world_verts'], atol=2e-6)
    np.testing.assert_allclose(T[0].detach().cpu().numpy()[rows], G['vf_T'], atol=2e-6)
    rng = np.random.default_rng(77)
    g_w = torch.from_numpy(rng.normal(size=(6890, 3)).astype(np.float32)).to(device)
    g_T = torch.from_numpy((rng.normal(size=(6890, 4, 4)) * 0.3).astype(np.float32)).to(device)
    loss = (wv[0] * g_w).sum() + (T[0] * g_T).sum()
    assert abs(float(loss.detach()) - D["loss"]) < 2e-3
    loss.backward()
    for name, got, ref in (("pose", P.grad[0], D['d_pose']), ("betas", B.grad[0], D['d_betas']), ("alignment", A.grad, D['d_align'])):
        e = np.abs(got.cpu().numpy() - ref).max() / np.abs(ref).max()
        print(f"[smpl diff] d/d{name}: relative error vs the reference's autograd {e:.2e}")
        assert e < 2e-4, name
    # Rodrigues at (near) zero angle: finite gradients (the reference adds 1e-8 before the norm, smpl.py:420)
    z = torch.zeros((2, 3), device=device, requires_grad=True)
    body.rodrigues(z).sum().backward()
    assert torch.isfinite(z.grad).all()


def test_vertex_forward_and_gradients_cpu():
    check('cpu')


@pytest.mark.gpu
def test_vertex_forward_and_gradients_device():
    check('cuda')
    # the differentiable chain agrees with the forward-only HIP kernel (SMPL.frames) used by the renderers
    from neuman_hip import smpl, synthetic
    body, P, B, A = setup('cuda')
    fast = smpl.SMPL(synthetic.smpl_like_model(0), device='cuda')
    w2, T2 = smpl.vertex_forward(fast, P.detach(), B.detach(), A.detach().cpu().numpy(), 1.37)
    w1, T1 = body.vertex_forward(P, B, A, 1.37)
    assert (w1 - w2).abs().max().item() < 2e-6 and (T1 - T2).abs().max().item() < 2e-6
