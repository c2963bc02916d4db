"""-m gpu: early ray termination + compaction of the live rays in front of the MLP (north star; the reference evaluates every
sample, utils/render_utils.py:139-151, so the contract is: off = bit-identical, on = colour within eps of the full evaluation)."""
import numpy as np
import pytest
import torch

from oracle import compositing, nerf_mlp, ray_ops as O
from oracle.nerf_mlp import JoinerSpec

pytestmark = pytest.mark.gpu


def cu(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', torch.float32).contiguous()


@pytest.fixture(scope="module")
def scene():
    from neuman_hip import ray_utils, render_utils, synthetic
    net = synthetic.make_joiner(1, preset='opaque').cuda()
    cap = synthetic.SimpleCapture(800, 800)
    o, d = ray_utils.shot_all_rays_dev(cap, torch.device('cuda'))
    sel = torch.arange(390 * 800, 390 * 800 + 4096, device='cuda')
    return dict(net=net, o=o[sel].contiguous(), d=d[sel].contiguous(), render=render_utils, ray=ray_utils, syn=synthetic)


def fine_z(sc, S=128, NI=128):
    o, d, net = sc['o'], sc['d'], sc['net']
    R = o.shape[0]
    _, _, z = sc['ray'].sample_z(o, d, torch.zeros(R, device='cuda'), torch.full((R,), 3.14, device='cuda'), S)
    raw = net.forward_rays(o, d, z, sigma_only=True)
    w = sc['render'].raw2outputs(raw, z, d)[3]
    return sc['ray'].importance_z(z, w, NI)


@pytest.mark.parametrize("precision", ["i8x3", "fp16x3"])
@pytest.mark.parametrize("chunk", [32, 64, 100])
def test_marching_without_termination_is_bit_identical(scene, precision, chunk):
    """eps = 0: nothing is dropped; the chunked, indexed launches must reproduce the single launch bit for bit (every kernel
    quantity is per sample, so how samples are grouped into tiles cannot matter)"""
    z = fine_z(scene)
    full = scene['net'].forward_rays(scene['o'], scene['d'], z, precision=precision)
    stats = {}
    marched = scene['render'].march_pass_rays(scene['net'], scene['o'], scene['d'], z, 0.0, chunk=chunk, precision=precision, role=None, stats=stats)
    assert stats['evaluated'] == stats['total'] == z.numel()
    assert torch.equal(marched, full)


@pytest.mark.parametrize("eps", [1e-4, 1e-3])
def test_early_termination_vs_full_evaluation_and_oracle(scene, eps):
    """opaque workload, 4096 rays x (128 + 128) samples: a good share of the evaluations is skipped, every pixel stays within eps
    of the full evaluation, and within eps + 2e-5 of the CPU oracle (which evaluates everything) on the same sample positions"""
    o, d, net = scene['o'], scene['d'], scene['net']
    z = fine_z(scene)
    full = net.forward_rays(o, d, z, role='shading')
    rgb_full, _, acc_full, _, depth_full = scene['render'].raw2outputs(full, z, d)
    stats = {}
    marched = scene['render'].march_pass_rays(net, o, d, z, eps, stats=stats)
    rgb, _, acc, _, depth = scene['render'].raw2outputs(marched, z, d)
    frac = stats['evaluated'] / stats['total']
    e = (rgb - rgb_full).abs().max().item()
    print(f"[march] eps {eps:g}: evaluated {frac:.3f} of the samples, colour Linf vs the full evaluation {e:.2e}, acc Linf {(acc - acc_full).abs().max().item():.2e}, "
          f"rays fully opaque {(acc_full > 0.9999).float().mean().item():.2f}")
    assert frac < 0.8 and e <= eps
    assert (acc - acc_full).abs().max().item() <= eps and (depth - depth_full).abs().max().item() <= eps * 3.14
    # evaluated samples are bit-identical to the full launch's, skipped ones are exactly zero
    same = (marched == full).all(-1)
    zero = (marched == 0).all(-1)
    assert bool((same | zero).all()) and abs(float(zero.float().mean()) - (1 - frac)) < 1e-6
    # a ray is only ever cut at a chunk boundary, once its transmittance is below eps
    n = 1024
    sd = scene['syn'].state_numpy(net)
    on, dn, zn = o[:n].cpu().numpy(), d[:n].cpu().numpy(), z[:n].cpu().numpy()
    pts = (on[:, None, :] + dn[:, None, :] * zn[..., None]).astype(np.float32)
    o_raw = nerf_mlp.joiner_forward(sd, JoinerSpec(), pts, np.broadcast_to(dn[:, None, :], pts.shape))
    o_rgb = compositing.raw2outputs(o_raw, zn, dn)[0]
    eo = np.abs(rgb[:n].cpu().numpy() - o_rgb).max()
    print(f"[march] eps {eps:g}: vs the CPU oracle (every sample evaluated) on the same positions: Linf {eo:.2e}")
    assert eo <= eps + 2e-5


def test_renderer_switch(scene):
    """render_utils.TERMINATION_EPS routes the background renderers' fine pass through the march; 0 restores the plain path"""
    R = scene['render']
    o, d, net = scene['o'][:2048].contiguous(), scene['d'][:2048].contiguous(), scene['net']
    a, _ = R.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128)
    old = R.TERMINATION_EPS
    try:
        R.TERMINATION_EPS = 1e-4
        trace = {}
        b, _ = R.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128, trace=trace)
    finally:
        R.TERMINATION_EPS = old
    st = trace['march'][0]
    assert st['evaluated'] < 0.8 * st['total'] and (a - b).abs().max().item() <= 1e-4
    c, _ = R.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128)
    assert torch.equal(a, c)
