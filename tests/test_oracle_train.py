"""CPU: oracle/train.py against the reference's own training loss and autograd gradients (tests/golden/train.npz)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
from neuman_hip import synthetic  # noqa: E402
from oracle import train as OT  # noqa: E402


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(ROOT, "tests", "golden", "train.npz")))


def grad_errors(grads, g, prefix):
    """per parameter tensor: the larger of check_grads' two deviations (three full rows; sum / sum of magnitudes / a fixed random projection)"""
    out = {}
    for name, gr in grads.items():
        gr = np.asarray(gr, np.float64).reshape(gr.shape[0], -1)
        rows = sorted({0, gr.shape[0] // 2, gr.shape[0] - 1})
        ref_rows, stats = g[f'{prefix}/{name}/rows'], g[f'{prefix}/{name}/stats']
        scale = max(np.abs(gr).max(), 1e-12)
        proj = np.random.default_rng(sum(map(ord, name))).normal(size=gr.size)
        mine = np.array([gr.sum(), np.abs(gr).sum(), float(gr.reshape(-1) @ proj)])
        out[name] = max(float(np.abs(gr[rows] - ref_rows).max() / scale), float(np.abs(mine - stats).max() / max(np.abs(gr).sum(), 1e-12)))
    return out


def check_grads(grads, g, prefix, tol=1e-4):
    """per parameter tensor: three full rows, the sum, the sum of magnitudes and a fixed random projection"""
    worst = 0.0
    for name, gr in grads.items():
        gr = np.asarray(gr, np.float64).reshape(gr.shape[0], -1)
        rows = sorted({0, gr.shape[0] // 2, gr.shape[0] - 1})
        ref_rows, stats = g[f'{prefix}/{name}/rows'], g[f'{prefix}/{name}/stats']
        scale = max(np.abs(gr).max(), 1e-12)
        e_rows = np.abs(gr[rows] - ref_rows).max() / scale
        proj = np.random.default_rng(sum(map(ord, name))).normal(size=gr.size)
        mine = np.array([gr.sum(), np.abs(gr).sum(), float(gr.reshape(-1) @ proj)])
        e_stats = np.abs(mine - stats).max() / max(np.abs(gr).sum(), 1e-12)
        assert e_rows < tol and e_stats < tol, f"{prefix}/{name}: rows {e_rows:.2e}, stats {e_stats:.2e}"
        worst = max(worst, e_rows, e_stats)
    return worst


@pytest.mark.parametrize("tag,white,penalty", [("white", True, 0.0), ("black_penalty", False, 0.1)])
def test_training_pass_matches_reference(G, tag, white, penalty):
    for k, (name, seed) in enumerate((("coarse", 0), ("fine", 1))):
        w = synthetic.state_numpy(synthetic.make_joiner(seed))
        p = f'{tag}/{name}'
        r = OT.training_pass(w, G['origin'], G['direction'], G[f'{p}/z'], G['color'], white, penalty, G['depth'])
        assert np.abs(r['raw'] - G[f'{p}/raw']).max() < 2e-4 * max(1.0, np.abs(G[f'{p}/raw']).max())
        np.testing.assert_allclose(r['rgb_map'], G[f'{p}/rgb_map'], atol=2e-5)
        np.testing.assert_allclose([r['loss_rgb'], r['loss_empty']], G[f'{tag}/losses'][2 * k:2 * k + 2], rtol=2e-5, atol=1e-7)
        scale = np.abs(G[f'{p}/d_raw']).max()
        assert np.abs(r['d_raw'] - G[f'{p}/d_raw']).max() < 1e-4 * scale
        worst = check_grads(r['grads'], G, p)
        print(f"[oracle train] {p}: loss {r['loss_rgb']:.6f} + {r['loss_empty']:.6f}, worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("mapping", ["posenc", "rotate"])
def test_input_gradients_match_reference(G, mapping):
    w = synthetic.state_numpy(synthetic.make_joiner(2, mapping))
    out, dp, dd = OT.input_gradients(w, G[f'in/{mapping}/pts'], G[f'in/{mapping}/dirs'], G[f'in/{mapping}/g_out'], mapping)
    assert np.abs(out - G[f'in/{mapping}/out']).max() < 2e-4 * np.abs(G[f'in/{mapping}/out']).max()
    for mine, ref in ((dp, G[f'in/{mapping}/d_pts']), (dd, G[f'in/{mapping}/d_dirs'])):
        assert np.abs(mine - ref).max() < 2e-4 * np.abs(ref).max(), np.abs(mine - ref).max() / np.abs(ref).max()


def offset_weights(scale_type):
    import torch
    from neuman_hip import vanilla
    torch.manual_seed(11)
    net = vanilla.build_offset_net(synthetic.default_opt(offset_scale=0.7, offset_scale_type=scale_type))
    return net, {k: v.detach().numpy() for k, v in net.state_dict().items()}


@pytest.mark.parametrize("scale_type", ["linear", "tanh"])
def test_offset_net_matches_reference(G, scale_type):
    _, w = offset_weights(scale_type)
    p = f'off/{scale_type}'
    out, dx, grads = OT.offset_net_gradients(w, G[f'{p}/x'], G[f'{p}/g_out'], 0.7, scale_type)
    assert np.abs(out - G[f'{p}/out']).max() < 1e-5 and np.abs(dx - G[f'{p}/d_x']).max() < 1e-4 * np.abs(G[f'{p}/d_x']).max()
    check_grads(grads, G, p)


@pytest.mark.parametrize("tag,white", [("white", True), ("black", False)])
def test_composite_backward_matches_reference(G, tag, white):
    d = OT.composite_backward(G['c/raw'], G['c/z'], G['c/d'], white, G['c/g_rgb'], G['c/g_acc'], G['c/g_depth'], G['c/g_w'])
    ref = G[f'c/{tag}/d_raw']
    assert np.abs(d - ref).max() < 2e-5 * np.abs(ref).max()
