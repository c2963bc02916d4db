"""The fused regularisers of the human trainer (csrc/loss.hip through neuman_hip.loss_ops) against the reference's formulas
(trainers/human_nerf_trainer.py:280-380) spelled in float64 torch on the host: the value of each term and its gradient with respect to
the network outputs."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

OFFSET = 0.31326165795326233


@pytest.fixture(scope="module")
def ops():
    from neuman_hip import loss_ops
    assert loss_ops.FUSED
    return loss_ops


def _raw(n, seed, dev):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randn((n, 4), generator=g) * 1.5
    raw[:, 3] = raw[:, 3] * 2 - 0.5                                     # densities on both sides of the ReLU
    return raw.to(dev)


def _check(name, got, want, g_got, g_want, tol=2e-6):
    got, want = got.detach(), want.detach()
    assert abs(float(got) - float(want)) <= tol * max(1.0, abs(float(want))), (name, float(got), float(want))
    for a, b in zip(g_got, g_want):
        scale = float(b.abs().max())
        if scale == 0.0:                                                 # (a single row behind the ReLU)
            assert float(a.abs().max()) == 0.0, name
            continue
        err = float((a.detach().cpu().double() - b).abs().max()) / scale
        assert err <= 5e-6, (name, err)


@pytest.mark.parametrize("n,clamp", [(1, True), (257, True), (128 * 48 + 5, True), (300000, True), (4099, False)])
def test_bimodal_prior(ops, n, clamp):
    dev = torch.device('cuda:0')
    x = (torch.rand(n, generator=torch.Generator().manual_seed(n)) * 1.6 - 0.3).to(dev).requires_grad_(True)       # some of it outside [0, 1]
    if n > 4:
        with torch.no_grad():
            x[0], x[1], x[2] = 0.0, 1.0, 0.5                                     # the clamp's edges and the prior's kink
    shaped = x.reshape(-1, 1) if n % 2 else x
    val = ops.bimodal_prior(shaped, OFFSET, clamp01=clamp)
    (val * 0.7).backward()
    xr = x.detach().cpu().double().requires_grad_(True)
    y = xr.clamp(0.0, 1.0) if clamp else xr
    want = torch.mean(-torch.log(torch.exp(-y.abs()) + torch.exp(-(1 - y).abs())) + OFFSET)
    (want * 0.7).backward()
    _check("bimodal", val, want, [x.grad], [xr.grad])


@pytest.mark.parametrize("n", [1, 1000, 2048 * 24 + 3])
def test_color_range_and_symmetry(ops, n):
    dev = torch.device('cuda:0')
    a, b = _raw(n, 1, dev).requires_grad_(True), _raw(n, 2, dev).requires_grad_(True)
    ar, br = a.detach().cpu().double().requires_grad_(True), b.detach().cpu().double().requires_grad_(True)
    val = ops.color_range(a.reshape(1, n, 4) if n > 1 else a, b, 0.1)
    (val * 1.3).backward()
    want = 0.1 * F.mse_loss(torch.sigmoid(ar[:, :3]), torch.sigmoid(br[:, :3]))
    (want * 1.3).backward()
    _check("color range", val, want, [a.grad, b.grad], [ar.grad, br.grad])
    assert float(a.grad[:, 3].abs().max()) == 0.0 and float(b.grad[:, 3].abs().max()) == 0.0
    a.grad = b.grad = ar.grad = br.grad = None
    val = ops.symmetry(a, b, 0.25)                                       # (mirrored, targets)
    val.backward()
    squash = lambda raw: torch.tanh(torch.relu(raw[..., 3]))            # noqa: E731
    want = 0.25 * F.mse_loss(squash(br), squash(ar))
    want.backward()
    _check("symmetry", val, want, [a.grad, b.grad], [ar.grad, br.grad])
    assert float(a.grad[:, :3].abs().max()) == 0.0


def test_pair_terms_on_views_of_one_output(ops):
    """the trainer hands the terms row ranges of ONE network output (torch.split): gradients flow back into the parent"""
    dev = torch.device('cuda:0')
    parent = _raw(3000, 3, dev).requires_grad_(True)
    a, b, c = torch.split(parent * 1.0, [1000, 1000, 1000], 0)
    (ops.color_range(b, a, 0.1) + ops.symmetry(c, a, 0.2)).backward()
    pr = parent.detach().cpu().double().requires_grad_(True)
    ar, br, cr = torch.split(pr, [1000, 1000, 1000], 0)
    squash = lambda raw: torch.tanh(torch.relu(raw[..., 3]))            # noqa: E731
    (0.1 * F.mse_loss(torch.sigmoid(br[:, :3]), torch.sigmoid(ar[:, :3])) + 0.2 * F.mse_loss(squash(ar), squash(cr))).backward()
    err = float((parent.grad.cpu().double() - pr.grad).abs().max() / pr.grad.abs().max())
    assert err < 5e-6, err


def _shape_ref(pred, dist_h, dummy, dist_d, w_smpl, w_dummy, factor, exponent):
    occ = lambda raw: 1 - torch.exp(-torch.relu(raw[..., 3]))           # noqa: E731

    def masked_mean(x, m):
        return x[m].mean() if bool(m.any()) else x.sum() * 0

    out = w_smpl * masked_mean((1 - occ(pred)) ** 2, dist_h < 0)
    if dummy is not None:
        out = out + w_dummy * masked_mean((1 - occ(dummy)) ** 2, dist_d < 0)
        out = out + w_dummy * masked_mean((occ(dummy) * (dist_d.abs() * factor) ** exponent).abs(), dist_d > 0)
    return out


@pytest.mark.parametrize("nh,nd,case", [(2048 * 24, 2048 * 24, "mixed"), (700, 300, "mixed"), (700, 0, "no dummy"), (500, 500, "all outside"), (500, 500, "all inside"),
                                        (3, 2, "mixed")])
def test_shape_prior(ops, nh, nd, case):
    dev = torch.device('cuda:0')
    g = torch.Generator().manual_seed(nh + nd)
    pred = _raw(nh, 5, dev).requires_grad_(True)
    dist_h = torch.randn(nh, generator=g) * 0.2
    dummy = _raw(nd, 6, dev).requires_grad_(True) if nd else None
    dist_d = (torch.randn(nd, generator=g) * 0.4) if nd else None
    if case == "all outside":
        dist_h, dist_d = dist_h.abs() + 0.01, dist_d.abs() + 0.01
    if case == "all inside":
        dist_h, dist_d = -dist_h.abs() - 0.01, -dist_d.abs() - 0.01
    if nd > 100:
        dist_d[7] = 0.0                                                  # on the surface: in neither selection
    val = ops.shape_prior(pred.reshape(-1, 1, 4) if nh % 2 == 0 else pred, dist_h.to(dev), dummy, None if dist_d is None else dist_d.to(dev), 1.0, 0.7, 2.0, 2.0)
    (val * 0.9).backward()
    pr = pred.detach().cpu().double().requires_grad_(True)
    dr = dummy.detach().cpu().double().requires_grad_(True) if nd else None
    want = _shape_ref(pr, dist_h.double(), dr, None if dist_d is None else dist_d.double(), 1.0, 0.7, 2.0, 2.0)
    (want * 0.9).backward()
    assert abs(float(val) - float(want)) <= 2e-6 * max(1.0, abs(float(want))), (float(val), float(want))
    pairs = [(pred.grad, pr.grad)] + ([(dummy.grad, dr.grad)] if nd else [])
    for got, ref in pairs:
        ref = ref if ref is not None else torch.zeros_like(got.cpu().double())
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert float(got.abs().max()) == 0.0
        else:
            assert float((got.cpu().double() - ref).abs().max()) / scale < 5e-6
    assert np.isfinite(float(val))


def test_fused_terms_are_the_trainers_own_formulas(ops, monkeypatch):
    """NEUMAN_FUSED_LOSS=0's torch spelling inside human_trainer and the kernels give the same term on the same tensors"""
    from neuman_hip import human_trainer as ht
    dev = torch.device('cuda:0')
    x = torch.rand(4000, generator=torch.Generator().manual_seed(9)).to(dev) * 1.4 - 0.2
    fused = float(ops.bimodal_prior(x, ht.HARD_SURFACE_OFFSET))
    plain = float(ht._bimodal_prior(x.clamp(0.0, 1.0)))
    assert abs(fused - plain) < 2e-6


def test_a_nan_input_reaches_every_term(ops):
    """torch.relu / torch.clamp hand a NaN on; so must the kernels, or the trainer's NaN guard (human_nerf_trainer.py:476-478) never fires"""
    dev = torch.device('cuda:0')
    nan = float('nan')
    a, b = _raw(500, 1, dev), _raw(500, 2, dev)
    bad = a.clone()
    bad[17] = nan
    dist = -torch.ones(500, device=dev)
    x = torch.rand(500, device=dev)
    x[3] = nan
    vals = [ops.bimodal_prior(x, OFFSET), ops.color_range(bad, b, 0.1), ops.symmetry(b, bad, 0.1), ops.shape_prior(bad, dist, None, None, 1.0, 1.0, 2.0, 2.0),
            ops.shape_prior(a, dist, bad, dist, 1.0, 1.0, 2.0, 2.0), ops.shape_prior(a, dist, bad, -dist, 1.0, 1.0, 2.0, 2.0)]
    assert all(bool(torch.isnan(v)) for v in vals), [float(v) for v in vals]
