"""-m gpu: `--include_input ''` (Embedder(include_input=False), models/vanilla.py:56-58, 63-65, 87-88) on the kernels -- rendering (zero weight columns
in the packed images), the training step (derived parameters, neuman_hip/train.py _full_input), the offset net and the time-conditioned net --
against the reference's own outputs and autograd (tests/golden/no_input.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import nerf_mlp, render as OR
from oracle.nerf_mlp import JoinerSpec

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def N():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return dict(np.load(os.path.join(HERE, "golden", "no_input.npz")))


def cu(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', torch.float32).contiguous()


def variant(seed, **over):
    from neuman_hip import synthetic
    return synthetic.make_variant_joiner(seed, include_input=False, **over).cuda()


@pytest.mark.parametrize("mapping", ["posenc", "rotate"])
def test_forward_every_arithmetic(N, mapping):
    from neuman_hip import synthetic
    j = variant(11, posenc=mapping)
    assert j.nerf.pts_linears[0].weight.shape == (256, 60) and j.nerf.views_linears[0].weight.shape == (128, 280)
    ora = nerf_mlp.joiner_forward(synthetic.state_numpy(j), JoinerSpec(mapping=mapping, include_input=False), N['pts'], N['dirs'])
    s = 30 if mapping == 'rotate' else 1          # the rotate encoding's arguments reach ~1e3 rad (tests/test_oracle_golden.py)
    scale = max(1.0, float(np.abs(ora[:, 3]).max()))
    for prec, tol in (("fp32", 2e-5), ("fp16x3", 2e-5), ("bf16x3", 1e-4), (None, 2e-5)):
        got = j(cu(N['pts']), cu(N['dirs']), precision=prec).cpu().numpy()
        e_g, e_o = np.abs(got - N[f'{mapping}_out']).max(), np.abs(got - ora).max()
        print(f"[no_input] {mapping} {prec}: vs reference golden {e_g:.2e}, vs oracle {e_o:.2e}")
        assert e_g < tol * s * 2 * scale and e_o < tol * s * scale
    sh = j(cu(N['pts']), cu(N['dirs']), role='shading').cpu().numpy()     # the 16-bit fixed-point kernel (network-output tolerances of tests/test_hip_mlp.py)
    assert np.abs(sh[:, :3] - ora[:, :3]).max() < 4e-4 * s and np.abs(sh[:, 3] - ora[:, 3]).max() < 2e-3 * s * scale
    # the fused ray form reads the same packed image
    R, S = 16, 16
    o, d = cu(N['pts'][:R] * 0.2), cu(N['dirs'][:R])
    z = torch.linspace(0.5, 2.0, S, device='cuda')[None].repeat(R, 1).contiguous()
    full = j.forward_rays(o, d, z)
    ref = j(o[:, None, :] + d[:, None, :] * z[..., None], d[:, None, :].expand(R, S, 3))
    assert (full - ref).abs().max() < 2e-5 * max(1.0, ref[..., 3].abs().max().item())


@pytest.mark.parametrize("mapping", ["posenc", "rotate"])
def test_training_step_against_the_reference_autograd(N, mapping):
    j = variant(11, posenc=mapping).train()
    p, d = cu(N['pts']).requires_grad_(True), cu(N['dirs']).requires_grad_(True)
    out = j(p, d)
    s = 30 if mapping == 'rotate' else 1
    scale = max(1.0, float(np.abs(N[f'{mapping}_out'][:, 3]).max()))
    assert np.abs(out.detach().cpu().numpy() - N[f'{mapping}_out']).max() < 4e-5 * s * scale
    ((out - cu(N['tgt'])) ** 2).mean().backward()
    named = dict(j.named_parameters())
    worst, wname = 0.0, None
    for key in [k for k in N if k.startswith(f'{mapping}_grad/')]:
        name = key.split('/', 1)[1]
        got = (p.grad if name == 'pts' else d.grad if name == 'dirs' else named[name].grad).cpu().numpy()
        want = N[key]
        assert got.shape == want.shape, (name, got.shape, want.shape)          # the real parameters' shapes: 60 / 316 / 280 columns
        e = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
        if e > worst:
            worst, wname = e, name
    print(f"[no_input] {mapping}: training step vs the reference's autograd: worst gradient deviation {worst:.2e} of a tensor's largest entry ({wname})")
    assert worst < (2e-4 if mapping == 'posenc' else 3e-3), (worst, wname)
    for prm in j.parameters():
        assert prm.grad is not None and prm.grad.shape == prm.shape and torch.isfinite(prm.grad).all()


def test_large_batch_equals_the_full_encoding_net_with_zero_columns(N, monkeypatch):
    """36000 samples (the fused 16-bit training kernels): the include_input=False net against an ordinary net whose weights are the same numbers with
    zero columns under the raw inputs -- same kernels, same operands: outputs and gradients bit for bit"""
    from neuman_hip import synthetic, train, vanilla
    monkeypatch.setattr(train, "STORE16_MIN_ROWS", 32768)
    j = variant(11).train()
    k = synthetic.make_variant_joiner(3).cuda().train()
    wide = vanilla.with_absent_columns([q.detach() for q in j.nerf.ordered_params()], vanilla.absent_input_columns(j.pos_pe, j.dir_pe, j.nerf))
    with torch.no_grad():
        for dst, src in zip(k.nerf.ordered_params(), wide):
            dst.copy_(src)
    n = 36000
    g = torch.Generator(device='cuda').manual_seed(5)
    pts = (torch.rand((n, 3), device='cuda', generator=g) * 2 - 1)
    dirs = F.normalize(torch.randn((n, 3), device='cuda', generator=g), dim=-1)
    tgt = torch.rand((n, 4), device='cuda', generator=g)
    res = []
    for net in (j, k):
        out = net(pts, dirs)
        ((out - tgt) ** 2).mean().backward()
        res.append((out.detach(), [q.grad for q in net.nerf.ordered_params()]))
    assert torch.equal(res[0][0], res[1][0])
    pads = vanilla.absent_input_columns(j.pos_pe, j.dir_pe, j.nerf)
    for i, (a, b) in enumerate(zip(res[0][1], res[1][1])):
        if i in pads:
            at, cnt = pads[i]
            b = torch.cat([b[:, :at], b[:, at + cnt:]], 1)
        assert a.shape == b.shape and torch.equal(a, b), i
    # and the views head alone through forward_two_views' fallback: two plain calls
    assert j.forward_two_views(pts, dirs, dirs) is None
    # two forwards of one iteration before its one backward pass (the human trainer's query sets): the second must not disturb what the first's backward reads
    for q in j.parameters():
        q.grad = None
    a, b = j(pts[:33000], dirs[:33000]), j(pts[3000:], dirs[3000:])
    (((a - tgt[:33000]) ** 2).mean() + ((b - tgt[3000:]) ** 2).mean()).backward()
    assert all(q.grad is not None and torch.isfinite(q.grad).all() for q in j.parameters())


def test_plain_head_frames_and_time_conditioned(N):
    from neuman_hip import render_utils, synthetic
    j = variant(12, use_viewdirs=False)
    got = j(cu(N['pts']), None).cpu().numpy()
    e = np.abs(got - N['plain_out']).max()
    print(f"[no_input] plain head vs reference golden: {e:.2e}")
    assert e < 4e-5
    coarse, fine = variant(14), variant(15)
    cap = synthetic.SimpleCapture(24, 18, fx=30.)
    rgb1 = render_utils.render_vanilla(coarse, cap, None, rays_per_batch=256, samples_per_ray=16)
    e1 = np.abs(rgb1 - N['frame_coarse_only_rgb']).max()
    rgb, depth = render_utils.render_vanilla(coarse, cap, fine, rays_per_batch=256, samples_per_ray=16, importance_samples_per_ray=16, return_depth=True)
    err = np.abs(rgb - N['frame_rgb']).max(-1)
    print(f"[no_input] frames vs reference golden: coarse-only Linf {e1:.2e}; two-pass rays > 1e-4: {(err > 1e-4).sum()} / {err.size}, Linf {err.max():.2e}")
    assert e1 < 1e-4 and (err > 1e-4).mean() < 0.02 and err.max() < 2e-2
    spec = JoinerSpec(include_input=False)
    o_rgb = OR.render_vanilla((synthetic.state_numpy(coarse), spec), cap, (synthetic.state_numpy(fine), spec), rays_per_batch=256, samples_per_ray=16,
                              importance_samples_per_ray=16)
    assert (np.abs(rgb - o_rgb).max(-1) > 1e-4).mean() < 0.02
    # the time-conditioned net: 80-wide encoding; a frame's time folded into two bias vectors (vanilla.frozen_time_joiner), 4-D points on the GEMM chain
    tc = variant(16, raw_pos_dim=4)
    assert tc.pos_pe.out_dim == 80
    x4 = cu(np.concatenate([N['pts'], np.full((N['pts'].shape[0], 1), 0.35, np.float32)], 1))
    with torch.no_grad():
        e4 = np.abs(tc(x4, cu(N['dirs'])).cpu().numpy() - N['nerft_out']).max()
    cap.frame_id = {'frame_id': 7, 'total_frames': 20}
    rgbt = render_utils.render_vanilla(tc, cap, None, rays_per_batch=256, samples_per_ray=16, ablate_nerft=True)
    et = np.abs(rgbt - N['nerft_coarse_only_rgb']).max()
    print(f"[no_input] time-conditioned net: 4-D points {e4:.2e}, coarse-only frame {et:.2e} vs reference golden")
    assert e4 < 2e-5 and et < 1e-4


@pytest.mark.parametrize("fused", [False, True])
def test_offset_net(N, fused, monkeypatch):
    """OffsetNet over the 80-wide space-time encoding (vanilla.py:180-205), tanh scale: output and the reference's autograd gradients; `fused`: the
    constant-time form on the 16-bit training kernels (a batch below their size threshold would take the chain: the threshold is lowered)"""
    from neuman_hip import synthetic, train, vanilla
    if fused:
        monkeypatch.setattr(train, "STORE16_MIN_ROWS", 256)
    torch.manual_seed(13)
    net = vanilla.build_offset_net(synthetic.default_opt(include_input=False, offset_scale=0.05, offset_scale_type='tanh')).cuda().train()
    cs = float(sum(v.abs().sum(dtype=torch.float64) for v in net.state_dict().values()))
    np.testing.assert_allclose(cs, N['offset_checksum'][0], rtol=1e-6)
    assert net.pos_pe.out_dim == 80 and net.nerf.pts_linears[0].weight.shape == (256, 80)
    x4 = cu(N['offset_x4'])
    if fused:
        assert train._offset_fused_ok(net, x4, x4.shape[0])
    out = net(x4, const_time=0.35 if fused else None)
    e = np.abs(out.detach().cpu().numpy() - N['offset_out']).max() / np.abs(N['offset_out']).max()
    (out * cu(N['tgt'][:, :3])).sum().backward()
    named = dict(net.named_parameters())
    worst, wname = 0.0, None
    for key in [k for k in N if k.startswith('offset_grad/')]:
        name = key.split('/', 1)[1]
        got, want = named[name].grad.cpu().numpy(), N[key]
        assert got.shape == want.shape
        ee = np.abs(got - want).max() / np.abs(want).max()
        if ee > worst:
            worst, wname = ee, name
    print(f"[no_input] offset net (fused={fused}): output {e:.2e} of its largest value, worst gradient {worst:.2e} ({wname}) vs the reference's autograd")
    # (260 samples, default-initialised weights: a gradient entry is a short cancelling sum; the chain's products are the default mixed16 ones)
    assert e < 2e-5 and worst < (2e-3 if fused else 5e-4), (e, worst, wname)
