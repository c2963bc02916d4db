"""-m gpu: the fused PE + MLP kernel (MFMA split-bf16 and exact-f32 modes) vs the CPU oracle, layer by layer."""
import numpy as np
import pytest
import torch

from oracle import nerf_mlp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_nets(nets):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return {k: (j.cuda(), sd, spec) for k, (j, sd, spec) in nets.items()}


def cu(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', torch.float32).contiguous()


def sample_inputs(n, seed=11):
    rng = np.random.default_rng(seed)
    pts = rng.uniform(-1.5, 1.5, size=(n, 3)).astype(np.float32)
    dirs = rng.normal(size=(n, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    return pts, dirs


def report(tag, got, ref):
    e = np.abs(got - ref)
    print(f"[mlp] {tag}: max abs err {e.max():.3e} (ref scale {np.abs(ref).max():.3e}), mean {e.mean():.3e}")
    return e.max()


@pytest.mark.parametrize("seed", [0, 2])
@pytest.mark.parametrize("prec", ["fp32", "bf16x3", "fp16x3"])
def test_stage_by_stage(gpu_nets, seed, prec):
    """Every intermediate the kernel can dump (PE, the eight hidden layers, feature, views) vs the oracle."""
    j, sd, spec = gpu_nets[seed]
    pts, dirs = sample_inputs(300)
    out, hidden = nerf_mlp.joiner_forward(sd, spec, pts, dirs, return_hidden=True)
    x_pe = nerf_mlp.embed(pts, spec.mapping, *spec.pos)
    pe = j.forward_debug(cu(pts), cu(dirs), -1, precision=prec).cpu().numpy()
    # rotate PE: the argument x.B^T reaches ~1e3 rad, one f32 ulp of it is 6e-5 and the summation order is BLAS's
    pe_tol = (3e-4 if spec.mapping == 'rotate' else 2e-6) + (1.6e-5 if prec == "bf16x3" else 0)
    assert report(f"seed{seed} {prec} PE", pe[:, :63], x_pe) < pe_tol
    assert np.abs(pe[:, 63]).max() == 0
    scale = 30 if spec.mapping == 'rotate' else 1          # error inherited from the rotate PE argument
    for st in range(10):
        got = j.forward_debug(cu(pts), cu(dirs), st, precision=prec).cpu().numpy()
        ref = hidden[st]
        assert got.shape == ref.shape
        tol = (6e-5 if prec == "bf16x3" else 2e-5) * scale * max(1.0, np.abs(ref).max())      # fp16x3 is held to the f32 kernel's bound
        assert report(f"seed{seed} {prec} stage {st}", got, ref) < tol, f"stage {st}"
    got = j(cu(pts), cu(dirs), precision=prec).cpu().numpy()
    assert report(f"seed{seed} {prec} rgb", got[:, :3], out[:, :3]) < 1e-4 * scale
    assert report(f"seed{seed} {prec} sigma", got[:, 3], out[:, 3]) < 2e-4 * scale * max(1.0, np.abs(out[:, 3]).max())


def test_golden_outputs(gpu_nets, golden):
    g = golden['mlp']
    for seed, mapping in [(0, 'posenc'), (2, 'rotate')]:
        j = gpu_nets[seed][0]
        got = j(cu(g['pts']), cu(g['dirs'])).cpu().numpy()
        ref = g[f'{mapping}_out']
        s = 30 if mapping == 'rotate' else 1
        assert report(f"golden {mapping} rgb", got[:, :3], ref[:, :3]) < 1e-4 * s
        assert report(f"golden {mapping} sigma", got[:, 3], ref[:, 3]) < 2e-4 * s * max(1.0, np.abs(ref[:, 3]).max())


@pytest.mark.parametrize("n", [1, 31, 127, 128, 129, 1000, 128 * 300 + 5])
def test_ragged_sizes_and_entry_points(gpu_nets, n):
    j, sd, spec = gpu_nets[0]
    pts, dirs = sample_inputs(n, seed=n)
    a = j(cu(pts), cu(dirs))
    b = j(cu(pts), cu(dirs), precision="fp32")
    assert a.shape == (n, 4) and torch.isfinite(a).all()
    assert (a[:, :3] - b[:, :3]).abs().max() < 5e-5
    assert ((a[:, 3] - b[:, 3]).abs() / (1 + b[:, 3].abs())).max() < 2e-4
    if n <= 1000:
        ref = nerf_mlp.joiner_forward(sd, spec, pts, dirs)
        assert np.abs(a.cpu().numpy()[:, :3] - ref[:, :3]).max() < 1e-4
    # [..., 3] leading shapes are kept
    if n == 1000:
        c = j(cu(pts).reshape(10, 100, 3), cu(dirs).reshape(10, 100, 3))
        assert c.shape == (10, 100, 4) and torch.equal(c.reshape(-1, 4), a)


def test_forward_rays_equals_forward_on_built_points(gpu_nets):
    """The fused point construction must be bit-identical to ray_to_samples + forward, and sigma_scale = `*= interval_comp`."""
    from neuman_hip import ray_utils
    j = gpu_nets[2][0]
    rng = np.random.default_rng(3)
    R, S = 77, 48
    o = cu(rng.normal(size=(R, 3)) * 0.3)
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d = cu(d / np.linalg.norm(d, axis=1, keepdims=True))
    near, far = cu(rng.uniform(0.1, 0.5, R)), cu(rng.uniform(1.0, 2.0, R))
    pts, dirs, z = ray_utils.sample_z(o, d, near, far, S, want_points=True)
    a = j.forward_rays(o, d, z)
    b = j(pts, dirs)
    assert torch.equal(a, b)
    c = j.forward_rays(o, d, z, sigma_scale=0.7)
    assert torch.equal(c[..., :3], a[..., :3]) and torch.equal(c[..., 3], a[..., 3] * 0.7)
    for prec in ("fp32",):
        assert torch.equal(j.forward_rays(o, d, z, precision=prec), j(pts, dirs, precision=prec))


def test_weight_cache_tracks_parameter_updates(gpu_nets):
    j, sd, spec = gpu_nets[1]
    pts, dirs = sample_inputs(200)
    a = j(cu(pts), cu(dirs))
    saved = j.nerf.rgb_linear.bias.detach().clone()
    with torch.no_grad():
        j.nerf.rgb_linear.bias.add_(0.25)
    b = j(cu(pts), cu(dirs))
    assert (b[:, :3] - a[:, :3] - 0.25).abs().max() < 1e-5
    with torch.no_grad():
        j.nerf.rgb_linear.bias.copy_(saved)
    assert torch.equal(j(cu(pts), cu(dirs)), a)


def test_bf16_fast_mode_is_sane_but_not_parity_grade(gpu_nets):
    j, sd, spec = gpu_nets[0]
    pts, dirs = sample_inputs(4096)
    ref = j(cu(pts), cu(dirs), precision="fp32")
    fast = j(cu(pts), cu(dirs), precision="bf16")
    par = j(cu(pts), cu(dirs), precision="bf16x3")
    e_fast = (fast[:, :3] - ref[:, :3]).abs().max().item()
    e_par = (par[:, :3] - ref[:, :3]).abs().max().item()
    print(f"[mlp] rgb(raw) max err vs fp32: bf16 {e_fast:.3e}, bf16x3 {e_par:.3e}")
    assert e_par < 5e-5 and e_fast < 0.5 and e_par < e_fast


def test_full_size_bf16x3_vs_fp32_kernel(gpu_nets):
    """BASELINE-size sample count through two independent device implementations (MFMA split-bf16 vs VALU f32)."""
    j = gpu_nets[0][0]
    g = torch.Generator(device='cuda').manual_seed(0)
    n = 1 << 20
    pts = (torch.rand((n, 3), device='cuda', generator=g) * 3 - 1.5).contiguous()
    dirs = torch.nn.functional.normalize(torch.randn((n, 3), device='cuda', generator=g), dim=-1).contiguous()
    a = j(pts, dirs, precision="bf16x3")
    b = j(pts, dirs, precision="fp32")
    e_rgb = (a[:, :3] - b[:, :3]).abs().max().item()
    e_sig = ((a[:, 3] - b[:, 3]).abs() / (1 + b[:, 3].abs())).max().item()
    print(f"[mlp] 1M samples: raw rgb max err {e_rgb:.3e}, sigma rel err {e_sig:.3e}")
    assert e_rgb < 1e-4 and e_sig < 3e-4
    # linearity in the last layer: rgb(raw) is affine in rgb_linear.bias -- a size-independent property
    saved = j.nerf.rgb_linear.bias.detach().clone()
    with torch.no_grad():
        j.nerf.rgb_linear.bias.add_(1.0)
    c = j(pts, dirs, precision="bf16x3")
    with torch.no_grad():
        j.nerf.rgb_linear.bias.copy_(saved)
    assert (c[:, :3] - a[:, :3] - 1.0).abs().max() < 1e-5 and torch.equal(c[:, 3], a[:, 3])


# ---------------------------------------------------------------------------------------------------------------------
# NM_PREC_I8X3: hidden layers as per-row-scaled int16 (two int8 limbs) on the i8 MFMA
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 2])
def test_i8x3_stage_by_stage(gpu_nets, seed):
    j, sd, spec = gpu_nets[seed]
    pts, dirs = sample_inputs(300)
    out, hidden = nerf_mlp.joiner_forward(sd, spec, pts, dirs, return_hidden=True)
    scale = 30 if spec.mapping == 'rotate' else 1
    for st in range(10):
        got = j.forward_debug(cu(pts), cu(dirs), st, precision="i8x3").cpu().numpy()
        ref = hidden[st]
        # 16-bit fixed point per row: absolute error ~ rowmax * 2^-16 per requantisation, accumulated over the layers
        tol = 3e-4 * scale * max(1.0, np.abs(ref).max())
        assert report(f"seed{seed} i8x3 stage {st}", got, ref) < tol, f"stage {st}"
    got = j(cu(pts), cu(dirs), precision="i8x3").cpu().numpy()
    assert report(f"seed{seed} i8x3 rgb", got[:, :3], out[:, :3]) < 4e-4 * scale
    assert report(f"seed{seed} i8x3 sigma", got[:, 3], out[:, 3]) < 2e-3 * scale * max(1.0, np.abs(out[:, 3]).max())


@pytest.mark.parametrize("n", [1, 127, 129, 1000, 128 * 300 + 5])
def test_i8x3_ragged_sizes_vs_fp32(gpu_nets, n):
    j = gpu_nets[0][0]
    pts, dirs = sample_inputs(n, seed=n)
    a = j(cu(pts), cu(dirs), precision="i8x3")
    b = j(cu(pts), cu(dirs), precision="fp32")
    assert a.shape == (n, 4) and torch.isfinite(a).all()
    assert (a[:, :3] - b[:, :3]).abs().max() < 4e-4
    assert ((a[:, 3] - b[:, 3]).abs() / (1 + b[:, 3].abs())).max() < 2e-3


def test_i8x3_composited_parity_and_full_size(gpu_nets):
    """What the contract is about: composited colours.  i8x3 vs the exact-f32 device kernel on 1 M samples (16384 rays x 64)."""
    from neuman_hip import ray_utils, render_utils
    j = gpu_nets[0][0]
    g = torch.Generator(device='cuda').manual_seed(1)
    R, S = 16384, 64
    o = torch.zeros((R, 3), device='cuda')
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g) * torch.tensor([0.3, 0.3, 0.05], device='cuda')
                                      + torch.tensor([0., 0., 1.], device='cuda'), dim=-1).contiguous()
    near, far = torch.zeros(R, device='cuda'), torch.full((R,), 3.14, device='cuda')
    _, _, z = ray_utils.sample_z(o, d, near, far, S)
    res = {}
    for prec in ("fp32", "fp16x3", "bf16x3", "i8x3"):
        raw = j.forward_rays(o, d, z, precision=prec)
        res[prec] = render_utils.raw2outputs(raw, z, d)[0]
    e16 = (res["fp16x3"] - res["fp32"]).abs().max().item()
    e3 = (res["bf16x3"] - res["fp32"]).abs().max().item()
    e8 = (res["i8x3"] - res["fp32"]).abs().max().item()
    print(f"[mlp] composited RGB Linf vs f32 kernel over {R} rays: fp16x3 {e16:.3e}, bf16x3 {e3:.3e}, i8x3 {e8:.3e}")
    assert e16 < 5e-6 and e3 < 2e-5 and e8 < 1e-4


@pytest.mark.parametrize("prec", ["fp16x3", "bf16x3", "bf16", "i8x3", "fp32"])
@pytest.mark.parametrize("R,S", [(1, 1), (3, 43), (64, 128), (700, 37), (2100, 64), (4099, 127)])     # (the last two: several tiles per workgroup)
def test_sigma_only_is_bit_identical(gpu_nets, prec, R, S):
    """nm_mlp_sigma_rays (the coarse pass of a two-pass render): sigma bit-identical to the full evaluation for every
    precision and ragged size; the colours are 0 where the colour head is skipped (bf16x3 / bf16)."""
    j = gpu_nets[0][0]
    g = torch.Generator(device='cuda').manual_seed(R * 1000 + S)
    o = torch.randn((R, 3), device='cuda', generator=g) * 0.3
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g), dim=-1)
    z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0, dim=1).values.contiguous()
    full = j.forward_rays(o, d, z, precision=prec, sigma_scale=1.7)
    dens = j.forward_rays(o, d, z, precision=prec, sigma_scale=1.7, sigma_only=True)
    assert dens.shape == full.shape == (R, S, 4)
    assert torch.equal(dens[..., 3], full[..., 3])
    if prec in ("fp16x3", "bf16x3", "bf16"):
        assert (dens[..., :3] == 0).all()
    else:
        assert torch.equal(dens, full)


def f64_network(sd, spec, pts, dirs):
    sd64 = {k: v.astype(np.float64) for k, v in sd.items()}
    x_pe = nerf_mlp.embed(pts, spec.mapping, *spec.pos).astype(np.float64)
    d_pe = nerf_mlp.embed(dirs, spec.mapping, *spec.dir).astype(np.float64)
    lin = lambda h, n: h @ sd64[f'nerf.{n}.weight'].T + sd64[f'nerf.{n}.bias']      # noqa: E731
    h = x_pe
    for i in range(8):
        h = np.maximum(lin(h, f'pts_linears.{i}'), 0)
        if i == 4:
            h = np.concatenate([x_pe, h], -1)
    sigma = lin(h, 'alpha_linear')[:, 0]
    rgb = lin(np.maximum(lin(np.concatenate([lin(h, 'feature_linear'), d_pe], -1), 'views_linears.0'), 0), 'rgb_linear')
    return rgb, sigma


def test_fp16x3_is_float32_class(gpu_nets):
    """The sampling-pass arithmetic: sigma (what places the importance samples) as close to an f64 evaluation of the network
    as a float32 sgemm evaluation is -- the CPU oracle's own distance from f64 is printed beside it -- and an order of magnitude
    closer than split bf16."""
    j, sd, spec = gpu_nets[0]
    pts, dirs = sample_inputs(4096, seed=5)
    rgb64, sig64 = f64_network(sd, spec, pts, dirs)
    ora = nerf_mlp.joiner_forward(sd, spec, pts, dirs)
    err = {}
    for prec in ("fp16x3", "bf16x3", "fp32"):
        got = j(cu(pts), cu(dirs), precision=prec).cpu().numpy()
        err[prec] = (np.abs(got[:, 3] - sig64).max(), np.abs(got[:, :3] - rgb64).max())
    e_ora = (np.abs(ora[:, 3] - sig64).max(), np.abs(ora[:, :3] - rgb64).max())
    print(f"[mlp] vs an f64 evaluation (|sigma| max {np.abs(sig64).max():.2f}): sigma / rgb max abs error  fp16x3 {err['fp16x3'][0]:.2e} / {err['fp16x3'][1]:.2e}, "
          f"bf16x3 {err['bf16x3'][0]:.2e} / {err['bf16x3'][1]:.2e}, f32 device kernel {err['fp32'][0]:.2e} / {err['fp32'][1]:.2e}, "
          f"CPU oracle (f32 sgemm) {e_ora[0]:.2e} / {e_ora[1]:.2e}")
    assert err['fp16x3'][0] < 4e-6 * max(1.0, np.abs(sig64).max()) and err['fp16x3'][1] < 2e-6
    assert err['fp16x3'][0] < 0.3 * err['bf16x3'][0]


def test_fp16x3_operand_range(gpu_nets):
    """fp16's narrow exponent range: a layer whose activations are ~1e-3 (their lo parts fall below fp16's normal range even
    after the 2^5 scaling) and one whose activations are ~1e3 (close to the scaled overflow bound 2047) must stay parity grade."""
    import copy
    j, sd, spec = gpu_nets[1]
    pts, dirs = sample_inputs(2048, seed=9)
    for name, f in (("small", 1e-3), ("large", 400.0)):
        k = copy.deepcopy(j)
        with torch.no_grad():
            k.nerf.pts_linears[2].weight.mul_(f); k.nerf.pts_linears[2].bias.mul_(f)
            k.nerf.pts_linears[3].weight.mul_(1.0 / f)
        sdk = {n: v.detach().cpu().numpy().astype(np.float32) for n, v in k.state_dict().items()}
        rgb64, sig64 = f64_network(sdk, spec, pts, dirs)
        got = k(cu(pts), cu(dirs), precision="fp16x3").cpu().numpy()
        h2 = k.forward_debug(cu(pts), cu(dirs), 2, precision="fp32").abs()
        es, er = np.abs(got[:, 3] - sig64).max(), np.abs(got[:, :3] - rgb64).max()
        print(f"[mlp] fp16x3 with layer-2 activations scaled by {f:g} (mean {h2.mean().item():.2e}, max {h2.max().item():.2e}): "
              f"sigma err {es:.2e}, rgb err {er:.2e}")
        assert np.isfinite(got).all()
        assert es < 2e-4 * max(1.0, np.abs(sig64).max()) and er < 1e-4
