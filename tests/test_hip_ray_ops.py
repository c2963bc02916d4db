"""-m gpu: per-ray HIP kernels (through the C ABI) vs the CPU oracle and the reference goldens."""
import numpy as np
import pytest
import torch

from oracle import compositing, ray_ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def H():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from neuman_hip import ray_utils, render_utils
    import types
    return types.SimpleNamespace(ray=ray_utils, render=render_utils)


def cu(x, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', dtype).contiguous()


def rays(rng, R):
    o = rng.normal(size=(R, 3)).astype(np.float32)
    d = rng.normal(size=(R, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    near = rng.uniform(0.1, 1.0, size=(R, 1)).astype(np.float32)
    far = (near + rng.uniform(0.5, 3.0, size=(R, 1))).astype(np.float32)
    return o, d, near, far


def tie_aware(a, ref, atol, max_frac, max_jump):
    bad = np.abs(a - ref) > atol
    assert bad.mean() <= max_frac, f"{bad.sum()}/{bad.size} differ"
    assert np.abs(a - ref).max() <= max_jump
    return bad


@pytest.mark.parametrize("R,S", [(37, 32), (1, 3), (1000, 128), (3, 200)])
def test_ray_to_samples_vs_oracle(H, R, S):
    rng = np.random.default_rng(R * 1000 + S)
    o, d, near, far = rays(rng, R)
    batch = {'origin': cu(o), 'direction': cu(d), 'near': cu(near), 'far': cu(far)}
    t = torch.linspace(0., 1., steps=S, device='cuda').cpu().numpy()
    for kw in ({}, {'lindisp': True}):
        p, dd, z = H.ray.ray_to_samples(batch, S, device='cuda', **kw)
        op, od, oz = O.ray_to_samples(o, d, near, far, S, t_vals=t, **kw)
        np.testing.assert_array_equal(z.cpu().numpy(), oz)          # same t, same two-rounding lerp: bit exact
        np.testing.assert_array_equal(p.cpu().numpy(), op)
        np.testing.assert_array_equal(dd.cpu().numpy(), od)
    torch.manual_seed(3)
    t_rand = torch.clip(torch.rand((R, S), device='cuda'), min=0.01, max=0.99).cpu().numpy()
    torch.manual_seed(3)
    p, dd, z = H.ray.ray_to_samples(batch, S, perturb=1.0, device='cuda')
    op, od, oz = O.ray_to_samples(o, d, near, far, S, t_vals=t, t_rand=t_rand)
    np.testing.assert_array_equal(z.cpu().numpy(), oz)
    np.testing.assert_array_equal(p.cpu().numpy(), op)


def test_ray_to_samples_golden(H, golden):
    g = golden['ray_ops']
    batch = {k: cu(g[v]) for k, v in [('origin', 'rs_o'), ('direction', 'rs_d'), ('near', 'rs_near'), ('far', 'rs_far')]}
    for tag, kw in [('lin', {}), ('disp', {'lindisp': True})]:
        p, dd, z = H.ray.ray_to_samples(batch, 32, device='cuda', **kw)
        np.testing.assert_allclose(z.cpu().numpy(), g[f'rs_{tag}_z'], atol=5e-7)
        np.testing.assert_allclose(p.cpu().numpy(), g[f'rs_{tag}_pts'], atol=2e-6)


@pytest.mark.parametrize("R,S", [(29, 32), (1, 2), (5, 64), (7, 65), (300, 128), (11, 896)])   # S = 1 breaks the reference itself (:86)
def test_composite_vs_oracle(H, R, S):
    rng = np.random.default_rng(S)
    raw = (rng.normal(size=(R, S, 4)) * np.array([1, 1, 1, 5])).astype(np.float32)
    z = np.sort(rng.uniform(0, 3.14, size=(R, S)).astype(np.float32), axis=1)
    d = rng.normal(size=(R, 3)).astype(np.float32)
    for wb in (True, False):
        rgb, disp, acc, w, depth = [x.cpu().numpy() for x in H.render.raw2outputs(cu(raw), cu(z), cu(d), white_bkg=wb)]
        o_rgb, o_disp, o_acc, o_w, o_depth = compositing.raw2outputs(raw, z, d, white_bkg=wb)
        np.testing.assert_allclose(w, o_w, atol=5e-7)
        np.testing.assert_allclose(rgb, o_rgb, atol=3e-6)
        np.testing.assert_allclose(acc, o_acc, atol=3e-6)
        np.testing.assert_allclose(depth, o_depth, atol=1e-5)
        np.testing.assert_allclose(disp, o_disp, rtol=2e-5)
        assert (w >= 0).all() and acc.max() <= 1 + 1e-5


def test_composite_golden_and_edge_cases(H, golden):
    g = golden['ray_ops']
    for tag, wb in [('white', True), ('black', False)]:
        rgb, disp, acc, w, depth = [x.cpu().numpy() for x in H.render.raw2outputs(cu(g['c_raw']), cu(g['c_z']), cu(g['c_d']), white_bkg=wb)]
        np.testing.assert_allclose(w, g[f'c_{tag}_w'], atol=5e-7)
        np.testing.assert_allclose(rgb, g[f'c_{tag}_rgb'], atol=3e-6)
        np.testing.assert_allclose(depth, g[f'c_{tag}_depth'], atol=1e-5)
    # all-negative sigma: alpha = 0 everywhere -> acc 0, white background, disp = 1/max(1e-10, 0/0) = NaN like torch
    raw = np.zeros((4, 16, 4), np.float32)
    raw[..., 3] = -1.0
    z = np.tile(np.linspace(0, 1, 16, dtype=np.float32), (4, 1))
    d = np.tile(np.array([[0, 0, 1]], np.float32), (4, 1))
    rgb, disp, acc, w, depth = [x.cpu().numpy() for x in H.render.raw2outputs(cu(raw), cu(z), cu(d))]
    assert (acc == 0).all() and (rgb == 1).all() and np.isnan(disp).all()
    # sigma > 0 only on the last sample: the 1e10 interval makes it opaque (alpha = 1)
    raw[..., -1, 3] = 1e-3
    rgb, disp, acc, w, depth = [x.cpu().numpy() for x in H.render.raw2outputs(cu(raw), cu(z), cu(d))]
    np.testing.assert_allclose(acc, 1.0, atol=1e-6)
    np.testing.assert_allclose(w[:, -1], 1.0, atol=1e-6)
    # empty batch
    out = H.render.raw2outputs(torch.zeros((0, 8, 4), device='cuda'), torch.zeros((0, 8), device='cuda'), torch.zeros((0, 3), device='cuda'))
    assert out[0].shape == (0, 3)


def test_sample_pdf_vs_oracle_and_golden(H, golden):
    g = golden['ray_ops']
    s = H.ray.sample_pdf(cu(g['pdf_bins']), cu(g['pdf_w']), 16, det=True).cpu().numpy()
    jump = np.diff(g['pdf_bins'], axis=1).max()
    tie_aware(s, g['pdf_samples'], 2e-6, 0.015, jump)
    tie_aware(s, O.sample_pdf(g['pdf_bins'], g['pdf_w'], 16), 2e-6, 0.01, jump)
    with pytest.raises(NotImplementedError):
        H.ray.sample_pdf(cu(g['pdf_bins']), cu(g['pdf_w']), 16, det=False)


@pytest.mark.parametrize("R,S,N", [(37, 32, 24), (200, 128, 128), (5, 192, 128), (3, 3, 1), (9, 320, 128)])
def test_importance_z_vs_oracle(H, R, S, N):
    rng = np.random.default_rng(S + N)
    o, d, near, far = rays(rng, R)
    _, _, z = O.ray_to_samples(o, d, near, far, S)
    w = (rng.uniform(size=(R, S)) ** 6).astype(np.float32)
    w[0] = 0.0
    gap = np.diff(z, axis=1).max()
    for inc in (True, False):
        hz = H.ray.importance_z(cu(z), cu(w), N, including_old=inc).cpu().numpy()
        _, _, oz = O.ray_to_importance_samples(o, d, z, w, N, including_old=inc)
        assert hz.shape == oz.shape
        if inc:
            assert (np.diff(hz, axis=1) >= 0).all()                 # sortedness: size-independent property
            # the S old samples must all be present, exactly
            for r in range(min(R, 8)):
                assert np.isin(z[r], hz[r]).all()
        tie_aware(hz, oz, 3e-6, 0.01, gap)
    batch = {'origin': cu(o), 'direction': cu(d)}
    p, dd, hz = H.ray.ray_to_importance_samples(batch, cu(z), cu(w), N, device='cuda')
    np.testing.assert_array_equal(p.cpu().numpy(), (o[:, None, :] + d[:, None, :] * hz.cpu().numpy()[..., None]).astype(np.float32))


@pytest.mark.parametrize("S,N", [(128, 128), (32, 32), (192, 320), (65, 130)])
def test_importance_merge_is_exactly_torch_sort(H, S, N):
    """including_old=True must be sort(cat(z, z_samples)) of the kernel's OWN inverse-CDF samples, bit for bit --
    also where those samples come out with f32 inversions (flat, near-zero and spiky PDFs: the adversarial cases)."""
    rng = np.random.default_rng(S * 1000 + N)
    R = 600
    z = np.sort(rng.uniform(0.0, 3.14, size=(R, S)).astype(np.float32), axis=1)
    z[: R // 6] = np.linspace(0.0, 3.14, S, dtype=np.float32)[None]                          # the renderers' own spacing
    w = np.zeros((R, S), np.float32)
    k = R // 6
    w[:k] = 0.0                                                                               # flat: every bin at the 1e-5 floor
    w[k:2 * k] = rng.uniform(0, 1e-7, size=(k, S))                                            # near zero
    w[2 * k:3 * k] = rng.uniform(0, 1, size=(k, S)) ** 8                                      # spiky
    w[3 * k:4 * k, S // 2] = 1.0                                                              # one opaque sample
    w[4 * k:5 * k] = rng.uniform(0, 1 / S, size=(k, S))                                       # diffuse
    w[5 * k:] = np.where(rng.uniform(size=(R - 5 * k, S)) < 0.1, rng.uniform(0, 0.2, size=(R - 5 * k, S)), 0).astype(np.float32)
    zt, wt = cu(z), cu(w)
    merged = H.ray.importance_z(zt, wt, N, including_old=True)
    alone = H.ray.importance_z(zt, wt, N, including_old=False)
    inv = (alone[:, 1:] < alone[:, :-1]).any(1)
    ref = torch.sort(torch.cat([zt, alone], dim=1), dim=1).values
    print(f"[importance] S={S} N={N}: rays whose inverse-CDF samples carry an f32 inversion: {int(inv.sum())} / {R}")
    assert torch.equal(merged, ref)


def test_importance_golden(H, golden):
    g = golden['ray_ops']
    hz = H.ray.importance_z(cu(g['rs_lin_z']), cu(g['imp_w']), 24).cpu().numpy()
    tie_aware(hz, g['imp_z'], 3e-6, 0.015, np.diff(g['rs_lin_z'], axis=1).max())


def test_near_far_and_compaction(H, golden):
    g = golden['ray_ops']
    n, f = H.ray.geometry_guided_near_far(cu(g['nf_o']), cu(g['nf_d']), cu(g['nf_verts']), 0.2)
    n, f = n.cpu().numpy(), f.cpu().numpy()
    hit = g['nf_near_torch'] < g['nf_far_torch']
    np.testing.assert_array_equal(n < f, hit)
    np.testing.assert_allclose(n[hit], g['nf_near_torch'][hit], atol=2e-5)
    np.testing.assert_allclose(f[hit], g['nf_far_torch'][hit], atol=2e-5)
    assert np.isposinf(n[~hit]).all() and np.isneginf(f[~hit]).all()
    # numpy in -> numpy out (reference dispatch on the type of `orig`)
    n2, f2 = H.ray.geometry_guided_near_far(g['nf_o'], g['nf_d'], g['nf_verts'], 0.2)
    assert isinstance(n2, np.ndarray)
    np.testing.assert_array_equal(n2, n)
    # bigger, vs the oracle, with V = 6890
    from neuman_hip import synthetic
    rng = np.random.default_rng(5)
    verts = synthetic.human_vertex_cloud(0)
    R = 3000
    o = np.tile(np.array([[0, 0, -3.]], np.float32), (R, 1))
    d = rng.normal(size=(R, 3)).astype(np.float32) * np.array([0.25, 0.4, 0.05], np.float32) + np.array([0, 0, 1], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    n, f = H.ray.geometry_guided_near_far(cu(o), cu(d), cu(verts), 0.2)
    # the device's discriminant is float64 (csrc/nearfar.hip): held to the float64 evaluation of the reference's expression at float32 rounding of the
    # result, and to the float32 one at what float32's cancellation leaves (the reference's own torch and numpy branches differ by as much)
    on, of = O.geometry_guided_near_far(o, d, verts, 0.2, dtype=np.float64)
    on32, of32 = O.geometry_guided_near_far(o, d, verts, 0.2)
    hn, hf = n.cpu().numpy(), f.cpu().numpy()
    flips = (hn < hf) != (on < of)
    assert flips.sum() == 0
    both = (hn < hf) & (on < of)
    assert 0.2 < both.mean() < 0.95
    np.testing.assert_allclose(hn[both], on[both], atol=1e-6)
    np.testing.assert_allclose(hf[both], of[both], atol=1e-6)
    b32 = both & (on32 < of32)
    e32 = np.maximum(np.abs(hn - on32), np.abs(hf - of32))[b32]
    print(f"[near / far] device vs the float64 evaluation: {np.abs(hn[both] - on[both]).max():.1e}; vs the float32 one: 99 % {np.percentile(e32, 99):.1e}, max {e32.max():.1e}")
    assert ((on32 < of32) != (on < of)).mean() < 2e-3 and np.percentile(e32, 99) < 1e-4 and e32.max() < 1e-3
    hit_idx, miss_idx = H.ray.compact_hits(n, f)
    np.testing.assert_array_equal(hit_idx.cpu().numpy(), np.nonzero(hn < hf)[0])           # integer work: bit exact, ascending
    np.testing.assert_array_equal(miss_idx.cpu().numpy(), np.nonzero(~(hn < hf))[0])
    # gather / scatter round trip
    rows = H.ray.gather_rows(cu(d), hit_idx)
    np.testing.assert_array_equal(rows.cpu().numpy(), d[hn < hf])
    dst = torch.zeros((R, 3), device='cuda')
    H.ray.scatter_rows(dst, hit_idx, rows)
    exp = np.zeros((R, 3), np.float32)
    exp[hn < hf] = d[hn < hf]
    np.testing.assert_array_equal(dst.cpu().numpy(), exp)
    # all-miss and all-hit batches, and sizes around the block size
    for R2 in (1, 255, 256, 257, 1025):
        nn = torch.zeros(R2, device='cuda')
        ff = torch.ones(R2, device='cuda')
        h, m = H.ray.compact_hits(nn, ff)
        assert h.numel() == R2 and m.numel() == 0 and torch.equal(h.cpu(), torch.arange(R2, dtype=torch.int32))
        h, m = H.ray.compact_hits(ff, nn)
        assert h.numel() == 0 and m.numel() == R2


def test_near_far_wave_skip_is_bit_identical(H):
    """near_far_kernel skips the vertex loop for a wave whose 64 rays all pass the body's bounding sphere at more than radius + tau.
    The same rays once grouped by distance from the body (whole waves far away: skipped) and once interleaved with a ray through the
    body's centre in every wave (nothing skipped) must give the same near / far bit for bit -- grazing rays at the 0.2 shell included --
    and so must rays seen from far away, non-unit directions (never skipped) and a single vertex."""
    from neuman_hip import synthetic
    rng = np.random.default_rng(11)
    verts = (synthetic.human_vertex_cloud(0) + np.array([0.3, -0.1, 0.2], np.float32)).astype(np.float32)
    for cam_z, spread in ((-3.0, 0.6), (-40.0, 0.05), (-3.0, 0.25)):
        R = 64 * 300
        o = np.tile(np.array([[0.2, 0.0, cam_z]], np.float32), (R, 1))
        tgt = rng.normal(size=(R, 3)).astype(np.float32) * spread * abs(cam_z) * np.array([1.0, 1.0, 0.0], np.float32) + verts.mean(0)
        d = tgt - o
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        miss_dist = np.linalg.norm(np.cross(verts.mean(0) - o, d), axis=1)
        order = np.argsort(miss_dist)                                       # grouped: the last waves are far from the body
        o_g, d_g = o[order], d[order]
        n_g, f_g = H.ray.geometry_guided_near_far(cu(o_g), cu(d_g), cu(verts), 0.2)
        # interleaved: every wave of 64 holds one ray through the centre in lane 0
        R2 = (R // 63) * 64
        keep = R // 63 * 63
        o_i, d_i = np.empty((R2, 3), np.float32), np.empty((R2, 3), np.float32)
        o_i[:], d_i[:] = o[0], (verts.mean(0) - o[0]) / np.linalg.norm(verts.mean(0) - o[0])
        slots = np.arange(R2)[np.arange(R2) % 64 != 0]
        o_i[slots], d_i[slots] = o_g[:keep], d_g[:keep]
        n_i, f_i = H.ray.geometry_guided_near_far(cu(o_i), cu(d_i), cu(verts), 0.2)
        n_g, f_g, n_i, f_i = (x.cpu().numpy() for x in (n_g, f_g, n_i, f_i))
        np.testing.assert_array_equal(n_g[:keep], n_i[slots])
        np.testing.assert_array_equal(f_g[:keep], f_i[slots])
        hit = n_g < f_g
        assert 0.02 < hit.mean() < 0.999 and np.isposinf(n_g[~hit]).all() and np.isneginf(f_g[~hit]).all()
    # directions that are not unit vectors: the reference's expression as it is, never skipped
    d2 = (d_g * 1.7).astype(np.float32)
    n_a, f_a = H.ray.geometry_guided_near_far(cu(o_g), cu(d2), cu(verts), 0.2)
    on, of = O.geometry_guided_near_far(o_g, d2, verts, 0.2)
    assert ((n_a.cpu().numpy() < f_a.cpu().numpy()) != (on < of)).mean() < 2e-3
    # one vertex: the sphere has radius 0
    n_1, f_1 = H.ray.geometry_guided_near_far(cu(o_g), cu(d_g), cu(verts[:1]), 0.2)
    on, of = O.geometry_guided_near_far(o_g, d_g, verts[:1], 0.2)
    np.testing.assert_array_equal(n_1.cpu().numpy() < f_1.cpu().numpy(), on < of)


@pytest.mark.parametrize("R,Sa,Sb", [(50, 256, 128), (3, 1, 1), (17, 320, 192), (4, 512, 192), (2, 704, 192)])
def test_merge_sorted_vs_oracle(H, R, Sa, Sb):
    rng = np.random.default_rng(Sa + Sb)
    za = np.sort(rng.uniform(0, 3, size=(R, Sa)).astype(np.float32), axis=1)
    zb = np.sort(rng.uniform(1, 2, size=(R, Sb)).astype(np.float32), axis=1)
    ra = rng.normal(size=(R, Sa, 4)).astype(np.float32)
    rb = rng.normal(size=(R, Sb, 4)).astype(np.float32)
    z, raw = H.render.merge_sorted(cu(za), cu(ra), cu(zb), cu(rb))
    oz, oraw = compositing.merge_sorted([za, zb], [ra, rb])
    np.testing.assert_array_equal(z.cpu().numpy(), oz)              # a permutation of the inputs: bit exact
    np.testing.assert_array_equal(raw.cpu().numpy(), oraw)
    # ties: list a first
    zb2 = za[:, :Sb].copy() if Sb <= Sa else zb
    z, raw = H.render.merge_sorted(cu(za), cu(ra), cu(zb2), cu(rb))
    oz, oraw = compositing.merge_sorted([za, zb2], [ra, rb])
    np.testing.assert_array_equal(z.cpu().numpy(), oz)
    np.testing.assert_array_equal(raw.cpu().numpy(), oraw)


def test_near_far_cluster_skip_is_bit_identical(H):
    """near_far_kernel also skips, per wave, the clusters of 64 consecutive vertices none of its rays can reach.  near / far are a min / max
    over the vertices that pass the reference's test, so the vertex ORDER cannot matter: a random permutation of the vertices (clusters
    become body-sized: nothing is skipped) must give the same bits as the model's order (compact clusters: most are skipped)."""
    from neuman_hip import synthetic
    rng = np.random.default_rng(5)
    verts = synthetic.human_vertex_cloud(0).astype(np.float32)
    order = np.argsort(verts[:, 1] * 7 + verts[:, 0])                       # spatially coherent clusters (sorted along the body)
    v_sorted, v_shuf = verts[order].copy(), verts[rng.permutation(len(verts))].copy()
    R = 64 * 500
    o = np.tile(np.array([[0.1, 0.2, -2.5]], np.float32), (R, 1)) + rng.normal(size=(R, 3)).astype(np.float32) * 0.01
    tgt = verts[rng.integers(0, len(verts), R)] + rng.normal(size=(R, 3)).astype(np.float32) * 0.3
    d = tgt - o
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    res = [H.ray.geometry_guided_near_far(cu(o), cu(d), cu(v), 0.05) for v in (v_sorted, v_shuf, verts)]
    n0, f0 = (x.cpu().numpy() for x in res[0])
    for n, f in res[1:]:
        np.testing.assert_array_equal(n0, n.cpu().numpy())
        np.testing.assert_array_equal(f0, f.cpu().numpy())
    hit = n0 < f0
    assert 0.2 < hit.mean() < 0.98
