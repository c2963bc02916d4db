"""-m gpu: whole-frame renderers (reference signatures, numpy out) vs the CPU oracle and the reference goldens."""
import types

import numpy as np
import pytest
import torch

from oracle import compositing, nerf_mlp, ray_ops as O, render as OR, warp as OW

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(nets):
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from neuman_hip import ray_utils, render_utils, synthetic
    gn = {k: (j.cuda(), (sd, spec)) for k, (j, sd, spec) in nets.items()}
    return types.SimpleNamespace(ray=ray_utils, render=render_utils, syn=synthetic, nets=gn)


def psnr(a, b):
    return 10 * np.log10(1.0 / max(np.mean((a.astype(np.float64) - b) ** 2), 1e-30))


def cu(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', torch.float32).contiguous()


def test_c1_coarse_only_frame(G, golden):
    """Single-pass frame: no inverse-CDF step function anywhere -> the 1e-4 L-inf contract holds on every pixel."""
    g = golden['render']
    cap = G.syn.SimpleCapture(64, 64, c2w=g['c1_c2w'])
    rgb = G.render.render_vanilla(G.nets[0][0], cap, None, rays_per_batch=4096, samples_per_ray=32)
    assert rgb.shape == (64, 64, 3) and rgb.dtype == np.float32
    e = np.abs(rgb - g['c1_coarse_only_rgb']).max()
    print(f"[render] C1 coarse-only vs reference golden: Linf {e:.3e}, PSNR {psnr(rgb, g['c1_coarse_only_rgb']):.1f} dB")
    assert e < 1e-4
    o_rgb = OR.render_vanilla(G.nets[0][1], cap, None, rays_per_batch=4096, samples_per_ray=32)
    assert np.abs(rgb - o_rgb).max() < 1e-4


def displacement_rank(err, z_dev, z_ora, tag, keep=0.80, max_bad_frac=0.02):
    """The end-to-end statement for the small merged scenes: rays beyond 1e-4 are all among the (1 - keep) most displaced rays (largest
    max_s |z_dev - z_oracle| of the two-pass background samples) and the `keep` least displaced rays are within 1e-4 -- the
    quantitative form (binding on 80 % of the rays of these 400-ray, 16 + 16-sample scenes, where one coarse bin is 0.2 wide) of "the inverse CDF moved a sample"; the full statements (a)-(d) are made on
    BASELINE-sized workloads by oracle/attribution.py (test_c1_two_pass_frame, tests/test_hip_configs.py) and against the
    reference's own frames in tests/test_hip_posed_golden.py."""
    dz = np.abs(z_dev - z_ora).max(-1)
    quiet = dz <= np.percentile(dz, 100 * keep)
    bad = err > 1e-4
    print(f"[{tag}] rays > 1e-4: {bad.sum()} / {err.size}, of which among the {100 * keep:.0f} % least displaced: {(bad & quiet).sum()}; Linf over those "
          f"{err[quiet].max():.2e}, overall {err.max():.2e}")
    assert (bad & quiet).sum() == 0 and bad.mean() < max_bad_frac


def test_c1_two_pass_frame(G, golden):
    """BASELINE configuration 1 (64x64, 32 + 32 samples) end to end, in the package's default arithmetic.

    The inverse CDF (ray_utils.py:164-194) turns a coarse-weight difference d into a sample displacement d / pdf, so two
    float32 evaluations of the reference disagree on a few rays (the CPU oracle vs the reference's own golden: 18 of 4096).
    With the float32-class coarse pass (fp16x3) the device sits at that level, and the deviation is attributed quantitatively
    (oracle/attribution.py)."""
    g = golden['render']
    cap = G.syn.SimpleCapture(64, 64, c2w=g['c1_c2w'])
    coarse, fine = G.nets[0][0], G.nets[1][0]
    rgb, depth = G.render.render_vanilla(coarse, cap, fine, rays_per_batch=2048, samples_per_ray=32,
                                         importance_samples_per_ray=32, return_depth=True)
    o_t, d_t = G.ray.shot_all_rays_dev(cap, torch.device('cuda'))             # the rays render_vanilla itself generates
    o, d = o_t.cpu().numpy(), d_t.cpu().numpy()
    R = o.shape[0]
    trace = {}
    rgb_t, _ = G.render.render_vanilla_rays(coarse, fine, o_t, d_t, 0.0, 3.14, 32, 32, trace=trace)
    zf = trace['bkg_z'][0].cpu().numpy()
    assert np.array_equal(rgb_t.cpu().numpy(), rgb.reshape(-1, 3))           # the traced run IS the renderer's run
    # (1) end to end vs the reference's own output
    err = np.abs(rgb - g['c1_rgb']).max(-1).reshape(-1)
    print(f"[render] C1 two-pass vs reference golden: Linf {err.max():.3e}, rays > 1e-4: {(err > 1e-4).sum()} / {err.size}, "
          f"PSNR {psnr(rgb, g['c1_rgb']):.1f} dB")
    assert (err > 1e-4).sum() <= 25 and psnr(rgb, g['c1_rgb']) >= 80.0
    # (2) end to end vs the oracle with the deviation attributed: statements (a)-(d) of oracle/attribution.py -- both conditional
    #     parities on every ray, the coarse weights, the displacement rank and first-order bound, the count against the floor
    from oracle import attribution
    ora = attribution.oracle_two_pass([G.nets[0][1], G.nets[1][1]], o, d, 0.0, 3.14, 32, 32)
    rgb_a, zf, w_a, rgb_on = attribution.device_two_pass(G.render, coarse, fine, o_t, d_t, 0.0, 3.14, 32, 32, cu(ora["z"]))
    assert np.array_equal(rgb_a, rgb.reshape(-1, 3))
    rep, fails = attribution.two_pass(rgb_a, zf, w_a, rgb_on, ora["rgb"], ora["z"], ora["w"], ora["fine_on"], arbiter=attribution.load_arbiter('c1'), tag="C1 two-pass vs oracle")
    assert not fails, fails
    z = O.ray_to_samples(o, d, np.zeros((R, 1), np.float32), np.full((R, 1), 3.14, np.float32), 32)[2]
    c_depth = compositing.raw2outputs(nerf_mlp.joiner_forward(*G.nets[1][1], (o[:, None, :] + d[:, None, :] * zf[..., None]).astype(np.float32),
                                                              np.broadcast_to(d[:, None, :], (R, zf.shape[1], 3))), zf, d)[4]
    assert np.abs(depth.reshape(-1) - c_depth).max() < 5e-4
    # (4) and the inverse CDF itself, given identical coarse weights: tie-aware sample positions
    zt = cu(z)
    raw = coarse.forward_rays(cu(o), cu(d), zt)
    _, _, _, w, _ = G.render.raw2outputs(raw, zt, cu(d))
    z_fine = G.ray.importance_z(zt, w, 32).cpu().numpy()
    _, _, oz2 = O.ray_to_importance_samples(o, d, z, w.cpu().numpy(), 32)
    diff = np.abs(z_fine - oz2) > 3e-6
    print(f"[render] C1 fine sample positions differing from the oracle's on identical weights: {diff.sum()} / {diff.size}")
    assert diff.mean() < 0.01


def test_c3_canonical_human_frame(G, golden):
    g = golden['render']
    cap = G.syn.SimpleCapture(48, 48, fx=float(g['c3_fx']), c2w=g['c3_c2w'])
    net = types.SimpleNamespace(coarse_human_net=G.nets[2][0], parameters=G.nets[2][0].parameters)
    verts = G.syn.human_vertex_cloud(0)
    rgb, depth, acc = G.render.render_smpl_nerf(net, cap, verts, None, None, rays_per_batch=1024, samples_per_ray=32,
                                                render_can=True, geo_threshold=0.2, return_depth=True, return_mask=True,
                                                interval_comp=0.7)
    hit = g['c3_acc'] > 0
    flips = (acc > 0) != hit
    ok = ~flips
    e = np.abs(rgb - g['c3_rgb'])[ok].max()
    print(f"[render] C3 canonical vs reference golden: hit/miss flips {flips.sum()}, Linf {e:.3e}, PSNR {psnr(rgb[ok], g['c3_rgb'][ok]):.1f} dB")
    assert flips.mean() < 2e-3
    assert (rgb[~(acc > 0)] == 1).all() and (depth[~(acc > 0)] == 0).all()          # misses: white, depth 0 (render_utils.py:199-205)
    # the canonical net uses the 'rotate' PE whose argument (x.B^T, up to ~1e3 rad) carries f32 summation-order noise of
    # ~1e-4 rad in the reference itself; the oracle shows the same spread against the golden (test_oracle_golden)
    assert e < 1e-4 and np.abs(acc - g['c3_acc'])[ok].max() < 1e-4                       # measured 4.6e-5 / (r01)
    # (the device's near / far are a float64 evaluation rounded once, csrc/nearfar.hip: the oracle renders on those bounds, its own float32 ones are
    #  checked against them in tests/test_hip_ray_ops.py)
    oo, dd = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, O.all_pixel_coords(cap.shape))
    nf64 = [tuple(x.astype(np.float32) for x in O.geometry_guided_near_far(oo, dd, verts, 0.2, dtype=np.float64))]
    o_rgb, o_depth, o_acc = OR.render_smpl_nerf(G.nets[2][1], cap, verts, None, None, rays_per_batch=4096, samples_per_ray=32,
                                                render_can=True, geo_threshold=0.2, return_depth=True, return_mask=True,
                                                interval_comp=0.7, given={'near_far': nf64})
    ok = (acc > 0) == (o_acc > 0)
    e = np.abs(rgb - o_rgb)[ok].max()
    print(f"[render] C3 canonical vs oracle: Linf {e:.3e}")
    assert ok.mean() > 0.998 and e < 2e-5                                                 # measured 1.6e-6 (r01)


def small_scene(G, S=16):
    verts_c, faces = G.syn.capsule_mesh(n_rings=10, n_seg=12)
    posed, T = G.syn.twist_transforms(verts_c)
    cap = G.syn.SimpleCapture(20, 20, fx=40., c2w=G.syn.spherical_c2w(20., -10., 3.0), near=0.5, far=4.0)
    return cap, posed, faces, T


def test_posed_human_frame_with_warp(G):
    cap, posed, faces, T = small_scene(G)
    net = types.SimpleNamespace(coarse_human_net=G.nets[2][0], parameters=G.nets[2][0].parameters)
    rgb, depth, acc = G.render.render_smpl_nerf(net, cap, posed, faces, T, samples_per_ray=16, render_can=False,
                                                geo_threshold=0.2, return_depth=True, return_mask=True)
    oo, dd = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, O.all_pixel_coords(cap.shape))
    nf64 = [tuple(x.astype(np.float32) for x in O.geometry_guided_near_far(oo, dd, posed, 0.2, dtype=np.float64))]       # (the device's bounds: float64 discriminant)
    o_rgb, o_depth, o_acc = OR.render_smpl_nerf(G.nets[2][1], cap, posed, faces, T, samples_per_ray=16, render_can=False,
                                                geo_threshold=0.2, return_depth=True, return_mask=True, given={'near_far': nf64})
    ok = (acc > 0) == (o_acc > 0)
    e = np.abs(rgb - o_rgb)[ok].max()
    print(f"[render] posed human (warp) vs oracle: hit rays {(o_acc > 0).sum()}, Linf {e:.3e}")
    assert (o_acc > 0).mean() > 0.05 and ok.mean() > 0.99 and e < 3e-5                    # measured 3.0e-6 (r01)


def test_hybrid_and_multi_person_frames(G):
    """The two hybrid renderers through their reference-named entry points on a small scene, (i) against the oracle's own
    rendering with the rays beyond 1e-4 confined to the most displaced fifth (displacement_rank), (ii) conditional on the device's samples and warped
    points at 1e-4 on every pixel (the full-size sample counts are in tests/test_hip_configs.py)."""
    from test_hip_configs import conditional_hybrid
    cap, posed, faces, T = small_scene(G)
    coarse, fine, human = G.nets[0], G.nets[1], G.nets[2]
    net = types.SimpleNamespace(coarse_bkg_net=coarse[0], fine_bkg_net=fine[0], coarse_human_net=human[0],
                                parameters=coarse[0].parameters)
    kw = dict(samples_per_ray=16, importance_samples_per_ray=16, geo_threshold=0.2, return_depth=True)
    o_t, d_t = G.render._pixel_rays(cap, torch.device('cuda'))                # the rays the renderers themselves generate
    o, d = o_t.cpu().numpy(), d_t.cpu().numpy()
    faces3 = np.ascontiguousarray(np.asarray(faces)[:, :3], np.int32)

    def bkg_oracle_z():
        R = o.shape[0]
        pts, dd, z = O.ray_to_samples(o, d, np.full((R, 1), cap.near['bkg'], np.float32), np.full((R, 1), cap.far['bkg'], np.float32), 16)
        w = compositing.raw2outputs(nerf_mlp.joiner_forward(*coarse[1], pts, dd), z, d)[3]
        return O.ray_to_importance_samples(o, d, z, w, 16)[2]

    oz = bkg_oracle_z()
    # ---- one actor
    rgb, depth = G.render.render_hybrid_nerf(net, cap, posed, faces, T, **kw)
    o_rgb, o_depth = OR.render_hybrid_nerf(coarse[1], fine[1], human[1], cap, posed, faces, T, **kw)
    err = np.abs(rgb - o_rgb).max(-1).reshape(-1)
    trace = {}
    mesh = G.ray.mesh_to_device(posed, faces3, T, 'cuda')
    rgb_t, _, _ = G.render.render_hybrid_rays(coarse[0], fine[0], human[0], o_t, d_t, cap.near['bkg'], cap.far['bkg'], cu(posed), mesh, 16, 16,
                                              trace=trace)
    assert np.array_equal(rgb_t.cpu().numpy(), rgb.reshape(-1, 3))
    displacement_rank(err, trace['bkg_z'][0].cpu().numpy(), oz, "render: hybrid vs oracle")
    c_rgb, _ = conditional_hybrid(G, {'fine': fine[1], 'human': human[1]}, o, d, trace, 1, 16)
    e = np.abs(rgb_t.cpu().numpy() - c_rgb).max()
    print(f"[render] hybrid, oracle on the device's samples and warped points: Linf {e:.2e}")
    assert e < 1e-4
    # ---- two actors
    posed2 = (posed + np.array([0.35, 0.0, 0.2], np.float32)).astype(np.float32)
    T2 = T.copy()
    T2[:, :3, 3] += np.array([0.35, 0.0, 0.2])
    rgb, depth = G.render.render_hybrid_nerf_multi_persons(net, cap, [net, net], [posed, posed2], [faces, faces], [T, T2], **kw)
    o_rgb, o_depth = OR.render_hybrid_nerf_multi_persons(coarse[1], fine[1], [human[1], human[1]], cap, [posed, posed2],
                                                         [faces, faces], [T, T2], **kw)
    err = np.abs(rgb - o_rgb).max(-1).reshape(-1)
    trace = {}
    meshes = [G.ray.mesh_to_device(p, faces3, t, 'cuda') for p, t in ((posed, T), (posed2, T2))]
    rgb_t, _ = G.render.render_multi_rays(coarse[0], fine[0], [human[0]] * 2, o_t, d_t, cap.near['bkg'], cap.far['bkg'], [cu(posed), cu(posed2)],
                                          meshes, 16, 16, trace=trace)
    assert np.array_equal(rgb_t.cpu().numpy(), rgb.reshape(-1, 3))
    displacement_rank(err, trace['bkg_z'][0].cpu().numpy(), oz, "render: multi-person vs oracle")
    c_rgb, _ = conditional_hybrid(G, {'fine': fine[1], 'human': human[1]}, o, d, trace, 2, 16, far=cap.far['bkg'])
    e = np.abs(rgb_t.cpu().numpy() - c_rgb).max()
    print(f"[render] multi-person, oracle on the device's samples and warped points: Linf {e:.2e}")
    assert e < 1e-4


def test_warp_vs_oracle(G):
    verts_c, faces = G.syn.capsule_mesh(n_rings=14, n_seg=16)
    posed, T = G.syn.twist_transforms(verts_c)
    rng = np.random.default_rng(0)
    R, S = 40, 24
    o = np.tile(np.array([[0.05, 0.0, -2.0]], np.float32), (R, 1))
    d = rng.normal(size=(R, 3)).astype(np.float32) * np.array([0.1, 0.25, 0.02], np.float32) + np.array([0, 0, 1], np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    z = np.linspace(1.5, 2.6, S, dtype=np.float32)[None].repeat(R, 0)
    pts = (o[:, None] + d[:, None] * z[..., None]).astype(np.float32)
    cp, cd, cl = G.ray.warp_samples_to_canonical(pts, posed, faces, T)
    ocp, ocd, ocl = OW.warp_samples_to_canonical(pts, posed, faces, T)
    assert isinstance(cp, np.ndarray) and cp.shape == (R, S, 3)
    print(f"[warp] closest Linf {np.abs(cl - ocl).max():.3e}, can_pts Linf {np.abs(cp - ocp).max():.3e}, can_dirs Linf {np.abs(cd - ocd).max():.3e}")
    # the invariant of a closest-point query is the DISTANCE; inside the concave side of the surface a point can have two
    # nearly equidistant feet on adjacent faces, and f32 (device) vs f64 (oracle) may pick either
    np.testing.assert_allclose(np.linalg.norm(cl - pts, axis=-1), np.linalg.norm(ocl - pts, axis=-1), atol=2e-6)
    same = np.abs(cl - ocl).max(-1) < 2e-5
    assert same.mean() > 0.99 and np.abs(cl - ocl).max() < 2e-3
    np.testing.assert_allclose(cp[same], ocp[same], atol=1e-5)
    np.testing.assert_allclose(cp, ocp, atol=5e-4)
    # distance property: the returned closest point is no farther than any vertex (igl semantics)
    dmin = np.sqrt(((pts[:, :, None, :] - posed[None, None]) ** 2).sum(-1)).min(-1)
    assert (np.linalg.norm(cl - pts, axis=-1) <= dmin + 1e-6).all()
    np.testing.assert_allclose(np.linalg.norm(cd, axis=-1), 1.0, atol=1e-5)
    assert np.abs(cd - ocd).max() < 5e-3
    # tree search vs the all-triangles loop: both are exact searches, so they must agree bit for bit on everything
    m_grid = G.ray.Mesh(posed, faces, T, 'cuda', search='tree')
    m_brute = G.ray.Mesh(posed, faces, T, 'cuda', search='all')
    info = m_grid.info()
    print(f"[warp] tree {info}")
    assert info['nodes'] > 0 and 4 ** info['levels'] >= faces.shape[0]
    a = G.ray.warp_to_canonical_dev(cu(pts), m_grid, want_closest=True)
    b = G.ray.warp_to_canonical_dev(cu(pts), m_brute, want_closest=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    # a query far outside the mesh box still gets the exact answer
    far_pts = pts + np.array([5.0, -3.0, 2.0], np.float32)
    fc = G.ray.warp_to_canonical_dev(cu(far_pts), m_grid, want_closest=True)[2].cpu().numpy()
    _, _, ofc = OW.closest_point_on_mesh(far_pts.reshape(-1, 3), posed, faces[:, :3])
    np.testing.assert_allclose(np.linalg.norm(fc.reshape(-1, 3) - far_pts.reshape(-1, 3), axis=-1),
                               np.linalg.norm(ofc - far_pts.reshape(-1, 3), axis=-1), rtol=1e-5)
    # the identity warp: T = I returns the points themselves
    I = np.tile(np.eye(4), (posed.shape[0], 1, 1))
    cp, cd, _ = G.ray.warp_samples_to_canonical(pts, posed, faces, I)
    np.testing.assert_allclose(cp, pts, atol=1e-6)
    np.testing.assert_allclose(cd, np.broadcast_to(d[:, None], cd.shape), atol=2e-4)


def test_full_size_frame_properties(G):
    """BASELINE config 2 (800x800, 128 + 128 samples): size-independent properties + chunking invariance."""
    coarse, fine = G.nets[0][0], G.nets[1][0]
    cap = G.syn.SimpleCapture(800, 800)
    o, d = G.ray.shot_all_rays(cap)
    o, d = cu(o), cu(d)
    sub = slice(0, 800 * 40)                      # 40 rows for the chunking comparison
    a_rgb, a_depth = G.render.render_vanilla_rays(coarse, fine, o[sub], d[sub], 0.0, 3.14, 128, 128)
    old = G.render.MAX_RAYS_PER_LAUNCH
    try:
        G.render.MAX_RAYS_PER_LAUNCH = 5000       # not a multiple of anything
        b_rgb, b_depth = G.render.render_vanilla_rays(coarse, fine, o[sub], d[sub], 0.0, 3.14, 128, 128)
    finally:
        G.render.MAX_RAYS_PER_LAUNCH = old
    assert torch.equal(a_rgb, b_rgb) and torch.equal(a_depth, b_depth)       # rays are independent: chunking is bit-invariant
    rgb, depth = G.render.render_vanilla_rays(coarse, fine, o, d, 0.0, 3.14, 128, 128)
    assert torch.isfinite(rgb).all() and rgb.min() >= -1e-5 and rgb.max() <= 1 + 1e-5
    assert depth.min() >= 0 and depth.max() <= 3.14 * (1 + 1e-5)
    assert torch.equal(rgb[sub], a_rgb)
    # pipeline pieces at full size: sorted fine samples that contain every coarse sample; weights sum <= 1
    R = o.shape[0]
    near, far = torch.zeros(R, device='cuda'), torch.full((R,), 3.14, device='cuda')
    _, _, z = G.ray.sample_z(o, d, near, far, 128)
    raw = coarse.forward_rays(o, d, z)
    _, _, acc, w, _ = G.render.raw2outputs(raw, z, d)
    assert acc.max() <= 1 + 1e-5 and w.min() >= 0
    zf = G.ray.importance_z(z, w, 128)
    assert (zf[:, 1:] >= zf[:, :-1]).all() and zf.shape == (R, 256)
    chk = torch.randint(0, R, (16,), device='cuda')
    for r in chk.tolist():
        assert torch.isin(z[r], zf[r]).all()
    # exact-f32 device kernel on a slice of the frame as an independent second opinion on the coarse pass
    sl = slice(R // 2, R // 2 + 8192)
    a = coarse.forward_rays(o[sl].contiguous(), d[sl].contiguous(), z[sl].contiguous(), precision="fp32")
    assert (a[..., :3] - raw[sl][..., :3]).abs().max() < 1e-4


def test_mixed_precision_policy_is_parity_grade(G):
    """The package default: coarse (sampling) pass fp16x3, shading passes i8x3.  On a C2-sized slice (8192 rays of the 800x800
    frame, 128 + 128 samples, dense weights): (i) the fine sample positions are bit-identical to the all-fp16x3 path,
    (ii) every pixel is within 1e-4 of the all-fp16x3 frame and (iii) of the exact-f32 kernel evaluated on the same fine
    sample positions -- the north-star contract, conditional on the samples.  Direct calls of a
    net stay fp16x3; only passes tagged role='shading' change arithmetic."""
    from neuman_hip import synthetic
    coarse, fine = synthetic.make_joiner(0).cuda(), synthetic.make_joiner(1).cuda()
    cap = synthetic.SimpleCapture(800, 800)
    o, d = G.ray.shot_all_rays_dev(cap, torch.device('cuda'))
    sel = torch.arange(300 * 800, 300 * 800 + 8192, device='cuda')
    o, d = o[sel].contiguous(), d[sel].contiguous()
    near, far = torch.zeros(8192, device='cuda'), torch.full((8192,), 3.14, device='cuda')
    out = {}
    for p in ("mixed", "fp16x3"):
        coarse.precision = fine.precision = p
        raw, z = G.render.bkg_pass_rays(coarse, fine, o, d, near, far, 128, 128, True)
        out[p] = (G.render.raw2outputs(raw, z, d)[0], z)
    assert torch.equal(out["mixed"][1], out["fp16x3"][1])                              # (i)
    e = (out["mixed"][0] - out["fp16x3"][0]).abs().max().item()
    raw32 = fine.forward_rays(o, d, out["mixed"][1], precision="fp32")
    e32 = (out["mixed"][0] - G.render.raw2outputs(raw32, out["mixed"][1], d)[0]).abs().max().item()
    print(f"[render] mixed vs all-fp16x3 on 8192 C2 rays: Linf {e:.3e}; vs the f32 kernel on the same samples: Linf {e32:.3e}")
    assert e < 1e-4 and e32 < 1e-4                                                     # (ii), (iii)
    coarse.precision = "mixed"
    pts = torch.rand((1000, 3), device='cuda') * 2 - 1
    dirs = torch.nn.functional.normalize(torch.randn((1000, 3), device='cuda'), dim=-1)
    assert torch.equal(coarse(pts, dirs), coarse(pts, dirs, precision="fp16x3"))
    assert torch.equal(coarse(pts, dirs, role='shading'), coarse(pts, dirs, precision="i8x3"))
    assert not torch.equal(coarse(pts, dirs, role='shading'), coarse(pts, dirs))


@pytest.mark.parametrize("case", ["smpl_size", "tiny", "one_triangle", "offset_scene", "duplicate_faces"])
def test_tree_search_is_bit_identical_to_all_triangles(G, case):
    """The tree search prunes with a rounding margin, so it must return exactly what the all-triangles loop returns:
    queries on the surface (ties between the faces around a vertex -> lowest face id), inside, near, far, huge, NaN."""
    rng = np.random.default_rng(7)
    if case == "smpl_size":
        verts_c, faces = G.syn.capsule_mesh()                       # V = 6890, F = 13776
    elif case == "tiny":
        verts_c, faces = G.syn.capsule_mesh(n_rings=2, n_seg=3)
    elif case == "one_triangle":
        verts_c = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
        faces = np.array([[0, 1, 2]], np.int32)
    else:
        verts_c, faces = G.syn.capsule_mesh(n_rings=20, n_seg=24)
    verts_c = np.asarray(verts_c, np.float32)
    if case == "one_triangle":
        posed, T = verts_c, np.tile(np.eye(4), (3, 1, 1))
    else:
        posed, T = G.syn.twist_transforms(verts_c)
    posed = np.asarray(posed, np.float32)
    if case == "offset_scene":                                       # a body 40 m from the origin: ulp(40) = 3.8e-6
        shift = np.array([40.0, -25.0, 31.0])
        posed = (posed + shift).astype(np.float32)
        T = T.copy()
        T[:, :3, 3] += shift
    if case == "duplicate_faces":                                    # every face twice: exact ties everywhere
        faces = np.concatenate([faces[::-1], faces], 0)
    faces = np.ascontiguousarray(np.asarray(faces)[:, :3], np.int32)
    V = posed.shape[0]
    tri = posed[faces]
    S = 32
    q = []
    q.append(posed[rng.integers(0, V, 20 * S)])                                               # exactly on vertices
    w = rng.dirichlet([1, 1, 1], 20 * S).astype(np.float32)
    q.append((tri[rng.integers(0, len(faces), 20 * S)] * w[..., None]).sum(1))                # on faces
    q.append((tri[rng.integers(0, len(faces), 10 * S), :2].mean(1)))                          # on edge midpoints
    c = posed.mean(0)
    for sc in (1e-4, 1e-2, 0.2, 2.0, 50.0):
        q.append(posed[rng.integers(0, V, 20 * S)] + rng.normal(size=(20 * S, 3)) * sc)
    q.append(np.tile(c, (S, 1)))                                                               # the centre (deep inside)
    q.append(c + rng.normal(size=(S, 3)) * 1e6)
    bad = np.tile(c, (S, 1)).astype(np.float32)
    bad[0, 0] = np.nan; bad[1, 1] = np.inf; bad[2, 2] = -np.inf; bad[3] = 3e38; bad[4] = 1e20
    q.append(bad)
    pts = np.concatenate(q, 0).astype(np.float32).reshape(-1, S, 3)
    m_tree = G.ray.Mesh(posed, faces, T, 'cuda', search='tree')
    m_all = G.ray.Mesh(posed, faces, T, 'cuda', search='all')
    a = G.ray.warp_to_canonical_dev(cu(pts), m_tree, want_closest=True)
    b = G.ray.warp_to_canonical_dev(cu(pts), m_all, want_closest=True)
    w = G.ray.warp_to_canonical_dev(cu(pts), G.ray.Mesh(posed, faces, T, 'cuda', search='tree_wide'), want_closest=True)
    for x, y, z, name in zip(a, b, w, ("can_pts", "can_dirs", "closest")):
        for other, what in ((y, "all-triangles"), (z, "wide-stack tree")):
            same = (x == other) | (x.isnan() & other.isnan())
            assert bool(same.all()), f"{case}: {name} differs from the {what} search at {int((~same).sum())} of {same.numel()} values"
    # and the answer is the right one: distance vs the f64 oracle on the finite, moderate queries
    fin = np.isfinite(pts).all(-1) & (np.abs(pts).max(-1) < 1e3)
    cl = a[2].cpu().numpy()[fin]
    _, _, ocl = OW.closest_point_on_mesh(pts[fin][:2000], posed, faces)
    dist = np.linalg.norm(cl[:2000] - pts[fin][:2000], axis=-1)
    np.testing.assert_allclose(dist, np.linalg.norm(ocl - pts[fin][:2000], axis=-1), atol=2e-5 * (1 + np.abs(posed).max()), rtol=1e-5)


@pytest.mark.parametrize("case", ["smpl_size", "small"])
def test_signed_distance(G, case):
    """nm_signed_distance vs the oracle's pseudonormal sign and, independently, the winding number of the closed mesh"""
    rng = np.random.default_rng(3)
    verts_c, faces = G.syn.capsule_mesh() if case == "smpl_size" else G.syn.capsule_mesh(n_rings=10, n_seg=12)
    posed, T = G.syn.twist_transforms(np.asarray(verts_c, np.float32))
    faces = np.ascontiguousarray(np.asarray(faces)[:, :3], np.int32)
    n = 600 if case == "smpl_size" else 3000
    pts = (posed[rng.integers(0, len(posed), n)] + rng.normal(size=(n, 3)) * rng.choice([0.003, 0.03, 0.2], size=(n, 1))).astype(np.float32)
    S, I, C = G.ray.signed_distance(pts, posed, faces)
    assert S.shape == (n,) and I.shape == (n,) and C.shape == (n, 3)
    wn = OW.winding_number(pts, posed, faces)
    sure = np.abs(S) > 1e-5                                            # a point within rounding of the surface has no defined side
    assert ((S < 0) == (wn > 0.5))[sure].all(), f"{(((S < 0) != (wn > 0.5)) & sure).sum()} wrong signs"
    print(f"[sdf] {case}: {n} queries, {(S < 0).mean():.2f} inside, signs agree with the winding number on all {sure.sum()} off-surface points")
    m = min(n, 400)
    oS, oI, oC = OW.signed_distance(pts[:m], posed, faces)
    np.testing.assert_allclose(np.abs(S[:m]), np.abs(oS), atol=2e-6)
    assert (np.sign(S[:m]) == np.sign(oS))[np.abs(oS) > 1e-5].all()
    np.testing.assert_allclose(np.linalg.norm(C[:m] - pts[:m], axis=1), np.abs(oS), atol=2e-6)
    # the returned face does contain the closest point
    tri = posed[faces[I[:m]]]
    bary = OW.barycentric_coordinates_tri(C[:m].astype(np.float64), tri[:, 0].astype(np.float64), tri[:, 1].astype(np.float64), tri[:, 2].astype(np.float64))
    assert (bary > -1e-4).all()


def test_warp_samples_to_canonical_diff(G):
    """reference ray_utils.py:69-93 on the device: T_interp_inv maps a posed surface point back to canonical space, and
    gradients reach verts and T"""
    verts_c, faces = G.syn.capsule_mesh(n_rings=10, n_seg=12)
    posed, T = G.syn.twist_transforms(np.asarray(verts_c, np.float32))
    faces = np.ascontiguousarray(np.asarray(faces)[:, :3], np.int32)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, len(posed), 64)
    pts = posed[idx].astype(np.float32)
    verts_t = torch.tensor(posed, device='cuda', requires_grad=True)
    T_t = torch.tensor(T, device='cuda', dtype=torch.float32, requires_grad=True)
    Ti, f_id, sd = G.ray.warp_samples_to_canonical_diff(pts, verts_t, faces, T_t)
    assert Ti.shape == (64, 4, 4) and f_id.shape == (64,) and sd.shape == (64,)
    can = (Ti @ torch.cat([torch.tensor(pts, device='cuda'), torch.ones((64, 1), device='cuda')], 1)[..., None])[:, :3, 0]
    np.testing.assert_allclose(can.detach().cpu().numpy(), np.asarray(verts_c, np.float32)[idx], atol=2e-4)
    can.sum().backward()
    assert torch.isfinite(T_t.grad).all() and T_t.grad.abs().sum() > 0 and verts_t.grad is not None
