"""-m gpu: one oracle-checked slice per BASELINE.json configuration at its REAL sample counts.

    C2  800x800 background, 128 coarse + 128 importance (merged 256)            2048 rays of the frame
    C3  512x512 canonical human, 128 samples                                     >= 1024 hit rays
    C4  hybrid: background 128 + 128 and human 128, merged 384                   hit and miss rays
    C5  three actors: background 192 + 128 and 3 x 192, merged 896

Two kinds of statement are made:

* end to end, with the deviation ATTRIBUTED quantitatively (oracle/attribution.py): the device's shading pass on the oracle's samples and
  the oracle's on the device's both within 1e-4 on every ray, the coarse weights within 2e-5, every ray beyond 1e-4 among the 6 % most
  displaced ones, and -- against the ARBITER, the reference's own renderer run in float64 (tests/golden/arbiter.npz) -- no more rays beyond
  1e-4 than the reference's own float32 run leaves (+ a quarter);
* conditional parity at 1e-4 on EVERY pixel: the oracle evaluates its networks, merge and compositing on the device's own
  sample positions and (for posed humans) the device's own warped points -- the renderer's product code path is what
  runs on the device (the `trace` hook records its intermediates), the warp itself is checked against the oracle in
  tests/test_hip_render.py.
"""
import numpy as np
import pytest
import torch

from oracle import compositing, nerf_mlp, ray_ops as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(nets):
    import types
    from neuman_hip import ray_utils, render_utils, synthetic
    gn = {k: (j.cuda(), (sd, spec)) for k, (j, sd, spec) in nets.items()}
    return types.SimpleNamespace(ray=ray_utils, render=render_utils, syn=synthetic, nets=gn)


def cu(x):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', torch.float32).contiguous()


def psnr(a, b):
    return 10 * np.log10(1.0 / max(np.mean((a.astype(np.float64) - b) ** 2), 1e-30))


def oracle_two_pass(nets, o, d, near, far, S, NI):
    """reference render_utils.py:131-151 on the CPU oracle -> rgb, z_fine"""
    R = o.shape[0]
    pts, dd, z = O.ray_to_samples(o, d, np.full((R, 1), near, np.float32), np.full((R, 1), far, np.float32), S)
    out = nerf_mlp.joiner_forward(*nets[0], pts, dd)
    _, _, _, w, _ = compositing.raw2outputs(out, z, d)
    pts, dd, zf = O.ray_to_importance_samples(o, d, z, w, NI)
    out = nerf_mlp.joiner_forward(*nets[1], pts, dd)
    return compositing.raw2outputs(out, zf, d)[0], zf


@pytest.mark.parametrize("precision", ["mixed", "fp16x3", "bf16x3"])
def test_c2_slice_vs_oracle(G, precision):
    """2048 rays from the middle of the 800x800 frame, 128 + 128 samples, end to end: statements (a)-(c) of oracle/attribution.py against the
    oracle, statement (d) against the ARBITER -- the reference's own render_vanilla run in float64 on these very rays (tests/golden/arbiter.npz,
    case 'c2'): the device may be beyond 1e-4 of it on no more rays than the reference's own float32 run is (+ margin) -- and, conditional on
    the REFERENCE's sample positions (its float64 run's), every ray within 1e-4 of the reference's float64 pixels."""
    from oracle import attribution
    coarse, fine = G.syn.make_joiner(0).cuda(), G.syn.make_joiner(1).cuda()
    cap = G.syn.SimpleCapture(800, 800)
    o, d = O.shot_all_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, cap.shape)
    arb = attribution.load_arbiter('c2')
    first = int(arb['first'])
    sl = slice(first, first + 2048)
    assert first == 400 * 800 + 100
    o, d = o[sl].astype(np.float32), d[sl].astype(np.float32)
    ora = attribution.oracle_two_pass([G.nets[0][1], G.nets[1][1]], o, d, 0.0, 3.14, 128, 128)
    rgb, z, w, rgb_on = attribution.device_two_pass(G.render, coarse, fine, cu(o), cu(d), 0.0, 3.14, 128, 128, cu(ora["z"]), precision=precision)
    print(f"[C2 {precision}] PSNR vs oracle {psnr(rgb, ora['rgb']):.1f} dB, vs the reference in float64 {psnr(rgb, arb['rgb64']):.1f} dB "
          f"(the reference's own float32 frame: {psnr(arb['rgb32'], arb['rgb64']):.1f} dB)")
    rep, fails = attribution.two_pass(rgb, z, w, rgb_on, ora["rgb"], ora["z"], ora["w"], ora["fine_on"], arbiter=arb, tag=f"C2 {precision}")
    # conditional on the reference's own (float64) sample positions: the device's shading pass against the reference's float64 pixels, every ray
    rgb_on64 = G.render.render_vanilla_rays(coarse, fine, cu(o), cu(d), 0.0, 3.14, 128, 128, True, precision=precision, given={'bkg_z': cu(arb['z64'])})[0].cpu().numpy()
    e64 = np.abs(rgb_on64.astype(np.float64) - arb['rgb64']).max(-1)
    print(f"[C2 {precision}] device shading pass on the reference's float64 sample positions vs the reference's float64 pixels: Linf {e64.max():.2e} (every ray)")
    assert e64.max() < 1e-4
    if precision == "bf16x3":
        # round 1's parity mode, kept as a cross-check of the float32-class default: its sigma error (2e-5) displaces more samples,
        # so statements (d) and (c)-rank are relaxed for it (measured 3x the fp16x3 count); everything conditional on the samples and the
        # coarse weights still binds
        fails = [f for f in fails if not (f.startswith("(d)") or "most displaced" in f)]
        assert rep["d_device_vs_reference_f64"]["rays_gt_1e-4"] <= 4 * rep["d_allowed_rays_gt_1e-4"]
    assert not fails, fails
    assert psnr(rgb, ora["rgb"]) > (80.0 if precision != "bf16x3" else 70.0)


def test_c3_canonical_slice_at_128_samples(G):
    """>= 1024 hit rays of the 512x512 canonical-human frame at 128 samples (rotate encoding, interval_comp): single pass, so
    the 1e-4 contract holds on every pixel against the oracle's own rendering"""
    human = G.nets[2]
    verts = G.syn.human_vertex_cloud(0)
    cap = G.syn.SimpleCapture(512, 512, c2w=G.syn.spherical_c2w(40., 0., 3.0))
    coords = O.all_pixel_coords(cap.shape)
    band = coords[(coords[:, 1] >= 250) & (coords[:, 1] < 256)]                    # six rows through the body
    o, d = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, band)
    o, d = o.astype(np.float32), d.astype(np.float32)
    trace = {}
    rgb, depth, acc = G.render.render_smpl_nerf_rays(human[0], cu(o), cu(d), cu(verts), None, 128, True, True, 0.2, 0.7, trace=trace)
    hit = trace['hit'][0].cpu().numpy()
    assert hit.size >= 1024, hit.size
    near, far = O.geometry_guided_near_far(o, d, verts, 0.2)
    o_hit = np.nonzero(near < far)[0]
    both = np.intersect1d(hit, o_hit)
    assert both.size >= 0.995 * max(hit.size, o_hit.size)                          # silhouette rays may flip on an ulp
    pts, dd, z = O.ray_to_samples(o[both], d[both], near[both][:, None], far[both][:, None], 128)
    out = nerf_mlp.joiner_forward(*human[1], pts, dd).copy()
    out[..., -1] *= np.float32(0.7)
    o_rgb, _, o_acc, _, o_depth = compositing.raw2outputs(out, z, d[both])
    e = np.abs(rgb.cpu().numpy()[both] - o_rgb).max()
    print(f"[C3] {both.size} hit rays at 128 samples vs oracle: rgb Linf {e:.2e}, acc Linf {np.abs(acc.cpu().numpy()[both] - o_acc).max():.2e}")
    # the rotate encoding evaluates sin / cos at arguments up to ~1e3 rad, where one float32 ulp of the argument is 6e-5: the
    # oracle itself sits ~5e-5 from the reference's output there (tests/test_oracle_golden.py); 1e-4 is the contract
    assert e < 1e-4
    assert np.abs(acc.cpu().numpy()[both] - o_acc).max() < 1e-4
    miss = np.setdiff1d(np.arange(o.shape[0]), hit)
    assert (rgb.cpu().numpy()[miss] == 1).all() and (depth.cpu().numpy()[miss] == 0).all()


def small_body(G):
    verts_c, faces = G.syn.capsule_mesh(n_rings=20, n_seg=24)
    posed, T = G.syn.twist_transforms(verts_c)
    return posed, np.ascontiguousarray(faces[:, :3], np.int32), T


def conditional_hybrid(G, nets_o, o, d, trace, n_actors, S_h, white=True, far=3.14):
    """The oracle's fine background net, human nets, merge and compositing on the device's sample positions / warped points
    (reference render_utils.py:313-353 for one actor, :390-456 for several) -> rgb per ray for the rays it covers, index list."""
    bkg_z = trace['bkg_z'][0].cpu().numpy()
    R = o.shape[0]
    pts = (o[:, None, :] + d[:, None, :] * bkg_z[..., None]).astype(np.float32)
    bkg_raw = nerf_mlp.joiner_forward(*nets_o['fine'], pts, np.broadcast_to(d[:, None, :], pts.shape))
    zs, raws = [bkg_z], [bkg_raw]
    hits = []
    for a in range(n_actors):
        hit = trace['hit'][a].cpu().numpy()
        hits.append(hit)
        h_raw = np.zeros((R, S_h, 4), np.float32)
        h_z = np.stack([O.linspace_f32(far * 2, far * 3, S_h)] * R) if n_actors > 1 else None
        if hit.size:
            cp, cdirs = trace['can_pts'][a].cpu().numpy(), trace['can_dirs'][a].cpu().numpy()
            raw = nerf_mlp.joiner_forward(*nets_o['human'], cp, cdirs)
            if n_actors > 1:
                h_raw[hit] = raw
                h_z[hit] = trace['human_z'][a].cpu().numpy()
            else:
                h_raw, h_z = raw, trace['human_z'][a].cpu().numpy()
        zs.append(h_z)
        raws.append(h_raw)
    if n_actors == 1:                                                              # hit rays merged, misses background only
        hit = hits[0]
        rgb = compositing.raw2outputs(bkg_raw, bkg_z, d, white_bkg=white)[0]
        if hit.size:
            z_all, raw_all = compositing.merge_sorted([bkg_z[hit], zs[1]], [bkg_raw[hit], raws[1]])
            rgb[hit] = compositing.raw2outputs(raw_all, z_all, d[hit], white_bkg=white)[0]
        return rgb, hits
    z_all, raw_all = compositing.merge_sorted(zs, raws)
    return compositing.raw2outputs(raw_all, z_all, d, white_bkg=white)[0], hits


def test_c4_hybrid_slice_128_128_128(G):
    """hybrid render (background 128 + 128, human 128 -> 384 merged samples on hit rays) on 1536 rays through a posed body"""
    posed, faces, T = small_body(G)
    cap = G.syn.SimpleCapture(96, 96, fx=190., c2w=G.syn.spherical_c2w(20., -10., 3.0), near=0.5, far=4.0)
    coords = O.all_pixel_coords(cap.shape)
    coords = coords[(coords[:, 1] >= 40) & (coords[:, 1] < 56)]
    o, d = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, coords)
    o, d = o.astype(np.float32), d.astype(np.float32)
    coarse, fine, human = G.nets[0], G.nets[1], G.nets[2]
    mesh = G.ray.mesh_to_device(posed, faces, T, 'cuda')
    trace = {}
    rgb, depth, acc = G.render.render_hybrid_rays(coarse[0], fine[0], human[0], cu(o), cu(d), cap.near['bkg'], cap.far['bkg'], cu(posed), mesh,
                                                  128, 128, trace=trace)
    assert trace['bkg_z'][0].shape[1] == 256 and trace['human_z'][0].shape[1] == 128
    c_rgb, hits = conditional_hybrid(G, {'fine': fine[1], 'human': human[1]}, o, d, trace, 1, 128)
    assert 200 < hits[0].size < o.shape[0] - 200, hits[0].size                     # hit and miss rays both well represented
    e = np.abs(rgb.cpu().numpy() - c_rgb).max(-1)
    print(f"[C4] {o.shape[0]} rays ({hits[0].size} hit, merged 384 samples): oracle nets + merge + compositing on the device's samples and "
          f"warped points: Linf {e.max():.2e} (hit rays {e[hits[0]].max():.2e})")
    assert e.max() < 1e-4
    # hit list and the human samples are what the oracle derives from the same rays
    near, far = (x.astype(np.float32) for x in O.geometry_guided_near_far(o, d, posed, 0.2, dtype=np.float64))       # (the device's discriminant is float64)
    o_hit = np.nonzero(near < far)[0]
    assert np.intersect1d(o_hit, hits[0]).size >= 0.99 * max(o_hit.size, hits[0].size)
    common = np.intersect1d(o_hit, hits[0])
    _, _, hz = O.ray_to_samples(o[common], d[common], near[common][:, None], far[common][:, None], 128)
    dev_hz = trace['human_z'][0].cpu().numpy()[np.searchsorted(hits[0], common)]
    assert np.abs(dev_hz - hz).max() < 2e-6                                        # both are float64 bounds rounded once


def test_c5_three_actor_slice_192_128_3x192(G):
    """three actors (background 192 + 128 and 3 x 192 -> 896 merged samples, zero-raw placeholders at z in [2 far, 3 far] for
    actors a ray misses, render_utils.py:418-419) on 768 rays"""
    posed, faces, T = small_body(G)
    shifts = [np.zeros(3), np.array([0.35, 0.0, 0.2]), np.array([-0.3, 0.05, -0.15])]
    posed_l = [(posed + s).astype(np.float32) for s in shifts]
    T_l = []
    for s in shifts:
        t = T.copy()
        t[:, :3, 3] += s
        T_l.append(t)
    cap = G.syn.SimpleCapture(96, 96, fx=150., c2w=G.syn.spherical_c2w(20., -10., 3.0), near=0.5, far=3.14)
    coords = O.all_pixel_coords(cap.shape)
    coords = coords[(coords[:, 1] >= 44) & (coords[:, 1] < 52)]
    o, d = O.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, coords)
    o, d = o.astype(np.float32), d.astype(np.float32)
    coarse, fine, human = G.nets[0], G.nets[1], G.nets[2]
    meshes = [G.ray.mesh_to_device(p, faces, t, 'cuda') for p, t in zip(posed_l, T_l)]
    trace = {}
    rgb, depth = G.render.render_multi_rays(coarse[0], fine[0], [human[0]] * 3, cu(o), cu(d), cap.near['bkg'], cap.far['bkg'],
                                            [cu(p) for p in posed_l], meshes, 192, 128, trace=trace)
    assert trace['bkg_z'][0].shape[1] == 320
    c_rgb, hits = conditional_hybrid(G, {'fine': fine[1], 'human': human[1]}, o, d, trace, 3, 192)
    n_hit = [h.size for h in hits]
    assert all(n > 50 for n in n_hit), n_hit
    e = np.abs(rgb.cpu().numpy() - c_rgb).max(-1)
    print(f"[C5] {o.shape[0]} rays, hits per actor {n_hit}, merged 896 samples: conditional Linf {e.max():.2e}")
    assert e.max() < 1e-4
    # the merge reads compact per-actor lists through row indices (one shared placeholder row for every missed ray): bit-identical to the full [R, S] arrays
    assert G.render.MULTI_COMPACT
    G.render.MULTI_COMPACT = False
    try:
        rgb_full, depth_full = G.render.render_multi_rays(coarse[0], fine[0], [human[0]] * 3, cu(o), cu(d), cap.near['bkg'], cap.far['bkg'],
                                                          [cu(p) for p in posed_l], meshes, 192, 128)
    finally:
        G.render.MULTI_COMPACT = True
    assert torch.equal(rgb, rgb_full) and torch.equal(depth, depth_full)
    # ... and with an actor nobody hits (its list is the placeholder row alone)
    far_l = [cu(p) for p in posed_l[:2]] + [cu(posed_l[2] + np.array([50., 0., 0.], np.float32))]
    far_T = T_l[2].copy()
    far_T[:, :3, 3] += np.array([50., 0., 0.])
    far_mesh = meshes[:2] + [G.ray.mesh_to_device(posed_l[2] + np.array([50., 0., 0.], np.float32), faces, far_T, 'cuda')]
    a = G.render.render_multi_rays(coarse[0], fine[0], [human[0]] * 3, cu(o), cu(d), cap.near['bkg'], cap.far['bkg'], far_l, far_mesh, 192, 128)
    G.render.MULTI_COMPACT = False
    try:
        b = G.render.render_multi_rays(coarse[0], fine[0], [human[0]] * 3, cu(o), cu(d), cap.near['bkg'], cap.far['bkg'], far_l, far_mesh, 192, 128)
    finally:
        G.render.MULTI_COMPACT = True
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_c4_c5_full_size_frame_properties(G):
    """BASELINE configs 4 and 5 at their full frame sizes (1280x720 hybrid, merged 384; 1920x1080 three actors, merged 896) through the
    size-independent properties the domain offers: rays are independent (a frame equals its slices, however it is cut into launches),
    rays that miss every body are the background render, colours / depths stay in range."""
    posed, T = G.syn.twist_transforms(G.syn.capsule_mesh()[0])
    faces = np.ascontiguousarray(G.syn.capsule_mesh()[1][:, :3], np.int32)
    coarse, fine, human = G.nets[0][0], G.nets[1][0], G.nets[2][0]
    # ---- C4
    cap = G.syn.SimpleCapture(1280, 720, fx=1.25 * 1280, c2w=G.syn.spherical_c2w(20., -10., 3.0), near=0.5, far=4.0)
    o, d = G.ray.shot_all_rays_dev(cap, 'cuda')
    mesh = G.ray.mesh_to_device(posed, faces, T, 'cuda')
    verts = cu(posed)
    trace = {}
    rgb, depth, acc = G.render.render_hybrid_rays(coarse, fine, human, o, d, cap.near['bkg'], cap.far['bkg'], verts, mesh, 128, 128, trace=trace)
    hit = trace['hit'][0]
    R = o.shape[0]
    assert 0.02 * R < hit.numel() < 0.6 * R, hit.numel()
    assert torch.isfinite(rgb).all() and rgb.min() >= -1e-5 and rgb.max() <= 1 + 1e-5 and depth.min() >= 0
    rows = slice(1280 * 330, 1280 * 390)                                           # 60 rows through the body
    old = G.render.MAX_RAYS_PER_LAUNCH
    try:
        G.render.MAX_RAYS_PER_LAUNCH = 7001
        s_rgb, s_depth, s_acc = G.render.render_hybrid_rays(coarse, fine, human, o[rows].contiguous(), d[rows].contiguous(), cap.near['bkg'],
                                                            cap.far['bkg'], verts, mesh, 128, 128)
    finally:
        G.render.MAX_RAYS_PER_LAUNCH = old
    assert torch.equal(s_rgb, rgb[rows]) and torch.equal(s_depth, depth[rows]) and torch.equal(s_acc, acc[rows])
    miss = torch.ones(R, dtype=torch.bool, device='cuda')
    miss[hit.long()] = False
    idx = torch.nonzero(miss)[:, 0][:: max(1, int(miss.sum()) // 20000)]          # ~20 000 miss rays across the frame
    b_rgb, b_depth = G.render.render_vanilla_rays(coarse, fine, o[idx].contiguous(), d[idx].contiguous(), cap.near['bkg'], cap.far['bkg'], 128, 128)
    assert torch.equal(b_rgb, rgb[idx]) and torch.equal(b_depth, depth[idx])      # render_utils.py:300-319: misses are the background pass
    print(f"[C4 full] 1280x720: {hit.numel()} of {R} rays hit the body; frame = slices bit for bit, {idx.numel()} miss rays = background render")
    # ---- C5
    cap = G.syn.SimpleCapture(1920, 1080, fx=1.25 * 1920, c2w=G.syn.spherical_c2w(20., -10., 3.0), near=0.5, far=3.14)
    o, d = G.ray.shot_all_rays_dev(cap, 'cuda')
    shifts = [np.zeros(3), np.array([0.35, 0.0, 0.2]), np.array([-0.3, 0.05, -0.15])]
    posed_l, meshes = [], []
    for s in shifts:
        t = T.copy()
        t[:, :3, 3] += s
        posed_l.append(cu((posed + s).astype(np.float32)))
        meshes.append(G.ray.mesh_to_device((posed + s).astype(np.float32), faces, t, 'cuda'))
    trace = {}
    rgb, depth = G.render.render_multi_rays(coarse, fine, [human] * 3, o, d, cap.near['bkg'], cap.far['bkg'], posed_l, meshes, 192, 128, trace=trace)
    assert torch.isfinite(rgb).all() and rgb.min() >= -1e-5 and rgb.max() <= 1 + 1e-5 and depth.min() >= 0
    assert all(h.numel() > 1000 for h in trace['hit'])
    rows = slice(1920 * 520, 1920 * 550)
    try:
        G.render.MAX_RAYS_PER_LAUNCH = 9973
        s_rgb, s_depth = G.render.render_multi_rays(coarse, fine, [human] * 3, o[rows].contiguous(), d[rows].contiguous(), cap.near['bkg'],
                                                    cap.far['bkg'], posed_l, meshes, 192, 128)
    finally:
        G.render.MAX_RAYS_PER_LAUNCH = old
    assert torch.equal(s_rgb, rgb[rows]) and torch.equal(s_depth, depth[rows])
    print(f"[C5 full] 1920x1080, hits per (launch, actor) {[h.numel() for h in trace['hit']]}: frame = slices bit for bit")
