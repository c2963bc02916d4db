"""-m gpu: the fused per-pass entry points (include/neuman_hip.h: nm_render_rays_bkg, nm_render_rays_human, nm_merge_composite -- one C call
per pass of the reference's renderers, utils/render_utils.py:131-151, 213-229, 330-345) against the step-by-step entry points: the same
kernels enqueued by one call, so every output is equal BIT FOR BIT; empty batches; their composites against the oracle."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import compositing

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(nets):
    import types
    from neuman_hip import _lib, ray_utils, render_utils, synthetic
    return types.SimpleNamespace(lib=_lib, ray=ray_utils, render=render_utils, syn=synthetic, nets={k: j.cuda() for k, (j, sd, spec) in nets.items()})


def rays(G, R, seed=0):
    g = torch.Generator(device='cuda').manual_seed(seed)
    o = torch.randn((R, 3), device='cuda', generator=g) * 0.2 + torch.tensor([0., 0., -3.], device='cuda')
    d = torch.nn.functional.normalize(torch.randn((R, 3), device='cuda', generator=g) * 0.15 + torch.tensor([0., 0., 1.], device='cuda'), dim=-1)
    return o.contiguous(), d.contiguous()


@pytest.mark.parametrize("R,S,N", [(1, 4, 4), (700, 37, 21), (3000, 128, 128), (513, 64, 0)])
def test_background_pass_in_one_call(G, R, S, N):
    o, d = rays(G, R)
    near, far = torch.full((R,), 0.5, device='cuda'), torch.full((R,), 5.0, device='cuda')
    fine = G.nets[1] if N else None
    raw_a, z_a = G.render.bkg_pass_rays(G.nets[0], fine, o, d, near, far, S, N, True)
    raw_b, z_b = G.render.bkg_pass_rays_fused(G.nets[0], fine, o, d, near, far, S, N, True)
    assert torch.equal(z_a, z_b) and torch.equal(raw_a, raw_b)
    # and with the composite inside the call
    lib = G.lib.lib()
    ws = torch.empty(int(lib.nm_render_rays_bkg_workspace_floats(R, S, N)) + 4, device='cuda')
    raw_c, z_c = torch.empty_like(raw_a), torch.empty_like(z_a)
    rgb, depth, acc = torch.empty((R, 3), device='cuda'), torch.empty(R, device='cuda'), torch.empty(R, device='cuda')
    t_vals = torch.linspace(0., 1., steps=S, device='cuda')
    u = torch.linspace(0., 1., steps=N, device='cuda') if N else None
    P = G.lib.dev_ptr
    G.lib.check(lib.nm_render_rays_bkg(G.nets[0].handle(), fine.handle() if fine is not None else None, P(o), P(d), P(near), P(far), R, S, N, P(t_vals), P(u), 1,
                                       G.nets[0]._prec(None, None if fine is not None else 'shading'), fine._prec(None, 'shading') if fine is not None else 0,
                                       P(ws), P(raw_c), P(z_c), P(rgb), P(depth), P(acc), G.lib.stream_ptr()), "nm_render_rays_bkg")
    r2 = G.render.raw2outputs(raw_a, z_a, d, want_weights=False)
    assert torch.equal(raw_c, raw_a) and torch.equal(rgb, r2[0]) and torch.equal(acc, r2[2]) and torch.equal(depth, r2[4])
    o_rgb = compositing.raw2outputs(raw_a.cpu().numpy(), z_a.cpu().numpy(), d.cpu().numpy())[0]
    assert np.abs(rgb.cpu().numpy() - o_rgb).max() < 2e-5


def test_human_pass_and_merge_composite_in_one_call(G):
    verts_c, faces = G.syn.capsule_mesh(n_rings=20, n_seg=24)
    posed, T = G.syn.twist_transforms(verts_c)
    mesh = G.ray.mesh_to_device(posed, np.ascontiguousarray(faces[:, :3], np.int32), T, 'cuda')
    o, d = rays(G, 600, 3)
    near, far = G.ray.geometry_guided_near_far(o, d, torch.as_tensor(posed).cuda(), 0.2)
    hit, _ = G.ray.compact_hits(near, far)
    assert 50 < hit.numel() < 600
    ho, hd, hn, hf = (G.ray.gather_rows(x, hit) for x in (o, d, near, far))
    human = G.nets[2]
    for render_can in (False, True):
        trace = {}
        raw, z = G.render.human_pass_rays(human, ho, hd, hn, hf, 48, mesh, render_can, 0.7, trace=trace)
        # step by step
        if render_can:
            z2 = G.ray.sample_z(ho, hd, hn, hf, 48)[2]
            raw2 = human.forward_rays(ho, hd, z2, sigma_scale=0.7, role='shading')
        else:
            pts, _, z2 = G.ray.sample_z(ho, hd, hn, hf, 48, want_points=True)
            cp, cd, _ = G.ray.warp_to_canonical_dev(pts, mesh)
            raw2 = human(cp, cd, sigma_scale=0.7, role='shading')
            assert torch.equal(trace['can_pts'][0], cp) and torch.equal(trace['can_dirs'][0], cd)
        assert torch.equal(z, z2) and torch.equal(raw, raw2)
    # merge + composite
    bkg_raw, bkg_z = G.render.bkg_pass_rays(G.nets[0], G.nets[1], ho, hd, torch.full_like(hn, 0.5), torch.full_like(hn, 5.0), 32, 32, True)
    rgb, depth, acc = G.render.merge_composite(bkg_z, bkg_raw, z, raw, hd)
    z_all, raw_all = G.render.merge_sorted(bkg_z, bkg_raw, z, raw)
    r2 = G.render.raw2outputs(raw_all, z_all, hd, want_weights=False)
    assert torch.equal(rgb, r2[0]) and torch.equal(acc, r2[2]) and torch.equal(depth, r2[4])
    # empty batches are accepted
    e = torch.empty((0, 3), device='cuda')
    raw0, z0 = G.render.human_pass_rays(human, e, e, torch.empty(0, device='cuda'), torch.empty(0, device='cuda'), 16, mesh, False, 1.0)
    assert raw0.shape == (0, 16, 4) and z0.shape == (0, 16)


def test_hybrid_batch_as_one_call_is_bit_identical():
    """nm_render_rays_hybrid (render_utils.render_hybrid_rays_fused) = render_hybrid_rays' per-batch sequence of calls, bit for bit:
    a body in front of the background at 64 + 64 background and 64 human samples, hits and misses in one batch"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
    import posed_scene as PS
    from neuman_hip import ray_utils, render_utils, synthetic
    g = PS.load()
    c = PS.cap(g, 'hybrid')
    o, d = (torch.as_tensor(x).cuda().contiguous() for x in PS.frame_rays(c))
    mesh = ray_utils.mesh_to_device(g['posed_verts'], np.ascontiguousarray(g['faces'][:, :3], np.int32), g['T'], 'cuda')
    nets = [synthetic.make_joiner(0).cuda(), synthetic.make_joiner(1).cuda(), synthetic.make_joiner(2, 'rotate').cuda()]
    verts = torch.as_tensor(g['posed_verts']).cuda().float()
    a = render_utils.render_hybrid_rays(nets[0], nets[1], nets[2], o, d, c.near['bkg'], c.far['bkg'], verts, mesh, 64, 64, trace={})   # (trace: the step-by-step path)
    b = render_utils.render_hybrid_rays_fused(nets[0], nets[1], nets[2], o, d, c.near['bkg'], c.far['bkg'], verts, mesh, 64, 64)
    assert float(a[2].max()) > 0 and float((a[2] == 0).float().mean()) > 0.1                # hits and misses
    for x, y, what in zip(a, b, ("rgb", "depth", "acc")):
        assert torch.equal(x, y), what
    # one net, no fine pass (render_hybrid_nerf with a coarse-only background model)
    a = render_utils.render_hybrid_rays(nets[0], None, nets[2], o, d, c.near['bkg'], c.far['bkg'], verts, mesh, 64, 0, trace={})
    b = render_utils.render_hybrid_rays_fused(nets[0], None, nets[2], o, d, c.near['bkg'], c.far['bkg'], verts, mesh, 64, 0)
    for x, y, what in zip(a, b, ("rgb", "depth", "acc")):
        assert torch.equal(x, y), what


def _lists(R, sizes, seed, ties=True):
    """k sorted z lists with records; some exact cross-list ties (the reference's sort leaves their order to torch; ours is stable)"""
    g = torch.Generator(device='cuda').manual_seed(seed)
    zs, raws = [], []
    for i, S in enumerate(sizes):
        z = torch.sort(torch.rand((R, S), device='cuda', generator=g) * 3.0 + 0.3 * i, dim=1)[0].contiguous()
        if ties and i > 0:
            z[:, S // 3] = zs[0][:, min(sizes[0] - 1, S // 2)]                  # an exact tie with list 0
            z = torch.sort(z, dim=1)[0].contiguous()
        raw = (torch.randn((R, S, 4), device='cuda', generator=g) * torch.tensor([1., 1., 1., 4.], device='cuda')).contiguous()
        zs.append(z)
        raws.append(raw)
    return zs, raws


@pytest.mark.parametrize("R,sizes", [(1, (3, 2)), (777, (256, 128)), (301, (320, 192, 192, 192)), (130, (64, 17, 5)), (64, (384,))])
def test_merge_composite_as_one_kernel_is_bit_identical(G, R, sizes):
    """nm_merge_composite_lists (k-way merge + raw2outputs in one kernel, the merged list in LDS only) == nm_merge_sorted list by list +
    nm_composite, including 320 + 3 x 192 = 896 merged samples (BASELINE config 5) and exact z ties between lists"""
    _, d = rays(G, R, 5)
    zs, raws = _lists(R, sizes, 11)
    z_all, raw_all = zs[0], raws[0]
    for z, raw in zip(zs[1:], raws[1:]):
        z_all, raw_all = G.render.merge_sorted(z_all, raw_all, z, raw)
    ref = G.render.raw2outputs(raw_all, z_all, d, white_bkg=True, want_weights=False)
    for white in (True, False):
        ref = G.render.raw2outputs(raw_all, z_all, d, white_bkg=white, want_weights=False)
        rgb, depth, acc = G.render.merge_composite_lists(zs, raws, d, white)
        assert torch.equal(rgb, ref[0]) and torch.equal(acc, ref[2]) and torch.equal(depth, ref[4]), (sizes, white)
    if len(sizes) == 2:                                           # the two-list entry point is the same kernel
        rgb2, depth2, acc2 = G.render.merge_composite(zs[0], raws[0], zs[1], raws[1], d, False)
        assert torch.equal(rgb2, rgb) and torch.equal(depth2, depth) and torch.equal(acc2, acc)


def test_merge_composite_reads_a_list_in_place_through_a_row_index(G):
    """rows: the background list exists for ALL rays of a batch; the hit rays' rows are read in place (no gathered copy)"""
    R_all, R = 900, 333
    _, d_all = rays(G, R_all, 6)
    (zb,), (rawb,) = _lists(R_all, (256,), 3, ties=False)
    hit = torch.sort(torch.randperm(R_all, device='cuda')[:R])[0].to(torch.int32)
    (zh,), (rawh,) = _lists(R, (128,), 4, ties=False)
    hd = d_all[hit.long()].contiguous()
    a = G.render.merge_composite_lists([zb, zh], [rawb, rawh], hd, True, rows=[hit, None])
    b = G.render.merge_composite_lists([zb[hit.long()].contiguous(), zh], [rawb[hit.long()].contiguous(), rawh], hd, True)
    assert all(torch.equal(x, y) for x, y in zip(a, b))


@pytest.mark.parametrize("R,S,N", [(1, 3, 1), (1000, 128, 128), (257, 192, 128), (64, 33, 77)])
def test_coarse_tail_as_one_kernel_is_bit_identical(G, R, S, N):
    """nm_importance_from_raw (compositing weights -> inverse CDF -> sorted merge in one kernel, sigma read once) == nm_composite's weights +
    nm_importance_z: same sample positions, same weights, bit for bit"""
    _, d = rays(G, R, 7)
    (z,), (raw,) = _lists(R, (S,), 21, ties=False)
    w = G.render.raw2outputs(raw, z, d)[3]
    z_ref = G.ray.importance_z(z, w, N)
    z_one, w_one = G.ray.importance_z_from_raw(raw, z, d, N, want_weights=True)
    assert torch.equal(z_one, z_ref) and torch.equal(w_one, w)
    z_two, none = G.ray.importance_z_from_raw(raw, z, d, N)
    assert none is None and torch.equal(z_two, z_ref)
