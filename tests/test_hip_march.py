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


# ---- round 3: the coarse pass, the human passes and the hybrid renderers ------------------------------------------------------
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import posed_scene as PS  # noqa: E402


def test_density_only_chunks_are_bit_identical_to_the_density_only_launch(scene):
    """the coarse pass marched with eps = 0 (nm_mlp_sigma_ray_chunk) = nm_mlp_sigma_rays, bit for bit"""
    o, d, net = scene['o'], scene['d'], scene['net']
    R = o.shape[0]
    _, _, z = scene['ray'].sample_z(o, d, torch.zeros(R, device='cuda'), torch.full((R,), 3.14, device='cuda'), 128)
    full = net.forward_rays(o, d, z, sigma_only=True)
    marched = scene['render'].march_pass_rays(net, o, d, z, 0.0, chunk=48, role=None, sigma_only=True)
    assert torch.equal(marched, full)


def test_adaptive_march_stops_launching_and_stays_within_eps(scene):
    """the same frame slice as above with the adaptive schedule (one host read per chunk): fewer evaluations than fixed 32-sample
    chunks, still within eps; and the vanilla renderer with the coarse pass marched too"""
    o, d, net = scene['o'], scene['d'], scene['net']
    z = fine_z(scene)
    full = net.forward_rays(o, d, z, role='shading')
    rgb_full = scene['render'].raw2outputs(full, z, d)[0]
    st_fixed, st_adapt = {}, {}
    fixed = scene['render'].march_pass_rays(net, o, d, z, 1e-4, stats=st_fixed, adaptive=False)
    adapt = scene['render'].march_pass_rays(net, o, d, z, 1e-4, stats=st_adapt)
    for m in (fixed, adapt):
        assert (scene['render'].raw2outputs(m, z, d)[0] - rgb_full).abs().max().item() <= 1e-4
    print(f"[march] evaluated: fixed chunks {st_fixed['evaluated'] / st_fixed['total']:.3f} in {st_fixed['launches']} launches, adaptive "
          f"{st_adapt['evaluated'] / st_adapt['total']:.3f} in {st_adapt['launches']} launches")
    assert st_adapt['evaluated'] < st_fixed['evaluated']
    R = scene['render']
    a, _ = R.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128)
    old = R.TERMINATION_EPS
    try:
        R.TERMINATION_EPS = 1e-4
        tr = {}
        b, _ = R.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128, trace=tr)
    finally:
        R.TERMINATION_EPS = old
    tr0 = {}
    R.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128, trace=tr0)
    assert torch.equal(tr['bkg_z'][0], tr0['bkg_z'][0]), "the marched coarse pass must leave the importance samples bit-identical"
    c, f = tr['march_coarse'][0], tr['march'][0]
    print(f"[march] vanilla renderer, eps 1e-4: coarse pass {c['evaluated'] / c['total']:.3f} evaluated, fine pass {f['evaluated'] / f['total']:.3f}, "
          f"colour Linf vs every sample {(a - b).abs().max().item():.2e}")
    assert c['evaluated'] < 0.8 * c['total'] and (a - b).abs().max().item() <= 1e-4


@pytest.fixture(scope="module")
def body():
    from neuman_hip import ray_utils, render_utils, synthetic
    g = PS.load()
    g['R'], g['ray'] = render_utils, ray_utils
    g['mesh'] = ray_utils.mesh_to_device(g['posed_verts'], np.ascontiguousarray(g['faces'][:, :3], np.int32), g['T'], 'cuda')
    g['meshes'] = [ray_utils.mesh_to_device(v, np.ascontiguousarray(g['faces'][:, :3], np.int32), t, 'cuda') for v, t in zip(g['posed_l'], g['T_l'])]
    g['bkg'] = synthetic.make_joiner(1, preset='opaque').cuda()
    g['human'] = synthetic.make_joiner(2, 'rotate', preset='opaque').cuda()
    return g


def test_human_march_without_termination_is_bit_identical(body):
    """human_march_rays at eps = 0 = human_pass_rays (one fused call): every sample's warp, direction and network output"""
    c = PS.cap(body, 'posed')
    o, d = (cu(x) for x in PS.frame_rays(c))
    near, far = body['ray'].geometry_guided_near_far(o, d, cu(body['posed_verts']), 0.2)
    hit, _ = body['ray'].compact_hits(near, far)
    assert hit.numel() > 200
    ho, hd = body['ray'].gather_rows(o, hit), body['ray'].gather_rows(d, hit)
    hn, hf = body['ray'].gather_rows(near, hit), body['ray'].gather_rows(far, hit)
    for S_h, chunk in ((128, 16), (128, 40), (33, 16)):                                  # 33 = 2 x 16 + a lone last sample
        full, z = body['R'].human_pass_rays(body['human'], ho, hd, hn, hf, S_h, body['mesh'])
        marched, zm = body['R'].human_march_rays(body['human'], ho, hd, hn, hf, S_h, body['mesh'], 0.0, chunk=chunk)
        assert torch.equal(z, zm) and torch.equal(marched, full), (S_h, chunk)


@pytest.mark.parametrize("which", ["posed", "hybrid", "multi", "hybrid-semi", "multi-semi"])
def test_renderers_with_termination(body, which):
    """opaque body in front of an opaque background: eps = 1e-4 skips evaluations in every pass and moves no pixel by more than the
    bound (eps for the body alone, 2 eps for body + background, (1 + actors) eps for three bodies); eps = 0 is the plain path.
    '-semi': a SEMI-TRANSPARENT body (the dense preset: sigma of a few units, > 0 on the last sample of most rays) in front of the
    opaque background, which stays visible through it -- the cut is decided on the merged list's transmittance (on the body's list
    alone the terminal 1e10 interval would make every such ray opaque and the background behind it would be dropped)."""
    from neuman_hip import synthetic
    R = body['R']
    semi = which.endswith('-semi')
    which = which.split('-')[0]
    c = PS.cap(body, which)
    o, d = (cu(x) for x in PS.frame_rays(c))
    bkg, human = body['bkg'], body['human']
    if semi:
        human = synthetic.make_joiner(2, 'rotate').cuda()

    def run(trace=None):
        if which == 'posed':
            return R.render_smpl_nerf_rays(human, o, d, cu(body['posed_verts']), body['mesh'], 128, True, False, 0.2, 1.0, trace=trace)[0]
        if which == 'hybrid':
            return R.render_hybrid_rays(bkg, bkg, human, o, d, c.near['bkg'], c.far['bkg'], cu(body['posed_verts']), body['mesh'], 128, 128, trace=trace)[0]
        return R.render_multi_rays(bkg, bkg, [human] * 3, o, d, c.near['bkg'], c.far['bkg'], [cu(v) for v in body['posed_l']], body['meshes'], 192, 128,
                                   trace=trace)[0]

    a = run()
    old = R.TERMINATION_EPS
    try:
        R.TERMINATION_EPS = 1e-4
        tr = {}
        b = run(tr)
    finally:
        R.TERMINATION_EPS = old
    assert torch.equal(run(), a)
    bound = {'posed': 1e-4, 'hybrid': 2e-4, 'multi': 4e-4}[which]
    hs = tr['march_human']
    he, ht = sum(s_['human_evaluated'] for s_ in hs), sum(s_['human_total'] for s_ in hs)
    msg = f"[march] {which}: body passes {he / ht:.3f} evaluated"
    if which != 'posed':
        f_, c_ = tr['march'][0], tr['march_coarse'][0]
        msg += f", background coarse {c_['evaluated'] / c_['total']:.3f}, fine {f_['evaluated'] / f_['total']:.3f}"
        assert f_['evaluated'] < 0.9 * f_['total']
    e = (a - b).abs().max().item()
    print(msg + f", colour Linf vs every sample {e:.2e} (bound {bound:g})" + (" [semi-transparent body]" if semi else ""))
    assert (semi or he < 0.8 * ht) and e <= bound


@pytest.mark.parametrize("sizes", [(256, 128), (320, 192, 192, 192), (7,), (5, 1, 3)])
def test_merged_intervals_equal_the_sorted_lists_differences(sizes):
    """render_utils.merged_intervals (nm_merged_intervals: binary searches, nothing sorted) against the definition: stable argsort of
    cat(lists) (ties: the earlier list first, the order of render_utils.py:330-337), differences of the sorted values, 1e10 at the end,
    scattered back -- bit for bit, with exact cross-list ties and repeated values inside a list in the data"""
    from neuman_hip import render_utils
    rng = np.random.default_rng(sum(sizes))
    R = 777
    lists = [np.sort(rng.uniform(0.5, 4.0, size=(R, S)).astype(np.float32), 1) for S in sizes]
    if len(lists) > 1:
        lists[1][:, 0] = lists[0][:, min(2, sizes[0] - 1)]                      # a cross-list tie on every ray
        lists[-1][::3, -1] = lists[0][::3, -1]                                   # ... and one at the very end of the merged list
    if sizes[0] > 4:
        lists[0][:, 4] = lists[0][:, 3]                                          # a repeated value inside a list
    lists = [np.sort(x, 1) for x in lists]                                      # (the planted values keep their ties; every list sorted again)
    z = np.concatenate(lists, 1)
    order = np.argsort(z, 1, kind='stable')
    zs = np.take_along_axis(z, order, 1)
    dz_s = np.concatenate([zs[:, 1:] - zs[:, :-1], np.full((R, 1), 1e10, np.float32)], 1).astype(np.float32)
    want = np.empty_like(z)
    np.put_along_axis(want, order, dz_s, 1)
    got = render_utils.merged_intervals([cu(x) for x in lists])
    off = 0
    for S, g in zip(sizes, got):
        assert np.array_equal(g.cpu().numpy(), want[:, off:off + S])
        off += S
