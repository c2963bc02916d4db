"""-m gpu: the HIP renderers END TO END against frames the REFERENCE ITSELF rendered behind the warp (tests/golden/posed.npz: the
reference's own warp_samples_to_canonical, render_smpl_nerf(render_can=False), render_hybrid_nerf, render_hybrid_nerf_multi_persons,
unmodified, libigl's three calls supplied by tests/golden/igl_shim.py), at the BASELINE sample counts, on whole 40 x 32 frames around an
SMPL-sized body (V = 6890, F = 13776).  Statements, as in tests/test_oracle_posed_golden.py:

* CONDITIONAL on the reference's recorded float32-ill-conditioned intermediates (importance-sample positions; per-actor near / far),
  replayed through the product renderers' `given` hook: every pixel of a band of rows through the bodies within 1e-4 of the
  reference's frame, with two kinds of ray counted and listed instead: rays with an exact background / human z tie, whose order the
  reference leaves to torch.sort(stable=False), and rays with a sample whose closest-point FOOT the device (float32, like libigl on
  float32 input) and the float64 shim place on different faces -- inside the body, at the medial axis, two feet are equidistant and
  the canonical point jumps with the choice (found by running the oracle's warp on the device's own sample points);
* the device's own intermediates against the recordings, in their own units;
* END TO END, nothing replayed, against the ARBITER -- the reference's own renderer run in FLOAT64 on the same frame (tests/golden/arbiter.npz,
  make_golden_f64.py) --: the device is beyond 1e-4 of it on no more rays than the reference's own float32 frame is (+ a quarter), and every ray
  that deviates from the reference's float32 frame is accounted for by a changed merged sample order, a displaced importance sample or a displaced near / far.
"""
import numpy as np
import pytest
import torch

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers"))
import posed_scene as PS  # noqa: E402

pytestmark = pytest.mark.gpu
# END-TO-END counts are scored against the ARBITER (PS.arbiter: the reference's own renderer run in float64 on the frame): rays on which the device is
# beyond 1e-4 of it <= rays on which the reference's own float32 frame is, + a quarter (oracle.attribution.allowed_count)
TIE_CAP = {'small': 4, 'big': 6}            # measured 3 / 4
# rays beyond 1e-4 over the WHOLE frame in the conditional runs (flagged or not): measured + 3
COND_CAP = {'small': {'posed': 7, 'hybrid': 4, 'multi': 6}, 'big': {'posed': 11, 'hybrid': 4}}     # measured 4 / 1 / 3 and 8 / 1


def cu(x, dt=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(x)).to('cuda', dt).contiguous()


SUMMARY = {}               # the numbers of the last run of each check, for __graft_entry__.smoke()'s parity lines


@pytest.fixture(scope="module")
def S(nets):
    return build_scene({k: j for k, (j, sd, spec) in nets.items()})


def build_scene(joiners, with_big=True):
    """{seed: Joiner (0, 1: background coarse / fine, 2: human)} -> the scene of tests/golden/posed.npz (and posed_big.npz) on the device"""
    from neuman_hip import ray_utils, render_utils
    g = PS.load()
    g['R'], g['ray'] = render_utils, ray_utils
    g['dev_nets'] = {k: j.cuda() for k, j in joiners.items()}
    g['mesh'] = ray_utils.mesh_to_device(g['posed_verts'], np.ascontiguousarray(g['faces'][:, :3], np.int32), g['T'], 'cuda')
    g['meshes'] = [ray_utils.mesh_to_device(v, np.ascontiguousarray(g['faces'][:, :3], np.int32), t, 'cuda') for v, t in zip(g['posed_l'], g['T_l'])]
    g['nrays'] = PS.W * PS.H
    if with_big:
        b = PS.load_big()                                         # the same body and nets on 64 x 64 frames (tests/golden/posed_big.npz)
        for k in ('R', 'ray', 'dev_nets', 'mesh'):
            b[k] = g[k]
        b['nrays'] = b['W'] * b['H']
        g['big'] = b
    return g


def pick(S, size):
    """the 40 x 32 goldens with their 320-ray bands, or the 64 x 64 ones with the WHOLE frame as the band"""
    if size == 'small':
        return S, None
    return S['big'], (0, S['big']['nrays'])


def test_warp_vs_the_references_own_warp(S):
    """utils/ray_utils.py:48-66 as the reference executed it, 64 rays x 128 samples through the body's interior and shell.  The
    invariant of the closest-point query is the DISTANCE; where two feet on different faces are nearly equidistant (the medial axis
    inside the body) float32 (the device, like libigl on float32 input) and float64 (the shim's arithmetic) may pick either, and
    the canonical point jumps with the foot: such samples are counted, everything else is held tight."""
    pts = S['warp_pts']
    cp, cd, cl = S['ray'].warp_samples_to_canonical(pts, S['posed_verts'], S['faces'], S['T'])
    dist, r_dist = np.linalg.norm(cl - pts, axis=-1), np.linalg.norm(S['warp_closest'] - pts, axis=-1)
    same = np.abs(cl - S['warp_closest']).max(-1) < 2e-5
    e = [np.abs(cp - S['warp_can_pts'])[same].max(), np.abs(cd - S['warp_can_dirs'])[same[:, :-1] & same[:, 1:]].max() if False else None]
    pair = same.copy()
    pair[:, :-1] &= same[:, 1:]                                               # a direction is the difference of two consecutive samples
    pair[:, -1] = pair[:, -2]
    e_dir = np.abs(cd - S['warp_can_dirs'])[pair].max()
    print(f"[warp vs reference] distance Linf {np.abs(dist - r_dist).max():.2e}; same foot on {same.mean() * 100:.2f} % of {same.size} samples: can_pts {e[0]:.2e}, "
          f"can_dirs {e_dir:.2e} there; overall can_pts {np.abs(cp - S['warp_can_pts']).max():.2e}, closest {np.abs(cl - S['warp_closest']).max():.2e}")
    SUMMARY['warp'] = {'distance_linf': float(np.abs(dist - r_dist).max()), 'same_foot_pct': float(same.mean() * 100), 'can_pts_same_foot': float(e[0]),
                       'samples': int(same.size)}
    assert np.abs(dist - r_dist).max() < 2e-6 and same.mean() > 0.99
    assert e[0] < 1e-5 and e_dir < 2e-3 and np.abs(cp - S['warp_can_pts']).max() < 5e-4


BAND = (480, 800)          # eight rows through the body (the CPU test's rays)
FOOT_TOL = 1e-5          # measured on the band: 6 of 275 hit rays beyond it (the two rays off by 9e-4 / 3e-4 among them, at 1.3e-5); unflagged rays <= 4.7e-5
MULTI_BAND = (560, 720)




def foot_jump_rays(trace, k, o, d, verts, faces, T, band):
    """rays of actor k within `band` that hold a sample whose canonical point differs from the oracle's warp of the SAME point by more
    than FOOT_TOL: the closest-point foot is ill conditioned there -- deep inside the body the feet on neighbouring
    faces are equidistant to 1e-7 while lying 1e-5 apart (at the medial axis: on opposite sides of the body), and the float32 search
    (the device, like libigl on float32 input) and the float64 shim pick different ones.  The distance, the query's invariant, agrees to
    6e-8 (test_warp_vs_the_references_own_warp); the finite-difference directions divide the foot's displacement by the sample spacing.

    foot_jump_rays.last keeps (rays, can_pts deviation, can_dirs deviation, sample spacing) of the band's hit rays: the big-frame test lists them
    for the rays beyond 1e-4 (a foot displaced by 5e-6 over a spacing of 5e-3 turns the finite-difference direction by 1e-3 and the colour follows the
    view direction with a slope of ~0.2; on rays grazing the 0.2 shell the reference's OWN float32 rounding of the closest point turns its directions by
    1e-3..1e-2 -- the CPU oracle and the reference disagree on exactly those rays by the same amounts, ray 2427 of posed_big.npz)."""
    from oracle import warp
    hit = trace['hit'][k].cpu().numpy()
    if trace['can_pts'][k] is None:
        return np.zeros(0, np.int64)
    sel = np.nonzero((hit >= band[0]) & (hit < band[1]))[0]
    if sel.size == 0:
        return np.zeros(0, np.int64)
    rays = hit[sel]
    z = trace['human_z'][k].cpu().numpy()[sel]
    pts = (o[rays, None, :] + d[rays, None, :] * z[..., None]).astype(np.float32)
    ocp, ocd, _ = warp.warp_samples_to_canonical(pts, verts, faces, T)
    dev = np.abs(ocp - trace['can_pts'][k].cpu().numpy()[sel]).max((-1, -2))
    dev_d = np.abs(ocd - trace['can_dirs'][k].cpu().numpy()[sel]).max((-1, -2))
    spacing = (z[:, -1] - z[:, 0]) / (z.shape[1] - 1)
    foot_jump_rays.last = (rays, dev, dev_d, spacing)
    return rays[dev > FOOT_TOL]


@pytest.mark.parametrize("size", ["small", "big"])
def test_posed_human_frame(S, size):
    S, whole = pick(S, size)
    NR = S['nrays']
    c = PS.cap(S, 'posed') if size == 'small' else PS.cap_big(S)
    o, d = PS.frame_rays(c)
    net = S['dev_nets'][2]
    ref = S['posed_rgb'].reshape(-1, 3)
    given = {'near_far': [(cu(S['posed_near']), cu(S['posed_far']))]}
    trc = {}
    rgb, depth, acc = S['R'].render_smpl_nerf_rays(net, cu(o), cu(d), cu(S['posed_verts']), S['mesh'], 128, True, False, 0.2, 1.0, given=given, trace=trc)
    e = np.abs(rgb.cpu().numpy() - ref).max(-1)
    ea, ed = np.abs(acc.cpu().numpy() - S['posed_acc'].ravel()), np.abs(depth.cpu().numpy() - S['posed_depth'].ravel())
    hit = S['posed_near'] < S['posed_far']
    a, b = whole or BAND
    jump = foot_jump_rays(trc, 0, o, d, S['posed_verts'], S['faces'], S['T'], (a, b))
    ok = np.ones(NR, bool)
    ok[jump] = False
    band = np.zeros(NR, bool)
    band[a:b] = True
    n_bad = int((e > 1e-4).sum())
    if size == 'small':
        print(f"[posed 128 small, conditional on the reference's near / far] rays {a}..{b} ({(hit & band).sum()} hit): {jump.size} ray(s) with a sample whose foot is on "
              f"another face (Linf there {e[jump].max() if jump.size else 0:.2e}); every other ray: rgb Linf {e[band & ok].max():.2e}, acc {ea[band & ok].max():.2e}, "
              f"depth {ed[band & ok].max():.2e}; whole frame: rays > 1e-4 {n_bad} of {hit.sum()} hit, acc Linf {ea.max():.2e}")
        SUMMARY['posed_' + size] = {'rays': int(band.sum()), 'hit': int((hit & band).sum()), 'cond_linf': float(e[band & ok].max()), 'foot_jump_rays': int(jump.size)}
        assert e[band & ok].max() < 1e-4 and ea[band & ok].max() < 1e-4 and ed[band & ok].max() < 2e-4
        assert jump.size <= 0.05 * (hit & band).sum()                   # measured: 6 of 275 (2.2 %)
    else:
        # The 64 x 64 frame: what carries the statement is the COUNT over the whole frame -- 8 of 1896 hit rays beyond 1e-4 in round 4, cap 11 --
        # not a per-ray flag: round 4's three-criterion flag (foot on another face / direction turned by 5e-4 / spacing < 1e-3) set 285 rays
        # aside to account for 8 and is gone.  The deviating rays are LISTED with the three conditioning numbers of
        # their own samples, so that the log says what each one is; none of it is asserted beyond the count.
        rays_l, dev_l, devd_l, sp_l = foot_jump_rays.last
        at = {int(r): i for i, r in enumerate(rays_l)}
        listing = [(int(r), float(e[r]), float(dev_l[at[int(r)]]), float(devd_l[at[int(r)]]), float(sp_l[at[int(r)]])) for r in np.nonzero(e > 1e-4)[0] if int(r) in at]
        in_jump = sum(1 for r in np.nonzero(e > 1e-4)[0] if r in set(jump.tolist()))
        print(f"[posed 128 big, conditional on the reference's near / far] whole 64 x 64 frame, {hit.sum()} hit rays: rays > 1e-4 {n_bad} (cap {COND_CAP[size]['posed']}), "
              f"Linf {e.max():.2e}, acc Linf {ea.max():.2e}; of those {in_jump} hold a sample whose foot is on another face ({jump.size} such rays in the frame, their Linf "
              f"{e[jump].max() if jump.size else 0:.2e}); median over hit rays {np.median(e[hit]):.1e}; the rays beyond 1e-4 as (ray, rgb dev, can_pts dev, can_dirs dev, spacing): "
              + "; ".join(f"({r}, {x:.1e}, {a_:.1e}, {b_:.1e}, {c_:.1e})" for r, x, a_, b_, c_ in listing))
        SUMMARY['posed_' + size] = {'rays': int(band.sum()), 'hit': int((hit & band).sum()), 'cond_linf_whole_frame': float(e.max()), 'foot_jump_rays': int(jump.size),
                                    'gt_1e4_with_a_foot_jump': int(in_jump), 'median_dev': float(np.median(e[hit]))}
    assert ea.max() < 1e-4
    SUMMARY['posed_' + size]['cond_gt_1e4_whole_frame'] = n_bad
    assert n_bad <= COND_CAP[size]['posed']
    # the device's own near / far and the end-to-end frame
    tr = {}
    rgb, depth, acc = S['R'].render_smpl_nerf_rays(net, cu(o), cu(d), cu(S['posed_verts']), S['mesh'], 128, True, False, 0.2, 1.0, trace=tr)
    n, f = tr['near'][0].cpu().numpy(), tr['far'][0].cpu().numpy()
    both = (n < f) & hit
    flips = ((n < f) != hit).sum()
    dn, df = np.abs(n - S['posed_near'])[both], np.abs(f - S['posed_far'])[both]
    e2 = np.abs(rgb.cpu().numpy() - ref).max(-1)
    bad = (e2 > 1e-4) & both
    dnf = np.maximum(np.abs(n - S['posed_near']), np.abs(f - S['posed_far']))
    print(f"[posed near / far {size}] device vs the reference's torch branch on {both.sum()} hit rays: near 99 % {np.percentile(dn, 99):.1e} max {dn.max():.1e}, "
          f"far 99 % {np.percentile(df, 99):.1e} max {df.max():.1e}, hit / miss flips {flips}")
    print(f"[posed end to end {size}] rays > 1e-4: {bad.sum()} of {both.sum()} hit rays (Linf {e2[both].max():.2e}); their near / far displacement: min {dnf[bad].min() if bad.any() else 0:.1e}; "
          f"Linf over rays displaced < 2e-6: {e2[both & (dnf < 2e-6)].max() if (both & (dnf < 2e-6)).any() else 0:.2e} ({(both & (dnf < 2e-6)).sum()} rays)")
    assert np.percentile(dn, 99) < 1e-4 and np.percentile(df, 99) < 1e-4 and dn.max() < 1e-3 and df.max() < 1e-3 and flips <= 6
    from oracle import attribution
    arb, fails = attribution.against_arbiter(rgb.cpu().numpy(), PS.arbiter(S, 'posed', big=size == 'big'), tag=f"posed end to end {size}")
    SUMMARY['posed_' + size].update(e2e_gt_1e4=arb['vs_reference_f64']['rays_gt_1e-4'], e2e_reference_f32=arb['reference_f32_vs_reference_f64']['rays_gt_1e-4'],
                                    e2e_cap=arb['allowed_rays_gt_1e-4'], e2e_vs_reference_f32=int(bad.sum()))
    assert not fails, fails


def _hybrid_lists(S, which, bkg_z, near, far, S_h, multi):
    if not multi:
        hz, hit = PS.human_z(near[0], far[0], S_h)
        return [bkg_z, hz], [None, None], [hit]
    hz = [PS.human_z(near[k], far[k], S_h, placeholder_far=3.14) for k in range(3)]
    return [bkg_z] + [h[0] for h in hz], [None] + [~h[1] for h in hz], [h[1] for h in hz]


@pytest.mark.parametrize("which,size", [("hybrid", "small"), ("multi", "small"), ("hybrid", "big")])
def test_merged_frames(S, which, size):
    multi = which == 'multi'
    S, whole = pick(S, size)
    NR = S['nrays']
    c = PS.cap(S, which) if size == 'small' else PS.cap_big(S, which)
    o, d = PS.frame_rays(c)
    nets = S['dev_nets']
    ref = S[f'{which}_rgb'].reshape(-1, 3)
    S_b, S_h = (192, 192) if multi else (128, 128)
    r_near = S['multi_near'] if multi else S['hybrid_near'][None]
    r_far = S['multi_far'] if multi else S['hybrid_far'][None]
    r_z = S[f'{which}_bkg_z']

    def run(given=None, trace=None):
        if multi:
            rgb, depth = S['R'].render_multi_rays(nets[0], nets[1], [nets[2]] * 3, cu(o), cu(d), c.near['bkg'], c.far['bkg'], [cu(v) for v in S['posed_l']],
                                                  S['meshes'], S_b, 128, given=given, trace=trace)
            return rgb.cpu().numpy(), depth.cpu().numpy(), None
        rgb, depth, acc = S['R'].render_hybrid_rays(nets[0], nets[1], nets[2], cu(o), cu(d), c.near['bkg'], c.far['bkg'], cu(S['posed_verts']), S['mesh'],
                                                    S_b, 128, given=given, trace=trace)
        return rgb.cpu().numpy(), depth.cpu().numpy(), acc.cpu().numpy()

    # ---- conditional on the reference's recordings
    given = {'near_far': [(cu(n), cu(f)) for n, f in zip(r_near, r_far)], 'bkg_z': cu(r_z)}
    trc = {}
    rgb, depth, acc = run(given=given, trace=trc)
    zl, zero, hits = _hybrid_lists(S, which, r_z, r_near, r_far, S_h, multi)
    ties = PS.cross_list_ties(zl, zero)
    e = np.abs(rgb - ref).max(-1)
    a, b = whole or (MULTI_BAND if multi else BAND)
    ok = ~ties
    verts_l, T_l = (S['posed_l'], S['T_l']) if multi else ([S['posed_verts']], [S['T']])
    jumps = [foot_jump_rays(trc, k, o, d, verts_l[k], S['faces'], T_l[k], (a, b)) for k in range(len(verts_l))]
    for j in jumps:
        ok[j] = False
    band = np.zeros(NR, bool)
    band[a:b] = True
    n_hit_band = sum(int((h & band).sum()) for h in hits)
    print(f"[{which} {size}, conditional on the reference's bkg z and near / far] rays {a}..{b}, actor hits there {n_hit_band}: rgb Linf over rays without a cross-list z tie "
          f"or a foot on another face {e[band & ok].max():.2e}; tie rays (frame) {list(np.nonzero(ties)[0])} at {e[ties]}; rays with a displaced foot per actor "
          f"{[j.size for j in jumps]}; depth {np.abs(depth - S[f'{which}_depth'].ravel())[band & ok].max():.2e}; whole frame rays > 1e-4: {(e > 1e-4).sum()}")
    SUMMARY[f'{which}_{size}'] = {'rays': int(band.sum()), 'actor_hits': n_hit_band, 'cond_linf': float(e[band & ok].max()), 'tie_rays': int(ties.sum()),
                                  'foot_jump_rays': int(sum(j.size for j in jumps))}
    # measured on the 40 x 32 frames: 1-3 tie rays, displaced feet on 2.2 % of the band's hit rays
    # (the same foot flag on both frame sizes -- round 4's wider one on the big frame is gone: the whole-frame count below carries the statement)
    assert e[band & ok].max() < 1e-4 and ties.sum() <= TIE_CAP[size] and sum(j.size for j in jumps) <= 0.05 * max(1, n_hit_band)
    SUMMARY[f'{which}_{size}']['cond_gt_1e4_whole_frame'] = int((e > 1e-4).sum())
    assert (e > 1e-4).sum() <= COND_CAP[size][which]
    # ---- end to end
    tr = {}
    rgb, depth, acc = run(trace=tr)
    e2 = np.abs(rgb - ref).max(-1)
    z_dev = tr['bkg_z'][0].cpu().numpy()
    n_dev = np.stack([x.cpu().numpy() for x in tr['near']])
    f_dev = np.stack([x.cpu().numpy() for x in tr['far']])
    dz = np.abs(z_dev - r_z).max(-1)
    hit_flip = np.zeros(NR, bool)
    dnf = np.zeros(NR, np.float32)
    for k in range(len(r_near)):
        hr, hd = r_near[k] < r_far[k], n_dev[k] < f_dev[k]
        hit_flip |= hr != hd
        m = hr & hd
        dnf[m] = np.maximum(dnf[m], np.maximum(np.abs(n_dev[k] - r_near[k]), np.abs(f_dev[k] - r_far[k]))[m])
    zl_d, _, _ = _hybrid_lists(S, which, z_dev, n_dev, f_dev, S_h, multi)
    order_flip = (PS.merged_order(zl_d) != PS.merged_order(zl)).any(1)
    bad = e2 > 1e-4
    quiet = ~order_flip & ~hit_flip & (dz < 2e-6) & (dnf < 2e-6)
    print(f"[{which} end to end {size}] rays > 1e-4: {bad.sum()} of {NR} (Linf {e2.max():.2e}); of those: merged order changed {(bad & order_flip).sum()}, hit / miss flip "
          f"{(bad & hit_flip).sum()}, neither {(bad & ~order_flip & ~hit_flip).sum()} (their sample displacement >= {dz[bad & ~order_flip & ~hit_flip].min() if (bad & ~order_flip & ~hit_flip).any() else 0:.1e}); "
          f"rays with changed order {order_flip.sum()}, displacement percentiles 50/95 {np.median(dz):.1e}/{np.percentile(dz, 95):.1e}; "
          f"quiet rays (same order, nothing displaced by 2e-6): {quiet.sum()}, their Linf {e2[quiet].max() if quiet.any() else 0:.2e}")
    from oracle import attribution
    arb, fails = attribution.against_arbiter(rgb, PS.arbiter(S, which, big=size == 'big'), tag=f"{which} end to end {size}")
    SUMMARY[f'{which}_{size}'].update(e2e_gt_1e4=arb['vs_reference_f64']['rays_gt_1e-4'], e2e_reference_f32=arb['reference_f32_vs_reference_f64']['rays_gt_1e-4'],
                                      e2e_cap=arb['allowed_rays_gt_1e-4'], e2e_vs_reference_f32=int(bad.sum()),
                                      quiet_rays=int(quiet.sum()), quiet_linf=float(e2[quiet].max()) if quiet.any() else 0.0)
    assert not fails, fails
    assert (e2[quiet] < 1e-4).all()
