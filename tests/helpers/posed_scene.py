"""The scene of tests/golden/posed.npz (made by the reference itself, tests/golden/make_golden_posed.py) rebuilt from the same
synthetic definitions, and the bookkeeping the posed-render parity tests share (CPU: oracle vs the reference's frames; GPU: the
HIP renderers vs the same frames)."""
import os

import numpy as np

from neuman_hip import synthetic
from oracle import ray_ops as O
from oracle.nerf_mlp import JoinerSpec

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "posed.npz")
GOLDEN_BIG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "golden", "posed_big.npz")
W, H = 40, 32


def load():
    g = dict(np.load(GOLDEN))
    verts_c, faces = synthetic.capsule_mesh()
    posed, T = synthetic.twist_transforms(verts_c)
    g['posed_verts'], g['faces'], g['T'] = posed, faces, T
    g['posed_l'] = [(posed + s).astype(np.float32) for s in g['multi_shifts']]
    g['T_l'] = []
    for s in g['multi_shifts']:
        t = T.copy()
        t[:, :3, 3] += s
        g['T_l'].append(t)
    return g


def load_big():
    """tests/golden/posed_big.npz (make_golden_posed.py --big): the reference's posed and hybrid frames at 64 x 64 = 4096 rays.  The
    background's final sample positions are rebuilt from the recorded importance samples: sort(cat(stratified z, sample_pdf's z)) --
    what ray_to_importance_samples returned, bit for bit (asserted by the generator)."""
    g = dict(np.load(GOLDEN_BIG))
    verts_c, faces = synthetic.capsule_mesh()
    posed, T = synthetic.twist_transforms(verts_c)
    g['posed_verts'], g['faces'], g['T'] = posed, faces, T
    g['W'], g['H'] = int(g['big_wh'][0]), int(g['big_wh'][1])
    R = g['W'] * g['H']
    near, far = (float(x) for x in g['hybrid_near_far'])
    zero = np.zeros((R, 3), np.float32)
    z_c = O.ray_to_samples(zero, zero, np.full((R, 1), near, np.float32), np.full((R, 1), far, np.float32), 128)[2]
    g['hybrid_bkg_z'] = np.sort(np.concatenate([z_c, g['hybrid_z_samples']], -1), -1)
    return g


def _reference_rays(case):
    """the frame's rays exactly as the reference shot them (float32 arithmetic on its float32 camera centre, utils/ray_utils.py:23-29), recorded by
    tests/golden/make_golden_f64.py: the contract is parity on IDENTICAL rays, and a ray direction off by one float32 ulp moves a grazing ray's near / far
    by up to 1e-4 (the discriminant's root divides by 2 dz).  None when the fixture does not hold them."""
    from oracle import attribution
    try:
        a = attribution.load_arbiter(case)
    except KeyError:
        return None
    return (a['rays_o'].astype(np.float32), a['rays_d'].astype(np.float32)) if 'rays_o' in a else None


def cap_big(g, which='posed'):
    near, far = (float(x) for x in g['hybrid_near_far'])
    c = synthetic.SimpleCapture(g['W'], g['H'], fx=float(g['big_fx']), c2w=g['cam_c2w'], near=near, far=far)
    c._rays = _reference_rays(which + 'big')
    return c


def oracle_nets():
    return {seed: (synthetic.state_numpy(synthetic.make_joiner(seed, mp)), JoinerSpec(mapping=mp)) for seed, mp in ((0, 'posenc'), (1, 'posenc'), (2, 'rotate'))}


def cap(g, which):
    fx = float(g[f'{which}_fx'])
    near, far = (0.5, 4.0) if which == 'posed' else tuple(float(x) for x in g[f'{which}_near_far'])
    c = synthetic.SimpleCapture(W, H, fx=fx, c2w=g['cam_c2w'], near=near, far=far)
    c._rays = _reference_rays(which)
    return c


def frame_rays(c):
    """the capture's rays: the reference's own recording when the fixture holds it (cap / cap_big), else the oracle's shot_rays (1 ulp from it)"""
    if getattr(c, '_rays', None) is not None:
        return c._rays
    o, d = O.shot_rays(c.intrinsic_matrix, c.cam_pose.camera_to_world, O.all_pixel_coords(c.shape))
    return o.astype(np.float32), d.astype(np.float32)


def human_z(near, far, S, placeholder_far=None):
    """z of one actor's samples for every ray (ray_utils.py:96-135 on the hit rays; misses: NaN, or -- multi-person renderer,
    render_utils.py:418-419 -- the zero-density placeholders at linspace(2 far, 3 far))"""
    R = near.shape[0]
    hit = near < far
    z = np.full((R, S), np.nan, np.float32)
    if placeholder_far is not None:
        z[:] = O.linspace_f32(placeholder_far * 2, placeholder_far * 3, S)[None]
    if hit.any():
        o = np.zeros((hit.sum(), 3), np.float32)
        z[hit] = O.ray_to_samples(o, o, near[hit][:, None], far[hit][:, None], S)[2]
    return z, hit


def merged_order(z_lists):
    """argsort of cat(z lists) per ray, stable (the renderers' merge, render_utils.py:330-337); rows holding NaN (a miss in the
    single-actor renderer: nothing is merged) get -1"""
    z = np.concatenate(z_lists, 1)
    order = np.argsort(z, 1, kind='stable')
    order[np.isnan(z).any(1)] = -1
    return order


def cross_list_ties(z_lists, zero=None):
    """rays whose merged list holds two EQUAL z values from different lists: the reference sorts with torch.sort(stable=False)
    (render_utils.py:330, 441), so which of the two samples comes first -- and owns the zero-length interval -- is
    implementation-defined there (torch 2.10's CPU sort puts the human sample first in the case tests/golden/posed.npz holds;
    the stable merge here and on the device puts the earlier list first).  `zero[i]` [R] marks the rays for which list i is an
    actor's zero-density placeholder row (render_utils.py:418-419): placeholders of different missed actors coincide by
    construction, and the order of two zero-density samples cannot matter."""
    z = np.concatenate(z_lists, 1)
    src = np.concatenate([np.full(zl.shape[1], i) for i, zl in enumerate(z_lists)])
    order = np.argsort(z, 1, kind='stable')
    zs = np.take_along_axis(z, order, 1)
    ss = src[order]
    tie = (zs[:, 1:] == zs[:, :-1]) & (ss[:, 1:] != ss[:, :-1])
    if zero is not None:
        zmask = np.stack([np.asarray(m, bool) if m is not None else np.zeros(z.shape[0], bool) for m in zero], 1)   # [R, lists]
        isz = np.take_along_axis(zmask, ss, 1)
        tie &= ~(isz[:, 1:] & isz[:, :-1])
    return tie.any(1) & ~np.isnan(z).any(1)


def arbiter(g, which, big=False):
    """The ARBITER of an end-to-end statement on one of the posed scenes: the reference's own renderer run in float64 on the frame
    (tests/golden/arbiter.npz, tests/golden/make_golden_f64.py) beside its float32 run (posed.npz / posed_big.npz: `g`) -> the dict
    oracle.attribution.against_arbiter takes"""
    from oracle import attribution
    a = attribution.load_arbiter(which + ('big' if big else ''))
    return {'rgb64': a['rgb64'].reshape(-1, 3), 'rgb32': g[f'{which}_rgb'].reshape(-1, 3)}
