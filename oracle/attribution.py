"""Checker: quantitative attribution of a two-pass frame's deviation (test infrastructure only, like the rest of oracle/; callers: tests/,
bench.py's parity leg, __graft_entry__.smoke()).

A two-pass render (reference utils/render_utils.py:131-151) places its fine samples through the inverse CDF of the coarse pass's
compositing weights (utils/ray_utils.py:164-194), which turns a weight difference d into a position difference d / pdf -- in the
near-empty bins the 1e-5 floor creates, a 1e-6 difference is a visible fraction of a bin -- so float32 evaluations of the very same
algorithm disagree by more than 1e-4 on a fraction of a per cent of the rays.  How large that fraction is for THE REFERENCE ITSELF is
measured, not argued: tests/golden/arbiter.npz holds the reference's own renderers run twice on identical inputs, as shipped (float32)
and in float64 (tests/golden/make_golden_f64.py; nothing of this repository takes part in either run).  The float64 frame is the ARBITER --
the function the reference computes -- and the reference's float32 frame is the YARDSTICK: the device has to sit as close to the arbiter
as the reference's own float32 arithmetic does.  The deviation of the device's frame is split into parts that are each held to a bound:

(a)  reverse conditional  the device's shading pass on the ORACLE's sample positions vs the oracle's pixels       <= 1e-4, every ray
(a') forward conditional  the oracle's shading pass on the DEVICE's sample positions vs the device's pixels       <= 1e-4, every ray
(b)  the sampling pass    device coarse compositing weights vs the oracle's                                        <= w_tol, every ray
(c)  displacement         every ray beyond 1e-4 (of the oracle) is among the 6 % most displaced rays (max_s |z_dev - z_oracle|) and
                          the other 94 % are within 1e-4
(d)  the count            rays on which the device is beyond 1e-4 of the ARBITER (reference, float64)
                                 <=  rays on which the REFERENCE'S OWN float32 frame is beyond 1e-4 of it  +  margin(.)
                          (margin: allowed_count below -- a quarter of the yardstick's count, at least 3: two float32-class
                          evaluations of an ill-conditioned step do not fail on the same rays, only on as many).  Nothing here is read
                          from profiles/ and nothing refers to what the device measured in an earlier round.
"""
import os

import numpy as np

_ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
ARBITER = os.path.join(_ROOT, "tests", "golden", "arbiter.npz")
ARBITER_FULL = os.path.join(_ROOT, "tests", "golden", "arbiter_full.npz")
_CACHE = {}


def load_arbiter(case):
    """The reference-made float32 / float64 results of one case of tests/golden/arbiter.npz -> dict: 'rgb32', 'rgb64' [R,3] (and 'z64', 'w64',
    'z32', 'w32', 'first' where the generator kept them).  Cases: 'smoke' (32x32, 16+16), 'c1' (64x64, 32+32), 'c2' (2048 rays of the
    800x800 frame from ray 320100, 128+128), 'bench' (its first 4096 rays), 'wc_fog00' / 'wc_opaque00' (rows 5::10 of the frame: 64 000 rays;
    + 'rows'), 'posed' / 'hybrid' / 'multi' (the 40x32 frames of posed.npz in float64; their float32 frames are posed.npz's)."""
    if 'npz' not in _CACHE:
        _CACHE['npz'] = np.load(ARBITER)
    z = _CACHE['npz']
    out = {k[len(case) + 1:]: z[k] for k in z.files if k.startswith(case + "_")}
    if case.startswith('wc_'):
        out['rows'] = z['wc_rows']
    if not out:
        raise KeyError(f"no case {case!r} in {ARBITER}")
    return out


def load_arbiter_full():
    """tests/golden/arbiter_full.npz: one WHOLE 800x800 frame of the well-conditioned workload (`name`), the reference in float64 ('rgb64'
    [640000,3]) and its float32 run (rebuilt from the stored difference: 'rgb32')"""
    z = np.load(ARBITER_FULL)
    rgb64 = z['rgb64']
    return {'name': str(z['name']), 'rgb64': rgb64, 'rgb32': (rgb64.astype(np.float64) + z['diff32_x2e16'].astype(np.float64) / 65536.0).astype(np.float32)}


def allowed_count(yardstick_count):
    """statement (d)'s right-hand side: the yardstick's count + a quarter of it, at least + 3"""
    return int(yardstick_count) + max(3, int(np.ceil(0.25 * yardstick_count)))


def against_arbiter(rgb, arb, tag=""):
    """rgb [R,3] (anybody's float32 frame of the arbiter's rays) vs the reference's float64 frame, beside the reference's float32 frame vs the same
    -> (report, violated statements): statement (d)"""
    e = np.abs(rgb.astype(np.float64) - arb['rgb64']).max(-1)
    y = np.abs(arb['rgb32'].astype(np.float64) - arb['rgb64']).max(-1)
    rep = {"rays": int(e.size), "vs_reference_f64": {"rgb_linf": float(e.max()), "rays_gt_1e-4": int((e > 1e-4).sum()), "median": float(np.median(e))},
           "reference_f32_vs_reference_f64": {"rgb_linf": float(y.max()), "rays_gt_1e-4": int((y > 1e-4).sum()), "median": float(np.median(y))},
           "allowed_rays_gt_1e-4": allowed_count((y > 1e-4).sum())}
    fails = []
    if (e > 1e-4).sum() > rep["allowed_rays_gt_1e-4"]:
        fails.append(f"(d) {(e > 1e-4).sum()} rays beyond 1e-4 of the reference's float64 frame > {rep['allowed_rays_gt_1e-4']} = the reference's own float32 "
                     f"count {(y > 1e-4).sum()} + margin")
    if tag:
        print(f"[{tag}] vs the reference in float64: Linf {e.max():.2e}, rays > 1e-4: {(e > 1e-4).sum()} of {e.size} | the reference's own float32 frame: "
              f"Linf {y.max():.2e}, rays > 1e-4: {(y > 1e-4).sum()} | allowed {rep['allowed_rays_gt_1e-4']}")
    return rep, fails


def two_pass(rgb_dev, z_dev, w_dev, rgb_dev_on_oracle_z, rgb_ora, z_ora, w_ora, oracle_fine_on, arbiter=None, w_tol=2e-5,
             max_forward_rays=None, tag=""):
    """All arrays numpy, per ray: rgb [R,3], z [R,S'], w [R,S]; `oracle_fine_on(z [n,S']) -> rgb [n,3]` evaluates the oracle's
    shading network + compositing on given positions of the first n rays; `arbiter` = load_arbiter(case) of these very rays (None: statement
    (d) is not made -- no reference-made float64 frame exists for the rays).  Returns (report dict, list of violated statements)."""
    R = rgb_dev.shape[0]
    err = np.abs(rgb_dev - rgb_ora).max(-1)
    bad = err > 1e-4
    rep = {"rays": int(R), "rgb_linf": float(err.max()), "rays_gt_1e-4": int(bad.sum())}
    fails = []
    # (a)
    rev = np.abs(rgb_dev_on_oracle_z - rgb_ora).max(-1)
    rep["a_device_shading_on_oracle_samples_linf"] = float(rev.max())
    if rev.max() > 1e-4:
        fails.append(f"(a) device shading pass on the oracle's samples: {rev.max():.2e} > 1e-4")
    # (a')
    n = R if max_forward_rays is None else min(R, max_forward_rays)
    fwd = np.abs(rgb_dev[:n] - oracle_fine_on(z_dev[:n])).max(-1)
    rep["a2_oracle_shading_on_device_samples_linf"] = float(fwd.max())
    rep["a2_rays"] = int(n)
    if fwd.max() > 1e-4:
        fails.append(f"(a') oracle shading pass on the device's samples: {fwd.max():.2e} > 1e-4")
    # (b)
    dw = np.abs(w_dev - w_ora).max(-1)
    rep["b_coarse_weight_linf"] = float(dw.max())
    rep["b_tolerance"] = float(w_tol)                              # measured 5e-7 ... 6e-6 (16 samples); 2e-5 = 3.4 x the largest
    if dw.max() > w_tol:
        fails.append(f"(b) coarse weights: {dw.max():.2e} > {w_tol:.1e}")
    # (c) rank
    dz = np.abs(z_dev - z_ora).max(-1)
    cut = float(np.percentile(dz, 94.0))
    quiet = dz <= cut
    rep["c_displacement_percentiles_50_94_99"] = [float(x) for x in np.percentile(dz, [50, 94, 99])]
    rep["c_linf_over_the_94pct_least_displaced_rays"] = float(err[quiet].max())
    rep["c_bad_rays_outside_the_6pct_most_displaced"] = int((bad & quiet).sum())
    if (bad & quiet).sum():
        fails.append(f"(c) {(bad & quiet).sum()} rays beyond 1e-4 are not among the 6 % most displaced rays")
    # (d)
    if arbiter is not None:
        d_rep, d_fails = against_arbiter(rgb_dev, arbiter)
        rep["d_device_vs_reference_f64"] = d_rep["vs_reference_f64"]
        rep["d_reference_f32_vs_reference_f64"] = d_rep["reference_f32_vs_reference_f64"]
        rep["d_oracle_vs_reference_f64"] = against_arbiter(rgb_ora, arbiter)[0]["vs_reference_f64"]
        rep["d_allowed_rays_gt_1e-4"] = d_rep["allowed_rays_gt_1e-4"]
        fails += d_fails
        if 'z64' in arbiter:                                       # the intermediates beside the yardstick's, in their own units
            rep["d_displacement_vs_reference_f64_percentiles_50_99_max"] = {
                "device": [float(x) for x in np.percentile(np.abs(z_dev - arbiter['z64']).max(-1), [50, 99, 100])],
                "reference_f32": [float(x) for x in np.percentile(np.abs(arbiter['z32'] - arbiter['z64']).max(-1), [50, 99, 100])]}
            rep["d_coarse_weight_linf_vs_reference_f64"] = {"device": float(np.abs(w_dev - arbiter['w64']).max()),
                                                            "reference_f32": float(np.abs(arbiter['w32'] - arbiter['w64']).max())}
    else:
        rep["d_device_vs_reference_f64"] = None
    if tag:
        print(f"[{tag}] " + ", ".join(f"{k}={v:.2e}" if isinstance(v, float) else f"{k}={v}" for k, v in rep.items()))
    return rep, fails


def oracle_two_pass(nets, o, d, near, far, S, NI, batch=2048):
    """reference render_utils.py:131-151 on the CPU oracle, keeping what the attribution needs -> dict(rgb, z, w, fine_on)"""
    from . import compositing, nerf_mlp, ray_ops
    rgbs, zs, ws = [], [], []
    for i in range(0, o.shape[0], batch):
        oo, dd_ = o[i:i + batch], d[i:i + batch]
        R = oo.shape[0]
        pts, dd, z = ray_ops.ray_to_samples(oo, dd_, np.full((R, 1), near, np.float32), np.full((R, 1), far, np.float32), S)
        w = compositing.raw2outputs(nerf_mlp.joiner_forward(*nets[0], pts, dd), z, dd_)[3]
        pts, dd, zf = ray_ops.ray_to_importance_samples(oo, dd_, z, w, NI)
        rgbs.append(compositing.raw2outputs(nerf_mlp.joiner_forward(*nets[1], pts, dd), zf, dd_)[0])
        zs.append(zf)
        ws.append(w)

    def fine_on(z):
        n = z.shape[0]
        out = []
        for i in range(0, n, batch):
            zz = z[i:i + batch]
            pts = (o[i:i + zz.shape[0], None, :] + d[i:i + zz.shape[0], None, :] * zz[..., None]).astype(np.float32)
            raw = nerf_mlp.joiner_forward(*nets[1], pts, np.broadcast_to(d[i:i + zz.shape[0], None, :], pts.shape))
            out.append(compositing.raw2outputs(raw, zz, d[i:i + zz.shape[0]])[0])
        return np.concatenate(out)
    return {"rgb": np.concatenate(rgbs), "z": np.concatenate(zs), "w": np.concatenate(ws), "fine_on": fine_on}


def device_two_pass(render_utils, coarse, fine, o_t, d_t, near, far, S, NI, z_oracle_t, precision=None):
    """The product renderer on device tensors, twice: as shipped (its sample positions and coarse weights through `trace`), and with
    the oracle's sample positions replayed (`given`) -> rgb, z, w, rgb_on_oracle_z as numpy."""
    trace = {}
    rgb = render_utils.render_vanilla_rays(coarse, fine, o_t, d_t, near, far, S, NI, True, precision=precision, trace=trace)[0]
    rgb_on = render_utils.render_vanilla_rays(coarse, fine, o_t, d_t, near, far, S, NI, True, precision=precision, given={'bkg_z': z_oracle_t})[0]
    cat = lambda k: np.concatenate([x.cpu().numpy() for x in trace[k]])
    return rgb.cpu().numpy(), cat('bkg_z'), cat('coarse_w'), rgb_on.cpu().numpy()
