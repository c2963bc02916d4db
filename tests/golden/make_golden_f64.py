"""The ARBITER for end-to-end parity: the REFERENCE's own renderers executed twice on identical inputs -- as shipped (float32) and in
float64 -- so that the device's deviation can be scored against something neither the device nor oracle/ had a hand in: the
float64 frame is the function the reference computes, and the distance of the reference's OWN float32 frame from it is the yardstick
(how far float32 arithmetic of this algorithm sits from its exact value on these very rays).  Build container only:

    python tests/golden/make_golden_f64.py                 ->  tests/golden/arbiter.npz       (slices + strided rows + posed scenes; --only a,b: those parts)
    python tests/golden/make_golden_f64.py --full-frame    ->  tests/golden/arbiter_full.npz  (a whole 800 x 800 frame, hours of CPU)

The reference (apple/ml-neuman) is imported unmodified; absent wheels are stubbed as in make_golden.py, `igl` is
tests/golden/igl_shim.py as in make_golden_posed.py.  The float64 run changes NOTHING in the reference's code: the harness
  * sets torch's default dtype to float64 (torch.linspace / torch.ones / torch.Tensor([1e10]) inside the reference become float64),
  * casts the networks' parameters and the 'rotate' embedder's `bvals` to float64 (the same float32 VALUES, widened),
  * makes Tensor.float() a widening to float64 for the duration of the run (the renderers cast with .float() at 30 places,
    utils/render_utils.py:114-435),
  * rounds the rays the reference shoots (utils/ray_utils.py:23-38) and the capture's near / far to float32 first, so that both runs
    -- and the device -- consume IDENTICAL float32 rays, bounds, weights and vertices.

What is stored (float64 results as float32: 6e-8 of rounding against a 1e-4 gate):

  c2_*        BASELINE config 2's slice of tests/test_hip_configs.py: 2048 rays of the 800 x 800 frame, 128 + 128 samples, the
              synthetic-dense pair (seeds 0 / 1): rgb, the fine pass's sample positions, the coarse weights -- of both runs
  smoke_* c1_*   the whole 32 x 32 (16 + 16) and 64 x 64 (32 + 32) frames of the same pair: rgb of both runs
  bench_*     the first 4096 rays of the same frame (what bench.py's parity leg renders): rgb of both runs
  wc_<name>_* two-pass workloads whose coarse and fine network AGREE (as a trained pair does), 80 rows spread over the full 800 x 800
              frame (rows 5, 15, ... 795: 64 000 rays), rgb of both runs: `fog00` (synthetic.make_joiner(0, preset='fog'): density positive
              everywhere, the inverse CDF well conditioned on every ray -- the reference's float32 run sits 3e-7 from its float64 one) and
              `opaque00` (surfaces: 3 of the 64 000 rays beyond 1e-4 in the reference's own float32 run)
  posed / hybrid / multi   the 40 x 32 frames of make_golden_posed.py again in float64: rgb (+ depth), near / far, background z, and the frame's rays
                           exactly as the reference shot them (float32 arithmetic of utils/ray_utils.py:23-29 on its float32 camera centre)
  posedbig / hybridbig     the 64 x 64 frames of make_golden_posed.py --big in float64: rgb
"""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import make_golden_posed as MP  # noqa: E402  (installs the igl shim + stubs, imports the reference unmodified)

from make_golden_posed import R_ray, R_render, R_vanilla, synthetic, PinholeCamera, CameraPose, BasePinholeCapture  # noqa: E402

W = H = 800
ROWS = np.arange(5, 800, 10)                       # the strided rows of the well-conditioned full-frame workloads
# name -> (coarse (seed, preset), fine (seed, preset)); preset None = synthetic-dense
WELL_CONDITIONED = {"fog00": ((0, 'fog'), (0, 'fog')), "opaque00": ((0, 'opaque'), (0, 'opaque'))}


RAYS = []                                         # (origins, directions) of every shot_rays call under float64_mode, as float32


def f32_values(x):
    return np.asarray(x).astype(np.float32).astype(np.float64)


@contextlib.contextmanager
def float64_mode():
    """the harness of the module docstring; restores everything on exit"""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    t_float = torch.Tensor.float
    torch.Tensor.float = lambda self, *a, **k: self.double()
    shot_all, shot = R_ray.shot_all_rays, R_ray.shot_rays

    def shot_all_rounded(cap):
        o, d = shot_all(cap)
        return f32_values(o), f32_values(d)

    def shot_rounded(cap, xys):
        o, d = shot(cap, xys)
        RAYS.append((np.asarray(o, np.float32).copy(), np.asarray(d, np.float32).copy()))      # the very rays of the frame (posed_part stores them)
        return f32_values(o), f32_values(d)
    R_ray.shot_all_rays, R_ray.shot_rays = shot_all_rounded, shot_rounded
    try:
        yield
    finally:
        torch.set_default_dtype(old)
        torch.Tensor.float = t_float
        R_ray.shot_all_rays, R_ray.shot_rays = shot_all, shot


def ref_net(seed, mapping='posenc', preset=None, f64=False):
    ours = synthetic.make_joiner(seed, mapping, preset=preset)
    net, _ = R_vanilla.build_nerf(synthetic.default_opt(posenc=mapping))
    net.load_state_dict(ours.state_dict(), strict=True)
    if mapping == 'rotate':
        net.pos_pe.bvals, net.dir_pe.bvals = net.pos_pe.bvals.cpu(), net.dir_pe.bvals.cpu()
    net = net.eval()
    if f64:
        net = net.double()
        if mapping == 'rotate':
            net.pos_pe.bvals, net.dir_pe.bvals = net.pos_pe.bvals.double(), net.dir_pe.bvals.double()
    return net


class Recorder:
    """keeps what ray_to_importance_samples was given (the coarse weights) and returned (the fine pass's sample positions)"""

    def __init__(self):
        self.z, self.w = [], []
        self.orig = R_ray.ray_to_importance_samples

    def __enter__(self):
        def rec(ray_batch, z_vals, weights, *a, **k):
            r = self.orig(ray_batch, z_vals, weights, *a, **k)
            self.w.append(weights.detach().cpu().numpy().copy())
            self.z.append(r[2].detach().cpu().numpy().copy())
            return r
        R_ray.ray_to_importance_samples = rec
        return self

    def __exit__(self, *exc):
        R_ray.ray_to_importance_samples = self.orig


def rows_capture(rows, W=W, H=H):
    """a capture whose shot_all_rays yields exactly the rays of the given rows of the W x H synthetic camera: the reference's own
    shot_all_rays on the full camera, rows selected (render_vanilla only needs cap.shape to reshape its result)"""
    full = BasePinholeCapture(PinholeCamera(W, H, 1.25 * W, 1.25 * W, W / 2, H / 2), CameraPose.from_camera_to_world(np.eye(4)))
    sub = BasePinholeCapture(PinholeCamera(W, len(rows), 1.25 * W, 1.25 * W, W / 2, H / 2), CameraPose.from_camera_to_world(np.eye(4)))
    sub.near, sub.far = {'bkg': 0.0}, {'bkg': float(np.float32(3.14))}
    sub._full, sub._rows = full, np.asarray(rows)
    return sub


@contextlib.contextmanager
def row_selection():
    inner = R_ray.shot_all_rays

    def shot_all_rows(cap):
        if not hasattr(cap, '_rows'):
            return inner(cap)
        o, d = inner(cap._full)
        pick = lambda x: x.reshape(*cap._full.shape, 3)[cap._rows].reshape(-1, 3)
        return pick(o), pick(d)
    R_ray.shot_all_rays = shot_all_rows
    try:
        yield
    finally:
        R_ray.shot_all_rays = inner


def render_rows(pair, rows, S=128, NI=128, record=False, W=W, H=H):
    """render_vanilla of the reference on whole rows, float32 then float64 -> dict of flat per-ray arrays"""
    out = {}
    for tag, f64 in (("32", False), ("64", True)):
        nets = [ref_net(s, preset=p, f64=f64) for s, p in pair]
        cap = rows_capture(rows, W, H)
        t0 = time.time()
        with contextlib.ExitStack() as st:
            if f64:
                st.enter_context(float64_mode())
            st.enter_context(row_selection())                      # entered last: it wraps the rounding wrapper of float64_mode
            rec = st.enter_context(Recorder())
            st.enter_context(contextlib.redirect_stdout(io.StringIO()))
            rgb = R_render.render_vanilla(nets[0], cap, nets[1], rays_per_batch=2048, samples_per_ray=S, importance_samples_per_ray=NI)
        assert rgb.dtype == (np.float64 if f64 else np.float32), rgb.dtype
        out["rgb" + tag] = rgb.reshape(-1, 3)
        if record:
            out["z" + tag], out["w" + tag] = np.concatenate(rec.z), np.concatenate(rec.w)
        print(f"   float{tag}: {rgb.shape[0] * rgb.shape[1]} rays in {time.time() - t0:.0f} s", flush=True)
    return out


def slice_of_rows(first, n):
    r0, r1 = first // W, -(-(first + n) // W)
    return np.arange(r0, r1), first - r0 * W


def stats(a, b):
    e = np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max(-1)
    return f"Linf {e.max():.2e}, rays > 1e-4: {(e > 1e-4).sum()} of {e.size}"


def vanilla_part(out, only=None):
    dense01 = ((0, None), (1, None))
    for name, first, n, keep32 in (("c2", 400 * 800 + 100, 2048, True), ("bench", 0, 4096, False)):
        if only and name not in only:
            continue
        rows, a = slice_of_rows(first, n)
        print(f"[{name}] rays {first}..{first + n} of the 800 x 800 frame, 128 + 128, synthetic-dense seeds 0 / 1", flush=True)
        r = render_rows(dense01, rows, record=True)
        cut = lambda x: x[a:a + n]
        out[f"{name}_first"] = np.array(first)
        out[f"{name}_rgb32"], out[f"{name}_rgb64"] = cut(r["rgb32"]), cut(r["rgb64"]).astype(np.float32)
        if keep32:
            out[f"{name}_z64"], out[f"{name}_w64"] = cut(r["z64"]).astype(np.float32), cut(r["w64"]).astype(np.float32)
            out[f"{name}_z32"], out[f"{name}_w32"] = cut(r["z32"]), cut(r["w32"])
        print(f"   reference float32 vs reference float64: {stats(cut(r['rgb32']), cut(r['rgb64']))}", flush=True)
    for name, w_, S_ in (("smoke", 32, 16), ("c1", 64, 32)):      # whole small frames: __graft_entry__.smoke()'s and BASELINE config 1's (tests/test_hip_render.py)
        if only and name not in only:
            continue
        print(f"[{name}] {w_} x {w_} frame, {S_} + {S_}, synthetic-dense seeds 0 / 1", flush=True)
        r = render_rows(dense01, np.arange(w_), S=S_, NI=S_, W=w_, H=w_)
        out[f"{name}_rgb32"], out[f"{name}_rgb64"] = r["rgb32"], r["rgb64"].astype(np.float32)
        print(f"   reference float32 vs reference float64: {stats(r['rgb32'], r['rgb64'])}", flush=True)
    out["wc_rows"] = ROWS
    for name, pair in WELL_CONDITIONED.items():
        if only and f"wc_{name}" not in only:
            continue
        print(f"[wc_{name}] rows 5::10 of the 800 x 800 frame ({ROWS.size * W} rays), 128 + 128", flush=True)
        r = render_rows(pair, ROWS)
        out[f"wc_{name}_rgb32"], out[f"wc_{name}_rgb64"] = r["rgb32"], r["rgb64"].astype(np.float32)
        print(f"   reference float32 vs reference float64: {stats(r['rgb32'], r['rgb64'])}", flush=True)


def posed_part(out):
    """make_golden_posed.py's three frames (same body, cameras, sample counts) in float64"""
    posed, faces, T = MP.body()
    net = MP.Scene(ref_net(0, f64=True), ref_net(1, f64=True), ref_net(2, 'rotate', f64=True))
    near_far = []
    orig_nf = R_ray.geometry_guided_near_far

    def rec_nf(*a, **k):
        n, f = orig_nf(*a, **k)
        near_far.append((np.array(n, dtype=np.float64), np.array(f, dtype=np.float64)))
        return n, f

    def nf(n_actors=1):
        return (np.stack([np.concatenate([c[j] for c in near_far[a::n_actors]]) for a in range(n_actors)]).astype(np.float32) for j in (0, 1))
    g32 = dict(np.load(os.path.join(HERE, 'posed.npz')))
    R_ray.geometry_guided_near_far = rec_nf
    try:
        with float64_mode(), Recorder() as rec:
            cap = MP.ref_cap(100.0, 0.5, 4.0)
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                rgb, depth, acc = R_render.render_smpl_nerf(net, cap, posed, faces, T, rays_per_batch=512, samples_per_ray=128, white_bkg=True,
                                                            render_can=False, geo_threshold=0.2, return_depth=True, return_mask=True)
            n_, f_ = nf()
            out.update(posed_rays_o=RAYS[-1][0], posed_rays_d=RAYS[-1][1])
            out.update(posed_rgb64=rgb.astype(np.float32), posed_depth64=depth.astype(np.float32), posed_acc64=acc.astype(np.float32), posed_near64=n_[0], posed_far64=f_[0])
            print(f"[posed] float64 {time.time() - t0:.0f} s; reference float32 vs float64: {stats(g32['posed_rgb'].reshape(-1, 3), rgb.reshape(-1, 3))}", flush=True)
            near_far.clear()
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                rgb, depth = R_render.render_hybrid_nerf(net, cap, posed, faces, T, rays_per_batch=512, samples_per_ray=128, importance_samples_per_ray=128,
                                                         white_bkg=True, geo_threshold=0.2, return_depth=True)
            n_, f_ = nf()
            out.update(hybrid_rays_o=RAYS[-1][0], hybrid_rays_d=RAYS[-1][1])
            out.update(hybrid_rgb64=rgb.astype(np.float32), hybrid_depth64=depth.astype(np.float32), hybrid_bkg_z64=np.concatenate(rec.z).astype(np.float32),
                       hybrid_near64=n_[0], hybrid_far64=f_[0])
            print(f"[hybrid] float64 {time.time() - t0:.0f} s; reference float32 vs float64: {stats(g32['hybrid_rgb'].reshape(-1, 3), rgb.reshape(-1, 3))}", flush=True)
            near_far.clear()
            rec.z.clear()
            cap5 = MP.ref_cap(70.0, 0.5, float(np.float32(3.14)))
            posed_l = [(posed + s).astype(np.float32) for s in MP.SHIFTS]
            T_l = []
            for s in MP.SHIFTS:
                t = T.copy()
                t[:, :3, 3] += s
                T_l.append(t)
            t0 = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                rgb, depth = R_render.render_hybrid_nerf_multi_persons(net, cap5, [net] * 3, posed_l, [faces] * 3, T_l, rays_per_batch=512, samples_per_ray=192,
                                                                       importance_samples_per_ray=128, white_bkg=True, geo_threshold=0.2, return_depth=True)
            n_, f_ = nf(3)
            out.update(multi_rays_o=RAYS[-1][0], multi_rays_d=RAYS[-1][1])
            out.update(multi_rgb64=rgb.astype(np.float32), multi_depth64=depth.astype(np.float32), multi_bkg_z64=np.concatenate(rec.z).astype(np.float32),
                       multi_near64=n_, multi_far64=f_)
            print(f"[multi] float64 {time.time() - t0:.0f} s; reference float32 vs float64: {stats(g32['multi_rgb'].reshape(-1, 3), rgb.reshape(-1, 3))}", flush=True)
    finally:
        R_ray.geometry_guided_near_far = orig_nf


def posed_big_part(out):
    """make_golden_posed.py --big's two 64 x 64 frames (posed_big.npz) in float64"""
    posed, faces, T = MP.body()
    net = MP.Scene(ref_net(0, f64=True), ref_net(1, f64=True), ref_net(2, 'rotate', f64=True))
    g32 = dict(np.load(os.path.join(HERE, 'posed_big.npz')))
    WB, HB, FX = int(g32['big_wh'][0]), int(g32['big_wh'][1]), float(g32['big_fx'])
    with float64_mode():
        cap = MP.ref_cap(FX, 0.5, 4.0, WB, HB)
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            rgb = R_render.render_smpl_nerf(net, cap, posed, faces, T, rays_per_batch=1024, samples_per_ray=128, white_bkg=True, render_can=False, geo_threshold=0.2)
        out['posedbig_rgb64'] = rgb.astype(np.float32)
        out['posedbig_rays_o'], out['posedbig_rays_d'] = RAYS[-1]
        print(f"[posedbig] float64 {time.time() - t0:.0f} s; reference float32 vs float64: {stats(g32['posed_rgb'].reshape(-1, 3), rgb.reshape(-1, 3))}", flush=True)
        t0 = time.time()
        with contextlib.redirect_stdout(io.StringIO()):
            rgb = R_render.render_hybrid_nerf(net, cap, posed, faces, T, rays_per_batch=1024, samples_per_ray=128, importance_samples_per_ray=128, white_bkg=True,
                                              geo_threshold=0.2)
        out['hybridbig_rgb64'] = rgb.astype(np.float32)
        out['hybridbig_rays_o'], out['hybridbig_rays_d'] = RAYS[-1]
        print(f"[hybridbig] float64 {time.time() - t0:.0f} s; reference float32 vs float64: {stats(g32['hybrid_rgb'].reshape(-1, 3), rgb.reshape(-1, 3))}", flush=True)


def full_frame(name="fog00", threads=None):
    """one whole 800 x 800 frame of a well-conditioned workload, float32 and float64 (hours of CPU): every ray of the frame"""
    rows = np.arange(H)
    r = render_rows(WELL_CONDITIONED[name], rows)
    e = (r["rgb32"].astype(np.float64) - r["rgb64"])
    print(f"[full frame {name}] reference float32 vs float64: {stats(r['rgb32'], r['rgb64'])}", flush=True)
    # the float32 run is stored as its difference from the float64 one, scaled into float16's range (3 significant digits of the difference)
    np.savez_compressed(os.path.join(HERE, 'arbiter_full.npz'), name=np.array(name), rgb64=r["rgb64"].astype(np.float32),
                        diff32_x2e16=np.clip(e * 65536.0, -60000, 60000).astype(np.float16))
    print('arbiter_full.npz', os.path.getsize(os.path.join(HERE, 'arbiter_full.npz')) // 1024, 'KiB')


def main():
    torch.set_num_threads(int(os.environ.get("ARBITER_THREADS", os.cpu_count() or 1)))
    if '--full-frame' in sys.argv:
        return full_frame()
    out, only = {}, None
    if '--only' in sys.argv:                                   # e.g. --only wc_fog00,posed: recompute those parts, keep the rest of the existing file
        only = set(sys.argv[sys.argv.index('--only') + 1].split(','))
        out = dict(np.load(os.path.join(HERE, 'arbiter.npz')))
        known = {"c2", "bench", "smoke", "c1", "posed", "hybrid", "multi", "posedbig", "hybridbig", "cam"} | {f"wc_{k}" for k in WELL_CONDITIONED}
        out = {k: v for k, v in out.items() if any(k == p or k.startswith(p + "_") for p in known)}      # (drops parts this script no longer makes)
    vanilla_part(out, only)
    if not only or 'posed' in only:
        posed_part(out)
    if not only or 'posedbig' in only:
        posed_big_part(out)
    np.savez_compressed(os.path.join(HERE, 'arbiter.npz'), **out)
    print('arbiter.npz', os.path.getsize(os.path.join(HERE, 'arbiter.npz')) // 1024, 'KiB')


if __name__ == '__main__':
    main()
