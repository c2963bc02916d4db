"""HIP-backed mirror of the reference's utils/render_utils.py:69-461.

Same public signatures and return conventions (numpy float32 frames) as the reference's raw2outputs and
its four frame renderers.  What differs is the execution plan:

* the reference loops over `rays_per_batch` chunks with 6-8 host<->device copies each; here a frame is
  rendered in launches of up to `MAX_RAYS_PER_LAUNCH` rays (rays are independent, so chunking never changes a
  pixel) and nothing returns to the host before the frame is assembled;
* pts/dirs are never materialised for camera-ray passes: the MLP kernel builds `o + d*z` itself;
* boolean-mask indexing is a ballot/prefix-sum compaction + row gather/scatter on the device;
* rays are generated on the device too (the reference's f64 chain per ray, `nm_shot_rays`): 25 doubles cross the
  boundary per frame instead of 15 MB of rays.

The `*_rays` functions work on device tensors (what bench.py and the multi-GPU path call); the reference
named functions wrap them with ray generation and the final download.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib, parallel, ray_utils
from .ray_utils import DEFAULT_GEO_THRESH

MAX_RAYS_PER_LAUNCH = int(os.environ.get("NEUMAN_MAX_RAYS_PER_LAUNCH", 1 << 20))
# Early ray termination in the shading pass of the background renderers (march_pass_rays): rays whose transmittance has fallen
# below this are not evaluated any further.  0 = off: every sample is evaluated, as the reference does (render_utils.py:139-151),
# and the frame is bit-identical to the unchunked path.  eps > 0 changes a pixel by at most eps per channel.
TERMINATION_EPS = float(os.environ.get("NEUMAN_TERMINATION_EPS", "0"))
TERMINATION_CHUNK = int(os.environ.get("NEUMAN_TERMINATION_CHUNK", "32"))
# With termination on, the COARSE pass of a two-pass render is marched too, at this transmittance whatever eps is: the weights it then
# never computes are all below it, and 4e-13 is less than half a float32 ulp of the 1e-5 that sample_pdf adds to every weight
# (ray_utils.py:168: spacing of float32 at 1e-5 = 9.1e-13) -- `weights + 1e-5` is the same float32 number with or without them, so
# the importance samples are BIT-IDENTICAL to the full evaluation's.  (A cut at eps * 1e-3 = 1e-7 was tried first: it perturbs the
# CDF at float32-rounding level, which the inverse CDF amplifies in near-empty bins: 1.6e-4 on the worst pixel of a frame.)
TERMINATION_COARSE = float(os.environ.get("NEUMAN_TERMINATION_COARSE", "4e-13"))
TERMINATION_MIN_CHUNK = 16
FUSED_HYBRID_RAYS = int(os.environ.get("NEUMAN_FUSED_HYBRID_RAYS", 1 << 17))


# ------------------------------------------------------------------------------------------------
# a5 raw2outputs
# ------------------------------------------------------------------------------------------------
def raw2outputs(raw, z_vals, rays_d, raw_noise_std=0, white_bkg=True, want_weights=True):
    """reference render_utils.py:69-105 -> (rgb_map, disp_map, acc_map, weights, depth_map)."""
    _lib.require_gpu()
    R, S = z_vals.shape
    dev = raw.device
    noise = None
    if raw_noise_std > 0.:
        noise = (torch.randn((R, S), device=dev) * raw_noise_std).contiguous()          # render_utils.py:93
    if torch.is_grad_enabled() and raw.requires_grad:                                   # a training step: autograd through `raw`
        from . import train
        return train.composite_train(raw, z_vals, rays_d, white_bkg, noise)
    raw = raw.to(torch.float32).contiguous()
    z_vals = z_vals.to(torch.float32).contiguous()
    rays_d = rays_d.to(torch.float32).contiguous()
    rgb = torch.empty((R, 3), device=dev, dtype=torch.float32)
    disp = torch.empty(R, device=dev, dtype=torch.float32)
    acc = torch.empty(R, device=dev, dtype=torch.float32)
    depth = torch.empty(R, device=dev, dtype=torch.float32)
    weights = torch.empty((R, S), device=dev, dtype=torch.float32) if want_weights else None
    _lib.check(_lib.lib().nm_composite(_lib.dev_ptr(raw, name='raw'), _lib.dev_ptr(z_vals, name='z_vals'),
                                       _lib.dev_ptr(rays_d, name='rays_d'), R, S, int(bool(white_bkg)), _lib.dev_ptr(noise),
                                       _lib.dev_ptr(rgb), _lib.dev_ptr(disp), _lib.dev_ptr(acc), _lib.dev_ptr(weights),
                                       _lib.dev_ptr(depth), _lib.stream_ptr()), "nm_composite")
    return rgb, disp, acc, weights, depth


def merge_sorted(za, rawa, zb, rawb):
    """sort(cat([za, zb])) + gather of cat([rawa, rawb]) (render_utils.py:330-337); both lists sorted per ray."""
    _lib.require_gpu()
    R, Sa = za.shape
    Sb = zb.shape[1]
    z = torch.empty((R, Sa + Sb), device=za.device, dtype=torch.float32)
    raw = torch.empty((R, Sa + Sb, 4), device=za.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_merge_sorted(_lib.dev_ptr(za.contiguous()), _lib.dev_ptr(rawa.contiguous()), Sa,
                                          _lib.dev_ptr(zb.contiguous()), _lib.dev_ptr(rawb.contiguous()), Sb, R,
                                          _lib.dev_ptr(z), _lib.dev_ptr(raw), _lib.stream_ptr()), "nm_merge_sorted")
    return z, raw


# ------------------------------------------------------------------------------------------------
# device-level ray renderers
# ------------------------------------------------------------------------------------------------
def _chunks(n):
    for i in range(0, n, MAX_RAYS_PER_LAUNCH):
        yield i, min(i + MAX_RAYS_PER_LAUNCH, n)


def _note(trace, **kw):
    """`trace` (a dict, or None) collects a renderer's intermediates -- sample positions, hit lists, warped points -- so that
    tests can hand exactly those to the CPU oracle (conditional parity) while running the product code path itself."""
    if trace is not None:
        for k, v in kw.items():
            trace.setdefault(k, []).append(v)


def _transmittance_chunk(raw, z, dz, d, live, counts, R, s0, c, S, T):
    """T[live] *= prod over samples s0 .. s0+c-1 of raw2outputs' factors; intervals from z, or the given ones (merged lists)"""
    if dz is None:
        _lib.check(_lib.lib().nm_transmittance_chunk(_lib.dev_ptr(raw), _lib.dev_ptr(z), _lib.dev_ptr(d), _lib.dev_ptr(live, torch.int32),
                                                     _lib.dev_ptr(counts, torch.int32), R, s0, c, S, _lib.dev_ptr(T), _lib.stream_ptr()),
                   "nm_transmittance_chunk")
    else:
        _lib.check(_lib.lib().nm_transmittance_chunk_dz(_lib.dev_ptr(raw), _lib.dev_ptr(dz), _lib.dev_ptr(d), _lib.dev_ptr(live, torch.int32),
                                                        _lib.dev_ptr(counts, torch.int32), R, s0, c, S, _lib.dev_ptr(T), _lib.stream_ptr()),
                   "nm_transmittance_chunk_dz")


def merged_intervals(z_lists):
    """For every sample of every list ([R, S_k] each, sorted per ray; k <= 4): the distance to its successor in the MERGED order of all the
    lists (stable, earlier list first -- the order of the reference's sort(cat(...)), render_utils.py:330-337, 441-448); the last
    sample of the merged list gets raw2outputs' 1e10 (render_utils.py:86).  -> one [R, S_k] tensor per list.  These are the
    intervals the samples will be composited with once the lists are merged: what an early-termination cut has to be decided on.
    One kernel (nm_merged_intervals: a binary search per foreign list, nothing sorted)."""
    _lib.require_gpu()
    k, R = len(z_lists), z_lists[0].shape[0]
    zs = [z.to(torch.float32).contiguous() for z in z_lists]
    dz = [torch.empty_like(z) for z in zs]
    arr = ctypes.c_void_p * k
    _lib.check(_lib.lib().nm_merged_intervals(k, arr(*[z.data_ptr() for z in zs]), (ctypes.c_int * k)(*[int(z.shape[1]) for z in zs]), R,
                                              arr(*[x.data_ptr() for x in dz]), _lib.stream_ptr()), "nm_merged_intervals")
    return dz


def march_pass_rays(net, o, d, z, eps, chunk=None, precision=None, role='shading', stats=None, sigma_only=False, occluder=None,
                    adaptive=None, dz=None):
    """A pass with early ray termination: the S sorted samples of every ray are evaluated front to back in chunks of `chunk`; after
    each chunk the rays whose transmittance (the running product of raw2outputs' factors, nm_transmittance_chunk) is below `eps`
    leave the list (ballot / prefix-sum compaction on the device, nm_compact_hits) and the next MLP launch covers the compacted live
    rays only (nm_mlp_forward_ray_chunk; `sigma_only`: nm_mlp_sigma_ray_chunk, a coarse pass).  Samples never evaluated keep
    sigma = 0, i.e. weight exactly 0 in raw2outputs; what they would have contributed is bounded by the transmittance at the cut:
    < eps per channel.

    occluder = (z_far [R], T_occ [R]): the ray's list will be merged with another whose samples all lie in front of z_far and whose
    total transmittance is T_occ (the hybrid renderers: the human samples).  Once the march is past z_far the transmittance of the
    MERGED list is T * T_occ, and that is what is compared with eps.  `dz` [R,S]: the samples' intervals in the merged list
    (merged_intervals) -- merging shortens the interval behind every sample that gets a foreign successor, so a transmittance
    taken over the list's OWN intervals is not an upper bound of the merged one; with `dz` (here and in T_occ) T * T_occ is the
    merged list's transmittance itself (before z_far: an upper bound of it, T_occ counted as 1).

    adaptive (default: eps > 0): ONE host read per chunk -- the live count.  No launch once nobody is live; the chunk is halved
    (not below TERMINATION_MIN_CHUNK) while more than 2 % of the live rays were cut by the last one -- rays are only ever cut at a chunk
    boundary, so near the cut a chunk is half its length of wasted evaluations per ray -- and doubled back otherwise.
    adaptive=False: fixed chunks, no host synchronisation between them (and with eps = 0, bit-identical to the unchunked launch).

    -> raw [R,S,4]; `stats` (a dict) receives the evaluation counts."""
    _lib.require_gpu()
    chunk = chunk or TERMINATION_CHUNK
    adaptive = (eps > 0) if adaptive is None else bool(adaptive)
    R, S = z.shape
    dev = z.device
    raw = torch.zeros((R, S, 4), device=dev, dtype=torch.float32)
    T = torch.ones(R, device=dev, dtype=torch.float32)
    thr = torch.full((R,), float(eps), device=dev, dtype=torch.float32)
    live = torch.arange(R, device=dev, dtype=torch.int32)
    counts = torch.tensor([R, 0], device=dev, dtype=torch.int32)
    ws = torch.empty(int(_lib.lib().nm_compact_workspace_ints(R)), device=dev, dtype=torch.int32)
    evaluated = torch.zeros(1, device=dev, dtype=torch.int64) if stats is not None else None
    o, d, z = o.contiguous(), d.contiguous(), z.contiguous()
    s0, n_live, launches = 0, R, 0
    while s0 < S:
        c = min(chunk, S - s0)
        if evaluated is not None:
            evaluated += counts[0].to(torch.int64) * c
        net.forward_ray_chunk(o, d, z, live, counts, s0, c, raw, precision=precision, role=role, sigma_only=sigma_only)
        launches += 1
        s0 += c
        if s0 >= S or eps <= 0:                                   # (eps = 0: nothing is ever dropped, not even rays whose T underflowed to 0)
            continue
        _transmittance_chunk(raw, z, dz, d, live, counts, R, s0 - c, c, S, T)
        T_eff = T if occluder is None else T * torch.where(z[:, s0] >= occluder[0], occluder[1], torch.ones_like(T))
        nxt = torch.empty(R, device=dev, dtype=torch.int32)
        counts = torch.zeros(2, device=dev, dtype=torch.int32)
        _lib.check(_lib.lib().nm_compact_hits(_lib.dev_ptr(thr), _lib.dev_ptr(T_eff.contiguous()), R, _lib.dev_ptr(nxt, torch.int32), None,
                                              _lib.dev_ptr(counts, torch.int32), _lib.dev_ptr(ws, torch.int32), _lib.stream_ptr()), "nm_compact_hits")
        live = nxt
        if adaptive:
            n_new = int(counts[0].item())
            if n_new == 0:
                break
            cut = 1.0 - n_new / max(1, n_live)
            chunk = max(TERMINATION_MIN_CHUNK, chunk // 2) if cut > 0.02 else min(max(chunk, TERMINATION_CHUNK), chunk * 2)
            n_live = n_new
    if stats is not None:
        n = int(evaluated.item())
        stats['evaluated'] = stats.get('evaluated', 0) + n
        stats['total'] = stats.get('total', 0) + R * S
        stats['launches'] = stats.get('launches', 0) + launches
    return raw


def transmittance_of(raw, z, d, dz=None):
    """prod_i (1 - alpha_i + 1e-10) over a whole list: raw2outputs' factors (render_utils.py:85-95) -> T [R].  With the list's own
    intervals (the last one 1e10), or -- `dz`, merged_intervals -- with the intervals it has once merged with other lists."""
    R, S = z.shape
    T = torch.ones(R, device=z.device, dtype=torch.float32)
    _transmittance_chunk(raw.contiguous(), z.contiguous(), None if dz is None else dz.contiguous(), d.contiguous(), None, None, R, 0, S, S, T)
    return T


def human_march_rays(human_net, o, d, near, far, samples_per_ray, mesh, eps, sigma_scale=1.0, precision=None, chunk=None, trace=None,
                     stats=None, dz=None):
    """human_pass_rays (posed) with early ray termination -> (raw [R,S,4], z [R,S]).  The warp is the expensive step here, and a ray that
    has entered an opaque body needs neither the closest-point queries nor the network behind the surface.  Front to back in chunks; the
    live rays' samples of a chunk are gathered, warped (one sample past the chunk: the canonical direction of a sample is the forward
    difference to the next warped point, ray_utils.py:62-64 -- the last sample of a ray repeats its predecessor's), evaluated and
    scattered back; rays whose transmittance over their own (human) samples is below eps are dropped (chunks of `chunk` samples, doubled
    after every chunk that cut fewer than 2 % of the live rays, halved otherwise).  `dz` [R,S] (the hybrid renderers): the samples'
    intervals in the list they will be MERGED into (merged_intervals) -- merging puts foreign samples inside the human intervals, which
    shortens them and RAISES the transmittance, so the cut is decided on the product of the human factors over the merged intervals:
    the merged list's transmittance can only be lower than that (the foreign samples' own factors are <= 1 + 1e-10), and what is skipped
    weighs < eps in the composite.  Without `dz` (render_smpl_nerf: the list is composited alone) the list's own intervals.  Every
    evaluated sample is bit-identical to human_pass_rays' (per-sample arithmetic; tests/test_hip_march.py)."""
    _lib.require_gpu()
    R, S = o.shape[0], int(samples_per_ray)
    dev = o.device
    chunk = chunk or TERMINATION_MIN_CHUNK
    pts, _, z = ray_utils.sample_z(o.contiguous(), d.contiguous(), near.reshape(-1).contiguous(), far.reshape(-1).contiguous(), S, want_points=True)
    raw = torch.zeros((R, S, 4), device=dev, dtype=torch.float32)
    T = torch.ones(R, device=dev, dtype=torch.float32)
    d = d.contiguous()
    live = torch.arange(R, device=dev)
    s0, evaluated, launches = 0, 0, 0
    while s0 < S and live.numel() > 0:
        c = min(chunk, S - s0)
        a = s0 - 1 if (s0 + c == S and c == 1) else s0                               # a lone last sample takes its predecessor along
        b = min(S, s0 + c + 1)
        can_pts, can_dirs, _ = ray_utils.warp_to_canonical_dev(pts[live, a:b].contiguous(), mesh)
        k = s0 - a
        out = human_net(can_pts[:, k:k + c].contiguous(), can_dirs[:, k:k + c].contiguous(), precision=precision, sigma_scale=sigma_scale,
                        role='shading')
        raw[live, s0:s0 + c] = out
        evaluated += live.numel() * c
        launches += 1
        s0 += c
        if s0 >= S or eps <= 0:
            continue
        idx = live.to(torch.int32)
        cnt = torch.tensor([idx.numel(), 0], device=dev, dtype=torch.int32)
        _transmittance_chunk(raw, z, dz, d, idx, cnt, R, s0 - c, c, S, T)
        n_live = live.numel()
        live = live[T[live] >= eps]
        # (march_pass_rays' rule: short chunks while rays are being cut, doubling when nothing happens)
        chunk = max(TERMINATION_MIN_CHUNK, chunk // 2) if live.numel() < 0.98 * n_live else chunk * 2
    if stats is not None:
        stats['human_evaluated'] = stats.get('human_evaluated', 0) + evaluated
        stats['human_total'] = stats.get('human_total', 0) + R * S
        stats['human_launches'] = stats.get('human_launches', 0) + launches
    _note(trace, human_z=z)
    return raw, z


def _given_near_far(given, actor, i, j, o, d, verts, geo_threshold):
    """near / far of one actor for rays [i, j) of the call: computed (nm_near_far), or replayed from `given` (see bkg_pass_rays)"""
    if given is not None and 'near_far' in given:
        n, f = given['near_far'][actor]
        return n[i:j].to(torch.float32).contiguous(), f[i:j].to(torch.float32).contiguous()
    return ray_utils.geometry_guided_near_far(o, d, verts, geo_threshold)


def _ws(n_floats, device):
    return torch.empty(max(int(n_floats), 4), device=device, dtype=torch.float32)


def bkg_pass_rays_fused(coarse_net, fine_net, o, d, near, far, samples_per_ray, importance_samples_per_ray, white_bkg, precision=None):
    """bkg_pass_rays as ONE C call (nm_render_rays_bkg: sample -> coarse density pass -> compositing weights -> importance samples ->
    fine pass, enqueued back to back; same kernels, same bits).  What the trainers' frozen background evaluation uses: at 2048 rays
    the five launches are shorter than the Python between them."""
    _lib.require_gpu()
    R, S = o.shape[0], int(samples_per_ray)
    N = int(importance_samples_per_ray) if fine_net is not None else 0
    dev = o.device
    for n_ in (coarse_net, fine_net):
        if n_ is not None:
            n_._guard(o, d, near)
    raw = torch.empty((R, S + N, 4), device=dev, dtype=torch.float32)
    z = torch.empty((R, S + N), device=dev, dtype=torch.float32)
    ws = _ws(_lib.lib().nm_render_rays_bkg_workspace_floats(R, S, N), dev)
    t_vals = torch.linspace(0., 1., steps=S, device=dev)
    u = torch.linspace(0., 1., steps=N, device=dev) if N else None
    _lib.check(_lib.lib().nm_render_rays_bkg(
        coarse_net.handle(), fine_net.handle() if fine_net is not None else None, _lib.dev_ptr(o.contiguous()), _lib.dev_ptr(d.contiguous()),
        _lib.dev_ptr(near.reshape(-1).contiguous()), _lib.dev_ptr(far.reshape(-1).contiguous()), R, S, N, _lib.dev_ptr(t_vals), _lib.dev_ptr(u),
        int(bool(white_bkg)), coarse_net._prec(precision, None if fine_net is not None else 'shading'),
        fine_net._prec(precision, 'shading') if fine_net is not None else 0, _lib.dev_ptr(ws), _lib.dev_ptr(raw), _lib.dev_ptr(z), None, None, None,
        _lib.stream_ptr()), "nm_render_rays_bkg")
    return raw, z


def bkg_place_z(coarse_net, fine_net, o, d, near, far, samples_per_ray, importance_samples_per_ray, white_bkg, precision=None, trace=None):
    """Where the background list's FINAL samples are (render_utils.py:131-147, 287-293): the stratified samples, or -- with a fine net --
    the coarse density pass, its compositing weights and the importance samples merged in.  -> (z [R,S'], raw of the coarse pass when it
    is the pass that is composited [no fine net; evaluated here unless termination is on], else None)"""
    _, _, z = ray_utils.sample_z(o, d, near, far, samples_per_ray)
    if fine_net is None:
        return z, (None if TERMINATION_EPS > 0 else coarse_net.forward_rays(o, d, z, precision=precision, role='shading'))
    # with a fine net the coarse pass only places the importance samples (only its density is used, render_utils.py:139-141: the
    # colour head is skipped) and is not the pass the mixed precision policy tags 'shading' (vanilla.Joiner._prec)
    if TERMINATION_EPS > 0:
        # marched at TERMINATION_COARSE on its own transmittance only (where the importance samples go must not depend on what the
        # list is merged with later)
        stats = {} if trace is not None else None
        raw = march_pass_rays(coarse_net, o, d, z, TERMINATION_COARSE, precision=precision, role=None, stats=stats, sigma_only=True)
        _note(trace, march_coarse=stats)
    else:
        raw = coarse_net.forward_rays(o, d, z, precision=precision, role=None, sigma_only=True)
    # raw2outputs' weights -> sample_pdf -> sorted merge: one kernel (the weights reach HBM only for a trace)
    z_fine, w = ray_utils.importance_z_from_raw(raw, z, d, importance_samples_per_ray, want_weights=trace is not None)
    _note(trace, coarse_z=z, coarse_w=w)
    return z_fine, None


def bkg_shade(net, o, d, z, precision=None, trace=None, occluder=None, dz=None):
    """The background pass that is composited, on its final samples (render_utils.py:148-151, 294-297): whole, or -- TERMINATION_EPS > 0 --
    marched front to back at eps (`occluder`, `dz`: march_pass_rays)"""
    if TERMINATION_EPS > 0:
        stats = {} if trace is not None else None
        raw = march_pass_rays(net, o, d, z, TERMINATION_EPS, precision=precision, stats=stats, occluder=occluder, dz=dz)
        _note(trace, march=stats)
        return raw
    return net.forward_rays(o, d, z, precision=precision, role='shading')


def bkg_pass_rays(coarse_net, fine_net, o, d, near, far, samples_per_ray, importance_samples_per_ray, white_bkg,
                  precision=None, trace=None, given_z=None, occluder=None, dz=None):
    """Coarse (+ fine) background evaluation of R rays -> (raw [R,S',4], z [R,S'])  (render_utils.py:131-151, 287-297):
    bkg_place_z + bkg_shade.

    `given_z` [R, S'] (tests only, like `trace`): replay recorded final sample positions instead of deriving them -- the shading
    network is evaluated on exactly those.  The renderers pass it from their `given` dict ({'bkg_z': [R,S'], 'near_far':
    [(near [R], far [R]) per actor]}, device tensors indexed like the call's rays): the two steps that are ill conditioned in
    float32 -- the inverse CDF behind the importance samples and the cancellation under geometry_guided_near_far's square root
     -- are taken from a recording of the reference's own run, so that everything else can be held to 1e-4
    on every pixel against the reference's frames (tests/test_hip_posed_golden.py)."""
    net = fine_net if fine_net is not None else coarse_net
    if given_z is not None:
        z = given_z.to(torch.float32).contiguous()
        _note(trace, bkg_z=z)
        return net.forward_rays(o, d, z, precision=precision, role='shading'), z
    z, raw = bkg_place_z(coarse_net, fine_net, o, d, near, far, samples_per_ray, importance_samples_per_ray, white_bkg, precision, trace)
    if raw is None:
        raw = bkg_shade(net, o, d, z, precision, trace, occluder, dz)
    _note(trace, bkg_z=z)
    return raw, z


def render_vanilla_rays(coarse_net, fine_net, o, d, near, far, samples_per_ray, importance_samples_per_ray, white_bkg=True,
                        precision=None, trace=None, given=None):
    """Device core of render_vanilla: o, d [R,3] CUDA f32, scalar near/far -> (rgb [R,3], depth [R]) CUDA."""
    R = o.shape[0]
    rgb = torch.empty((R, 3), device=o.device, dtype=torch.float32)
    depth = torch.empty(R, device=o.device, dtype=torch.float32)
    for i, j in _chunks(R):
        oc, dc = o[i:j], d[i:j]
        n = torch.full((j - i,), float(near), device=o.device, dtype=torch.float32)
        f = torch.full((j - i,), float(far), device=o.device, dtype=torch.float32)
        raw, z = bkg_pass_rays(coarse_net, fine_net, oc, dc, n, f, samples_per_ray, importance_samples_per_ray, white_bkg,
                               precision, trace, given['bkg_z'][i:j] if given is not None and 'bkg_z' in given else None)
        rgb[i:j], _, _, _, depth[i:j] = raw2outputs(raw, z, dc, white_bkg=white_bkg, want_weights=False)
    return rgb, depth


def human_pass_rays(human_net, o, d, near, far, samples_per_ray, mesh=None, render_can=False, sigma_scale=1.0, precision=None,
                    trace=None):
    """Human-net evaluation of (already compacted) hit rays -> (raw [R,S,4], z [R,S])  (render_utils.py:213-229, 320-329): ONE C call
    per actor pass (nm_render_rays_human: sample -> warp -> network, enqueued back to back)."""
    _lib.require_gpu()
    R, S = o.shape[0], int(samples_per_ray)
    dev = o.device
    human_net._guard(o, d, near)
    posed = not render_can
    raw = torch.empty((R, S, 4), device=dev, dtype=torch.float32)
    z = torch.empty((R, S), device=dev, dtype=torch.float32)
    ws = _ws(_lib.lib().nm_render_rays_human_workspace_floats(R, S, int(posed)), dev)
    t_vals = torch.linspace(0., 1., steps=S, device=dev)
    _lib.check(_lib.lib().nm_render_rays_human(
        human_net.handle(), mesh.handle if posed else None, _lib.dev_ptr(mesh.T, torch.float64, 'T') if posed else None, _lib.dev_ptr(o.contiguous()),
        _lib.dev_ptr(d.contiguous()), _lib.dev_ptr(near.reshape(-1).contiguous()), _lib.dev_ptr(far.reshape(-1).contiguous()), R, S, _lib.dev_ptr(t_vals), 1,
        float(sigma_scale), human_net._prec(precision, 'shading'), _lib.dev_ptr(ws), _lib.dev_ptr(raw), _lib.dev_ptr(z), None, None, None, _lib.stream_ptr()),
        "nm_render_rays_human")
    if trace is not None:
        if posed:
            n3 = (R * S * 3 + 3) & ~3
            _note(trace, human_z=z, can_pts=ws[n3:n3 + R * S * 3].reshape(R, S, 3).clone(), can_dirs=ws[2 * n3:2 * n3 + R * S * 3].reshape(R, S, 3).clone())
        else:
            _note(trace, human_z=z)
    return raw, z


def merge_composite(za, rawa, zb, rawb, rays_d, white_bkg=True):
    """merge_sorted + raw2outputs of the merged list as ONE C call (render_utils.py:330-345) -> (rgb [R,3], depth [R], acc [R])"""
    _lib.require_gpu()
    R, Sa = za.shape
    Sb = zb.shape[1]
    dev = za.device
    rgb = torch.empty((R, 3), device=dev, dtype=torch.float32)
    depth = torch.empty(R, device=dev, dtype=torch.float32)
    acc = torch.empty(R, device=dev, dtype=torch.float32)
    ws = _ws(_lib.lib().nm_merge_composite_workspace_floats(R, Sa, Sb), dev)
    _lib.check(_lib.lib().nm_merge_composite(_lib.dev_ptr(za.contiguous()), _lib.dev_ptr(rawa.contiguous()), Sa, _lib.dev_ptr(zb.contiguous()),
                                             _lib.dev_ptr(rawb.contiguous()), Sb, R, _lib.dev_ptr(rays_d.contiguous()), int(bool(white_bkg)), _lib.dev_ptr(ws),
                                             _lib.dev_ptr(rgb), _lib.dev_ptr(depth), _lib.dev_ptr(acc), _lib.stream_ptr()), "nm_merge_composite")
    return rgb, depth, acc


def merge_composite_lists(z_lists, raw_lists, rays_d, white_bkg=True, rows=None):
    """k <= 4 sorted lists per ray -> merged order -> raw2outputs' sums, ONE kernel, the merged list never in HBM
    (render_utils.py:330-345, 441-456).  rows[l] (int32 [R] or None): list l's tensors are indexed by rows[l][ray] -- a list that
    exists for ALL rays of a batch read in place for the compacted hit rays.  -> (rgb [R,3], depth [R], acc [R]); bit-identical to
    merge_sorted list by list + raw2outputs."""
    _lib.require_gpu()
    k = len(z_lists)
    R = rays_d.shape[0]
    dev = rays_d.device
    zs = [z.to(torch.float32).contiguous() for z in z_lists]
    raws = [r_.to(torch.float32).contiguous() for r_ in raw_lists]
    rows = [None] * k if rows is None else [None if x is None else x.to(torch.int32).contiguous() for x in rows]
    for l_ in range(k):
        if rows[l_] is None and zs[l_].shape[0] != R:
            raise _lib.NeumanHipError(f"merge_composite_lists: list {l_} has {zs[l_].shape[0]} rows for {R} rays")
    arr = ctypes.c_void_p * k
    rgb = torch.empty((R, 3), device=dev, dtype=torch.float32)
    depth = torch.empty(R, device=dev, dtype=torch.float32)
    acc = torch.empty(R, device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().nm_merge_composite_lists(
        k, arr(*[z.data_ptr() for z in zs]), arr(*[r_.data_ptr() for r_ in raws]), arr(*[None if x is None else x.data_ptr() for x in rows]),
        (ctypes.c_int * k)(*[int(z.shape[1]) for z in zs]), R, _lib.dev_ptr(rays_d.contiguous()), int(bool(white_bkg)), _lib.dev_ptr(rgb), _lib.dev_ptr(depth),
        _lib.dev_ptr(acc), _lib.stream_ptr()), "nm_merge_composite_lists")
    return rgb, depth, acc


def render_smpl_nerf_rays(human_net, o, d, posed_verts, mesh, samples_per_ray, white_bkg=True, render_can=False,
                          geo_threshold=DEFAULT_GEO_THRESH, interval_comp=1.0, precision=None, trace=None, given=None):
    """Device core of render_smpl_nerf -> (rgb [R,3], depth [R], acc [R]) CUDA.  `given`: see bkg_pass_rays."""
    R = o.shape[0]
    rgb = torch.full((R, 3), 1.0 if white_bkg else 0.0, device=o.device, dtype=torch.float32)    # misses, :199-205
    depth = torch.zeros(R, device=o.device, dtype=torch.float32)
    acc = torch.zeros(R, device=o.device, dtype=torch.float32)
    for i, j in _chunks(R):
        oc, dc = o[i:j].contiguous(), d[i:j].contiguous()
        near, far = _given_near_far(given, 0, i, j, oc, dc, posed_verts, geo_threshold)
        _note(trace, near=near, far=far)
        hit, _ = ray_utils.compact_hits(near, far)
        if hit.numel() == 0:
            continue
        ho, hd = ray_utils.gather_rows(oc, hit), ray_utils.gather_rows(dc, hit)
        hn, hf = ray_utils.gather_rows(near, hit), ray_utils.gather_rows(far, hit)
        _note(trace, hit=hit + i)
        if TERMINATION_EPS > 0 and not render_can:
            mstats = {} if trace is not None else None
            raw, z = human_march_rays(human_net, ho, hd, hn, hf, samples_per_ray, mesh, TERMINATION_EPS, interval_comp, precision, None, trace, mstats)
            _note(trace, march_human=mstats)
        else:
            raw, z = human_pass_rays(human_net, ho, hd, hn, hf, samples_per_ray, mesh, render_can, interval_comp, precision, trace)
        _rgb, _, _acc, _, _depth = raw2outputs(raw, z, hd, white_bkg=white_bkg, want_weights=False)
        ray_utils.scatter_rows(rgb[i:j], hit, _rgb)
        ray_utils.scatter_rows(depth[i:j], hit, _depth)
        ray_utils.scatter_rows(acc[i:j], hit, _acc)
    return rgb, depth, acc


def render_hybrid_rays(coarse_bkg, fine_bkg, human_net, o, d, bkg_near, bkg_far, posed_verts, mesh, samples_per_ray,
                       importance_samples_per_ray, white_bkg=True, geo_threshold=DEFAULT_GEO_THRESH, precision=None, trace=None,
                       given=None):
    """Device core of render_hybrid_nerf -> (rgb [R,3], depth [R], acc [R]) CUDA  (render_utils.py:276-356).  `given`: see bkg_pass_rays.
    With TERMINATION_EPS > 0 the background's sample positions are settled first, then the human pass and the background shading pass
    are marched, each on the transmittance its samples have in the MERGED list (merged_intervals: the body's factors over the merged
    intervals bound the merged transmittance from above; behind the body the background sees its own x the body's): a pixel moves
    by < 2 eps, for bodies of any opacity (tests/test_hip_march.py: opaque and semi-transparent).
    The plain case (no trace, no replay, no termination) is ONE C call per batch: render_hybrid_rays_fused, bit-identical."""
    if TERMINATION_EPS <= 0 and trace is None and given is None:
        return render_hybrid_rays_fused(coarse_bkg, fine_bkg, human_net, o, d, bkg_near, bkg_far, posed_verts, mesh, samples_per_ray,
                                        importance_samples_per_ray, white_bkg, geo_threshold, precision)
    R = o.shape[0]
    rgb = torch.empty((R, 3), device=o.device, dtype=torch.float32)
    depth = torch.empty(R, device=o.device, dtype=torch.float32)
    acc = torch.zeros(R, device=o.device, dtype=torch.float32)                                   # misses: acc forced 0, :311
    for i, j in _chunks(R):
        oc, dc = o[i:j].contiguous(), d[i:j].contiguous()
        n = torch.full((j - i,), float(bkg_near), device=o.device, dtype=torch.float32)
        f = torch.full((j - i,), float(bkg_far), device=o.device, dtype=torch.float32)
        given_z = given['bkg_z'][i:j] if given is not None and 'bkg_z' in given else None

        def human_lists(merge_z=None):
            near, far = _given_near_far(given, 0, i, j, oc, dc, posed_verts, geo_threshold)
            _note(trace, near=near, far=far)
            hit, _ = ray_utils.compact_hits(near, far)
            if hit.numel() == 0:
                return hit, None, None, None, None
            ho, hd = ray_utils.gather_rows(oc, hit), ray_utils.gather_rows(dc, hit)
            hn, hf = ray_utils.gather_rows(near, hit), ray_utils.gather_rows(far, hit)
            _note(trace, hit=hit + i)
            if TERMINATION_EPS > 0:
                mstats = {} if trace is not None else None
                dz_h = None
                if merge_z is not None:                                                            # the intervals the body's samples have once merged
                    _, _, h_z0 = ray_utils.sample_z(ho, hd, hn, hf, samples_per_ray)
                    dz_h = merged_intervals([ray_utils.gather_rows(merge_z, hit), h_z0])[1]
                h_raw, h_z = human_march_rays(human_net, ho, hd, hn, hf, samples_per_ray, mesh, TERMINATION_EPS, 1.0, precision, None, trace, mstats,
                                              dz=dz_h)
                _note(trace, march_human=mstats)
            else:
                h_raw, h_z = human_pass_rays(human_net, ho, hd, hn, hf, samples_per_ray, mesh, False, 1.0, precision, trace)
            return hit, hd, hf, h_raw, h_z

        if TERMINATION_EPS > 0 and given_z is None:
            # sample positions first (they do not depend on the body), then the body -- it may hide the background -- then the
            # background shading pass; both marched on the transmittance of the MERGED list (merged_intervals): a pixel moves < 2 eps
            bkg_z, _ = bkg_place_z(coarse_bkg, fine_bkg, oc, dc, n, f, samples_per_ray, importance_samples_per_ray, white_bkg, precision, trace)
            _note(trace, bkg_z=bkg_z)
            dz_b = torch.cat([bkg_z[:, 1:] - bkg_z[:, :-1], torch.full_like(bkg_z[:, :1], 1e10)], 1)
            hit, hd, hf, h_raw, h_z = human_lists(bkg_z)
            occluder = None
            if hit.numel() > 0:
                z_far = torch.full((j - i,), float('inf'), device=o.device, dtype=torch.float32)
                T_occ = torch.ones(j - i, device=o.device, dtype=torch.float32)
                dz_bh, dz_h = merged_intervals([ray_utils.gather_rows(bkg_z, hit), h_z])
                ray_utils.scatter_rows(dz_b, hit, dz_bh)
                ray_utils.scatter_rows(z_far, hit, hf.reshape(-1))
                ray_utils.scatter_rows(T_occ, hit, transmittance_of(h_raw, h_z, hd, dz_h))
                occluder = (z_far, T_occ)
            bkg_raw = bkg_shade(fine_bkg if fine_bkg is not None else coarse_bkg, oc, dc, bkg_z, precision, trace, occluder, dz_b.contiguous())
        else:
            bkg_raw, bkg_z = bkg_pass_rays(coarse_bkg, fine_bkg, oc, dc, n, f, samples_per_ray, importance_samples_per_ray,
                                           white_bkg, precision, trace, given_z)
            hit, hd, hf, h_raw, h_z = human_lists()
        # every ray first gets the background-only composite (what the reference does for misses, :303-311) ...
        rgb[i:j], _, _, _, depth[i:j] = raw2outputs(bkg_raw, bkg_z, dc, white_bkg=white_bkg, want_weights=False)
        if hit.numel() == 0:
            continue
        # ... and hit rays are overwritten by the merged human + background composite (:313-353)
        _rgb, _depth, _ = merge_composite_lists([bkg_z, h_z], [bkg_raw, h_raw], hd, white_bkg, rows=[hit, None])     # (the background rows in place)
        _, _, _acc, _, _ = raw2outputs(h_raw, h_z, hd, white_bkg=white_bkg, want_weights=False)          # :345-350
        ray_utils.scatter_rows(rgb[i:j], hit, _rgb)
        ray_utils.scatter_rows(depth[i:j], hit, _depth)
        ray_utils.scatter_rows(acc[i:j], hit, _acc)
    return rgb, depth, acc


def render_hybrid_rays_fused(coarse_bkg, fine_bkg, human_net, o, d, bkg_near, bkg_far, posed_verts, mesh, samples_per_ray,
                             importance_samples_per_ray, white_bkg=True, geo_threshold=DEFAULT_GEO_THRESH, precision=None):
    """render_hybrid_rays' batch body as ONE C call (nm_render_rays_hybrid): same kernels, same bits; no trace / replay / early
    termination hooks -- the plain path of a frame render."""
    _lib.require_gpu()
    R = o.shape[0]
    dev = o.device
    rgb = torch.empty((R, 3), device=dev, dtype=torch.float32)
    depth = torch.empty(R, device=dev, dtype=torch.float32)
    acc = torch.empty(R, device=dev, dtype=torch.float32)
    S, N, Sh = int(samples_per_ray), int(importance_samples_per_ray) if fine_bkg is not None else 0, int(samples_per_ray)
    verts = posed_verts.to(dev, torch.float32).contiguous()
    t_vals = torch.linspace(0., 1., steps=S, device=dev)
    u = torch.linspace(0., 1., steps=N, device=dev) if N else None
    # the C call sizes its hit-ray buffers for every ray of the batch (it cannot read the hit count back without a host
    # synchronisation): ~28 KB per ray at 128 + 128 + 128 samples, so batches of FUSED_HYBRID_RAYS rays (3.7 GB) and ONE workspace
    step = min(MAX_RAYS_PER_LAUNCH, FUSED_HYBRID_RAYS)
    ws = _ws(_lib.lib().nm_render_rays_hybrid_workspace_floats(min(R, step), S, N, Sh), dev)
    for i in range(0, R, step):
        j = min(i + step, R)
        oc, dc = o[i:j].contiguous(), d[i:j].contiguous()
        n = j - i
        r_, d_, a_ = rgb[i:j], depth[i:j], acc[i:j]
        _lib.check(_lib.lib().nm_render_rays_hybrid(
            coarse_bkg.handle(), fine_bkg.handle() if fine_bkg is not None else None, human_net.handle(), mesh.handle, _lib.dev_ptr(mesh.T, torch.float64, 'T'),
            _lib.dev_ptr(verts), verts.shape[0], float(geo_threshold), _lib.dev_ptr(oc), _lib.dev_ptr(dc), n, float(bkg_near), float(bkg_far), S, N, Sh,
            _lib.dev_ptr(t_vals), _lib.dev_ptr(u), _lib.dev_ptr(t_vals), int(bool(white_bkg)),
            coarse_bkg._prec(precision, None if fine_bkg is not None else 'shading'), fine_bkg._prec(precision, 'shading') if fine_bkg is not None else 0,
            human_net._prec(precision, 'shading'), _lib.dev_ptr(ws), _lib.dev_ptr(r_), _lib.dev_ptr(d_), _lib.dev_ptr(a_), _lib.stream_ptr()),
            "nm_render_rays_hybrid")
    return rgb, depth, acc


def render_multi_rays(coarse_bkg, fine_bkg, human_nets, o, d, bkg_near, bkg_far, posed_verts, meshes, samples_per_ray,
                      importance_samples_per_ray, white_bkg=True, geo_threshold=DEFAULT_GEO_THRESH, precision=None, trace=None,
                      given=None):
    """Device core of render_hybrid_nerf_multi_persons -> (rgb [R,3], depth [R]) CUDA  (render_utils.py:390-456).  `given`: see
    bkg_pass_rays.  TERMINATION_EPS > 0: as render_hybrid_rays (sample positions of all lists, then the actors, then the background
    shading pass, each marched on the transmittance of its samples over their intervals in the MERGED list; the background behind the
    farthest body a ray hits on its own x all the bodies'); a pixel moves by < (1 + actors) eps."""
    R = o.shape[0]
    rgb = torch.empty((R, 3), device=o.device, dtype=torch.float32)
    depth = torch.empty(R, device=o.device, dtype=torch.float32)
    for i, j in _chunks(R):
        oc, dc = o[i:j].contiguous(), d[i:j].contiguous()
        nr = j - i
        n = torch.full((nr,), float(bkg_near), device=o.device, dtype=torch.float32)
        f = torch.full((nr,), float(bkg_far), device=o.device, dtype=torch.float32)
        far_z = torch.linspace(float(bkg_far) * 2, float(bkg_far) * 3, samples_per_ray, device=o.device)   # :418-419

        def actor_rays(a_):
            """one actor's hit list and the z of its list over ALL rays of the batch: its samples where it is hit, the zero-density
            placeholders elsewhere (render_utils.py:405-419)"""
            near, far = _given_near_far(given, a_, i, j, oc, dc, posed_verts[a_], geo_threshold)
            _note(trace, near=near, far=far)
            h_z = far_z[None].repeat(nr, 1).contiguous()
            hit, _ = ray_utils.compact_hits(near, far)
            _note(trace, hit=hit + i)
            if hit.numel() == 0:
                return hit, None, h_z
            ho, hd = ray_utils.gather_rows(oc, hit), ray_utils.gather_rows(dc, hit)
            hn, hf = ray_utils.gather_rows(near, hit), ray_utils.gather_rows(far, hit)
            return hit, (ho, hd, hn, hf), h_z

        def actor_lists_compact(a_):
            """the actor's list as COMPACT arrays for merge_composite_lists' `rows` indirection: the evaluated rows of the hit rays followed by ONE placeholder row
            (the zero-density samples at linspace(2 far, 3 far), render_utils.py:418-419) that every missed ray points at -> (z [n_hit + 1, S], raw [n_hit + 1, S, 4],
            rows [nr] int32).  The same values the full [nr, S] arrays hold, without filling, scattering and streaming 16 KB per missed ray and actor."""
            near, far = _given_near_far(given, a_, i, j, oc, dc, posed_verts[a_], geo_threshold)
            _note(trace, near=near, far=far)
            hit, _ = ray_utils.compact_hits(near, far)
            _note(trace, hit=hit + i)
            n_hit = int(hit.numel())
            rows = torch.full((nr,), n_hit, device=o.device, dtype=torch.int32)
            pad_raw = torch.zeros((1, samples_per_ray, 4), device=o.device, dtype=torch.float32)
            if n_hit == 0:
                _note(trace, human_z=None, can_pts=None, can_dirs=None)
                return far_z[None].contiguous(), pad_raw, rows
            ho, hd = ray_utils.gather_rows(oc, hit), ray_utils.gather_rows(dc, hit)
            hn, hf = ray_utils.gather_rows(near, hit), ray_utils.gather_rows(far, hit)
            r_, z_ = human_pass_rays(human_nets[a_], ho, hd, hn, hf, samples_per_ray, meshes[a_], False, 1.0, precision, trace)
            rows[hit.to(torch.int64)] = torch.arange(n_hit, device=o.device, dtype=torch.int32)
            return torch.cat([z_, far_z[None]], 0), torch.cat([r_, pad_raw], 0), rows

        def actor_lists(a_, rays=None, dz=None):
            """the actor's list evaluated -> (h_z [nr,S], h_raw [nr,S,4], (hit, far, transmittance over `dz`) when marched)"""
            hit, hr, h_z = rays if rays is not None else actor_rays(a_)
            h_raw = torch.zeros((nr, samples_per_ray, 4), device=o.device, dtype=torch.float32)
            occ = None
            if hit.numel() > 0:
                ho, hd, hn, hf = hr
                if TERMINATION_EPS > 0:
                    mstats = {} if trace is not None else None
                    dz_h = ray_utils.gather_rows(dz, hit) if dz is not None else None
                    r_, z_ = human_march_rays(human_nets[a_], ho, hd, hn, hf, samples_per_ray, meshes[a_], TERMINATION_EPS, 1.0, precision, None, trace,
                                              mstats, dz=dz_h)
                    _note(trace, march_human=mstats)
                    occ = (hit, hf.reshape(-1), transmittance_of(r_, z_, hd, dz_h))
                else:
                    r_, z_ = human_pass_rays(human_nets[a_], ho, hd, hn, hf, samples_per_ray, meshes[a_], False, 1.0, precision, trace)
                ray_utils.scatter_rows(h_raw.reshape(nr, -1), hit, r_.reshape(hit.shape[0], -1))
                ray_utils.scatter_rows(h_z, hit, z_)
            else:
                _note(trace, human_z=None, can_pts=None, can_dirs=None)
            return h_z, h_raw, occ

        lists = None
        given_z = given['bkg_z'][i:j] if given is not None and 'bkg_z' in given else None
        if TERMINATION_EPS > 0 and given_z is None:
            # sample positions of every list first (none depends on another list's densities), then the bodies -- they may hide the
            # background -- then the background shading pass, each marched on the transmittance its samples have in the MERGED list
            z_all, _ = bkg_place_z(coarse_bkg, fine_bkg, oc, dc, n, f, samples_per_ray, importance_samples_per_ray, white_bkg, precision, trace)
            _note(trace, bkg_z=z_all)
            rays_l = [actor_rays(a_) for a_ in range(len(human_nets))]
            for hit, hr, h_z in rays_l:                                                              # (z of the hit rows, as human_march_rays will sample them)
                if hit.numel() > 0:
                    ray_utils.scatter_rows(h_z, hit, ray_utils.sample_z(hr[0], hr[1], hr[2], hr[3], samples_per_ray)[2])
            dz_l = merged_intervals([z_all] + [h_z for _, _, h_z in rays_l])
            lists = [actor_lists(a_, rays_l[a_], dz_l[1 + a_]) for a_ in range(len(human_nets))]
            z_far = torch.full((nr,), float('-inf'), device=o.device, dtype=torch.float32)
            T_occ = torch.ones(nr, device=o.device, dtype=torch.float32)
            for _, _, occ in lists:
                if occ is not None:
                    hit, hf, Th = occ
                    idx = hit.to(torch.int64)
                    z_far[idx] = torch.maximum(z_far[idx], hf)
                    T_occ[idx] = T_occ[idx] * Th
            z_far = torch.where(torch.isinf(z_far), torch.full_like(z_far, float('inf')), z_far)  # rays that hit nobody: never
            raw_all = bkg_shade(fine_bkg if fine_bkg is not None else coarse_bkg, oc, dc, z_all, precision, trace, (z_far, T_occ), dz_l[0])
        else:
            raw_all, z_all = bkg_pass_rays(coarse_bkg, fine_bkg, oc, dc, n, f, samples_per_ray, importance_samples_per_ray,
                                           white_bkg, precision, trace, given_z)
        # In this renderer the terminal 1e10 interval sits on an actor's zero-density placeholder whenever a ray misses one
        # (render_utils.py:418-419), so the LAST background sample is followed by a finite interval of ~far..2 far instead:
        # alpha = 1 - exp(-sigma * 3.14..) is then ~300x as sensitive to that one sigma as a sample inside the ray is.  Under the
        # mixed policy (i8x3 shading passes) that single sample per ray is re-evaluated in the float32-class arithmetic.
        last_net = fine_bkg if fine_bkg is not None else coarse_bkg
        if last_net._prec(precision, 'shading') == _lib.NM_PREC_I8X3 and (precision or last_net.precision) == 'mixed':
            last = last_net.forward_rays(oc, dc, z_all[:, -1:].contiguous(), precision='fp16x3')[:, 0, :]
            if TERMINATION_EPS > 0:                                                              # a sample the march never reached stays unevaluated
                last = torch.where((raw_all[:, -1, :] == 0).all(-1, keepdim=True), raw_all[:, -1, :], last)
            raw_all[:, -1, :] = last
        if lists is None and len(human_nets) <= 3 and MULTI_COMPACT:
            cl = [actor_lists_compact(a_) for a_ in range(len(human_nets))]
            rgb[i:j], depth[i:j], _ = merge_composite_lists([z_all] + [c[0] for c in cl], [raw_all] + [c[1] for c in cl], dc, white_bkg, rows=[None] + [c[2] for c in cl])
            continue
        if lists is None:
            lists = [actor_lists(a_) for a_ in range(len(human_nets))]
        if len(lists) <= 3:                                                                      # :441-456 as one kernel: 4-way merge + composite
            rgb[i:j], depth[i:j], _ = merge_composite_lists([z_all] + [l_[0] for l_ in lists], [raw_all] + [l_[1] for l_ in lists], dc, white_bkg)
        else:                                                                                    # more than three actors: list by list
            for h_z, h_raw, _ in lists:
                z_all, raw_all = merge_sorted(z_all, raw_all, h_z, h_raw)
            rgb[i:j], _, _, _, depth[i:j] = raw2outputs(raw_all, z_all, dc, white_bkg=white_bkg, want_weights=False)
    return rgb, depth


# ------------------------------------------------------------------------------------------------
# the reference's frame renderers (same signatures, numpy out)
# ------------------------------------------------------------------------------------------------
def _device_of(net):
    dev = next(net.parameters()).device
    if dev.type != 'cuda':
        raise _lib.NeumanHipError("the network must live on the HIP device (net.cuda()): there is no CPU render path")
    return dev


MULTI_COMPACT = os.environ.get("NEUMAN_MULTI_COMPACT", "1") != "0"   # the multi-person merge reads compact per-actor lists through row indices (0: full [R, S] arrays, A/B)
HOST_RAYS = os.environ.get("NEUMAN_HOST_RAYS", "0") == "1"     # A/B switch: generate rays with the host (numpy) mirror and upload them


def _pixel_rays(cap, device):
    """shot_rays over every pixel, (x, y) row-major (render_utils.py:185-186)."""
    h, w = cap.shape
    if HOST_RAYS:
        coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
        o, d = ray_utils.shot_rays(cap, coords)
        return (torch.from_numpy(o).to(device, torch.float32).contiguous(),
                torch.from_numpy(d).to(device, torch.float32).contiguous())
    i = torch.arange(h * w, device=device, dtype=torch.int32)
    return ray_utils.shot_rays_dev(cap, torch.stack([i % w, i // w], dim=1))


def _all_rays(cap, device):
    """shot_all_rays (render_utils.py:125-128)."""
    if HOST_RAYS:
        o, d = ray_utils.shot_all_rays(cap)
        return (torch.from_numpy(o).to(device, torch.float32).contiguous(),
                torch.from_numpy(d).to(device, torch.float32).contiguous())
    return ray_utils.shot_all_rays_dev(cap, device)


def _frame(rays_fn, o, d):
    """The frame's rays through `rays_fn` -> tuple of per-ray tensors: on this device alone, or -- under an initialised process
    group (parallel.sharding_active) -- the rays of this rank's interleaved tiles only, with ONE gather assembling the frame on
    rank 0 (None on the other ranks; parallel.render_frame_sharded)."""
    if parallel.sharding_active():
        return parallel.render_frame_sharded(rays_fn, o, d)
    return rays_fn(o, d)


# ------------------------------------------------------------------------------------------------
# frame egress (reference render_test_views.py:83-92, 27-41): uint8 pixels and their PSNR, on the device
# ------------------------------------------------------------------------------------------------
def frame_to_uint8(frame):
    """What `imageio.imsave(path, out)` makes of a renderer's float frame before encoding: clip to [0, 1],
    `uint8(x * 255 + 0.499999999)`.  CUDA f32 tensor of any shape -> uint8 tensor of the same shape."""
    _lib.require_gpu()
    src = frame.to(torch.float32).contiguous()
    dst = torch.empty(src.shape, device=src.device, dtype=torch.uint8)
    _lib.check(_lib.lib().nm_frame_to_uint8(_lib.dev_ptr(src), src.numel(), ctypes.c_void_p(dst.data_ptr()), _lib.stream_ptr()),
               "nm_frame_to_uint8")
    return dst


def psnr_uint8(gt, pred):
    """skimage.metrics.peak_signal_noise_ratio(gt, pred) for uint8 frames (render_test_views.py:35): 10 log10(255^2 / mse)."""
    _lib.require_gpu()
    if gt.dtype != torch.uint8 or pred.dtype != torch.uint8 or gt.shape != pred.shape or not (gt.is_cuda and pred.is_cuda):
        raise _lib.NeumanHipError("psnr_uint8 takes two CUDA uint8 tensors of the same shape")
    gt, pred = gt.contiguous(), pred.contiguous()
    ssd = torch.empty(1, device=gt.device, dtype=torch.int64)
    _lib.check(_lib.lib().nm_ssd_u8(ctypes.c_void_p(gt.data_ptr()), ctypes.c_void_p(pred.data_ptr()), gt.numel(),
                                    ctypes.c_void_p(ssd.data_ptr()), _lib.stream_ptr()), "nm_ssd_u8")
    mse = float(ssd.item()) / gt.numel()
    return float('inf') if mse == 0 else 10.0 * float(np.log10(255.0 ** 2 / mse))


def save_png(path, frame):
    """imageio.imsave(path, frame) for the renderers' output (render_test_views.py:83-88, render_360.py:77-81): a float frame goes
    through `frame_to_uint8` on the device, then an 8-bit RGB / grey PNG is written with the standard library (zlib): same
    pixels as imageio's file, not the same bytes (deflate settings differ)."""
    import struct
    import zlib
    if isinstance(frame, np.ndarray):
        frame = torch.as_tensor(np.ascontiguousarray(frame))
    if frame.dtype != torch.uint8:
        frame = frame_to_uint8(frame.to('cuda', torch.float32))
    a = frame.cpu().numpy()
    if a.ndim == 2:
        a = a[..., None]
    h, w, c = a.shape
    if c not in (1, 3, 4):
        raise ValueError(f"save_png: {c} channels")
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], 1).tobytes()          # filter type 0 on every row

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, {1: 0, 3: 2, 4: 6}[c], 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def ssim_uint8(pred, gt):
    """skimage.metrics.structural_similarity(pred, gt, multichannel=True) for uint8 [H,W,C] frames (render_test_views.py:33)."""
    _lib.require_gpu()
    if gt.dtype != torch.uint8 or pred.dtype != torch.uint8 or gt.shape != pred.shape or gt.dim() != 3 or not (gt.is_cuda and pred.is_cuda):
        raise _lib.NeumanHipError("ssim_uint8 takes two CUDA uint8 tensors [H,W,C] of the same shape")
    gt, pred = gt.contiguous(), pred.contiguous()
    out = torch.empty(1, device=gt.device, dtype=torch.float64)
    ws = torch.empty(4096, device=gt.device, dtype=torch.float64)
    H, W, C = gt.shape
    _lib.check(_lib.lib().nm_ssim_u8(ctypes.c_void_p(pred.data_ptr()), ctypes.c_void_p(gt.data_ptr()), H, W, C, ctypes.c_void_p(out.data_ptr()),
                                     ctypes.c_void_p(ws.data_ptr()), _lib.stream_ptr()), "nm_ssim_u8")
    return float(out.item())


def render_vanilla(coarse_net, cap, fine_net=None, rays_per_batch=32768, samples_per_ray=64, importance_samples_per_ray=128,
                   white_bkg=True, near_far_source='bkg', return_depth=False, ablate_nerft=False):
    """reference render_utils.py:108-161."""
    device = _device_of(coarse_net)
    if ablate_nerft and not NERFT_GEMM_CHAIN:
        # the frame's time is one constant (render_utils.py:134-148): the 4-D-encoding nets at that time ARE 3-D-encoding nets with two
        # other bias vectors (vanilla.frozen_time_joiner) -- everything below, sharding included, is the ordinary path on the MFMA kernels
        from . import vanilla
        cur_time = cap.frame_id['frame_id'] / cap.frame_id['total_frames']
        coarse_net = vanilla.frozen_time_joiner(coarse_net, cur_time)
        fine_net = vanilla.frozen_time_joiner(fine_net, cur_time) if fine_net is not None else None
    elif ablate_nerft:
        return _render_vanilla_with_time(coarse_net, cap, fine_net, samples_per_ray, importance_samples_per_ray, white_bkg, near_far_source,
                                         return_depth, device)
    with torch.no_grad():
        o, d = _all_rays(cap, device)
        out = _frame(lambda oo, dd: render_vanilla_rays(coarse_net, fine_net, oo, dd, cap.near[near_far_source], cap.far[near_far_source],
                                                        samples_per_ray, importance_samples_per_ray, white_bkg), o, d)
        if out is None:                                                                          # a rank other than 0 of a sharded render
            return None
        rgb, depth = out
        rgb = rgb.reshape(*cap.shape, -1).cpu().numpy()
        depth = depth.reshape(*cap.shape).cpu().numpy()
    return (rgb, depth) if return_depth else rgb


# NEUMAN_NERFT_GEMM=1: render_vanilla(ablate_nerft=True) sample by sample with the time as a fourth input coordinate on the float32 GEMM chain
# (the form that also serves a time varying from sample to sample) instead of the folded-time nets on the MFMA kernels
NERFT_GEMM_CHAIN = os.environ.get('NEUMAN_NERFT_GEMM', '0') == '1'


def _render_vanilla_with_time(coarse_net, cap, fine_net, samples_per_ray, importance_samples_per_ray, white_bkg, near_far_source, return_depth,
                              device, rays_per_chunk=8192):
    """render_vanilla(ablate_nerft=True), reference render_utils.py:134-151: every sample point carries the frame's time
    `frame_id / total_frames` as a fourth coordinate (ray_utils.py:133-134, 158-159) and the nets encode 4-vectors.  Spelled with
    the reference-named functions; the nets run on the float32 GEMM chain (Joiner.forward), the sampling and compositing on the
    usual kernels.  Chunked: that forward keeps its layer outputs while it runs (8 KB per sample)."""
    with torch.no_grad():
        o, d = _all_rays(cap, device)
        R = o.shape[0]
        cur_time = cap.frame_id['frame_id'] / cap.frame_id['total_frames']
        rgb = torch.empty((R, 3), device=device, dtype=torch.float32)
        depth = torch.empty(R, device=device, dtype=torch.float32)
        for i in range(0, R, rays_per_chunk):
            j = min(i + rays_per_chunk, R)
            n = j - i
            batch = {'origin': o[i:j], 'direction': d[i:j],
                     'near': torch.full((n, 1), float(cap.near[near_far_source]), device=device),
                     'far': torch.full((n, 1), float(cap.far[near_far_source]), device=device)}
            coarse_time = torch.ones((n, samples_per_ray, 1), device=device) * cur_time
            pts, dirs, z = ray_utils.ray_to_samples(batch, samples_per_ray, device=device, append_t=coarse_time)
            out = coarse_net(pts, dirs)
            rgb_c, _, _, w, depth_c = raw2outputs(out, z, dirs[:, 0, :].contiguous(), white_bkg=white_bkg)
            if fine_net is not None:
                fine_time = torch.ones((n, samples_per_ray + importance_samples_per_ray, 1), device=device) * cur_time
                pts, dirs, z = ray_utils.ray_to_importance_samples(batch, z, w, importance_samples_per_ray, device=device, append_t=fine_time)
                out = fine_net(pts, dirs)
                rgb_c, _, _, _, depth_c = raw2outputs(out, z, dirs[:, 0, :].contiguous(), white_bkg=white_bkg, want_weights=False)
            rgb[i:j], depth[i:j] = rgb_c, depth_c
        rgb = rgb.reshape(*cap.shape, -1).cpu().numpy()
        depth = depth.reshape(*cap.shape).cpu().numpy()
    return (rgb, depth) if return_depth else rgb


def render_smpl_nerf(net, cap, posed_verts, faces, Ts, rays_per_batch=32768, samples_per_ray=64, white_bkg=True,
                     render_can=False, geo_threshold=DEFAULT_GEO_THRESH, return_depth=False, return_mask=False,
                     interval_comp=1.0):
    """reference render_utils.py:164-246."""
    device = _device_of(net)
    with torch.no_grad():
        o, d = _pixel_rays(cap, device)
        verts = torch.as_tensor(np.ascontiguousarray(posed_verts, dtype=np.float32)).to(device)
        mesh = None if render_can else ray_utils.mesh_to_device(posed_verts, faces, Ts, device)
        out = _frame(lambda oo, dd: render_smpl_nerf_rays(net.coarse_human_net, oo, dd, verts, mesh, samples_per_ray, white_bkg, render_can,
                                                          geo_threshold, interval_comp), o, d)
        if out is None:
            return None
        rgb, depth, acc = out
        rgb = rgb.reshape(*cap.shape, -1).cpu().numpy()
        depth = depth.reshape(*cap.shape).cpu().numpy()
        acc = acc.reshape(*cap.shape).cpu().numpy()
    if return_depth and return_mask:
        return rgb, depth, acc
    if return_depth:
        return rgb, depth
    if return_mask:
        return rgb, acc
    return rgb


def render_hybrid_nerf(net, cap, posed_verts, faces, Ts, rays_per_batch=32768, samples_per_ray=64,
                       importance_samples_per_ray=128, white_bkg=True, geo_threshold=DEFAULT_GEO_THRESH, return_depth=False):
    """reference render_utils.py:249-362."""
    device = _device_of(net)
    with torch.no_grad():
        o, d = _pixel_rays(cap, device)
        verts = torch.as_tensor(np.ascontiguousarray(posed_verts, dtype=np.float32)).to(device)
        mesh = ray_utils.mesh_to_device(posed_verts, faces, Ts, device)
        out = _frame(lambda oo, dd: render_hybrid_rays(net.coarse_bkg_net, net.fine_bkg_net, net.coarse_human_net, oo, dd, cap.near['bkg'],
                                                       cap.far['bkg'], verts, mesh, samples_per_ray, importance_samples_per_ray, white_bkg,
                                                       geo_threshold)[:2], o, d)
        if out is None:
            return None
        rgb, depth = out
        rgb = rgb.reshape(*cap.shape, -1).cpu().numpy()
        depth = depth.reshape(*cap.shape).cpu().numpy()
    return (rgb, depth) if return_depth else rgb


def render_hybrid_nerf_multi_persons(bkg_model, cap, human_models, posed_verts, faces, Ts, rays_per_batch=32768,
                                     samples_per_ray=64, importance_samples_per_ray=128, white_bkg=True,
                                     geo_threshold=DEFAULT_GEO_THRESH, return_depth=False):
    """reference render_utils.py:365-461."""
    device = _device_of(bkg_model)
    with torch.no_grad():
        o, d = _pixel_rays(cap, device)
        verts = [torch.as_tensor(np.ascontiguousarray(v, dtype=np.float32)).to(device) for v in posed_verts]
        meshes = [ray_utils.mesh_to_device(v, f, t, device) for v, f, t in zip(posed_verts, faces, Ts)]
        out = _frame(lambda oo, dd: render_multi_rays(bkg_model.coarse_bkg_net, bkg_model.fine_bkg_net,
                                                      [m.coarse_human_net for m in human_models], oo, dd, cap.near['bkg'], cap.far['bkg'], verts,
                                                      meshes, samples_per_ray, importance_samples_per_ray, white_bkg, geo_threshold), o, d)
        if out is None:
            return None
        rgb, depth = out
        rgb = rgb.reshape(*cap.shape, -1).cpu().numpy()
        depth = depth.reshape(*cap.shape).cpu().numpy()
    return (rgb, depth) if return_depth else rgb
