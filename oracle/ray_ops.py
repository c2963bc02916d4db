"""Oracle: ray generation, stratified / hierarchical sampling, SMPL-guided bounds.

numpy float32 restatement of reference utils/ray_utils.py (test infrastructure,
see oracle/__init__.py).  Every function cites the reference lines it follows.
All arithmetic is float32 unless the reference itself computes in float64.
"""
import numpy as np

F32 = np.float32
PERTURB_EPSILON = 0.01        # reference utils/constant.py:15
DEFAULT_GEO_THRESH = 0.2      # reference utils/constant.py:14


def linspace_f32(start, end, steps):
    """torch.linspace(start, end, steps) in float32.

    torch computes ``step = (end-start)/(steps-1)`` in float32 and mirrors the
    upper half from ``end`` (ATen RangeFactories); its CPU kernel is vectorised
    per ISA, so individual elements can differ from this scalar form by 1 ulp.
    Consumers compare with a tolerance, never bitwise.
    """
    start, end = F32(start), F32(end)
    if steps == 1:
        return np.array([start], dtype=F32)
    step = F32((end - start) / F32(steps - 1))
    idx = np.arange(steps)
    lo = (start + step * idx.astype(F32)).astype(F32)
    hi = (end - step * (steps - idx - 1).astype(F32)).astype(F32)
    return np.where(idx < steps // 2, lo, hi).astype(F32)


# ----------------------------------------------------------------------------
# ray generation (host side, numpy f64 -> f32)
# ----------------------------------------------------------------------------
def pcd_2d_to_pcd_3d(xy, depth, intrinsic, cam2world):
    """reference geometry/pcd_projector.py:85-120 (float64)."""
    x, y, z = xy[:, 0], xy[:, 1], depth[:, 0]
    xyz = np.stack([x, y, np.ones_like(x)], axis=1)
    xyz = np.matmul(np.linalg.inv(intrinsic), xyz.T).T * z[..., None]
    xyz = xyz[np.where(xyz[:, 2] > 0)]
    xyzw = np.concatenate([xyz, np.ones_like(xyz[:, 0:1])], axis=1)
    xyzw = np.matmul(cam2world, xyzw.T).T
    xyzw = xyzw[np.where(xyzw[:, 3] != 0)]
    xyzw /= xyzw[:, 3:4]
    return xyzw[:, 0:3]


def shot_rays(intrinsic, cam2world, xys):
    """reference utils/ray_utils.py:23-29.

    Returns (orig f64 [N,3], dir f32 [N,3]) exactly like the reference (the
    world points are cast to f32 before the centre is subtracted; the centre
    stays f64 so ``dir`` is promoted to f64, normalised, and callers cast).
    """
    z = np.ones((xys.shape[0], 1))
    pcd_3d = pcd_2d_to_pcd_3d(xys.astype(np.float64), z, intrinsic, cam2world).astype(F32)
    center = cam2world[:3, 3]
    orig = np.stack([center] * xys.shape[0])
    d = pcd_3d - orig
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return orig, d


def all_pixel_coords(shape):
    """``np.argwhere(np.ones(cap.shape))[:, ::-1]`` (reference render_utils.py:185): (x, y), row-major."""
    return np.argwhere(np.ones(shape))[:, ::-1]


def shot_all_rays(intrinsic, cam2world, shape):
    """reference utils/ray_utils.py:32-38 via pcd_projector.img_to_pcd_3d (123-153, 209-227). float64 out."""
    h, w = shape
    x, y = np.meshgrid(np.linspace(0, w - 1, num=w), np.linspace(0, h - 1, num=h))
    xy = np.concatenate([x.reshape(-1, 1), y.reshape(-1, 1)], axis=1)
    z = np.ones((h * w, 1))
    pcd = pcd_2d_to_pcd_3d(xy, z, intrinsic, cam2world)
    center = cam2world[:3, 3]
    dirs = pcd - center
    dirs = dirs / np.linalg.norm(dirs, axis=1, keepdims=True)
    origs = np.stack([center] * dirs.shape[0], axis=0)
    return origs, dirs


def to_homogeneous(pts):
    """reference utils/ray_utils.py:41-45."""
    return np.concatenate([pts, np.ones_like(pts[..., 0:1])], axis=-1)


# ----------------------------------------------------------------------------
# sampling
# ----------------------------------------------------------------------------
def ray_to_samples(origin, direction, near, far, samples_per_ray, lindisp=False, t_rand=None, t_vals=None, append_t=None):
    """reference utils/ray_utils.py:96-135.

    origin/direction [R,3] f32, near/far [R,1] f32.  ``t_rand`` (already clipped
    to [eps, 1-eps], shape [R,S]) replaces the reference's torch.rand draw for
    perturb>0.  Returns pts [R,S,3], dirs [R,S,3], z_vals [R,S] (all f32).
    """
    o, d = origin.astype(F32), direction.astype(F32)
    near, far = near.astype(F32), far.astype(F32)
    assert near.shape[0] == far.shape[0] == o.shape[0]
    t = linspace_f32(0., 1., samples_per_ray) if t_vals is None else t_vals.astype(F32)
    if not lindisp:
        z = (near * (F32(1.) - t) + far * t).astype(F32)
    else:
        z = (F32(1.) / (F32(1.) / near * (F32(1.) - t) + F32(1.) / far * t)).astype(F32)
    if t_rand is not None:
        mids = (F32(.5) * (z[..., 1:] + z[..., :-1])).astype(F32)
        upper = np.concatenate([mids, z[..., -1:]], -1)
        lower = np.concatenate([z[..., :1], mids], -1)
        z = (lower + (upper - lower) * t_rand.astype(F32)).astype(F32)
    pts = (o[..., None, :] + d[..., None, :] * z[..., :, None]).astype(F32)
    dirs = np.stack([d] * samples_per_ray, axis=1)
    if append_t is not None:                                    # ray_utils.py:133-134
        pts = np.concatenate([pts, append_t.astype(F32)], -1)
    return pts, dirs, z


def sample_pdf(bins, weights, n_samples, u=None, cdf_ulps=0):
    """reference utils/ray_utils.py:164-194 with det=True (the only mode the reference uses, :149).

    bins [R,B] f32, weights [R,B-1] f32 -> samples [R,N] f32.

    Rounding convention (the reference leaves it to torch): the normaliser is the correctly rounded f32 sum
    (torch.sum's f32 result is within an ulp or two of it, order unspecified) and the running sum accumulates
    in float64 and rounds each entry to f32 -- exactly what torch's CPU cumsum does (acc_type<float,cpu> = double).
    The inverse-CDF lookup below is a step function of (u - cdf[i]); ``cdf_ulps`` nudges every cdf entry by that many
    f32 ulps so that callers can measure how ill-conditioned a ray is (tests flag rays whose colour moves under +-1 ulp).
    """
    weights = (weights.astype(F32) + F32(1e-5)).astype(F32)
    pdf = (weights / np.sum(weights.astype(np.float64), -1, keepdims=True).astype(F32)).astype(F32)
    cdf = np.cumsum(pdf.astype(np.float64), -1).astype(F32)
    for _ in range(abs(int(cdf_ulps))):
        cdf = np.nextafter(cdf, F32(np.inf if cdf_ulps > 0 else -np.inf)).astype(F32)
    cdf = np.concatenate([np.zeros_like(cdf[..., :1]), cdf], -1)
    if u is None:
        u = linspace_f32(0., 1., n_samples)
    u = np.broadcast_to(u.astype(F32), cdf.shape[:-1] + (n_samples,))
    nb = cdf.shape[-1]
    inds = np.empty(u.shape, dtype=np.int64)
    for r in range(cdf.shape[0]):
        inds[r] = np.searchsorted(cdf[r], u[r], side='right')
    below = np.maximum(0, inds - 1)
    above = np.minimum(nb - 1, inds)
    cdf_g0 = np.take_along_axis(cdf, below, -1)
    cdf_g1 = np.take_along_axis(cdf, above, -1)
    bins = bins.astype(F32)
    bins_g0 = np.take_along_axis(bins, below, -1)
    bins_g1 = np.take_along_axis(bins, above, -1)
    denom = (cdf_g1 - cdf_g0).astype(F32)
    denom = np.where(denom < F32(1e-5), np.ones_like(denom), denom)
    t = ((u - cdf_g0) / denom).astype(F32)
    return (bins_g0 + t * (bins_g1 - bins_g0)).astype(F32)


def ray_to_importance_samples(origin, direction, z_vals, weights, n_importance, including_old=True, cdf_ulps=0, append_t=None):
    """reference utils/ray_utils.py:138-160."""
    o, d = origin.astype(F32), direction.astype(F32)
    z_mid = (F32(.5) * (z_vals[..., 1:] + z_vals[..., :-1])).astype(F32)
    z_samples = sample_pdf(z_mid, weights[..., 1:-1], n_importance, cdf_ulps=cdf_ulps)
    if including_old:
        z = np.sort(np.concatenate([z_vals, z_samples], -1), -1)
    else:
        z = z_samples
    pts = (o[..., None, :] + d[..., None, :] * z[..., :, None]).astype(F32)
    dirs = np.stack([d] * pts.shape[1], axis=1)
    if append_t is not None:                                    # ray_utils.py:158-159
        pts = np.concatenate([pts, append_t.astype(F32)], -1)
    return pts, dirs, z.astype(F32)


# ----------------------------------------------------------------------------
# SMPL-guided near/far
# ----------------------------------------------------------------------------
def geometry_guided_near_far(orig, direction, vert, geo_threshold=DEFAULT_GEO_THRESH, chunk=512, dtype=F32):
    """reference utils/ray_utils.py:204-219 (torch branch, f32) == :222-233 (numpy branch).

    Union of radius-tau spheres round the vertices: per vertex
    ``z0 = (v-o).d``, ``dz = sqrt(tau^2 - (|v-o|^2 - z0^2))``; NaN -> +-inf;
    near = min(z0-dz), far = max(z0+dz).  A miss gives near=+inf > far=-inf.

    ``dtype``: the arithmetic.  float32 (default) is what the reference computes on the float32 rays its renderers hand it; float64 is the same
    expression on the same values without float32's cancellation error in the bracket (the reference's numpy branch given float64 arrays: pinned on
    tests/golden/arbiter.npz's recorded float64 near / far) -- the yardstick for the device's bounds, whose discriminant is float64 (csrc/nearfar.hip).
    """
    T = np.dtype(dtype).type
    o, d, v = orig.astype(F32).astype(T), direction.astype(F32).astype(T), vert.astype(F32).astype(T)
    tau2 = T(geo_threshold ** 2)
    near = np.empty(o.shape[0], T)
    far = np.empty(o.shape[0], T)
    with np.errstate(invalid='ignore'):
        for s in range(0, o.shape[0], chunk):
            ov = v[None, :, :] - o[s:s + chunk, None, :]                     # [r,V,3]
            z0 = np.einsum('rvi,ri->rv', ov, d[s:s + chunk]).astype(T)
            nrm = np.sqrt(np.sum(ov * ov, axis=2, dtype=T)).astype(T)        # torch.norm
            dz = np.sqrt(tau2 - (nrm * nrm - z0 * z0)).astype(T)
            n_ = z0 - dz
            f_ = z0 + dz
            n_[n_ != n_] = np.inf
            f_[f_ != f_] = -np.inf
            near[s:s + chunk] = n_.min(axis=1)
            far[s:s + chunk] = f_.max(axis=1)
    return near, far
