"""HIP-backed mirror of the reference's utils/ray_utils.py (same function names and argument meaning).

Ray generation (a1) exists twice: the reference-named host functions in float64 numpy exactly like the reference, and
`shot_all_rays_dev` / `shot_rays_dev`, the same chain per ray on the device (what the renderers use); everything per-sample
runs in libneuman_hip.so.  Functions that the reference defines on torch tensors take CUDA tensors here and raise
on CPU tensors -- there is no host fallback.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

DEFAULT_GEO_THRESH = 0.2      # reference utils/constant.py:14
PERTURB_EPSILON = 0.01        # reference utils/constant.py:15


# ------------------------------------------------------------------------------------------------
# a1 ray generation: host, float64 (reference ray_utils.py:13-38 + geometry/pcd_projector.py:85-153)
# ------------------------------------------------------------------------------------------------
def _unproject(xy, intrinsic, c2w):
    """Pixel (x, y) at depth 1 -> world point.  K^-1 [x,y,1], then the 4x4 camera-to-world, all f64."""
    pix = np.concatenate([xy.astype(np.float64), np.ones((xy.shape[0], 1))], axis=1)
    cam = (np.linalg.inv(intrinsic) @ pix.T).T
    world = (c2w @ np.concatenate([cam, np.ones((cam.shape[0], 1))], axis=1).T).T
    return world[:, :3] / world[:, 3:4]


def shot_rays(cap, xys):
    """reference ray_utils.py:23-29: world points are cast to f32 before the (f64) centre is subtracted."""
    c2w = cap.cam_pose.camera_to_world
    centre = cap.cam_pose.camera_center_in_world
    pts = _unproject(xys, cap.intrinsic_matrix, c2w).astype(np.float32)
    orig = np.repeat(centre[None], xys.shape[0], axis=0)
    d = pts - orig
    return orig, d / np.linalg.norm(d, axis=1, keepdims=True)


def shot_ray(cap, x, y):
    """reference ray_utils.py:13-20."""
    o, d = shot_rays(cap, np.array([[x, y]]))
    return o[0], d[0]


def shot_all_rays(cap):
    """reference ray_utils.py:32-38: every pixel, row-major, integer pixel centres, float64 throughout."""
    h, w = cap.size
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing='ij')
    xy = np.stack([xs.reshape(-1), ys.reshape(-1)], axis=1)
    centre = cap.cam_pose.camera_center_in_world
    d = _unproject(xy, cap.intrinsic_matrix, cap.cam_pose.camera_to_world) - centre
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    return np.repeat(centre[None], d.shape[0], axis=0), d


def _cam_matrices(cap):
    kinv = np.ascontiguousarray(np.linalg.inv(np.asarray(cap.intrinsic_matrix, dtype=np.float64)))
    c2w = np.ascontiguousarray(np.asarray(cap.cam_pose.camera_to_world, dtype=np.float64))
    as_p = lambda m: m.ctypes.data_as(ctypes.POINTER(ctypes.c_double))           # noqa: E731
    return kinv, c2w, as_p


def shot_all_rays_dev(cap, device):
    """shot_all_rays (reference ray_utils.py:32-38) on the device: f32 origins / directions [H*W,3], row-major pixels.
    The f64 chain of the reference runs per ray in nm_shot_rays; nothing but 25 doubles crosses the boundary."""
    _lib.require_gpu()
    h, w = cap.size
    kinv, c2w, as_p = _cam_matrices(cap)
    o = torch.empty((h * w, 3), device=device, dtype=torch.float32)
    d = torch.empty_like(o)
    _lib.check(_lib.lib().nm_shot_rays(None, h * w, w, 0, as_p(kinv), as_p(c2w), _lib.dev_ptr(o), _lib.dev_ptr(d), _lib.stream_ptr()),
               "nm_shot_rays")
    return o, d


def shot_rays_dev(cap, xys):
    """shot_rays (reference ray_utils.py:23-29) for a device int32 [N,2] list of (x, y) pixels."""
    _lib.require_gpu()
    xys = xys.to(torch.int32).contiguous()
    kinv, c2w, as_p = _cam_matrices(cap)
    o = torch.empty((xys.shape[0], 3), device=xys.device, dtype=torch.float32)
    d = torch.empty_like(o)
    # numpy's result type follows the pose matrix: the reference's CameraPose yields float32 (f32 subtraction and norm)
    mode = 2 if np.asarray(cap.cam_pose.camera_to_world).dtype == np.float32 else 1
    _lib.check(_lib.lib().nm_shot_rays(_lib.dev_ptr(xys, torch.int32), xys.shape[0], cap.size[1], mode, as_p(kinv), as_p(c2w),
                                       _lib.dev_ptr(o), _lib.dev_ptr(d), _lib.stream_ptr()), "nm_shot_rays")
    return o, d


def to_homogeneous(pts):
    """reference ray_utils.py:41-45."""
    if isinstance(pts, torch.Tensor):
        return torch.cat([pts, torch.ones_like(pts[..., 0:1])], dim=-1)
    return np.concatenate([pts, np.ones_like(pts[..., 0:1])], axis=-1)


# ------------------------------------------------------------------------------------------------
# a4 / a6 / a7 sampling
# ------------------------------------------------------------------------------------------------
def _f32c(t, device=None):
    t = t.to(device=device, dtype=torch.float32) if device is not None else t.to(torch.float32)
    return t.contiguous()


def _batch_f32c(t):
    """A ray batch as the reference's DataLoader hands it to a trainer lives on the HOST (datasets/*.py; vanilla_nerf_trainer.py:51-60 samples first
    and moves the samples to the device afterwards): the two sampling entry points upload it.  An upload, not a host evaluation -- without a HIP
    device this raises like everything else."""
    _lib.require_gpu()
    if isinstance(t, np.ndarray):
        t = torch.from_numpy(t)
    if not t.is_cuda:
        t = t.to(torch.device('cuda', torch.cuda.current_device()))
    return _f32c(t)


def sample_z(origin, direction, near, far, samples_per_ray, lindisp=False, perturb=0., want_points=False):
    """Core of ray_to_samples on device tensors.  Returns (pts|None, dirs|None, z)."""
    _lib.require_gpu()
    dev = origin.device
    R = origin.shape[0]
    t_vals = torch.linspace(0., 1., steps=samples_per_ray, device=dev)                  # ray_utils.py:111
    t_rand = None
    if perturb > 0.:                                                                     # ray_utils.py:123-127
        t_rand = torch.clip(torch.rand((R, samples_per_ray), device=dev), min=PERTURB_EPSILON, max=1 - PERTURB_EPSILON)
    z = torch.empty((R, samples_per_ray), device=dev, dtype=torch.float32)
    pts = torch.empty((R, samples_per_ray, 3), device=dev, dtype=torch.float32) if want_points else None
    dirs = torch.empty((R, samples_per_ray, 3), device=dev, dtype=torch.float32) if want_points else None
    _lib.check(_lib.lib().nm_ray_to_samples(
        _lib.dev_ptr(origin, name='origin'), _lib.dev_ptr(direction, name='direction'), _lib.dev_ptr(near, name='near'),
        _lib.dev_ptr(far, name='far'), R, samples_per_ray, _lib.dev_ptr(t_vals), int(bool(lindisp)), _lib.dev_ptr(t_rand),
        _lib.dev_ptr(pts), _lib.dev_ptr(dirs), _lib.dev_ptr(z), _lib.stream_ptr()), "nm_ray_to_samples")
    return pts, dirs, z


def ray_to_samples(ray_batch, samples_per_ray, lindisp=False, perturb=0., device='cpu', append_t=None):
    """reference ray_utils.py:96-135.  The batch may live on the host (the reference's trainers sample before they move anything to the device):
    it is uploaded; the samples come back on the HIP device whatever `device` says (the callers' `.to(device)` is then a no-op)."""
    o, d = _batch_f32c(ray_batch['origin']), _batch_f32c(ray_batch['direction'])
    near, far = _batch_f32c(ray_batch['near']).reshape(-1), _batch_f32c(ray_batch['far']).reshape(-1)
    assert near.shape[0] == far.shape[0] == o.shape[0]
    pts, dirs, z = sample_z(o, d, near, far, samples_per_ray, lindisp, perturb, want_points=True)
    if append_t is not None:
        pts = torch.cat([pts, append_t.to(pts.device)], dim=-1)
    return pts, dirs, z


def z_to_points(origin, direction, z_vals):
    """pts = o + d*z, dirs = d repeated (ray_utils.py:153-155) for given z."""
    _lib.require_gpu()
    R, S = z_vals.shape
    pts = torch.empty((R, S, 3), device=z_vals.device, dtype=torch.float32)
    dirs = torch.empty((R, S, 3), device=z_vals.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_z_to_points(_lib.dev_ptr(origin), _lib.dev_ptr(direction), _lib.dev_ptr(z_vals), R, S,
                                         _lib.dev_ptr(pts), _lib.dev_ptr(dirs), _lib.stream_ptr()), "nm_z_to_points")
    return pts, dirs


def sample_pdf(bins, weights, N_samples, det=False, device='cpu'):
    """reference ray_utils.py:164-194.  Only det=True (the reference hard-codes it at :149) is implemented."""
    if not det:
        raise NotImplementedError("sample_pdf(det=False) is never used by the reference (ray_utils.py:149)")
    _lib.require_gpu()
    bins, weights = _f32c(bins), _f32c(weights)
    R, B = bins.shape
    assert weights.shape == (R, B - 1)
    u = torch.linspace(0., 1., steps=N_samples, device=bins.device)                     # ray_utils.py:173
    out = torch.empty((R, N_samples), device=bins.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_sample_pdf(_lib.dev_ptr(bins), _lib.dev_ptr(weights), R, B, _lib.dev_ptr(u), N_samples,
                                        _lib.dev_ptr(out), _lib.stream_ptr()), "nm_sample_pdf")
    return out


def importance_z(z_vals, weights, importance_samples_per_ray, including_old=True):
    """z-only core of ray_to_importance_samples: mid-points, weights[1:-1], inverse CDF, sorted merge (one kernel)."""
    _lib.require_gpu()
    z_vals, weights = _f32c(z_vals), _f32c(weights.detach())
    R, S = z_vals.shape
    n_out = S + importance_samples_per_ray if including_old else importance_samples_per_ray
    u = torch.linspace(0., 1., steps=importance_samples_per_ray, device=z_vals.device)
    out = torch.empty((R, n_out), device=z_vals.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_importance_z(_lib.dev_ptr(z_vals), _lib.dev_ptr(weights), R, S, _lib.dev_ptr(u),
                                          importance_samples_per_ray, int(bool(including_old)), _lib.dev_ptr(out),
                                          _lib.stream_ptr()), "nm_importance_z")
    return out


def importance_z_from_raw(raw, z_vals, rays_d, importance_samples_per_ray, want_weights=False):
    """The coarse tail of a two-pass render as ONE kernel (reference render_utils.py:139-147: raw2outputs' weights -> sample_pdf ->
    sort(cat)): raw [R,S,4] of the coarse pass -> z [R, S + N] (and the weights [R,S] when asked for).  Bit-identical to raw2outputs +
    importance_z."""
    _lib.require_gpu()
    raw, z_vals, rays_d = _f32c(raw.detach()), _f32c(z_vals), _f32c(rays_d)
    R, S = z_vals.shape
    N = int(importance_samples_per_ray)
    u = torch.linspace(0., 1., steps=N, device=z_vals.device)
    out = torch.empty((R, S + N), device=z_vals.device, dtype=torch.float32)
    w = torch.empty((R, S), device=z_vals.device, dtype=torch.float32) if want_weights else None
    _lib.check(_lib.lib().nm_importance_from_raw(_lib.dev_ptr(raw), _lib.dev_ptr(z_vals), _lib.dev_ptr(rays_d), R, S, _lib.dev_ptr(u), N, _lib.dev_ptr(out),
                                                 _lib.dev_ptr(w), _lib.stream_ptr()), "nm_importance_from_raw")
    return out, w


def ray_to_importance_samples(ray_batch, z_vals, weights, importance_samples_per_ray, device='cpu', including_old=True,
                              append_t=None):
    """reference ray_utils.py:138-160 (a host batch is uploaded, as in ray_to_samples)."""
    o, d = _batch_f32c(ray_batch['origin']), _batch_f32c(ray_batch['direction'])
    z = importance_z(_batch_f32c(z_vals), _batch_f32c(weights.detach()), importance_samples_per_ray, including_old)
    pts, dirs = z_to_points(o, d, z)
    if append_t is not None:
        pts = torch.cat([pts, append_t.to(pts.device)], dim=-1)
    return pts, dirs, z


# ------------------------------------------------------------------------------------------------
# a2 / a3 SMPL-guided bounds and hit compaction
# ------------------------------------------------------------------------------------------------
def _near_far_dev(orig, direction, vert, geo_threshold):
    R = orig.shape[0]
    near = torch.empty(R, device=orig.device, dtype=torch.float32)
    far = torch.empty(R, device=orig.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_near_far(_lib.dev_ptr(orig, name='orig'), _lib.dev_ptr(direction, name='dir'), R,
                                      _lib.dev_ptr(vert, name='vert'), vert.shape[0], float(geo_threshold),
                                      _lib.dev_ptr(near), _lib.dev_ptr(far), _lib.stream_ptr()), "nm_near_far")
    return near, far


def geometry_guided_near_far(orig, dir, vert, geo_threshold):
    """reference ray_utils.py:197-233.  Dispatches on the type of `orig` like the reference: CUDA tensors in ->
    tensors out; numpy in -> numpy out (the arrays make one round trip to the device, the arithmetic is the kernel's)."""
    _lib.require_gpu()
    if isinstance(orig, torch.Tensor):
        return _near_far_dev(_f32c(orig), _f32c(dir), _f32c(vert if isinstance(vert, torch.Tensor) else torch.from_numpy(vert), orig.device),
                             geo_threshold)
    dev = torch.device('cuda')
    n, f = _near_far_dev(_f32c(torch.from_numpy(np.ascontiguousarray(orig)), dev), _f32c(torch.from_numpy(np.ascontiguousarray(dir)), dev),
                         _f32c(torch.from_numpy(np.ascontiguousarray(vert)), dev), geo_threshold)
    return n.cpu().numpy(), f.cpu().numpy()


def geometry_guided_near_far_torch(orig, dir, vert, geo_threshold=DEFAULT_GEO_THRESH):
    """reference ray_utils.py:204-219 (the variant with a default threshold)."""
    return geometry_guided_near_far(orig, dir, vert, geo_threshold)


def geometry_guided_near_far_np(orig, dir, vert, geo_threshold=DEFAULT_GEO_THRESH):
    """reference ray_utils.py:222-233."""
    return geometry_guided_near_far(orig, dir, vert, geo_threshold)


def compact_hits(near, far):
    """Indices of rays with near < far (ascending) and of the others: the boolean masks of
    render_utils.py:199-212 as int32 index lists.  One host sync to read the two counts."""
    _lib.require_gpu()
    R = near.shape[0]
    dev = near.device
    hit = torch.empty(R, device=dev, dtype=torch.int32)
    miss = torch.empty(R, device=dev, dtype=torch.int32)
    counts = torch.zeros(2, device=dev, dtype=torch.int32)
    ws = torch.empty(int(_lib.lib().nm_compact_workspace_ints(R)), device=dev, dtype=torch.int32)
    _lib.check(_lib.lib().nm_compact_hits(_lib.dev_ptr(near), _lib.dev_ptr(far), R, _lib.dev_ptr(hit, torch.int32),
                                          _lib.dev_ptr(miss, torch.int32), _lib.dev_ptr(counts, torch.int32),
                                          _lib.dev_ptr(ws, torch.int32), _lib.stream_ptr()), "nm_compact_hits")
    n_hit, n_miss = counts.tolist()
    return hit[:n_hit], miss[:n_miss]


def gather_rows(src, idx):
    """src[idx] for a [N, W] (or [N]) f32 tensor and an int32 index list."""
    _lib.require_gpu()
    flat = src.reshape(src.shape[0], -1).contiguous()
    out = torch.empty((idx.shape[0], flat.shape[1]), device=src.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_gather_rows(_lib.dev_ptr(flat), _lib.dev_ptr(idx.contiguous(), torch.int32), None, idx.shape[0],
                                         flat.shape[1], _lib.dev_ptr(out), _lib.stream_ptr()), "nm_gather_rows")
    return out.reshape(idx.shape[0], *src.shape[1:])


def scatter_rows(dst, idx, src):
    """dst[idx] = src (in place) for f32 row tensors."""
    _lib.require_gpu()
    assert dst.is_contiguous()
    flat = src.reshape(src.shape[0], -1).contiguous()
    _lib.check(_lib.lib().nm_scatter_rows(_lib.dev_ptr(flat), _lib.dev_ptr(idx.contiguous(), torch.int32), None, idx.shape[0],
                                          flat.shape[1], _lib.dev_ptr(dst), _lib.stream_ptr()), "nm_scatter_rows")
    return dst


# ------------------------------------------------------------------------------------------------
# a11 observation -> canonical warp
# ------------------------------------------------------------------------------------------------
class Mesh:
    """A posed mesh on the device plus its closest-point search structure (nm_mesh_create).  Built once per frame and
    actor; `T` are the per-vertex canonical->observation transforms (f64 [>=V,4,4], joint rows beyond V never indexed)."""

    SEARCH = {'tree': 0, 'all': 1, 'tree_wide': 2}      # NM_SEARCH_TREE / NM_SEARCH_ALL (the all-triangles loop: tests and diagnostics)

    def __init__(self, verts, faces, T, device, search='tree'):
        """T = None: a mesh for closest-point / signed-distance queries only (no per-vertex transforms: nm_warp_to_canonical is not for it)"""
        import ctypes
        _lib.require_gpu()
        v = verts if isinstance(verts, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float32))
        f = faces[:, :3] if isinstance(faces, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(np.asarray(faces)[:, :3], dtype=np.int32))
        self.verts = v.to(device, torch.float32).contiguous()
        self.faces = f.to(device, torch.int32).contiguous()          # cols 3-5 of scene.faces are UV ids (utils/utils.py:213-221)
        if T is None:
            self.T = None
        else:
            t = T if isinstance(T, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(T, dtype=np.float64))
            self.T = t.to(device, torch.float64).reshape(-1, 16).contiguous()
        # the kernels index verts and T by face entries without bounds checks: refuse UV ids, 1-based or out-of-range indices
        # and a transform table shorter than the vertex list here, once per mesh (the check is a reduction on the device;
        # nm_mesh_create syncs for the bounding box anyway)
        V = self.verts.shape[0]
        if self.faces.numel() and (int(self.faces.min()) < 0 or int(self.faces.max()) >= V):
            raise _lib.NeumanHipError(f"Mesh: face indices must lie in [0, {V}) (got {int(self.faces.min())}..{int(self.faces.max())}): "
                                      "pass the vertex-id columns of a 0-based face array")
        if self.T is not None and self.T.shape[0] < V:
            raise _lib.NeumanHipError(f"Mesh: T has {self.T.shape[0]} rows for {V} vertices (one 4x4 per vertex is required)")
        self.handle = ctypes.c_void_p()
        _lib.check(_lib.lib().nm_mesh_create(_lib.dev_ptr(self.verts), self.verts.shape[0], _lib.dev_ptr(self.faces, torch.int32),
                                             self.faces.shape[0], self.SEARCH[search], ctypes.byref(self.handle), _lib.stream_ptr()),
                   "nm_mesh_create")

    def update(self, verts):
        """the same faces with moved vertices ([V,3] on the mesh's device): the tree is rebuilt inside the handle (nm_mesh_update) -- no allocation, no
        question to the host.  What a training iteration does with the posed body after every optimiser step."""
        v = verts.detach().to(self.verts.device, torch.float32).contiguous()
        if v.shape != self.verts.shape:
            raise _lib.NeumanHipError(f"Mesh.update: {tuple(v.shape)} vertices for a mesh of {tuple(self.verts.shape)}")
        self.verts = v
        _lib.check(_lib.lib().nm_mesh_update(self.handle, _lib.dev_ptr(v), _lib.stream_ptr()), "nm_mesh_update")
        return self

    def info(self):
        import ctypes
        levels = ctypes.c_int32()
        n = ctypes.c_int64()
        b = ctypes.c_int64()
        _lib.check(_lib.lib().nm_mesh_info(self.handle, ctypes.byref(levels), ctypes.byref(n), ctypes.byref(b)), "nm_mesh_info")
        return {"levels": levels.value, "nodes": n.value, "bytes": b.value}

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.lib().nm_mesh_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def mesh_to_device(verts, faces, T, device, search='tree'):
    return Mesh(verts, faces, T, device, search)


def warp_to_canonical_dev(pts, mesh, want_closest=False):
    """Device-tensor core: pts [R,S,3] f32 CUDA, mesh = Mesh -> can_pts, can_dirs (, closest) [R,S,3] f32."""
    _lib.require_gpu()
    R, S, _ = pts.shape
    dev = pts.device
    can_pts = torch.empty((R, S, 3), device=dev, dtype=torch.float32)
    can_dirs = torch.empty((R, S, 3), device=dev, dtype=torch.float32)
    closest = torch.empty((R, S, 3), device=dev, dtype=torch.float32) if want_closest else None
    _lib.check(_lib.lib().nm_warp_to_canonical(mesh.handle, _lib.dev_ptr(pts, name='pts'), R, S,
                                               _lib.dev_ptr(mesh.T, torch.float64, 'T'), _lib.dev_ptr(can_pts),
                                               _lib.dev_ptr(can_dirs), _lib.dev_ptr(closest), _lib.stream_ptr()),
               "nm_warp_to_canonical")
    return can_pts, can_dirs, closest


def warp_samples_to_canonical(pts, verts, faces, T):
    """reference ray_utils.py:48-66.  numpy in -> numpy out like the reference (float32 here: the reference's callers
    cast to float32 right away, render_utils.py:226-227); CUDA tensors in -> CUDA tensors out."""
    assert len(pts.shape) == 3, 'pts should have shape [num_rays, num_samples, 3]'
    assert pts.shape[-1] == 3
    as_numpy = not isinstance(pts, torch.Tensor)
    dev = torch.device('cuda') if as_numpy else pts.device
    p = torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32)).to(dev) if as_numpy else _f32c(pts)
    can_pts, can_dirs, closest = warp_to_canonical_dev(p, Mesh(verts, faces, T, dev), want_closest=True)
    if as_numpy:
        return can_pts.cpu().numpy(), can_dirs.cpu().numpy(), closest.cpu().numpy()
    return can_pts, can_dirs, closest


# ------------------------------------------------------------------------------------------------
# signed distance + the differentiable warp of the human trainer
# ------------------------------------------------------------------------------------------------
def signed_distance_dev(pts, mesh):
    """pts [N,3] f32 CUDA, mesh = Mesh -> (signed distance [N] f32, face id [N] int32, closest point [N,3] f32), CUDA tensors."""
    _lib.require_gpu()
    p = _f32c(pts.reshape(-1, 3))
    N = p.shape[0]
    s = torch.empty(N, device=p.device, dtype=torch.float32)
    f = torch.empty(N, device=p.device, dtype=torch.int32)
    c = torch.empty((N, 3), device=p.device, dtype=torch.float32)
    _lib.check(_lib.lib().nm_signed_distance(mesh.handle, _lib.dev_ptr(p), N, _lib.dev_ptr(s), _lib.dev_ptr(f, torch.int32), _lib.dev_ptr(c),
                                             _lib.stream_ptr()), "nm_signed_distance")
    return s, f, c


def signed_distance(pts, verts, faces):
    """igl.signed_distance(P, V, F) as the reference calls it (ray_utils.py:70, human_nerf_trainer.py:310): numpy in ->
    (S [N], I [N], C [N,3]) numpy out, pseudonormal sign (negative inside)."""
    dev = torch.device('cuda')
    v = torch.as_tensor(np.ascontiguousarray(verts, dtype=np.float32))
    T = torch.zeros((v.shape[0], 16), dtype=torch.float64)                                  # the search needs no transforms
    s, f, c = signed_distance_dev(torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32)).to(dev), Mesh(v, faces, T, dev))
    return s.cpu().numpy(), f.cpu().numpy(), c.cpu().numpy()


class _WarpApplyFn(torch.autograd.Function):
    """can = inv(sum_k bary_k T[tri_k]) [p; 1] per sample (csrc/warp.hip nm_warp_apply_forward / _backward): one kernel each way for the
    blend, the 4x4 inverse and the product that the reference leaves to autograd (ray_utils.py:85-93, human_nerf_trainer.py:262-266)."""

    @staticmethod
    def forward(ctx, T, bary, tri, pts):
        Tc, bc = T.detach().reshape(-1, 16).contiguous().float(), bary.detach().contiguous().float()
        N = pts.shape[0]
        can = torch.empty((N, 3), device=pts.device, dtype=torch.float32)
        _lib.check(_lib.lib().nm_warp_apply_forward(_lib.dev_ptr(Tc), _lib.dev_ptr(tri, torch.int32), _lib.dev_ptr(bc), _lib.dev_ptr(pts), N,
                                                    _lib.dev_ptr(can), _lib.stream_ptr()), "nm_warp_apply_forward")
        ctx.save_for_backward(Tc, bc, tri, pts)
        ctx.T_shape = T.shape
        return can

    @staticmethod
    def backward(ctx, g_can):
        Tc, bc, tri, pts = ctx.saved_tensors
        N, V = pts.shape[0], Tc.shape[0]
        g_T = torch.empty_like(Tc)
        g_b = torch.empty_like(bc)
        _lib.check(_lib.lib().nm_warp_apply_backward(_lib.dev_ptr(Tc), _lib.dev_ptr(tri, torch.int32), _lib.dev_ptr(bc), _lib.dev_ptr(pts),
                                                     _lib.dev_ptr(g_can.contiguous().float()), N, V, _lib.dev_ptr(g_T), _lib.dev_ptr(g_b),
                                                     _lib.stream_ptr()), "nm_warp_apply_backward")
        return g_T.reshape(ctx.T_shape), g_b, None, None


class _BaryFn(torch.autograd.Function):
    """barycentric coordinates of the (constant) closest points in their triangles, reference ray_utils.py:72-84, with the adjoint to the
    vertices: one kernel each way (csrc/warp.hip nm_bary_forward / _backward) instead of ~20 + ~40 elementwise launches and a sorted
    index_put for the gather's backward"""

    @staticmethod
    def forward(ctx, verts, tri, closest):
        vc = verts.detach().contiguous().float()
        N = tri.shape[0]
        bary = torch.empty((N, 3), device=vc.device, dtype=torch.float32)
        _lib.check(_lib.lib().nm_bary_forward(_lib.dev_ptr(vc), _lib.dev_ptr(tri, torch.int32), _lib.dev_ptr(closest), N, _lib.dev_ptr(bary), _lib.stream_ptr()),
                   "nm_bary_forward")
        ctx.save_for_backward(vc, tri, closest)
        return bary

    @staticmethod
    def backward(ctx, g_bary):
        vc, tri, closest = ctx.saved_tensors
        g_v = torch.empty_like(vc)
        _lib.check(_lib.lib().nm_bary_backward(_lib.dev_ptr(vc), _lib.dev_ptr(tri, torch.int32), _lib.dev_ptr(closest), _lib.dev_ptr(g_bary.contiguous().float()),
                                               tri.shape[0], vc.shape[0], _lib.dev_ptr(g_v), _lib.stream_ptr()), "nm_bary_backward")
        return g_v, None, None


BARY_KERNELS = os.environ.get('NEUMAN_BARY_KERNELS', '1') != '0'       # 0: the reference's torch lines under autograd (the check of the kernels)


# The differentiable warp is called once per training iteration with the SAME face array and moved vertices: the device copies of the faces and the
# search handle are kept (keyed by the face array object) and the handle is updated in place -- building a Mesh costs eight allocations, a
# host-to-device copy of the faces, three read-backs and as many frees, every one of which stalls the queue the iteration's kernels wait in.
_DIFF_CACHE = {}
_DIFF_CACHE_MAX = 4


def _diff_faces(faces, dev):
    """-> (faces int64 [F,3] on dev, faces int32 [F,3] on dev, cache entry) for a numpy / tensor face array, by object identity"""
    key = (id(faces), str(dev))
    e = _DIFF_CACHE.get(key)
    if e is None or e['src'] is not faces:
        f64 = torch.as_tensor(np.ascontiguousarray(np.asarray(faces.cpu() if isinstance(faces, torch.Tensor) else faces)[:, :3]).astype(np.int64)).to(dev)
        while len(_DIFF_CACHE) >= _DIFF_CACHE_MAX:
            _DIFF_CACHE.pop(next(iter(_DIFF_CACHE)))
        e = _DIFF_CACHE[key] = {'src': faces, 'f64': f64, 'f32': f64.to(torch.int32).contiguous(), 'mesh': None}
    return e['f64'], e['f32'], e


def _diff_mesh(e, verts):
    """the cache entry's search handle on the vertices `verts` (built on first use, updated in place afterwards)"""
    m = e['mesh']
    if m is None or m.verts.shape != verts.shape or m.verts.device != verts.device:
        m = e['mesh'] = Mesh(verts.detach(), e['f32'], None, verts.device)
    else:
        m.update(verts)
    return m


def _closest_barycentric(p, verts, f3, mesh=None):
    """the closest-point query (libneuman_hip) and the reference's differentiable barycentric lines (ray_utils.py:70-84) ->
    (barycentric [N,3] with autograd to `verts`, face ids [N] long, signed distance [N])"""
    mesh = mesh or Mesh(verts.detach(), f3.to(torch.int32), None, verts.device)
    signed_dist, f_id, closest = signed_distance_dev(p, mesh)
    f_id = f_id.long()
    if BARY_KERNELS and verts.dtype == torch.float32:
        return _BaryFn.apply(verts, f3[f_id].to(torch.int32).contiguous(), closest.contiguous()), f_id, signed_dist
    return _barycentric_torch(verts[f3[f_id]], closest), f_id, signed_dist


def _barycentric_torch(corners, p):
    """Barycentric coordinates of points p [N,3] in triangles corners [N,3,3] as ratios of signed areas: the sub-triangle opposite corner k
    (spanned from corner k+1 to corner k+2 and to p), projected on the triangle's normal, over the triangle's own -- the quantity
    ray_utils.py:73-84 computes; plain torch under autograd (the check of nm_bary_forward / _backward, and the path for non-float32 vertices)."""
    nxt, prv = corners.roll(-1, 1), corners.roll(-2, 1)
    normal = torch.linalg.cross(corners[:, 1] - corners[:, 0], corners[:, 2] - corners[:, 0])
    sub = torch.linalg.cross(prv - nxt, p[:, None, :] - nxt)
    uv = (sub[:, :2] * normal[:, None, :]).sum(-1) / (normal * normal).sum(-1, keepdim=True)
    return torch.cat([uv, 1 - uv.sum(1, keepdim=True)], 1)


def warp_points_to_canonical_diff(pts, verts, faces, T):
    """What the human trainer makes of warp_samples_to_canonical_diff (human_nerf_trainer.py:262-266: the inverse blended transforms
    applied to the very points they were found for), fused: pts [N,3] (detached), verts [V,3] / T [V,4,4] CUDA tensors that may
    require grad -> (canonical points [N,3] with autograd to T and, through the barycentric coordinates, to verts; f_id; signed_dist)."""
    dev = verts.device
    f3, _, entry = _diff_faces(faces, dev)
    p = pts.detach().to(dev, torch.float32).contiguous()
    bary, f_id, signed_dist = _closest_barycentric(p, verts, f3, _diff_mesh(entry, verts))
    tri = f3[f_id].to(torch.int32).contiguous()
    return _WarpApplyFn.apply(T.to(dev), bary, tri, p), f_id, signed_dist


def warp_samples_to_canonical_diff(pts, verts, faces, T):
    """reference ray_utils.py:69-93: pts [N,3] numpy (detached), verts [V,3] / T [V,4,4] torch tensors that may require grad
    -> (T_interp_inv [N,4,4], f_id, signed_dist).  The closest-point query, its sign and the barycentric coordinates (with their adjoint to
    the vertices) run in libneuman_hip (_closest_barycentric); the blend of the three corner transforms and the 4x4 inverse are
    differentiable torch operations on the device (callers that only apply the inverse to the points: warp_points_to_canonical_diff)."""
    dev = verts.device if isinstance(verts, torch.Tensor) and verts.is_cuda else torch.device('cuda')
    verts, T = verts.to(dev), T.to(dev)
    f3, _, entry = _diff_faces(faces, dev)
    p = pts.detach().to(dev, torch.float32) if isinstance(pts, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(pts, dtype=np.float32)).to(dev)
    bary, f_id, signed_dist = _closest_barycentric(p.contiguous(), verts, f3, _diff_mesh(entry, verts))
    blended = torch.einsum('nk,nkij->nij', bary.to(T.dtype), T[f3[f_id]])
    return torch.linalg.inv(blended), f_id.cpu().numpy(), signed_dist.cpu().numpy()
