"""Oracle: whole-frame renderers.

numpy restatement of reference utils/render_utils.py:108-461.  A "net" here is a
pair ``(weights, JoinerSpec)`` (oracle/nerf_mlp.py); ``cap`` is the reference
duck-type (``.shape``, ``.intrinsic_matrix``, ``.cam_pose.camera_to_world``,
``.near['bkg']``, ``.far['bkg']``).  Test infrastructure only.
"""
import numpy as np

from . import ray_ops, warp
from .compositing import raw2outputs, merge_sorted
from .nerf_mlp import joiner_forward

F32 = np.float32


def _net(net, pts, dirs):
    return joiner_forward(net[0], net[1], pts, dirs)


def render_vanilla(coarse_net, cap, fine_net=None, rays_per_batch=32768, samples_per_ray=64,
                   importance_samples_per_ray=128, white_bkg=True, near_far_source='bkg', return_depth=False,
                   max_rays=None, cdf_ulps=0, ablate_nerft=False):
    """reference utils/render_utils.py:108-161.  ``max_rays`` renders only a prefix of the frame (bounded CPU baseline);
    ``cdf_ulps`` is the conditioning probe of ray_ops.sample_pdf; ``ablate_nerft`` appends the frame time
    cap.frame_id['frame_id'] / cap.frame_id['total_frames'] to every sample point (:134-148)."""
    origins, dirs = ray_ops.shot_all_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, cap.shape)
    total = origins.shape[0] if max_rays is None else min(max_rays, origins.shape[0])
    rgbs, depths = [], []
    for i in range(0, total, rays_per_batch):
        j = min(i + rays_per_batch, total)
        o, d = origins[i:j].astype(F32), dirs[i:j].astype(F32)
        near = np.full((j - i, 1), cap.near[near_far_source], F32)
        far = np.full((j - i, 1), cap.far[near_far_source], F32)
        ct = ft = None
        if ablate_nerft:
            cur_time = cap.frame_id['frame_id'] / cap.frame_id['total_frames']
            ct = np.ones((j - i, samples_per_ray, 1), F32) * F32(cur_time)
            ft = np.ones((j - i, samples_per_ray + importance_samples_per_ray, 1), F32) * F32(cur_time)
        pts, dd, z = ray_ops.ray_to_samples(o, d, near, far, samples_per_ray, append_t=ct)
        out = _net(coarse_net, pts, dd)
        rgb, _, _, w, depth = raw2outputs(out, z, dd[:, 0, :], white_bkg=white_bkg)
        if fine_net is not None:
            pts, dd, z = ray_ops.ray_to_importance_samples(o, d, z, w, importance_samples_per_ray, cdf_ulps=cdf_ulps, append_t=ft)
            out = _net(fine_net, pts, dd)
            rgb, _, _, w, depth = raw2outputs(out, z, dd[:, 0, :], white_bkg=white_bkg)
        rgbs.append(rgb)
        depths.append(depth)
    rgb = np.concatenate(rgbs)
    depth = np.concatenate(depths)
    if max_rays is None:
        rgb, depth = rgb.reshape(*cap.shape, -1), depth.reshape(*cap.shape)
    return (rgb, depth) if return_depth else rgb


def _frame_rays(cap, ray_range=None):
    if getattr(cap, '_rays', None) is not None:                   # the reference's own recorded rays of this frame (tests/helpers/posed_scene.py): identical inputs
        o, d = cap._rays
        return (o, d) if ray_range is None else (o[ray_range[0]:ray_range[1]], d[ray_range[0]:ray_range[1]])
    coords = ray_ops.all_pixel_coords(cap.shape)
    if ray_range is not None:
        coords = coords[ray_range[0]:ray_range[1]]
    o, d = ray_ops.shot_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, coords)
    return o, d


def render_smpl_nerf(net, cap, posed_verts, faces, Ts, rays_per_batch=32768, samples_per_ray=64, white_bkg=True,
                     render_can=False, geo_threshold=ray_ops.DEFAULT_GEO_THRESH, return_depth=False,
                     return_mask=False, interval_comp=1.0, ray_range=None, given=None):
    """reference utils/render_utils.py:164-246.  ``net`` = coarse_human_net pair.  ``ray_range=(a, b)`` renders rays [a, b) of the
    frame only and returns flat arrays (bounded CPU tests; rays are independent).  ``given`` replays recorded intermediates of the
    float32-ill-conditioned steps instead of recomputing them -- {'near_far': [(near [R], far [R]) per actor], 'bkg_z': [R, S']},
    indexed by frame ray -- which is how parity CONDITIONAL on those steps is stated (tests/golden/make_golden_posed.py)."""
    origins, dirs = _frame_rays(cap, ray_range)
    r0 = 0 if ray_range is None else ray_range[0]
    total = origins.shape[0]
    rgbs, depths, accs = [], [], []
    for i in range(0, total, rays_per_batch):
        o, d = origins[i:i + rays_per_batch].astype(F32), dirs[i:i + rays_per_batch].astype(F32)
        rgb = np.zeros_like(o)
        depth = np.zeros(o.shape[0], F32)
        acc = np.zeros(o.shape[0], F32)
        near, far = _near_far(given, 0, r0 + i, o, d, posed_verts, geo_threshold)
        miss, hit = near >= far, near < far
        if miss.any():
            rgb[miss] = 1.0 if white_bkg else 0.0
        if hit.any():
            pts, dd, z = ray_ops.ray_to_samples(o[hit], d[hit], near[hit][:, None], far[hit][:, None], samples_per_ray)
            if render_can:
                can_pts, can_dirs = pts, dd
            else:
                can_pts, can_dirs, _ = warp.warp_samples_to_canonical(pts, posed_verts, faces, Ts)
            out = _net(net, can_pts.astype(F32), can_dirs.astype(F32)).copy()
            out[..., -1] *= F32(interval_comp)
            _rgb, _, _acc, _, _depth = raw2outputs(out, z, dd[:, 0, :], white_bkg=white_bkg)
            rgb[hit], depth[hit], acc[hit] = _rgb, _depth, _acc
        rgbs.append(rgb)
        depths.append(depth)
        accs.append(acc)
    rgb, depth, acc = np.concatenate(rgbs), np.concatenate(depths), np.concatenate(accs)
    if ray_range is None:
        rgb, depth, acc = rgb.reshape(*cap.shape, -1), depth.reshape(*cap.shape), acc.reshape(*cap.shape)
    if return_depth and return_mask:
        return rgb, depth, acc
    if return_depth:
        return rgb, depth
    if return_mask:
        return rgb, acc
    return rgb


def _near_far(given, actor, start, o, d, verts, geo_threshold):
    if given is not None and 'near_far' in given:
        n, f = given['near_far'][actor]
        return n[start:start + o.shape[0]].astype(F32), f[start:start + o.shape[0]].astype(F32)
    return ray_ops.geometry_guided_near_far(o, d, verts, geo_threshold)


def _bkg_pass(coarse, fine, o, d, near_v, far_v, samples_per_ray, n_importance, white_bkg, given_z=None):
    if given_z is not None:                                       # the recorded positions: the fine network on them (ray_utils.py:153-156)
        z = given_z.astype(F32)
        pts = (o[:, None, :] + d[:, None, :] * z[..., None]).astype(F32)
        return _net(fine, pts, np.broadcast_to(d[:, None, :], pts.shape)), z
    near = np.full((o.shape[0], 1), near_v, F32)
    far = np.full((o.shape[0], 1), far_v, F32)
    pts, dd, z = ray_ops.ray_to_samples(o, d, near, far, samples_per_ray)
    out = _net(coarse, pts, dd)
    if fine is not None:
        _, _, _, w, _ = raw2outputs(out, z, dd[:, 0, :], white_bkg=white_bkg)
        pts, dd, z = ray_ops.ray_to_importance_samples(o, d, z, w, n_importance)
        out = _net(fine, pts, dd)
    return out, z


def render_hybrid_nerf(coarse_bkg, fine_bkg, human, cap, posed_verts, faces, Ts, rays_per_batch=32768,
                       samples_per_ray=64, importance_samples_per_ray=128, white_bkg=True,
                       geo_threshold=ray_ops.DEFAULT_GEO_THRESH, return_depth=False, ray_range=None, bkg_z_out=None, given=None):
    """reference utils/render_utils.py:249-362.  ``ray_range`` / ``given`` as in render_smpl_nerf; ``bkg_z_out`` (a list) receives the
    background pass's final sample positions per batch."""
    origins, dirs = _frame_rays(cap, ray_range)
    r0 = 0 if ray_range is None else ray_range[0]
    total = origins.shape[0]
    rgbs, depths = [], []
    for i in range(0, total, rays_per_batch):
        o, d = origins[i:i + rays_per_batch].astype(F32), dirs[i:i + rays_per_batch].astype(F32)
        rgb = np.zeros_like(o)
        depth = np.zeros(o.shape[0], F32)
        bkg_out, bkg_z = _bkg_pass(coarse_bkg, fine_bkg, o, d, cap.near['bkg'], cap.far['bkg'],
                                   samples_per_ray, importance_samples_per_ray, white_bkg,
                                   given['bkg_z'][r0 + i:r0 + i + o.shape[0]] if given is not None and 'bkg_z' in given else None)
        if bkg_z_out is not None:
            bkg_z_out.append(bkg_z)
        near, far = _near_far(given, 0, r0 + i, o, d, posed_verts, geo_threshold)
        miss, hit = near >= far, near < far
        if miss.any():
            _rgb, _, _, _, _depth = raw2outputs(bkg_out[miss], bkg_z[miss], d[miss], white_bkg=white_bkg)
            rgb[miss], depth[miss] = _rgb, _depth
        if hit.any():
            pts, dd, hz = ray_ops.ray_to_samples(o[hit], d[hit], near[hit][:, None], far[hit][:, None], samples_per_ray)
            can_pts, can_dirs, _ = warp.warp_samples_to_canonical(pts, posed_verts, faces, Ts)
            h_out = _net(human, can_pts.astype(F32), can_dirs.astype(F32))
            z_all, raw_all = merge_sorted([bkg_z[hit], hz], [bkg_out[hit], h_out])
            _rgb, _, _, _, _depth = raw2outputs(raw_all, z_all, d[hit], white_bkg=white_bkg)
            rgb[hit], depth[hit] = _rgb, _depth
        rgbs.append(rgb)
        depths.append(depth)
    rgb, depth = np.concatenate(rgbs), np.concatenate(depths)
    if ray_range is None:
        rgb, depth = rgb.reshape(*cap.shape, -1), depth.reshape(*cap.shape)
    return (rgb, depth) if return_depth else rgb


def render_hybrid_nerf_multi_persons(coarse_bkg, fine_bkg, humans, cap, posed_verts, faces, Ts, rays_per_batch=32768,
                                     samples_per_ray=64, importance_samples_per_ray=128, white_bkg=True,
                                     geo_threshold=ray_ops.DEFAULT_GEO_THRESH, return_depth=False, ray_range=None, bkg_z_out=None,
                                     given=None):
    """reference utils/render_utils.py:365-461.  ``ray_range`` / ``bkg_z_out`` / ``given`` as in render_hybrid_nerf."""
    origins, dirs = _frame_rays(cap, ray_range)
    r0 = 0 if ray_range is None else ray_range[0]
    total = origins.shape[0]
    rgbs, depths = [], []
    for i in range(0, total, rays_per_batch):
        o, d = origins[i:i + rays_per_batch].astype(F32), dirs[i:i + rays_per_batch].astype(F32)
        n = o.shape[0]
        bkg_out, bkg_z = _bkg_pass(coarse_bkg, fine_bkg, o, d, cap.near['bkg'], cap.far['bkg'],
                                   samples_per_ray, importance_samples_per_ray, white_bkg,
                                   given['bkg_z'][r0 + i:r0 + i + o.shape[0]] if given is not None and 'bkg_z' in given else None)
        if bkg_z_out is not None:
            bkg_z_out.append(bkg_z)
        outs, zs = [bkg_out], [bkg_z]
        for a_, (net, v, f, T) in enumerate(zip(humans, posed_verts, faces, Ts)):
            near, far = _near_far(given, a_, r0 + i, o, d, v, geo_threshold)
            h_out = np.zeros((n, samples_per_ray, 4), F32)
            h_z = np.stack([ray_ops.linspace_f32(cap.far['bkg'] * 2, cap.far['bkg'] * 3, samples_per_ray)] * n)
            hit = near < far
            if hit.any():
                pts, dd, hz = ray_ops.ray_to_samples(o[hit], d[hit], near[hit][:, None], far[hit][:, None], samples_per_ray)
                can_pts, can_dirs, _ = warp.warp_samples_to_canonical(pts, v, f, T)
                h_out[hit] = _net(net, can_pts.astype(F32), can_dirs.astype(F32))
                h_z[hit] = hz
            outs.append(h_out)
            zs.append(h_z)
        z_all, raw_all = merge_sorted(zs, outs)
        rgb, _, _, _, depth = raw2outputs(raw_all, z_all, d, white_bkg=white_bkg)
        rgbs.append(rgb)
        depths.append(depth)
    rgb, depth = np.concatenate(rgbs), np.concatenate(depths)
    if ray_range is None:
        rgb, depth = rgb.reshape(*cap.shape, -1), depth.reshape(*cap.shape)
    return (rgb, depth) if return_depth else rgb
