"""Oracle: alpha compositing and sorted merging of sample lists.

numpy float32 restatement of reference utils/render_utils.py:69-105 (raw2outputs)
and :330-337 / :441-448 (sort + gather merge).  Test infrastructure only.
"""
import numpy as np

F32 = np.float32


def sigmoid_f32(x):
    x = x.astype(F32)
    return (F32(1.) / (F32(1.) + np.exp(-x, dtype=F32))).astype(F32)


def raw2outputs(raw, z_vals, rays_d, noise=None, white_bkg=True):
    """reference utils/render_utils.py:69-105.

    raw [R,S,4] (r,g,b,sigma), z_vals [R,S], rays_d [R,3]; ``noise`` [R,S]
    replaces the reference's ``torch.randn * raw_noise_std`` draw.
    Returns rgb_map [R,3], disp_map [R], acc_map [R], weights [R,S], depth_map [R].
    """
    raw, z_vals, rays_d = raw.astype(F32), z_vals.astype(F32), rays_d.astype(F32)
    dists = (z_vals[..., 1:] - z_vals[..., :-1]).astype(F32)
    dists = np.concatenate([dists, np.full(dists[..., :1].shape, 1e10, F32)], -1)
    dists = (dists * np.sqrt(np.sum(rays_d * rays_d, -1, dtype=F32))[..., None]).astype(F32)
    rgb = sigmoid_f32(raw[..., :3])
    sigma = raw[..., 3]
    if noise is not None:
        sigma = (sigma + noise.astype(F32)).astype(F32)
    with np.errstate(over='ignore'):
        alpha = (F32(1.) - np.exp(-np.maximum(sigma, F32(0.)) * dists, dtype=F32)).astype(F32)
    # torch's CPU cumprod accumulates float32 inputs in float64 and rounds each output to float32
    trans = np.cumprod(np.concatenate([np.ones((alpha.shape[0], 1), F32),
                                       (F32(1.) - alpha + F32(1e-10)).astype(F32)], -1).astype(np.float64), -1).astype(F32)[:, :-1]
    weights = (alpha * trans).astype(F32)
    rgb_map = np.sum(weights[..., None] * rgb, -2, dtype=F32)
    depth_map = np.sum(weights * z_vals, -1, dtype=F32)
    acc_map = np.sum(weights, -1, dtype=F32)
    with np.errstate(divide='ignore', invalid='ignore'):
        disp_map = (F32(1.) / np.maximum(F32(1e-10), depth_map / acc_map)).astype(F32)
    if white_bkg:
        rgb_map = (rgb_map + (F32(1.) - acc_map[..., None])).astype(F32)
    return rgb_map, disp_map, acc_map, weights, depth_map


def merge_sorted(z_list, raw_list):
    """reference utils/render_utils.py:330-337: sort(cat(z)) and gather cat(raw) by the sort order."""
    z = np.concatenate(z_list, -1).astype(F32)
    raw = np.concatenate(raw_list, 1).astype(F32)
    order = np.argsort(z, -1, kind='stable')
    return np.take_along_axis(z, order, -1), np.take_along_axis(raw, order[..., None], 1)
