"""Oracle: one training step of the background NeRF -- loss and parameter gradients.

Restates in torch float64 (autograd supplies the adjoint; this is the one oracle that is not numpy, because what it checks is
a gradient):
  * models/vanilla.py:60-79 (posenc), :120-152 (NeRF.forward), :162-166 (Joiner.forward)
  * utils/render_utils.py:69-105 (raw2outputs)
  * trainers/vanilla_nerf_trainer.py:66-78, 85-95 (MSE on rgb_map + the empty-space penalty on sigma)

PINNED: tests/golden/train.npz holds the reference's own losses, raw outputs, d loss / d raw and parameter gradients from
NeRFTrainer.loss_func + backward() (tests/golden/make_golden_train.py); the reference computes in float32, so agreement is to
float32 rounding of a 600 k-term reduction (1e-4 of each tensor's largest gradient).  Test infrastructure only.
"""
import numpy as np
import torch

F64 = torch.float64


def posenc(x, n_freqs, min_freq=0, max_freq=None):
    max_freq = n_freqs - 1 if max_freq is None else max_freq
    bands = (2.0 ** torch.linspace(min_freq, max_freq, n_freqs, dtype=torch.float32)).to(F64)
    out = [x]
    for f in bands:
        out += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(out, -1)


def rotenc(x, n_freqs, min_freq=0, max_freq=None):
    """models/vanilla.py:83-89: [x, sin(x B^T), cos(x B^T)], B from :44-55 (float32 values)"""
    from .nerf_mlp import rotate_bvals
    B = torch.tensor(rotate_bvals(min_freq, n_freqs - 1 if max_freq is None else max_freq, n_freqs), dtype=F64)
    proj = x @ B.T
    return torch.cat([x, torch.sin(proj), torch.cos(proj)], -1)


def joiner_forward(W, pts, dirs, pos_freqs=10, dir_freqs=4, depth=8, skips=(4,), mapping='posenc'):
    """W: {state_dict name: float64 tensor requiring grad}; pts, dirs [...,3] float64 -> raw [...,4]."""
    enc = rotenc if mapping == 'rotate' else posenc
    x_pe, d_pe = enc(pts, pos_freqs), enc(dirs, dir_freqs)
    h = x_pe
    for i in range(depth):
        h = torch.relu(h @ W[f'nerf.pts_linears.{i}.weight'].T + W[f'nerf.pts_linears.{i}.bias'])
        if i in skips:
            h = torch.cat([x_pe, h], -1)
    alpha = h @ W['nerf.alpha_linear.weight'].T + W['nerf.alpha_linear.bias']
    feature = h @ W['nerf.feature_linear.weight'].T + W['nerf.feature_linear.bias']
    h = torch.relu(torch.cat([feature, d_pe], -1) @ W['nerf.views_linears.0.weight'].T + W['nerf.views_linears.0.bias'])
    rgb = h @ W['nerf.rgb_linear.weight'].T + W['nerf.rgb_linear.bias']
    return torch.cat([rgb, alpha], -1)


def raw2outputs(raw, z_vals, rays_d, white_bkg=True):
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], 1e10)], -1) * torch.norm(rays_d[..., None, :], dim=-1)
    rgb = torch.sigmoid(raw[..., :3])
    alpha = 1. - torch.exp(-torch.relu(raw[..., 3]) * dists)
    T = torch.cumprod(torch.cat([torch.ones_like(alpha[:, :1]), 1. - alpha + 1e-10], -1), -1)[:, :-1]
    weights = alpha * T
    rgb_map = torch.sum(weights[..., None] * rgb, -2)
    depth_map = torch.sum(weights * z_vals, -1)
    acc_map = torch.sum(weights, -1)
    if white_bkg:
        rgb_map = rgb_map + (1. - acc_map[..., None])
    return rgb_map, acc_map, weights, depth_map


def training_pass(weights, origin, direction, z_vals, color, white_bkg=True, penalty=0.0, depth=None, margin=0.9):
    """One net's share of NeRFTrainer.loss_func at given sample depths.  weights: {name: numpy}.
    -> dict(raw, rgb_map, acc_map, weights, depth_map, loss_rgb, loss_empty, d_raw, grads {name: numpy float64})"""
    W = {k: torch.tensor(np.asarray(v), dtype=F64, requires_grad=True) for k, v in weights.items()}
    o, d, z = (torch.tensor(np.asarray(a), dtype=F64) for a in (origin, direction, z_vals))
    pts = o[:, None, :] + d[:, None, :] * z[..., None]
    dirs = d[:, None, :].expand(pts.shape)
    raw = joiner_forward(W, pts, dirs)
    raw.retain_grad()
    rgb_map, acc, wts, dep = raw2outputs(raw, z, d, white_bkg)
    loss_rgb = torch.mean((rgb_map - torch.tensor(np.asarray(color), dtype=F64)) ** 2)
    loss_empty = torch.zeros((), dtype=F64)
    if penalty > 0:
        closer = z < (torch.tensor(np.asarray(depth), dtype=F64)[:, None] * margin)
        loss_empty = torch.mean(torch.tanh(torch.relu(raw[closer][:, 3])) ** 2) * penalty
    (loss_rgb + loss_empty).backward()
    n = lambda t: t.detach().numpy()
    return dict(raw=n(raw), rgb_map=n(rgb_map), acc_map=n(acc), weights=n(wts), depth_map=n(dep), loss_rgb=float(loss_rgb.detach()),
                loss_empty=float(loss_empty.detach()), d_raw=n(raw.grad), grads={k: n(v.grad) for k, v in W.items()})


def input_gradients(weights, pts, dirs, g_out, mapping='posenc'):
    """d (sum of g_out . net(pts, dirs)) / d pts, d dirs -> (out, d_pts, d_dirs), float64"""
    W = {k: torch.tensor(np.asarray(v), dtype=F64) for k, v in weights.items()}
    p = torch.tensor(np.asarray(pts), dtype=F64, requires_grad=True)
    d = torch.tensor(np.asarray(dirs), dtype=F64, requires_grad=True)
    out = joiner_forward(W, p, d, mapping=mapping)
    (out * torch.tensor(np.asarray(g_out), dtype=F64)).sum().backward()
    return out.detach().numpy(), p.grad.numpy(), d.grad.numpy()


def composite_backward(raw, z_vals, rays_d, white_bkg, g_rgb, g_acc, g_depth, g_w):
    """d (sum of g_k . output_k) / d raw through raw2outputs, float64."""
    r = torch.tensor(np.asarray(raw), dtype=F64, requires_grad=True)
    rgb, acc, wts, dep = raw2outputs(r, torch.tensor(np.asarray(z_vals), dtype=F64), torch.tensor(np.asarray(rays_d), dtype=F64), white_bkg)
    t = lambda a: torch.tensor(np.asarray(a), dtype=F64)
    ((rgb * t(g_rgb)).sum() + (acc * t(g_acc)).sum() + (dep * t(g_depth)).sum() + (wts * t(g_w)).sum()).backward()
    return r.grad.numpy()


def offset_net_gradients(weights, x, g_out, scale=1.0, scale_type='linear', n_freqs=10, depth=8, skips=(4,)):
    """The offset net (models/vanilla.py:169-205: 4-D posenc, trunk, output_linear, scale) and the adjoint of sum(g_out . out):
    -> (out, d_x, {name: grad}) float64."""
    W = {k: torch.tensor(np.asarray(v), dtype=F64, requires_grad=True) for k, v in weights.items()}
    xt = torch.tensor(np.asarray(x), dtype=F64, requires_grad=True)
    x_pe = posenc(xt, n_freqs)
    h = x_pe
    for i in range(depth):
        h = torch.relu(h @ W[f'nerf.pts_linears.{i}.weight'].T + W[f'nerf.pts_linears.{i}.bias'])
        if i in skips:
            h = torch.cat([x_pe, h], -1)
    out = h @ W['nerf.output_linear.weight'].T + W['nerf.output_linear.bias']
    out = out * scale if scale_type == 'linear' else (torch.tanh(out) * scale if scale_type == 'tanh' else out)
    (out * torch.tensor(np.asarray(g_out), dtype=F64)).sum().backward()
    return out.detach().numpy(), xt.grad.numpy(), {k: v.grad.numpy() for k, v in W.items()}
