"""Oracle: positional encodings and the 8x256 NeRF MLP.

float32 restatement of reference models/vanilla.py (Embedder :17-92, NeRF :95-152, Joiner :155-166): the
encodings in numpy, the dense layers through torch's CPU `linear` -- the BLAS call the reference itself makes.
Test infrastructure only.

Weights are a dict keyed by the reference Joiner's state_dict names
(``nerf.pts_linears.0.weight`` ... ``nerf.rgb_linear.bias``), values numpy f32.
"""
import numpy as np
import torch
import torch.nn.functional as TF

F32 = np.float32


def _t(a):
    return a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=F32))


def rotate_bvals(min_freq, max_freq, n_freqs):
    """reference models/vanilla.py:44-55: B = (I3 (x) bands) . Rz(45)^T . Rx(45)^T, cast to f32. [3N,3]."""
    bvals = 2. ** np.linspace(min_freq, max_freq, num=n_freqs)
    bvals = np.reshape(np.eye(3) * bvals[:, None, None], [len(bvals) * 3, 3])
    rot = np.array([[(2 ** .5) / 2, -(2 ** .5) / 2, 0], [(2 ** .5) / 2, (2 ** .5) / 2, 0], [0, 0, 1]])
    bvals = bvals @ rot.T
    rot = np.array([[1, 0, 0], [0, (2 ** .5) / 2, -(2 ** .5) / 2], [0, (2 ** .5) / 2, (2 ** .5) / 2]])
    bvals = bvals @ rot.T
    return bvals.astype(F32)


def posenc_bands(min_freq, max_freq, n_freqs):
    """reference models/vanilla.py:67-68: 2**torch.linspace(min,max,N) in f32 (exact powers of two for the defaults)."""
    return (F32(2.) ** np.linspace(min_freq, max_freq, n_freqs).astype(F32)).astype(F32)


def embed_posenc(x, min_freq, max_freq, n_freqs, include_input=True):
    """reference models/vanilla.py:60-79, 92: [x, sin(f0 x), cos(f0 x), sin(f1 x), ...] -- with torch's CPU sin / cos, the
    functions the reference calls (numpy's differ from them by an ulp here and there)."""
    xt = _t(x)
    out = [xt] if include_input else []
    for f in torch.from_numpy(posenc_bands(min_freq, max_freq, n_freqs)):
        xf = xt * f
        out.append(torch.sin(xf))
        out.append(torch.cos(xf))
    return torch.cat(out, -1).numpy()


def embed_rotate(x, min_freq, max_freq, n_freqs, include_input=True):
    """reference models/vanilla.py:83-89: [x, sin(x B^T), cos(x B^T)]."""
    xt = _t(x)
    proj = xt @ torch.from_numpy(rotate_bvals(min_freq, max_freq, n_freqs)).T
    out = torch.cat([torch.sin(proj), torch.cos(proj)], -1)
    if include_input:
        out = torch.cat([xt, out], -1)
    return out.numpy()


def embed(x, mapping, min_freq, max_freq, n_freqs, include_input=True):
    return (embed_rotate if mapping == 'rotate' else embed_posenc)(x, min_freq, max_freq, n_freqs, include_input)


def _linear(h, w, b):
    """nn.Linear on CPU tensors: the very sgemm call (torch's CPU BLAS, float32, bias added by addmm) the reference's
    `self.pts_linears[i](h)` makes, so the restatement runs at the reference's speed and rounds like it."""
    return TF.linear(h, w, b)


def nerf_forward(weights, x_pe, d_pe, depth=8, skips=(4,), return_hidden=False):
    """reference models/vanilla.py:120-152 (use_viewdirs=True, scale_type='no').

    x_pe [N,63], d_pe [N,27] -> [N,4] = (r,g,b,sigma).  With return_hidden the
    post-activation output of every layer is returned too (for layer-by-layer
    kernel debugging).  numpy in, numpy out; the dense layers run as torch CPU ops (see _linear).
    """
    W = {k: _t(v) for k, v in weights.items()}
    hidden = []
    x_pe, d_pe = _t(x_pe), _t(d_pe)
    h = x_pe
    with torch.no_grad():
        for i in range(depth):
            h = torch.relu(_linear(h, W[f'nerf.pts_linears.{i}.weight'], W[f'nerf.pts_linears.{i}.bias']))
            hidden.append(h)
            if i in skips:
                h = torch.cat([x_pe, h], -1)
        if 'nerf.output_linear.weight' in W:                      # use_viewdirs=False: outputs = output_linear(h), vanilla.py:145
            out = _linear(h, W['nerf.output_linear.weight'], W['nerf.output_linear.bias']).numpy()
            return (out, [x.numpy() for x in hidden]) if return_hidden else out
        alpha = _linear(h, W['nerf.alpha_linear.weight'], W['nerf.alpha_linear.bias'])
        feature = _linear(h, W['nerf.feature_linear.weight'], W['nerf.feature_linear.bias'])
        hidden.append(feature)
        h = torch.cat([feature, d_pe], -1)
        h = torch.relu(_linear(h, W['nerf.views_linears.0.weight'], W['nerf.views_linears.0.bias']))
        hidden.append(h)
        rgb = _linear(h, W['nerf.rgb_linear.weight'], W['nerf.rgb_linear.bias'])
        out = torch.cat([rgb, alpha], -1).numpy()
    return (out, [x.numpy() for x in hidden]) if return_hidden else out


class JoinerSpec:
    """What the reference's Joiner(pos_pe, dir_pe, nerf) is configured with (options/options.py:60-71)."""

    def __init__(self, mapping='posenc', pos_min_freq=0, pos_max_freq=9, pos_n_freqs=10,
                 dir_max_freq=3, dir_n_freqs=4, depth=8, width=256, skips=(4,), include_input=True):
        self.mapping = mapping
        self.include_input = include_input                       # options.py:70; both Embedders take it (vanilla.py:208-227)
        self.pos = (pos_min_freq, pos_max_freq, pos_n_freqs)
        self.dir = (0, dir_max_freq, dir_n_freqs)
        self.depth, self.width, self.skips = depth, width, tuple(skips)


def joiner_forward(weights, spec, pts, dirs, chunk=65536, return_hidden=False):
    """reference models/vanilla.py:162-166: PE both inputs then NeRF.forward.  pts/dirs [...,3] -> [...,4]."""
    shp = pts.shape[:-1]
    p = pts.reshape(-1, pts.shape[-1]).astype(F32)             # 3, or 4 with the time channel (ray_utils.py:133-134)
    d = dirs.reshape(-1, 3).astype(F32)
    outs, hid = [], None
    for s in range(0, p.shape[0], chunk):
        x_pe = embed(p[s:s + chunk], spec.mapping, *spec.pos, include_input=getattr(spec, 'include_input', True))
        d_pe = embed(d[s:s + chunk], spec.mapping, *spec.dir, include_input=getattr(spec, 'include_input', True))
        r = nerf_forward(weights, x_pe, d_pe, spec.depth, spec.skips, return_hidden)
        if return_hidden:
            outs.append(r[0])
            hid = r[1] if hid is None else [np.concatenate([a, b], 0) for a, b in zip(hid, r[1])]
        else:
            outs.append(r)
    out = np.concatenate(outs, 0).reshape(*shp, 4)
    return (out, hid) if return_hidden else out
