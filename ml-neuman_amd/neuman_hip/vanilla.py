"""Host-side mirror of the reference's models/vanilla.py for the HIP path.

Same class names, constructor arguments and state_dict keys as the reference (Embedder vanilla.py:17-92,
NeRF :95-152, Joiner :155-166, build_nerf :208-250), so reference checkpoints load unchanged
(`nerf.pts_linears.{i}.weight` ...).  The arithmetic is NOT here: ``Joiner.forward`` hands raw device
pointers to ``nm_mlp_forward`` (fused PE + MLP kernel, csrc/mlp.hip).  There is no torch/CPU evaluation
of the network in this package; calling it with CPU tensors or with autograd enabled raises.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib

# "mixed" (Joiner._prec): split-fp16 x3 (float32-class results) everywhere except the passes the renderers tag as shading,
# which run i8x3 -- sample positions identical to all-fp16x3, every pixel within 1e-4 of the oracle on identical samples
# (measured <= 2.3e-5).  "fp16x3" | "bf16x3" | "i8x3" | "bf16" | "fp32" force one arithmetic for every call.
DEFAULT_PRECISION = os.environ.get("NEUMAN_PRECISION", "mixed")
# the plain-head net (use_viewdirs=False) in "mixed": shading passes on nerf_mlp_i8s_plain_kernel like the view-dependent net's (0: fp16x3 for them)
PLAIN_HEAD_I8 = os.environ.get("NEUMAN_PLAIN_I8", "1") != "0"


def weight_reset(m):
    """reference models/vanilla.py:11-14: re-draw a Linear layer's parameters (the trainer's restart of a dead network)"""
    if isinstance(m, nn.Linear):
        m.reset_parameters()


class Embedder(nn.Module):
    """Positional-encoding *specification* (the encoding itself is computed inside the MLP kernel)."""

    def __init__(self, input_dims, max_freq, N_freqs, log_sampling=True, include_input=True, min_freq=0, mapping='posenc'):
        super().__init__()
        if mapping not in ('posenc', 'rotate'):
            raise ValueError(mapping)
        # log_sampling=False: the reference's posenc constructor stops at `assert 0` (vanilla.py:69-71); its rotate encoding never reads the flag
        assert log_sampling or mapping == 'rotate', "log_sampling=False is not a configuration of the reference (models/vanilla.py:70)"
        self.input_dims, self.max_freq, self.min_freq, self.N_freqs = input_dims, max_freq, min_freq, N_freqs
        self.log_sampling, self.include_input, self.mapping = log_sampling, bool(include_input), mapping
        # include_input=False (vanilla.py:56-58, 63-65, 87-88): the encoding without its leading copy of the input.  The kernels always form the
        # full encoding; the weight columns of the absent inputs are zero in the packed image (Joiner.kernel_params)
        raw = input_dims if mapping == 'posenc' else 3
        self.out_dim = (raw if self.include_input else 0) + (2 * input_dims * N_freqs if mapping == 'posenc' else 6 * N_freqs)

    def table(self):
        """f32 table the kernel consumes: posenc -> bands[N]; rotate -> bvals[3N,3] (vanilla.py:44-55, 67-68)."""
        if self.mapping == 'posenc':
            return (2. ** torch.linspace(self.min_freq, self.max_freq, steps=self.N_freqs)).numpy().astype(np.float32)
        bands = 2. ** np.linspace(self.min_freq, self.max_freq, num=self.N_freqs)
        b = (np.eye(3)[None] * bands[:, None, None]).reshape(3 * self.N_freqs, 3)
        c = 2 ** .5 / 2
        b = b @ np.array([[c, -c, 0], [c, c, 0], [0, 0, 1]]).T
        b = b @ np.array([[1, 0, 0], [0, c, -c], [0, c, c]]).T
        return np.ascontiguousarray(b.astype(np.float32))

    @property
    def bvals(self):  # reference attribute name (vanilla.py:52-55)
        return torch.from_numpy(self.table()) if self.mapping == 'rotate' else None

    def forward(self, inputs, cur_iter=None):
        raise _lib.NeumanHipError("Embedder.forward is fused into the HIP MLP kernel: call the Joiner")


class NeRF(nn.Module):
    """Parameter container with the reference's layer names (vanilla.py:95-118)."""

    def __init__(self, depth=8, width=256, input_ch=3, input_ch_views=3, output_ch=4, skips=[4], use_viewdirs=False,
                 scale=1.0, scale_type='no'):
        super().__init__()
        self.depth, self.width, self.input_ch, self.input_ch_views = depth, width, input_ch, input_ch_views
        self.skips, self.use_viewdirs, self.scale, self.scale_type = skips, use_viewdirs, scale, scale_type
        layers = [nn.Linear(input_ch, width)]
        for i in range(depth - 1):
            layers.append(nn.Linear(width + input_ch if i in skips else width, width))
        self.pts_linears = nn.ModuleList(layers)
        if use_viewdirs:
            self.views_linears = nn.ModuleList([nn.Linear(input_ch_views + width, width // 2)])
            self.feature_linear = nn.Linear(width, width)
            self.alpha_linear = nn.Linear(width, 1)
            self.rgb_linear = nn.Linear(width // 2, 3)
        else:
            self.output_linear = nn.Linear(width, output_ch)

    def ordered_params(self):
        """The tensors nm_mlp_create expects (include/neuman_hip.h), reference state_dict order: 24 for the use_viewdirs=True net;
        the 16 pts_linears tensors + output_linear's two for the plain head (use_viewdirs=False, vanilla.py:116-117)."""
        heads = [self.views_linears[0], self.feature_linear, self.alpha_linear, self.rgb_linear] if self.use_viewdirs else [self.output_linear]
        if not self.use_viewdirs and self.output_linear.out_features != 4:
            raise NotImplementedError("the rendering kernels take the 4-output (r, g, b, sigma) plain head; other widths (the offset "
                                      "net's 3) run on the differentiable float32 path (neuman_hip/train.py)")
        out = []
        for lin in list(self.pts_linears) + heads:
            out += [lin.weight, lin.bias]
        return out

    def forward(self, input_pts, input_views=None):
        raise _lib.NeumanHipError("NeRF.forward on pre-encoded inputs is not exposed: the HIP kernel fuses PE + MLP (use Joiner)")


def absent_input_columns(pos_pe, dir_pe, nerf):
    """{index into NeRF.ordered_params(): (column, count)}: where zero columns go so that a net built over encodings WITHOUT the raw input
    (include_input=False: vanilla.py:56-58, 87-88 -- the encoding loses its leading `input_dims` columns) reads the full encoding the kernels
    form.  Layer 0 and the layer after a skip take cat([x_pe, h]) (vanilla.py:127-131), the views layer cat([feature, d_pe]) (:139)."""
    pads = {}
    if not pos_pe.include_input:
        raw = pos_pe.input_dims if pos_pe.mapping == 'posenc' else 3
        pads[0] = (0, raw)
        for s in nerf.skips:
            if s + 1 < nerf.depth:
                pads[2 * (s + 1)] = (0, raw)
    if nerf.use_viewdirs and not dir_pe.include_input:
        pads[2 * nerf.depth] = (nerf.width, dir_pe.input_dims if dir_pe.mapping == 'posenc' else 3)
    return pads


def with_absent_columns(tensors, pads):
    """ordered_params() widened by the zero columns of absent_input_columns (torch.cat: differentiable, the training path's derived parameters)"""
    out = list(tensors)
    for i, (at, cnt) in pads.items():
        w = out[i]
        out[i] = torch.cat([w[:, :at], torch.zeros((w.shape[0], cnt), device=w.device, dtype=w.dtype), w[:, at:]], 1)
    return out


class Joiner(nn.Module):
    """PE + MLP, evaluated by the fused HIP kernel (reference vanilla.py:155-166)."""

    def __init__(self, pos_pe, dir_pe, nerf):
        super().__init__()
        self.pos_pe, self.dir_pe, self.nerf = pos_pe, dir_pe, nerf
        self.precision = DEFAULT_PRECISION
        self._handle = None
        self._handle_key = None
        self._train_handle = None

    # ---- weight-pack cache: rebuilt whenever a parameter's storage or version changes -------------
    def _key(self):
        return tuple((p.data_ptr(), p._version) for p in self.nerf.ordered_params())

    def _release(self, train_too=True):
        if self._handle is not None:
            _lib.lib().nm_mlp_destroy(self._handle)
            self._handle = None
        if train_too and getattr(self, '_train_handle', None) is not None:
            _lib.lib().nm_mlp_destroy(self._train_handle)
            self._train_handle = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    # the packed-weight handle is a cache owned by this object: copies and pickles (copy.deepcopy, torch.save(model),
    # DataLoader workers) drop it and rebuild their own lazily, like the reference's plain nn.Module
    def __getstate__(self):
        state = self.__dict__.copy()
        state['_handle'] = None
        state['_handle_key'] = None
        state['_train_handle'] = None
        return state

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            new.__dict__[k] = None if k in ('_handle', '_handle_key', '_train_handle') else copy.deepcopy(v, memo)
        return new

    def handle(self):
        key = self._key()
        if self._handle is None or key != self._handle_key:
            self._release(train_too=False)                       # (the training handle refreshes its own image on the device: train_handle)
            n = self.nerf
            if n.scale_type != 'no':
                raise NotImplementedError("scale_type != 'no' (offset nets) is outside the HIP path")
            if self.pos_pe.mapping != self.dir_pe.mapping:
                raise NotImplementedError("mixed PE kinds")
            desc = _lib.MlpDesc(n.depth, n.width, n.skips[0] if len(n.skips) == 1 else -1,
                                _lib.NM_PE_ROTATE if self.pos_pe.mapping == 'rotate' else _lib.NM_PE_POSENC,
                                self.pos_pe.N_freqs, self.dir_pe.N_freqs, 0 if n.use_viewdirs else 1)
            # (include_input=False: zero weight columns where the kernels' encoding carries the raw input)
            host = [p.to('cpu', torch.float32).contiguous()
                    for p in with_absent_columns([q.detach() for q in n.ordered_params()], absent_input_columns(self.pos_pe, self.dir_pe, n))]
            arr = (ctypes.c_void_p * 24)(*([t.data_ptr() for t in host] + [None] * (24 - len(host))))
            pos_tab, dir_tab = self.pos_pe.table(), self.dir_pe.table()
            out = ctypes.c_void_p()
            _lib.check(_lib.lib().nm_mlp_create(ctypes.byref(desc), arr, pos_tab.ctypes.data, dir_tab.ctypes.data,
                                                ctypes.byref(out)), "nm_mlp_create")
            self._handle, self._handle_key = out, key
        return self._handle

    def train_handle(self):
        """A second handle for the training forward (neuman_hip/train.py): created once and never rebuilt -- its split-fp16 weight
        image is rewritten on the device from the live parameters before every use (nm_mlp_refresh_f16), so an optimiser step costs
        three small kernels, not a host repack."""
        if self._train_handle is None:
            keep, keep_key = self._handle, self._handle_key
            self._handle = None                                  # build a fresh handle through the same code path, then restore the cache
            try:
                self._train_handle = self.handle()
            finally:
                self._handle, self._handle_key = keep, keep_key
        return self._train_handle

    def _prec(self, precision, role=None):
        """'mixed': a pass the caller tags role='shading' -- its output is composited into the frame and nothing else --
        runs in i8x3; every other call (the coarse pass whose compositing weights place the importance samples, and any
        direct call) runs in fp16x3, whose sigma is float32 class (the inverse CDF amplifies a coarse-pass error by 1 / pdf:
        DESIGN.md section 5).  Sample positions are then bit-identical to the all-fp16x3 path and the colours differ from it
        by the i8x3 compositing error (<= 2e-5, tests/test_hip_mlp.py) on every pixel."""
        p = precision or self.precision
        if p == 'mixed':
            p = 'i8x3' if role == 'shading' else 'fp16x3'
        if p == 'i8x3' and not self.nerf.use_viewdirs and (precision or self.precision) == 'mixed' and not PLAIN_HEAD_I8:
            p = 'fp16x3'                                     # NEUMAN_PLAIN_I8=0: the plain-head net's shading passes stay float32 class
        return _lib.PRECISIONS[p]

    @staticmethod
    def _guard(*tensors):
        _lib.require_gpu()
        if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
            raise _lib.NeumanHipError("the rendering kernels are forward-only: wrap in torch.no_grad(), or put the Joiner in train() mode "
                                       "for the differentiable float32 forward (neuman_hip/train.py)")

    def forward(self, input_pts, input_views=None, precision=None, sigma_scale=1.0, role=None):
        """input_pts [..., 3], input_views [..., 3] (CUDA f32) -> [..., 4] = (r, g, b, sigma).  The plain-head net
        (use_viewdirs=False) ignores the views like the reference (vanilla.py:122-123) and accepts None."""
        if input_views is None:
            if self.nerf.use_viewdirs:
                raise _lib.NeumanHipError("input_views is required by the use_viewdirs=True net (vanilla.py:133-134)")
            input_views = torch.zeros_like(input_pts)
        if self.pos_pe.input_dims != 3:
            # the time-conditioned ablation net (`--ablate_nerft`, raw_pos_dim = 4: points carry the frame time, ray_utils.py:133-134,
            # render_utils.py:134-148): the fused kernels encode 3-vectors only, so this net runs on the float32 MFMA GEMM chain of
            # the training slice (exact float32 products; with or without autograd) -- correct, not the tuned path
            from . import train
            if sigma_scale != 1.0:
                raise _lib.NeumanHipError("sigma_scale is a canonical-render option (render_utils.py:229)")
            if input_pts.shape[-1] != self.pos_pe.input_dims:
                raise _lib.NeumanHipError(f"input_pts has {input_pts.shape[-1]} components, the position encoding takes {self.pos_pe.input_dims}")
            return train.mlp_forward_train(self, input_pts, input_views)
        if self.training and torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            # a training step (trainers/vanilla_nerf_trainer.py:66): float32 MFMA forward that keeps its activations
            from . import train
            if sigma_scale != 1.0:
                raise _lib.NeumanHipError("sigma_scale is a render-time option (render_utils.py:229)")
            return train.mlp_forward_train(self, input_pts, input_views)
        self._guard(input_pts, input_views)
        shp = input_pts.shape[:-1]
        p = input_pts.detach().reshape(-1, 3).contiguous()
        d = input_views.detach().reshape(-1, 3).contiguous()
        out = torch.empty((p.shape[0], 4), device=p.device, dtype=torch.float32)
        _lib.check(_lib.lib().nm_mlp_forward(self.handle(), _lib.dev_ptr(p, name='input_pts'), _lib.dev_ptr(d, name='input_views'),
                                             p.shape[0], self._prec(precision, role), float(sigma_scale), _lib.dev_ptr(out),
                                             _lib.stream_ptr()), "nm_mlp_forward")
        return out.reshape(*shp, 4)

    def forward_two_views(self, input_pts, input_views, other_views):
        """(self(input_pts, input_views), self(input_pts, other_views) with a zero density column), the trunk evaluated once -- a training step only
        (neuman_hip/train.py two_views; None when this net / batch cannot take it: call forward twice then)"""
        from . import train
        n = input_pts.reshape(-1, input_pts.shape[-1]).shape[0]
        if not (self.training and torch.is_grad_enabled() and self.nerf.use_viewdirs and self.pos_pe.input_dims == 3 and train.two_views_ok(self, n)):
            return None
        return train.two_views(self, input_pts, input_views, other_views)

    def forward_rays(self, origin, direction, z_vals, precision=None, sigma_scale=1.0, role=None, sigma_only=False):
        """Fused ray_to_samples point construction + forward: origin/direction [R,3], z_vals [R,S] -> [R,S,4].
        sigma_only: the caller uses nothing but out[..., 3] (a coarse pass that only places importance samples) -- the
        colour head is skipped where the kernel can (out[..., :3] = 0), sigma is bit-identical either way."""
        self._guard(origin, direction, z_vals)
        R, S = z_vals.shape
        out = torch.empty((R, S, 4), device=z_vals.device, dtype=torch.float32)
        entry = _lib.lib().nm_mlp_sigma_rays if sigma_only else _lib.lib().nm_mlp_forward_rays
        _lib.check(entry(self.handle(), _lib.dev_ptr(origin, name='origin'), _lib.dev_ptr(direction, name='direction'),
                         _lib.dev_ptr(z_vals, name='z_vals'), R, S, self._prec(precision, role), float(sigma_scale),
                         _lib.dev_ptr(out), _lib.stream_ptr()), "nm_mlp_sigma_rays" if sigma_only else "nm_mlp_forward_rays")
        return out

    def forward_ray_chunk(self, origin, direction, z_vals, ray_idx, n_rays_dev, s0, chunk, out, precision=None, sigma_scale=1.0, role=None,
                          sigma_only=False):
        """One chunk of a front-to-back march (nm_mlp_forward_ray_chunk): samples s0 .. s0+chunk-1 of the rays listed in
        `ray_idx` (int32; only its first *n_rays_dev entries are live -- the count stays on the device) are evaluated and written
        into `out` [R,S,4]; nothing else of `out` is touched.  sigma_only: as forward_rays."""
        self._guard(origin, direction, z_vals)
        R, S = z_vals.shape
        entry = _lib.lib().nm_mlp_sigma_ray_chunk if sigma_only else _lib.lib().nm_mlp_forward_ray_chunk
        _lib.check(entry(
            self.handle(), _lib.dev_ptr(origin, name='origin'), _lib.dev_ptr(direction, name='direction'), _lib.dev_ptr(z_vals, name='z_vals'), S,
            _lib.dev_ptr(ray_idx, torch.int32, 'ray_idx'), _lib.dev_ptr(n_rays_dev, torch.int32, 'n_rays_dev'), ray_idx.shape[0], int(s0), int(chunk),
            self._prec(precision, role), float(sigma_scale), _lib.dev_ptr(out), _lib.stream_ptr()), "nm_mlp_forward_ray_chunk")
        return out

    def forward_debug(self, input_pts, input_views, stage, precision=None):
        """Activations after `stage` (include/neuman_hip.h: nm_mlp_forward_debug)."""
        self._guard(input_pts, input_views)
        p = input_pts.detach().reshape(-1, 3).contiguous()
        d = input_views.detach().reshape(-1, 3).contiguous()
        width = 64 if stage == -1 else (128 if stage == 9 else 256)
        out = torch.zeros((p.shape[0], width), device=p.device, dtype=torch.float32)
        _lib.check(_lib.lib().nm_mlp_forward_debug(self.handle(), _lib.dev_ptr(p), _lib.dev_ptr(d), p.shape[0],
                                                   self._prec(precision), int(stage), _lib.dev_ptr(out), _lib.stream_ptr()),
                   "nm_mlp_forward_debug")
        return out


def time_columns(pe):
    """Of a space-time posenc Embedder (input_dims = 4: x, y, z, t; vanilla.py:60-79 layout [v, sin(f_0 v), cos(f_0 v), ...] with v the 4-vector):
    -> (columns of the three spatial coordinates in the order of the 3-D encoding, columns of the time coordinate)"""
    if pe.mapping != 'posenc' or pe.input_dims != 4:
        raise _lib.NeumanHipError("time_columns: a 4-D posenc encoding is expected (raw_pos_dim = 4)")
    N = pe.N_freqs
    raw = 4 if pe.include_input else 0                                  # (include_input=False: no leading copy of v, vanilla.py:63-65)
    spatial = [0, 1, 2][:raw] + [raw + 8 * b + k for b in range(N) for k in (0, 1, 2, 4, 5, 6)]
    timecol = [3][:raw] + [raw + 8 * b + k for b in range(N) for k in (3, 7)]
    return spatial, timecol


def time_encoding(pe, t):
    """the encoded time coordinate alone, in the order of time_columns()[1]: [t, sin(f_0 t), cos(f_0 t), ...] with the float32 arguments the
    reference forms (vanilla.py:73-76)"""
    bands = pe.table().astype(np.float32)
    tt = np.float32(t)
    out = [float(tt)] if pe.include_input else []
    for b in range(pe.N_freqs):
        a = np.float64(np.float32(tt * bands[b]))
        out += [float(np.sin(a)), float(np.cos(a))]
    return np.asarray(out, np.float64)


def frozen_time_joiner(joiner, t):
    """The time-conditioned net of `--ablate_nerft` (raw_pos_dim = 4: every sample point carries the frame's time, ray_utils.py:133-134,
    158-159) at ONE time t, as an ordinary 3-D-position Joiner the fused MFMA kernels evaluate.

    Within a rendered frame the time is a constant (render_utils.py:134-148: `ones * cur_time`), so the 21 encoded-time inputs of layer 0
    and of the skip layer contribute a constant vector to those layers' pre-activations: W[:, time columns] . pe(t), folded into the
    bias.  What is left is exactly the reference's 3-D network shape (63-wide encoding, the same hidden weights -- shared, not copied)
    with two bias vectors that depend on t.  Cached per (weights version, t): a sequence rendered frame by frame repacks two bias
    vectors' worth of images per frame."""
    pe = joiner.pos_pe
    key = (joiner._key(), float(t))
    cache = joiner.__dict__.setdefault('_frozen_time', {})
    if key in cache:
        return cache[key]
    spatial, timecol = time_columns(pe)
    n = joiner.nerf
    with torch.no_grad():
        pe3 = Embedder(3, pe.max_freq, pe.N_freqs, pe.log_sampling, pe.include_input, min_freq=pe.min_freq, mapping='posenc')
        nerf = NeRF(depth=n.depth, width=n.width, input_ch=pe3.out_dim, input_ch_views=n.input_ch_views, use_viewdirs=n.use_viewdirs, skips=list(n.skips),
                    output_ch=(n.output_linear.out_features if not n.use_viewdirs else 4))
        dev = n.pts_linears[0].weight.device
        pt = torch.as_tensor(time_encoding(pe, t), dtype=torch.float64, device=dev)
        sp, tc = torch.as_tensor(spatial, device=dev), torch.as_tensor(timecol, device=dev)
        for i, lin in enumerate(n.pts_linears):
            takes_pe = i == 0 or (i - 1) in n.skips
            if not takes_pe:
                nerf.pts_linears[i] = lin                                  # shared: the same Parameter objects
                continue
            w = lin.weight.detach()
            new = nn.Linear(pe3.out_dim + (0 if i == 0 else n.width), n.width)
            new.weight.copy_(torch.cat([w[:, sp], w[:, pe.out_dim:]], 1) if i else w[:, sp])
            new.bias.copy_((lin.bias.detach().double() + w[:, tc].double() @ pt).to(torch.float32))
            nerf.pts_linears[i] = new.to(dev)
        for name in (('views_linears', 'feature_linear', 'alpha_linear', 'rgb_linear') if n.use_viewdirs else ('output_linear',)):
            setattr(nerf, name, getattr(n, name))
        out = Joiner(pe3, joiner.dir_pe, nerf).to(dev).eval()
        out.precision = joiner.precision
    while len(cache) >= 4:
        cache.pop(next(iter(cache)))
    cache[key] = out
    return out


class OffsetNet(nn.Module):
    """reference vanilla.py:169-178: space-time encoding + a NeRF trunk without view directions, used by the human trainer only
    (human_nerf_trainer.py:261).  Training-only here: it runs on the differentiable float32 path (neuman_hip/train.py)."""

    def __init__(self, pos_pe, nerf):
        super().__init__()
        self.pos_pe, self.nerf = pos_pe, nerf

    def forward(self, input_pts, cur_iter=None, const_time=None):
        """const_time (an extension): the caller's promise that input_pts[..., 3] is this one number on every row -- the human trainer's
        batches are one frame (human_nerf_trainer.py:258-261) -- which lets a large batch run on the fused training kernels"""
        assert cur_iter is None                                      # as the reference's posenc Embedder (vanilla.py:91)
        from . import train
        out = train.offset_forward_train(self, input_pts, const_time=const_time)
        if self.nerf.scale_type == 'no':                             # vanilla.py:146-152
            return out
        if self.nerf.scale_type == 'linear':
            return out * self.nerf.scale
        if self.nerf.scale_type == 'tanh':
            return torch.tanh(out) * self.nerf.scale
        raise ValueError(self.nerf.scale_type)


def build_offset_net(opt):
    """reference vanilla.py:180-205"""
    st_pe = Embedder(opt.raw_pos_dim + 1, opt.pos_max_freq, opt.pos_N_freqs, opt.log_sampling, opt.include_input, min_freq=opt.pos_min_freq)
    nerf = NeRF(depth=opt.nerf_depth, width=opt.nerf_width, input_ch=st_pe.out_dim, input_ch_views=0, output_ch=3, use_viewdirs=False,
                scale=opt.offset_scale, scale_type=opt.offset_scale_type)
    net = OffsetNet(st_pe, nerf)
    return net.cuda() if opt.use_cuda else net


def build_nerf(opt):
    """Same construction order as the reference (vanilla.py:208-250): coarse NeRF first, then fine."""
    mapping = opt.posenc if hasattr(opt, 'posenc') else 'posenc'
    pos_pe = Embedder(opt.raw_pos_dim, opt.pos_max_freq, opt.pos_N_freqs, opt.log_sampling, opt.include_input,
                      min_freq=opt.pos_min_freq, mapping=mapping)
    dir_pe = Embedder(opt.raw_dir_dim, opt.dir_max_freq, opt.dir_N_freqs, opt.log_sampling, opt.include_input, mapping=mapping)
    nets = []
    for _ in range(2):
        nerf = NeRF(depth=opt.nerf_depth, width=opt.nerf_width, input_ch=pos_pe.out_dim, input_ch_views=dir_pe.out_dim,
                    use_viewdirs=opt.use_viewdirs)
        nets.append(nerf)
    coarse, fine = Joiner(pos_pe, dir_pe, nets[0]), Joiner(pos_pe, dir_pe, nets[1])
    if opt.use_cuda:
        coarse, fine = coarse.cuda(), fine.cuda()
    return coarse, fine
