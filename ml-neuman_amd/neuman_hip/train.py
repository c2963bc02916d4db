"""SURVEY 8f-1, first slice: differentiable forward of the Joiner (PE + NeRF MLP) and of raw2outputs on the device, so that the
reference's training loss (trainers/vanilla_nerf_trainer.py:45-96) runs through `loss.backward()` unchanged:

    out = coarse_net(pts, dirs)                      # Joiner in train() mode with grad enabled -> mlp_forward_train
    rgb_map, _, _, weights, _ = raw2outputs(out, z_vals, dirs[:, 0, :], ...)        # -> composite_train
    F.mse_loss(rgb_map, color).backward()            # fills .grad of the 24 parameters

Every matrix product is one of `nm_gemm_fp16x3 / nm_gemm_bf16x3 / nm_gemm_f32` (csrc/train.hip; GEMM_PRECISION below), the encodings `nm_pe_encode`, the compositing adjoint
`nm_composite_backward`; torch supplies memory and the autograd graph.  The layer loop below is the reference's
models/vanilla.py:120-152 with the two concatenations (skip connection :130-131, views layer :139-140) written as two
products into one output.  Gradients reach the parameters and, when asked for, the sample positions and view directions
(what the human trainer's pose / offset optimisation differentiates: the differentiable warp, the offset nets and the skinning
upstream of them are neuman_hip.ray_utils.warp_samples_to_canonical_diff, OffsetNet and smpl.SMPLDiff).
"""
import ctypes
import os

import torch

from . import _lib

ACC, BIAS, RELU, MASK, COLSUM = 1, 2, 4, 8, 16        # NM_GEMM_* (include/neuman_hip.h)
PE_KINDS = {'posenc': 0, 'rotate': 1}                 # NM_PE_POSENC / NM_PE_ROTATE
# arithmetic of the matrix products:
#   'f32'     f32 MFMA everywhere: gradients within 1e-6 of the reference's float32 autograd (NEUMAN_TRAIN_GEMM=f32)
#   'mixed16' (default) forward products split-fp16 x3 on the fp16 MFMA (float32 class: no ReLU decided differently from the f32 forward beyond
#             what float32 reordering already does), backward products split-bf16 x3 on the bf16 MFMA (range-safe for gradients of any
#             magnitude; 2^-17 per product, a smooth error): parameter gradients 6e-6 from the reference's float32 autograd, 1.25x faster
#             per iteration (the reference itself runs TF32 products on the GPUs it was written for: torch 1.8's default)
#   'bf16x3'  split-bf16 x3 everywhere (its 1e-5 forward error flips a ReLU here and there: single gradient entries move by ~1e-3 of
#             the tensor's largest -- fine for SGD, not for the parity tests);  'fp16x3': split-fp16 x3 everywhere (forward-safe only)
GEMM_PRECISION = os.environ.get('NEUMAN_TRAIN_GEMM', 'mixed16')
# the forward of a standard Joiner (3-D encodings, 8 x 256, view directions) in ONE kernel that keeps a tile's activations on chip across
# the layers and only writes the copies the backward pass reads (nm_mlp_forward_save; mixed16 only).  NEUMAN_TRAIN_FUSED=0: the GEMM chain.
FUSED_FORWARD = os.environ.get('NEUMAN_TRAIN_FUSED', '1') != '0'
# ... and the backward-data chain of its trunk in one kernel as well (nm_mlp_backward_chain: dZ of a tile stays on chip from layer 7 to layer 0,
# bias gradients out of the same pass); the weight-gradient products stay per layer.  NEUMAN_TRAIN_FUSED_BWD=0: the GEMM chain.
FUSED_BACKWARD = os.environ.get('NEUMAN_TRAIN_FUSED_BWD', '1') != '0'
# ... and what the two keep between them as fp16 instead of float32 wherever it is only an operand of a weight-gradient product (the trunk's
# activations: nm_mlp_forward_save16; dZ of every layer: nm_mlp_backward_chain16; the encoded position: nm_pe_encode16), consumed by nm_wgrad16 --
# half the bytes of the HBM-bound kernels of an iteration, one MFMA per product instead of three.  The 2^-12 roundings of independent samples
# average out under the sum over the batch, so the form is taken from STORE16_MIN_ROWS samples on (the trainers' batches are 10^5 .. 10^6);
# smaller calls keep the float32 copies.  NEUMAN_TRAIN_STORE16=0: float32 copies always.
STORE16 = os.environ.get('NEUMAN_TRAIN_STORE16', '1') != '0'
STORE16_MIN_ROWS = int(os.environ.get('NEUMAN_TRAIN_STORE16_MIN', '32768'))


def _gemm(a_kmajor, b_kmajor, M, N, K, A, lda, B, ldb, C, ldc, bias=None, mask=None, ldmask=0, flags=0, ws=None, precision=None):
    prec = precision or GEMM_PRECISION
    if prec == 'mixed16':
        prec = 'fp16x3' if (a_kmajor, b_kmajor) == (0, 0) else 'bf16x3'        # (0, 0) = the forward products Z = A W^T
    fn = {'f32': _lib.lib().nm_gemm_f32, 'bf16x3': _lib.lib().nm_gemm_bf16x3, 'fp16x3': _lib.lib().nm_gemm_fp16x3}[prec]
    _lib.check(fn(a_kmajor, b_kmajor, M, N, K, _lib.dev_ptr(A), lda, _lib.dev_ptr(B), ldb, _lib.dev_ptr(C), ldc,
                                      _lib.dev_ptr(bias), _lib.dev_ptr(mask), ldmask, flags, _lib.dev_ptr(ws), 0 if ws is None else ws.numel(),
                                      _lib.stream_ptr()), "nm_gemm")


def _pad_cols(w, cols):
    """[rows, c] -> [rows, cols] zero-padded, contiguous f32"""
    out = torch.zeros((w.shape[0], cols), device=w.device, dtype=torch.float32)
    out[:, :w.shape[1]] = w
    return out


def _pad_rows(w, rows, at=0):
    out = torch.zeros((rows, w.shape[1]), device=w.device, dtype=torch.float32)
    out[at:at + w.shape[0]] = w
    return out


class _Packed:
    """The parameters in the shapes the products want (reference layout [out, in], inputs padded to multiples of 4) -- each made on first
    use: the fused forward / backward read the live parameters themselves and need a handful of these at most."""

    def __init__(self, nerf, n_pos, n_dir):
        self.nerf = nerf
        self.n_pos, self.n_dir = n_pos, n_dir                               # 63 / 84, 27
        self.kp, self.kd = (n_pos + 3) // 4 * 4, (n_dir + 3) // 4 * 4       # 64 / 84, 28
        self.n_layers = len(nerf.pts_linears)
        self.skip = [i > 0 and lin.weight.shape[1] == nerf.width + n_pos for i, lin in enumerate(nerf.pts_linears)]   # the layer after the skip: cat([x_pe, h]) (vanilla.py:130)
        if not nerf.use_viewdirs:
            self.n_out = nerf.output_linear.weight.shape[0]                  # output_linear, <= 4 outputs (vanilla.py:117, 150)
        self._c = {}

    def _get(self, key, make):
        if key not in self._c:
            with torch.no_grad():
                self._c[key] = make()
        return self._c[key]

    def Wpe(self, i):
        """[width, kp]: the encoded-position columns of layer 0 / of the skip layer, zero-padded"""
        return self._get(('pe', i), lambda: _pad_cols(self.nerf.pts_linears[i].weight.detach().float()[:, :self.n_pos], self.kp))

    def Wh(self, i):
        """[width, width]: the hidden columns of layer i >= 1"""
        w = self.nerf.pts_linears[i].weight
        return self._get(('h', i), lambda: w.detach().float()[:, self.n_pos:].contiguous() if self.skip[i] else w.detach().float().contiguous())

    def b(self, i):
        return self._get(('b', i), lambda: self.nerf.pts_linears[i].bias.detach().float().contiguous())

    @property
    def W(self):                                                             # the GEMM chain's view: per layer (W,) or, after the skip, (W_pe, W_hidden)
        return [(self.Wpe(i),) if i == 0 else ((self.Wpe(i), self.Wh(i)) if self.skip[i] else (self.Wh(i),)) for i in range(self.n_layers)]

    @property
    def Wv(self):
        n = self.nerf
        return self._get('Wv', lambda: (n.views_linears[0].weight.detach().float()[:, :n.width].contiguous(),       # cat([feature, d_pe]) (vanilla.py:139)
                                        _pad_cols(n.views_linears[0].weight.detach().float()[:, n.width:], self.kd)))

    @property
    def bv(self):
        return self._get('bv', lambda: self.nerf.views_linears[0].bias.detach().float().contiguous())

    @property
    def Wf(self):
        return self._get('Wf', lambda: self.nerf.feature_linear.weight.detach().float().contiguous())

    @property
    def bf(self):
        return self._get('bf', lambda: self.nerf.feature_linear.bias.detach().float().contiguous())

    @property
    def Wa4(self):
        return self._get('Wa4', lambda: _pad_rows(self.nerf.alpha_linear.weight.detach().float(), 4, at=3))           # raw[:, 3] = sigma

    @property
    def Wr4(self):
        return self._get('Wr4', lambda: _pad_rows(self.nerf.rgb_linear.weight.detach().float(), 4))                   # raw[:, :3] = rgb

    @property
    def b4(self):
        return self._get('b4', lambda: torch.cat([self.nerf.rgb_linear.bias.detach().float(), self.nerf.alpha_linear.bias.detach().float()]).contiguous())

    @property
    def Wo4(self):
        return self._get('Wo4', lambda: _pad_rows(self.nerf.output_linear.weight.detach().float(), 4))

    @property
    def bo4(self):
        def make():
            out = torch.zeros(4, device=self.nerf.output_linear.weight.device)
            out[:self.n_out] = self.nerf.output_linear.bias.detach().float()
            return out
        return self._get('bo4', make)


def _pe_table(emb, dev):
    """the encoding's frequency table on the device, uploaded once per embedder (a pageable host-to-device copy per call waits for the queue to drain)"""
    cache = emb.__dict__.setdefault('_dev_tables', {})
    key = str(dev)
    if key not in cache:
        cache[key] = torch.from_numpy(emb.table()).to(dev).contiguous()
    return cache[key]


def upload(array, dev, dtype=torch.float32):
    """a small host array -> device tensor through pinned memory, without waiting for the device (what changes every iteration: a frame time's encoding,
    the pixels of the canonical rays)"""
    t = torch.as_tensor(array)
    if t.dtype != dtype:
        t = t.to(dtype)
    return t.contiguous().pin_memory().to(dev, non_blocking=True)


def _encode(emb, x, ld):
    dev = x.device
    tab = _pe_table(emb, dev)
    out = torch.empty((x.shape[0], ld), device=dev, dtype=torch.float32)
    _lib.check(_lib.lib().nm_pe_encode(_lib.dev_ptr(x), x.shape[0], x.shape[1], PE_KINDS[emb.mapping], emb.N_freqs, _lib.dev_ptr(tab),
                                       _lib.dev_ptr(out), ld, _lib.stream_ptr()), "nm_pe_encode")
    return out


def _encode16(emb, x, ld=64, ones_col=-1):
    """the encoding as fp16 of 32 x value, rows of `ld` (an operand of nm_wgrad16); ones_col: a padding column that holds 1 instead of 0"""
    dev = x.device
    tab = _pe_table(emb, dev)
    out = torch.empty((x.shape[0], ld), device=dev, dtype=torch.float16)
    _lib.check(_lib.lib().nm_pe_encode16(_lib.dev_ptr(x), x.shape[0], x.shape[1], PE_KINDS[emb.mapping], emb.N_freqs, _lib.dev_ptr(tab),
                                         ctypes.c_void_p(out.data_ptr()), ld, ones_col, _lib.stream_ptr()), "nm_pe_encode16")
    return out


def _encode_backward(emb, x, g):
    dev = x.device
    tab = _pe_table(emb, dev)
    dx = torch.empty_like(x)
    _lib.check(_lib.lib().nm_pe_backward(_lib.dev_ptr(x), x.shape[0], x.shape[1], PE_KINDS[emb.mapping], emb.N_freqs, _lib.dev_ptr(tab),
                                         _lib.dev_ptr(g), g.shape[1], _lib.dev_ptr(dx), _lib.stream_ptr()), "nm_pe_backward")
    return dx


def _pad4(x):
    n = x.shape[0]
    if n % 4 == 0 and x.dtype == torch.float32 and x.is_contiguous():
        return x.detach()                                                  # (the trainers' batches: nothing to pad, no fill + copy)
    out = torch.zeros(((n + 3) // 4 * 4, x.shape[1]), device=x.device, dtype=torch.float32)
    out[:n] = x
    return out


def _input_versions(*xs):
    """_pad4 hands back an ALIAS of the caller's tensor when nothing needs padding (detach() shares the version counter): the versions are
    recorded at forward and compared at backward, so an in-place edit of pts / dirs between the two passes (`can_pts += offset`) is an error
    instead of a silently wrong input gradient -- what ctx.save_for_backward would have checked had the tensors gone through it"""
    return tuple(None if x is None else x._version for x in xs)


def _check_input_versions(ctx, *xs):
    if getattr(ctx, 'in_versions', None) is not None and _input_versions(*xs) != ctx.in_versions:
        raise _lib.NeumanHipError("an input of the net (points / directions) was modified in place between the forward and the backward pass of a "
                                  "training step: the backward pass reads the tensor the forward was given (clone it before editing)")


def _fused_ok(net):
    nerf = net.nerf
    return (FUSED_FORWARD and GEMM_PRECISION == 'mixed16' and hasattr(net, 'train_handle') and nerf.use_viewdirs and net.pos_pe.input_dims == 3
            and nerf.depth == 8 and nerf.width == 256 and list(nerf.skips) == [4] and getattr(nerf, 'scale_type', 'no') == 'no'
            and net.pos_pe.mapping == net.dir_pe.mapping
            and all(p.dtype == torch.float32 and p.is_contiguous() and p.is_cuda for p in nerf.ordered_params()))


class _MLP(torch.autograd.Function):
    """net = Joiner (use_viewdirs: pts, dirs -> [rgb, sigma]) or OffsetNet (dirs is None: x -> output_linear)."""

    @staticmethod
    def forward(ctx, net, pts, dirs, *params):
        # net may be a pair (net, 'feature'): the call then also returns feature_linear's output [n, 256] as a differentiable tensor, and a
        # gradient that reaches it is added to the kernel's own d_feat in the backward pass (two_views below)
        want_feat = isinstance(net, tuple)
        if want_feat:
            net = net[0]
        nerf = net.nerf
        dev = pts.device
        n = pts.shape[0]
        width, half = nerf.width, nerf.width // 2
        views = nerf.use_viewdirs
        pk = _Packed(nerf, net.pos_pe.out_dim, net.dir_pe.out_dim if views else 0)
        p4 = _pad4(pts)
        n4 = p4.shape[0]
        fused = _fused_ok(net) and dirs is not None
        use16 = fused and STORE16 and FUSED_BACKWARD and n4 >= STORE16_MIN_ROWS and pk.kp <= 64
        X0 = None if use16 else _encode(net.pos_pe, p4, pk.kp)
        if fused:
            d4 = _pad4(dirs)
            D0 = None if use16 else _encode(net.dir_pe, d4, pk.kd)
            handle = net.train_handle()
            plist = nerf.ordered_params()
            ptrs = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in plist])
            _lib.check(_lib.lib().nm_mlp_refresh_f16(handle, ptrs, _lib.stream_ptr()), "nm_mlp_refresh_f16")
            hv = torch.empty((n4, half), device=dev, dtype=torch.float32)
            raw = torch.empty((n4, 4), device=dev, dtype=torch.float32)
            bits = torch.empty((8, n4, 8), device=dev, dtype=torch.int32) if FUSED_BACKWARD else None     # (what the backward-data chain masks with)
            ctx.h16 = ctx.x0h = None
            if use16:
                h16 = torch.empty((8, n4, width), device=dev, dtype=torch.float16)
                feat16 = torch.empty((n4, width), device=dev, dtype=torch.float16)
                hvbits = torch.empty((n4, 4), device=dev, dtype=torch.int32)
                # the encodings as fp16 operands of the weight-gradient products, written by the forward kernel from its own encoding buffers; the
                # view encoding's last padding column holds 1: its product with d_hv is the views layer's bias gradient
                x0h = torch.empty((n4, 64), device=dev, dtype=torch.float16)
                d0h = torch.empty((n4, 64), device=dev, dtype=torch.float16)
                feat32 = torch.empty((n4, width), device=dev, dtype=torch.float32) if want_feat else None
                _lib.check(_lib.lib().nm_mlp_forward_save16(handle, _lib.dev_ptr(p4), _lib.dev_ptr(d4), n4, ctypes.c_void_p(h16.data_ptr()), _lib.dev_ptr(feat32),
                                                            ctypes.c_void_p(feat16.data_ptr()), _lib.dev_ptr(hv), ctypes.c_void_p(bits.data_ptr()),
                                                            ctypes.c_void_p(hvbits.data_ptr()), ctypes.c_void_p(x0h.data_ptr()), ctypes.c_void_p(d0h.data_ptr()),
                                                            _lib.dev_ptr(raw), _lib.stream_ptr()), "nm_mlp_forward_save16")
                ctx.h16, ctx.x0h, ctx.d0h, ctx.feat16, ctx.hvbits = h16, x0h, d0h, feat16, hvbits
                acts = H = feat = D0 = None
            elif want_feat:
                raise _lib.NeumanHipError("the feature output of a training forward exists in the fp16-storage form only (train.two_views_ok)")
            else:
                acts = torch.empty((9, n4, width), device=dev, dtype=torch.float32)
                _lib.check(_lib.lib().nm_mlp_forward_save_bits(handle, _lib.dev_ptr(p4), _lib.dev_ptr(d4), n4, _lib.dev_ptr(acts), _lib.dev_ptr(hv),
                                                               ctypes.c_void_p(bits.data_ptr() if bits is not None else 0), _lib.dev_ptr(raw),
                                                               _lib.stream_ptr()), "nm_mlp_forward_save_bits")
                H, feat = [acts[i] for i in range(8)], acts[8]
            ctx.pk, ctx.X0, ctx.D0, ctx.H, ctx.feat, ctx.hv, ctx.n, ctx.net = pk, X0, D0, H, feat, hv, n, net
            ctx.p4, ctx.d4 = p4, d4
            ctx.in_versions = _input_versions(p4, d4)
            ctx.acts, ctx.bits = acts, bits
            ctx.versions = [p._version for p in plist]          # the backward pass repacks W^T from the live parameters: they must still be these
            ctx.want_feat = want_feat
            if want_feat:
                ctx.set_materialize_grads(False)
                return raw[:n], feat32[:n]
            return raw[:n]
        if want_feat:
            raise _lib.NeumanHipError("the feature output of a training forward exists in the fused fp16-storage form only (train.two_views_ok)")
        H = []
        h, kh = X0, pk.kp
        for i, Ws in enumerate(pk.W):
            o = torch.empty((n4, width), device=dev, dtype=torch.float32)
            if len(Ws) == 2:                                                     # skip layer: x_pe W_a^T, then + h W_b^T + b, ReLU
                _gemm(0, 0, n4, width, pk.kp, X0, pk.kp, Ws[0], pk.kp, o, width)
                _gemm(0, 0, n4, width, width, h, width, Ws[1], width, o, width, bias=pk.b(i), flags=ACC | BIAS | RELU)
            else:
                _gemm(0, 0, n4, width, kh, h, kh, Ws[0], kh, o, width, bias=pk.b(i), flags=BIAS | RELU)
            H.append(o)
            h, kh = o, width
        raw = torch.empty((n4, 4), device=dev, dtype=torch.float32)
        d4 = D0 = feat = hv = None
        if views:
            d4 = _pad4(dirs)
            D0 = _encode(net.dir_pe, d4, pk.kd)
            _gemm(0, 0, n4, 4, width, h, width, pk.Wa4, width, raw, 4, bias=pk.b4, flags=BIAS)         # sigma (+ all four biases)
            feat = torch.empty((n4, width), device=dev, dtype=torch.float32)
            _gemm(0, 0, n4, width, width, h, width, pk.Wf, width, feat, width, bias=pk.bf, flags=BIAS)
            hv = torch.empty((n4, half), device=dev, dtype=torch.float32)
            _gemm(0, 0, n4, half, width, feat, width, pk.Wv[0], width, hv, half)
            _gemm(0, 0, n4, half, pk.kd, D0, pk.kd, pk.Wv[1], pk.kd, hv, half, bias=pk.bv, flags=ACC | BIAS | RELU)
            _gemm(0, 0, n4, 4, half, hv, half, pk.Wr4, half, raw, 4, flags=ACC)                           # + rgb
        else:
            _gemm(0, 0, n4, 4, width, h, width, pk.Wo4, width, raw, 4, bias=pk.bo4, flags=BIAS)
        ctx.pk, ctx.X0, ctx.D0, ctx.H, ctx.feat, ctx.hv, ctx.n, ctx.net = pk, X0, D0, H, feat, hv, n, net
        ctx.p4, ctx.d4 = p4, d4
        ctx.in_versions = _input_versions(p4, d4)
        ctx.h16 = ctx.x0h = ctx.versions = None
        return raw[:n] if views else raw[:n, :pk.n_out]

    @staticmethod
    def backward(ctx, g_raw, g_feat=None):
        pk, X0, D0, H, feat, hv, n, net = ctx.pk, ctx.X0, ctx.D0, ctx.H, ctx.feat, ctx.hv, ctx.n, ctx.net
        nerf = net.nerf
        if g_raw is None:                                                        # (the feature output alone was used)
            g_raw = torch.zeros((n, 4), device=ctx.p4.device, dtype=torch.float32)
        views = nerf.use_viewdirs
        want_in = ctx.needs_input_grad[1] or (views and ctx.needs_input_grad[2])
        dev = ctx.p4.device
        _check_input_versions(ctx, ctx.p4, ctx.d4)
        n4, width, half = ctx.p4.shape[0], nerf.width, nerf.width // 2
        if getattr(ctx, 'versions', None) is not None and (FUSED_BACKWARD or ctx.h16 is not None):
            # nm_mlp_backward_chain repacks W^T from the LIVE parameters while the masks and the weight-gradient operands are the forward's:
            # an in-place edit between the two passes would make the halves disagree silently
            now = [p._version for p in nerf.ordered_params()]
            if now != ctx.versions:
                raise _lib.NeumanHipError("a parameter of the net was modified in place between the forward and the backward pass of a training step "
                                          "(the fused backward reads the live weights): run backward before optimizer.step() / weight edits")
        if n == n4 and g_raw.shape[1] == 4 and g_raw.dtype == torch.float32 and g_raw.is_contiguous() and (g_raw.data_ptr() & 15) == 0:
            d_raw = g_raw
        else:
            d_raw = torch.zeros((n4, 4), device=dev, dtype=torch.float32)
            d_raw[:n, :g_raw.shape[1]] = g_raw
        if getattr(ctx, 'h16', None) is not None:
            return _backward16(ctx, d_raw, want_in, g_feat)
        ws = [torch.empty(4, device=dev, dtype=torch.float32)]

        def workspace(m, k):                                                     # split-K partials, grown to the largest product
            need = int(_lib.lib().nm_gemm_workspace_floats(m, k, n4))
            if need > ws[0].numel():
                ws[0] = torch.empty(need, device=dev, dtype=torch.float32)
            return ws[0]

        def wgrad(dz, m, a, k):                                                  # dW [m,k] = dz^T a, a reduction over the n4 samples
            out = torch.empty((m, k), device=dev, dtype=torch.float32)
            _gemm(1, 1, m, k, n4, dz, dz.shape[1], a, a.shape[1], out, k, ws=workspace(m, k))
            return out

        def bgrad(dz, m):                                                        # db = column sums of dz (pad rows are zero)
            out = torch.empty(m, device=dev, dtype=torch.float32)
            need = int(_lib.lib().nm_colsum_workspace_floats(n4, m))
            if need > ws[0].numel():
                ws[0] = torch.empty(need, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib().nm_colsum(_lib.dev_ptr(dz), n4, m, dz.shape[1], _lib.dev_ptr(out), _lib.dev_ptr(ws[0]), ws[0].numel(),
                                            _lib.stream_ptr()), "nm_colsum")
            return out

        bands = (n4 + 63) // 64
        cs_buf = torch.empty((bands, width), device=dev, dtype=torch.float32)      # per-64-row column sums out of a product's epilogue

        def band_sum(m):                                                         # -> db [m] = the bands summed (nm_colsum on [bands, m])
            out = torch.empty(m, device=dev, dtype=torch.float32)
            need = int(_lib.lib().nm_colsum_workspace_floats(bands, m))
            if need > ws[0].numel():
                ws[0] = torch.empty(need, device=dev, dtype=torch.float32)
            _lib.check(_lib.lib().nm_colsum(_lib.dev_ptr(cs_buf), bands, m, m, _lib.dev_ptr(out), _lib.dev_ptr(ws[0]), ws[0].numel(),
                                            _lib.stream_ptr()), "nm_colsum")
            return out

        h7 = H[-1]
        dX0 = dD0 = None
        dz = torch.empty((n4, width), device=dev, dtype=torch.float32)
        use_chain = FUSED_BACKWARD and getattr(ctx, 'acts', None) is not None and pk.n_layers == 8 and width == 256
        d_feat = None
        if views:
            g = {}
            gWr4, gb4 = wgrad(d_raw, 4, hv, half), bgrad(d_raw, 4)
            g['rgb_w'], g['rgb_b'], g['alpha_b'] = gWr4[:3], gb4[:3], gb4[3:4]
            d_hv = torch.empty((n4, half), device=dev, dtype=torch.float32)
            _gemm(0, 1, n4, half, 4, d_raw, 4, pk.Wr4, half, d_hv, half, mask=hv, ldmask=half, flags=MASK | COLSUM, ws=cs_buf)
            g['views_b'] = band_sum(half)                                       # column sums of d_hv, out of that product's epilogue
            g['views_w'] = torch.cat([wgrad(d_hv, half, feat, width), wgrad(d_hv, half, D0, pk.kd)[:, :pk.n_dir]], 1)
            if want_in:                                                          # gradient of the encoded view direction
                dD0 = torch.empty((n4, pk.kd), device=dev, dtype=torch.float32)
                _gemm(0, 1, n4, pk.kd, half, d_hv, half, pk.Wv[1], pk.kd, dD0, pk.kd)
            d_feat = torch.empty((n4, width), device=dev, dtype=torch.float32)
            _gemm(0, 1, n4, width, half, d_hv, half, pk.Wv[0], width, d_feat, width, flags=COLSUM, ws=cs_buf)
            g['feature_b'] = band_sum(width)
            g['feature_w'] = wgrad(d_feat, width, h7, width)
            g['alpha_w'] = wgrad(d_raw, 4, h7, width)[3:4]
            if not use_chain:                                                   # (the chain kernel's first stage forms dZ_7 itself)
                _gemm(0, 1, n4, width, width, d_feat, width, pk.Wf, width, dz, width)
                _gemm(0, 1, n4, width, 4, d_raw, 4, pk.Wa4, width, dz, width, mask=h7, ldmask=width, flags=ACC | MASK | COLSUM, ws=cs_buf)
            head = [g['views_w'].contiguous(), g['views_b'].contiguous(), g['feature_w'], g['feature_b'].contiguous(),
                    g['alpha_w'].contiguous(), g['alpha_b'].contiguous(), g['rgb_w'].contiguous(), g['rgb_b'].contiguous()]
        else:
            head = [wgrad(d_raw, 4, h7, width)[:pk.n_out].contiguous(), bgrad(d_raw, 4)[:pk.n_out].contiguous()]
            _gemm(0, 1, n4, width, 4, d_raw, 4, pk.Wo4, width, dz, width, mask=h7, ldmask=width, flags=MASK | COLSUM, ws=cs_buf)
        gw, gb = [None] * pk.n_layers, [None] * pk.n_layers
        chain = None
        if use_chain:
            from_feat = views and d_feat is not None
            ns = 8 if from_feat else 7
            if not from_feat:
                gb[7] = band_sum(width)                                          # of dz_7: left in cs_buf by the product that made it
            chain = torch.empty((ns, n4, width), device=dev, dtype=torch.float32)    # dZ of layers (7,) 6 .. 0
            gbs = torch.empty((ns, width), device=dev, dtype=torch.float32)
            need = int(_lib.lib().nm_mlp_backward_chain_workspace_floats(n4))
            if need > ws[0].numel():
                ws[0] = torch.empty(need, device=dev, dtype=torch.float32)
            ptrs = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in nerf.ordered_params()])
            _lib.check(_lib.lib().nm_mlp_backward_chain(net.train_handle(), ptrs, _lib.dev_ptr(None if from_feat else dz), _lib.dev_ptr(d_feat if from_feat else None),
                                                        _lib.dev_ptr(d_raw if from_feat else None), _lib.dev_ptr(ctx.acts),
                                                        ctypes.c_void_p(ctx.bits.data_ptr() if ctx.bits is not None else 0), n4, _lib.dev_ptr(chain),
                                                        _lib.dev_ptr(gbs), _lib.dev_ptr(ws[0]), ws[0].numel(), _lib.stream_ptr()), "nm_mlp_backward_chain")
            off = 1 if from_feat else 0                                               # chain[off + 6 - i] = dZ_i for i <= 6
            if from_feat:
                dz, gb[7] = chain[0], gbs[0]
            for i in range(7):
                gb[i] = gbs[off + 6 - i]
        W_all = pk.W
        for i in range(pk.n_layers - 1, -1, -1):
            Ws = W_all[i]
            if chain is None:
                gb[i] = band_sum(width)                                          # of dz: left in cs_buf by the product that made it
            elif i < 7:
                dz = chain[off + 6 - i]
            if want_in and (i == 0 or len(Ws) == 2):                             # gradient of the encoded position: both layers it feeds
                first = dX0 is None
                if first:
                    dX0 = torch.empty((n4, pk.kp), device=dev, dtype=torch.float32)
                _gemm(0, 1, n4, pk.kp, width, dz, width, Ws[0], pk.kp, dX0, pk.kp, flags=0 if first else ACC)
            if i == 0:
                gw[i] = wgrad(dz, width, X0, pk.kp)[:, :pk.n_pos]
                break
            prev = H[i - 1]
            if len(Ws) == 2:
                gw[i] = torch.cat([wgrad(dz, width, X0, pk.kp)[:, :pk.n_pos], wgrad(dz, width, prev, width)], 1)
                Wb = Ws[1]
            else:
                gw[i] = wgrad(dz, width, prev, width)
                Wb = Ws[0]
            if chain is None:
                nz = torch.empty((n4, width), device=dev, dtype=torch.float32)
                _gemm(0, 1, n4, width, width, dz, width, Wb, width, nz, width, mask=prev, ldmask=width, flags=MASK | COLSUM, ws=cs_buf)
                dz = nz
        grads = []
        for i in range(pk.n_layers):
            grads += [gw[i].contiguous(), gb[i].contiguous()]
        grads += head
        d_pts = _encode_backward(net.pos_pe, ctx.p4, dX0)[:n] if want_in else None
        d_dirs = _encode_backward(net.dir_pe, ctx.d4, dD0)[:n] if (want_in and views) else None
        return (None, d_pts, d_dirs) + tuple(grads)


def _products16(n4, amax, grow, p_cols, q_cols, items):
    """nm_wgrad16 over `items` = (dZ16 rows [n4, p_cols], activation16 rows [n4, q_cols or 64], gradient tensor, column offset into it)"""
    lib = _lib.lib()
    k = len(items)
    P = (ctypes.c_void_p * k)(*[a.data_ptr() for a, _, _, _ in items])
    Q = (ctypes.c_void_p * k)(*[b.data_ptr() for _, b, _, _ in items])
    C = (ctypes.c_void_p * k)(*[c.data_ptr() + 4 * off for _, _, c, off in items])
    L = (ctypes.c_int * k)(*[c.shape[1] for _, _, c, _ in items])
    w_ = grow(lib.nm_wgrad16_workspace_floats(k, n4, p_cols, q_cols))
    _lib.check(lib.nm_wgrad16(k, p_cols, q_cols, P, Q, C, L, n4, _lib.dev_ptr(amax), _lib.dev_ptr(w_), w_.numel(), _lib.stream_ptr()), "nm_wgrad16")


class _MLPPlain16(torch.autograd.Function):
    """A plain-head net with a 3-D position encoding -- the 8 x 256 trunk and output_linear [4][256] -- on the fused 16-bit training kernels, its 18
    parameter tensors GIVEN (they need not be anybody's nn.Parameters: the offset net's folded weights are functions of its parameters that autograd
    differentiates on its own, offset_forward_train).  `joiner` supplies the encoding's description and the kernel handle.  pts [n, 3] -> [n, 4]."""

    @staticmethod
    def forward(ctx, joiner, pts, *params):
        lib = _lib.lib()
        dev, n = pts.device, pts.shape[0]
        p4 = _pad4(pts)
        n4 = p4.shape[0]
        plist = [p.detach().to(torch.float32).contiguous() for p in params]
        ptrs = (ctypes.c_void_p * 24)(*([p.data_ptr() for p in plist] + [None] * 6))
        handle = joiner.train_handle()
        _lib.check(lib.nm_mlp_refresh_f16(handle, ptrs, _lib.stream_ptr()), "nm_mlp_refresh_f16")
        h16 = torch.empty((8, n4, 256), device=dev, dtype=torch.float16)
        bits = torch.empty((8, n4, 8), device=dev, dtype=torch.int32)
        x0h = torch.empty((n4, 64), device=dev, dtype=torch.float16)
        raw = torch.empty((n4, 4), device=dev, dtype=torch.float32)
        _lib.check(lib.nm_mlp_forward_save16(handle, _lib.dev_ptr(p4), None, n4, ctypes.c_void_p(h16.data_ptr()), None, None, None, ctypes.c_void_p(bits.data_ptr()), None,
                                             ctypes.c_void_p(x0h.data_ptr()), None, _lib.dev_ptr(raw), _lib.stream_ptr()), "nm_mlp_forward_save16 (plain head)")
        ctx.joiner, ctx.plist, ctx.h16, ctx.bits, ctx.x0h, ctx.n, ctx.n4 = joiner, plist, h16, bits, x0h, n, n4
        ctx.n_pos = joiner.pos_pe.out_dim
        return raw[:n]

    @staticmethod
    def backward(ctx, g_raw):
        lib = _lib.lib()
        if ctx.needs_input_grad[1]:
            raise _lib.NeumanHipError("the fused plain-head path gives no gradient to its input points (the offset net's are not differentiated)")
        h16, n, n4, n_pos, plist = ctx.h16, ctx.n, ctx.n4, ctx.n_pos, ctx.plist
        dev = g_raw.device
        if n == n4 and g_raw.dtype == torch.float32 and g_raw.is_contiguous() and (g_raw.data_ptr() & 15) == 0:
            d_out = g_raw
        else:
            d_out = torch.zeros((n4, 4), device=dev, dtype=torch.float32)
            d_out[:n] = g_raw
        ws = [torch.empty(4, device=dev, dtype=torch.float32)]

        def grow(need):
            if need > ws[0].numel():
                ws[0] = torch.empty(int(need), device=dev, dtype=torch.float32)
            return ws[0]
        amax = torch.zeros(1, device=dev, dtype=torch.float32)
        head = torch.empty(1028, device=dev, dtype=torch.float32)           # output_linear's [4][256] and its four bias sums
        w = grow(lib.nm_wgrad_out16_workspace_floats(n4))
        _lib.check(lib.nm_wgrad_out16(_lib.dev_ptr(d_out), ctypes.c_void_p(h16[7].data_ptr()), n4, _lib.dev_ptr(head), _lib.dev_ptr(amax), _lib.dev_ptr(w), w.numel(),
                                      _lib.stream_ptr()), "nm_wgrad_out16")
        dz16 = torch.empty((8, n4, 256), device=dev, dtype=torch.float16)
        gbs = torch.empty((8, 256), device=dev, dtype=torch.float32)
        w = grow(lib.nm_mlp_backward_chain_workspace_floats(n4))
        ptrs = (ctypes.c_void_p * 24)(*([p.data_ptr() for p in plist] + [None] * 6))
        _lib.check(lib.nm_mlp_backward_plain16(ctx.joiner.train_handle(), ptrs, _lib.dev_ptr(d_out), ctypes.c_void_p(ctx.bits.data_ptr()), n4, _lib.dev_ptr(amax),
                                               ctypes.c_void_p(dz16.data_ptr()), None, None, _lib.dev_ptr(gbs), _lib.dev_ptr(w), w.numel(), _lib.stream_ptr()),
                   "nm_mlp_backward_plain16")
        skip = [i > 0 and plist[2 * i].shape[1] == 256 + n_pos for i in range(8)]
        gw = [torch.empty((256, (n_pos if i == 0 else 256) + (n_pos if skip[i] else 0)), device=dev, dtype=torch.float32) for i in range(8)]
        _products16(n4, amax, grow, 256, 256, [(dz16[7 - i], h16[i - 1], gw[i], n_pos if skip[i] else 0) for i in range(7, 0, -1)])
        _products16(n4, amax, grow, 256, n_pos, [(dz16[7 - i], ctx.x0h, gw[i], 0) for i in range(8) if i == 0 or skip[i]])
        grads = []
        for i in range(8):
            grads += [gw[i], gbs[7 - i]]
        grads += [head[:1024].view(4, 256), head[1024:1028]]
        return (None, None) + tuple(grads)


def _backward16(ctx, d_raw, want_in, g_feat=None):
    """The backward pass of a step whose forward kept fp16 copies (STORE16): ONE kernel for the whole backward-data pass from d_raw
    (nm_mlp_backward_net16: the views layer's adjoint, feature_linear's, the eight trunk layers'; dZ of every layer, d_feat and d_hv out as fp16),
    then the weight gradients as batched fp16 products (nm_wgrad16): the eight 256 x 256 ones in one launch, the encoded-position columns of layer 0
    and of the skip layer in a second, the views layer's two blocks in a third and fourth -- the last one against the encoded direction whose
    padding column holds 1, so that it carries the views layer's bias gradient as its 64th column.  What stays on the float32 kernels: the
    4-row heads (rgb_linear's weight, two bias sums) and, when the inputs want gradients, the three products behind d pts / d dirs."""
    pk, net, n = ctx.pk, ctx.net, ctx.n
    nerf = net.nerf
    lib = _lib.lib()
    dev = d_raw.device
    n4, width, half = d_raw.shape[0], nerf.width, nerf.width // 2
    h16, x0h, d0h, feat16, hv, n_pos, n_dir = ctx.h16, ctx.x0h, ctx.d0h, ctx.feat16, ctx.hv, pk.n_pos, pk.n_dir
    ws = [torch.empty(4, device=dev, dtype=torch.float32)]

    def grow(need):
        if need > ws[0].numel():
            ws[0] = torch.empty(int(need), device=dev, dtype=torch.float32)
        return ws[0]
    # ---- the 4-row heads in one pass over d_raw, H_7's fp16 copy and hv: alpha_linear's and rgb_linear's weight gradients, their bias gradients
    # (column sums of d_raw) and max |d_raw| -> the scale of every fp16 copy below
    amax = torch.zeros(1, device=dev, dtype=torch.float32)
    heads = torch.empty(644, device=dev, dtype=torch.float32)
    w = grow(lib.nm_wgrad_heads16_workspace_floats(n4))
    _lib.check(lib.nm_wgrad_heads16(_lib.dev_ptr(d_raw), ctypes.c_void_p(h16[7].data_ptr()), _lib.dev_ptr(hv), n4, _lib.dev_ptr(heads), _lib.dev_ptr(amax), _lib.dev_ptr(w),
                                    w.numel(), _lib.stream_ptr()), "nm_wgrad_heads16")
    alpha_w, rgb_w, rgb_b, alpha_b = heads[:256].view(1, 256), heads[256:640].view(3, half), heads[640:643], heads[643:644]
    # ---- the backward-data pass
    dz16 = torch.empty((8, n4, width), device=dev, dtype=torch.float16)    # dZ_7 .. dZ_0 (x scale, k-slot order)
    dfeat16 = torch.empty((n4, width), device=dev, dtype=torch.float16)
    dhv16 = torch.empty((n4, half), device=dev, dtype=torch.float16)
    dz32 = torch.empty((2, n4, width), device=dev, dtype=torch.float32) if want_in else None      # layers 5, 0: the input gradient's products
    dhv32 = torch.empty((n4, half), device=dev, dtype=torch.float32) if want_in else None
    gbs = torch.empty((9, width), device=dev, dtype=torch.float32)
    w = grow(lib.nm_mlp_backward_chain_workspace_floats(n4))
    ptrs = (ctypes.c_void_p * 24)(*[p.data_ptr() for p in nerf.ordered_params()])
    feat_add = None
    if g_feat is not None:                                                 # a second evaluation of the views head on these features (two_views): its share of d_feat
        if g_feat.shape[0] == n4 and g_feat.dtype == torch.float32 and g_feat.is_contiguous() and (g_feat.data_ptr() & 15) == 0:
            feat_add = g_feat
        else:
            feat_add = torch.zeros((n4, width), device=dev, dtype=torch.float32)
            feat_add[:g_feat.shape[0]] = g_feat
    _lib.check(lib.nm_mlp_backward_net16(net.train_handle(), ptrs, _lib.dev_ptr(d_raw), _lib.dev_ptr(feat_add), ctypes.c_void_p(ctx.bits.data_ptr()), ctypes.c_void_p(ctx.hvbits.data_ptr()), n4,
                                         _lib.dev_ptr(amax), ctypes.c_void_p(dz16.data_ptr()), ctypes.c_void_p(dfeat16.data_ptr()), ctypes.c_void_p(dhv16.data_ptr()),
                                         _lib.dev_ptr(dz32[0] if want_in else None), _lib.dev_ptr(dz32[1] if want_in else None), _lib.dev_ptr(dhv32),
                                         _lib.dev_ptr(gbs), _lib.dev_ptr(w), w.numel(), _lib.stream_ptr()), "nm_mlp_backward_net16")
    # ---- the weight gradients
    gw = [torch.empty((width, (n_pos if i == 0 else width) + (n_pos if pk.skip[i] else 0)), device=dev, dtype=torch.float32) for i in range(8)]
    feature_w = torch.empty((width, width), device=dev, dtype=torch.float32)
    views_w = torch.empty((half, width + n_dir), device=dev, dtype=torch.float32)
    views_x = torch.empty((half, 64), device=dev, dtype=torch.float32)      # [:, :n_dir] the encoded-direction columns, [:, 63] the bias gradient

    def products(p_cols, q_cols, items):
        _products16(n4, amax, grow, p_cols, q_cols, items)
    products(width, width, [(dfeat16, h16[7], feature_w, 0)] + [(dz16[7 - i], h16[i - 1], gw[i], n_pos if pk.skip[i] else 0) for i in range(7, 0, -1)])
    products(width, n_pos, [(dz16[7 - i], x0h, gw[i], 0) for i in range(8) if i == 0 or pk.skip[i]])
    products(half, width, [(dhv16, feat16, views_w, 0)])
    products(half, 64, [(dhv16, d0h, views_x, 0)])
    views_w[:, width:] = views_x[:, :n_dir]
    grads = []
    for i in range(8):
        grads += [gw[i], gbs[7 - i]]
    grads += [views_w, views_x[:, 63].contiguous(), feature_w, gbs[8], alpha_w, alpha_b, rgb_w, rgb_b]
    d_pts = d_dirs = None
    if want_in:                                                            # the encoded inputs' gradients: the layers they feed, then the encodings' adjoints
        dX0 = torch.empty((n4, pk.kp), device=dev, dtype=torch.float32)
        first = True
        for i, slot in ((5, 0), (0, 1)):
            if i == 0 or pk.skip[i]:
                _gemm(0, 1, n4, pk.kp, width, dz32[slot], width, pk.Wpe(i), pk.kp, dX0, pk.kp, flags=0 if first else ACC)
                first = False
        dD0 = torch.empty((n4, pk.kd), device=dev, dtype=torch.float32)
        _gemm(0, 1, n4, pk.kd, half, dhv32, half, pk.Wv[1], pk.kd, dD0, pk.kd)
        d_pts = _encode_backward(net.pos_pe, ctx.p4, dX0)[:n]
        d_dirs = _encode_backward(net.dir_pe, ctx.d4, dD0)[:n]
    return (None, d_pts, d_dirs) + tuple(grads)


def train_params(nerf):
    """the parameters in the order _MLP returns their gradients: pts_linears, then the heads (reference state_dict order)"""
    lins = list(nerf.pts_linears) + ([nerf.views_linears[0], nerf.feature_linear, nerf.alpha_linear, nerf.rgb_linear] if nerf.use_viewdirs
                                     else [nerf.output_linear])
    out = []
    for lin in lins:
        out += [lin.weight, lin.bias]
    return out


def _full_input(net):
    """A net over encodings WITHOUT the raw input (include_input=False, vanilla.py:56-58, 87-88) as one over the full encodings the kernels form:
    -> (shadow net, derived parameters).  The derived tensors are the real parameters with zero columns where the absent inputs would be read
    (torch.cat: autograd slices the kernels' gradients back onto the real parameters); the shadow holds their values for the kernels, which read
    parameter storage.  With include_input=True: (net, its own parameters)."""
    from . import vanilla
    nerf, pos, dpe = net.nerf, net.pos_pe, getattr(net, 'dir_pe', None)
    pads = vanilla.absent_input_columns(pos, dpe, nerf)
    if not pads:
        return net, train_params(nerf)
    dev = nerf.pts_linears[0].weight.device
    cache = net.__dict__.setdefault('_full_input_cache', {})
    if cache.get('dev') != dev:
        def full(pe):
            return vanilla.Embedder(pe.input_dims, pe.max_freq, pe.N_freqs, pe.log_sampling, True, min_freq=pe.min_freq, mapping=pe.mapping)
        fp, fd = full(pos), (full(dpe) if dpe is not None else None)
        body = vanilla.NeRF(depth=nerf.depth, width=nerf.width, input_ch=fp.out_dim, input_ch_views=fd.out_dim if fd is not None else 0, skips=list(nerf.skips),
                            output_ch=4 if nerf.use_viewdirs else nerf.output_linear.out_features, use_viewdirs=nerf.use_viewdirs,
                            scale=getattr(nerf, 'scale', 1.0), scale_type=getattr(nerf, 'scale_type', 'no'))
        shadow = vanilla.Joiner(fp, fd, body) if dpe is not None else vanilla.OffsetNet(fp, body)
        cache.update(dev=dev, shadow=shadow.to(dev).train())
    shadow = cache['shadow']
    real = train_params(nerf)
    derived = vanilla.with_absent_columns(real, pads)
    # the shadow's storage follows the real parameters' VERSIONS: a second forward of the same iteration (the human trainer evaluates a net on several
    # query sets before its one backward pass) must not touch it -- the fused backward checks that the weights it reads are the forward's
    key = tuple((p.data_ptr(), p._version) for p in real)
    if cache.get('key') != key:
        with torch.no_grad():
            for sp, dp in zip(train_params(shadow.nerf), derived):
                sp.copy_(dp)
        cache['key'] = key
    return shadow, derived


def _offset_fused_ok(net, x, n):
    nerf, pe = net.nerf, net.pos_pe
    return (STORE16 and FUSED_FORWARD and FUSED_BACKWARD and GEMM_PRECISION == 'mixed16' and n >= STORE16_MIN_ROWS and not x.requires_grad and x.is_cuda
            and pe.mapping == 'posenc' and pe.input_dims == 4 and nerf.depth == 8 and nerf.width == 256 and list(nerf.skips) == [4] and not nerf.use_viewdirs
            and nerf.output_linear.out_features <= 4 and 3 + 6 * pe.N_freqs <= 64)


def _offset_fused(net, pts, t):
    """OffsetNet's network at ONE time t on the fused kernels.  The time coordinate of every sample of a batch is the same number
    (trainers/human_nerf_trainer.py:258-261: `ones * cur_view_f`), so the 21 encoded-time inputs of layer 0 and of the skip layer add a constant vector
    W[:, time columns] . pe(t) to those layers' pre-activations: with it folded into the bias the net is a plain-head net over the 63-wide 3-D
    encoding.  The folded tensors are built with differentiable torch operations from the real parameters, so autograd carries the fused kernels'
    gradients back to them (the time columns get bias gradient x pe(t), as they must)."""
    from . import vanilla
    nerf, pe = net.nerf, net.pos_pe
    dev = pts.device
    cache = net.__dict__.setdefault('_fused_cache', {})
    if cache.get('dev') != dev:
        sp, tc = vanilla.time_columns(pe)
        pe3 = vanilla.Embedder(3, pe.max_freq, pe.N_freqs, pe.log_sampling, True, min_freq=pe.min_freq, mapping='posenc')
        dpe = vanilla.Embedder(3, 3, 4)
        body = vanilla.NeRF(depth=nerf.depth, width=nerf.width, input_ch=pe3.out_dim, input_ch_views=dpe.out_dim, output_ch=4, skips=list(nerf.skips), use_viewdirs=False)
        cache.update(dev=dev, sp=torch.as_tensor(sp, device=dev), tc=torch.as_tensor(tc, device=dev), joiner=vanilla.Joiner(pe3, dpe, body).to(dev))
    sp, tc = cache['sp'], cache['tc']
    pt = upload(vanilla.time_encoding(pe, t), dev)                   # (the one thing that changes with the frame)
    params = []
    for i, lin in enumerate(nerf.pts_linears):
        W, b = lin.weight, lin.bias
        if i == 0 or (i - 1) in nerf.skips:
            Wf = W[:, sp] if i == 0 else torch.cat([W[:, sp], W[:, pe.out_dim:]], 1)
            params += [Wf, b + (W[:, tc] * pt[None, :]).sum(1)]
        else:
            params += [W, b]
    Wo, bo = nerf.output_linear.weight, nerf.output_linear.bias
    k = Wo.shape[0]
    if k < 4:                                                           # the kernels' head has four rows (r, g, b, sigma): the offset's three + a zero one
        Wo = torch.cat([Wo, torch.zeros((4 - k, Wo.shape[1]), device=dev, dtype=Wo.dtype)], 0)
        bo = torch.cat([bo, torch.zeros(4 - k, device=dev, dtype=bo.dtype)], 0)
    params += [Wo, bo]
    if not pe.include_input:                                            # the folded net has no raw-input columns: zero ones for the kernels' full encoding
        params = vanilla.with_absent_columns(params, {0: (0, 3), **{2 * (s_ + 1): (0, 3) for s_ in nerf.skips}})
    return _MLPPlain16.apply(cache['joiner'], pts, *params)[:, :k]


def offset_forward_train(net, x, const_time=None):
    """OffsetNet.forward (vanilla.py:169-178) with autograd: x [..., 4] (point + time) -> offset [..., 3], before the scale.  const_time: the caller's
    promise that x[..., 3] is this one number for every row (the human trainer's batches are one frame): large batches then run on the fused kernels."""
    _lib.require_gpu()
    shp = x.shape[:-1]
    xf = x.reshape(-1, x.shape[-1]).to(torch.float32).contiguous()
    if const_time is not None and _offset_fused_ok(net, xf, xf.shape[0]):
        return _offset_fused(net, xf[:, :3].contiguous(), float(const_time)).reshape(*shp, -1)
    full, params = _full_input(net)
    return _MLP.apply(full, xf, None, *params).reshape(*shp, -1)


def mlp_forward_train(joiner, pts, dirs):
    """Joiner.forward with autograd: pts, dirs [..., 3] CUDA f32 -> raw [..., 4]; backward fills the parameters' .grad."""
    _lib.require_gpu()
    shp = pts.shape[:-1]
    p = pts.reshape(-1, pts.shape[-1]).to(torch.float32).contiguous()             # (3, or 4 with the time channel of ray_utils.py:133-134)
    d = dirs.reshape(-1, 3).to(torch.float32).contiguous()
    full, params = _full_input(joiner)
    return _MLP.apply(full, p, d, *params).reshape(*shp, 4)


class _ViewsHead(torch.autograd.Function):
    """The view-dependent head alone (models/vanilla.py:139-144): rgb = rgb_linear(relu(views_linears[0](cat([feature, embed(dirs)])))) on features
    an _MLP call returned -- a second colour of the same points seen from other directions without a second pass through the trunk.  Returns [n, 4]
    with a zero density column (the layout the loss terms read).  The per-layer products of the GEMM chain (nm_gemm_*)."""

    @staticmethod
    def forward(ctx, net, feat, dirs, Wv, bv, Wr, br):
        nerf = net.nerf
        dev = feat.device
        n = feat.shape[0]
        width, half = nerf.width, nerf.width // 2
        pk = _Packed(nerf, net.pos_pe.out_dim, net.dir_pe.out_dim)
        f4, d4 = _pad4(feat), _pad4(dirs)
        n4 = f4.shape[0]
        D0 = _encode(net.dir_pe, d4, pk.kd)
        hv = torch.empty((n4, half), device=dev, dtype=torch.float32)
        _gemm(0, 0, n4, half, width, f4, width, pk.Wv[0], width, hv, half)
        _gemm(0, 0, n4, half, pk.kd, D0, pk.kd, pk.Wv[1], pk.kd, hv, half, bias=pk.bv, flags=ACC | BIAS | RELU)
        raw = torch.empty((n4, 4), device=dev, dtype=torch.float32)
        b4 = torch.cat([br.detach().float(), torch.zeros(1, device=dev)])
        _gemm(0, 0, n4, 4, half, hv, half, pk.Wr4, half, raw, 4, bias=b4, flags=BIAS)
        ctx.net, ctx.pk, ctx.f4, ctx.d4, ctx.D0, ctx.hv, ctx.n = net, pk, f4, d4, D0, hv, n
        ctx.versions = [p._version for p in (Wv, bv, Wr, br)]
        ctx.live = (Wv, bv, Wr, br)
        ctx.in_versions = _input_versions(f4, d4)
        return raw[:n]

    @staticmethod
    def backward(ctx, g_raw):
        net, pk, f4, d4, D0, hv, n = ctx.net, ctx.pk, ctx.f4, ctx.d4, ctx.D0, ctx.hv, ctx.n
        if [p._version for p in ctx.live] != ctx.versions:
            raise _lib.NeumanHipError("a parameter of the views head was modified in place between the forward and the backward pass of a training step")
        _check_input_versions(ctx, f4, d4)
        nerf = net.nerf
        dev = f4.device
        n4, width, half = f4.shape[0], nerf.width, nerf.width // 2
        lib = _lib.lib()
        d_raw = torch.zeros((n4, 4), device=dev, dtype=torch.float32)
        d_raw[:n, :3] = g_raw[:, :3]                                             # (the density column is a constant zero)
        ws = [torch.empty(4, device=dev, dtype=torch.float32)]

        def grow(need):
            if need > ws[0].numel():
                ws[0] = torch.empty(int(need), device=dev, dtype=torch.float32)
            return ws[0]

        def wgrad(dz, m, a, k):
            out = torch.empty((m, k), device=dev, dtype=torch.float32)
            _gemm(1, 1, m, k, n4, dz, dz.shape[1], a, a.shape[1], out, k, ws=grow(lib.nm_gemm_workspace_floats(m, k, n4)))
            return out

        def colsum(x, rows, m):
            out = torch.empty(m, device=dev, dtype=torch.float32)
            w = grow(lib.nm_colsum_workspace_floats(rows, m))
            _lib.check(lib.nm_colsum(_lib.dev_ptr(x), rows, m, x.shape[1], _lib.dev_ptr(out), _lib.dev_ptr(w), w.numel(), _lib.stream_ptr()), "nm_colsum")
            return out
        rgb_w, rgb_b = wgrad(d_raw, 4, hv, half)[:3].contiguous(), colsum(d_raw, n4, 4)[:3].contiguous()
        bands = (n4 + 63) // 64
        cs_buf = torch.empty((bands, half), device=dev, dtype=torch.float32)      # per-64-row column sums out of the product's epilogue (row stride = its N)
        d_hv = torch.empty((n4, half), device=dev, dtype=torch.float32)
        _gemm(0, 1, n4, half, 4, d_raw, 4, pk.Wr4, half, d_hv, half, mask=hv, ldmask=half, flags=MASK | COLSUM, ws=cs_buf)
        views_b = colsum(cs_buf, bands, half)
        views_w = torch.cat([wgrad(d_hv, half, f4, width), wgrad(d_hv, half, D0, pk.kd)[:, :pk.n_dir]], 1)
        d_dirs = None
        if ctx.needs_input_grad[2]:
            dD0 = torch.empty((n4, pk.kd), device=dev, dtype=torch.float32)
            _gemm(0, 1, n4, pk.kd, half, d_hv, half, pk.Wv[1], pk.kd, dD0, pk.kd)
            d_dirs = _encode_backward(net.dir_pe, d4, dD0)[:n]
        d_feat = None
        if ctx.needs_input_grad[1]:
            d_feat = torch.empty((n4, width), device=dev, dtype=torch.float32)
            _gemm(0, 1, n4, width, half, d_hv, half, pk.Wv[0], width, d_feat, width)
            d_feat = d_feat[:n]
        return None, d_feat, d_dirs, views_w, views_b, rgb_w, rgb_b


def two_views_ok(joiner, n):
    """can `two_views` serve n points of this net?  (the fused fp16-storage step: the only form whose forward hands out feature_linear's output)"""
    return (_fused_ok(joiner) and STORE16 and FUSED_BACKWARD and (n + 3) // 4 * 4 >= STORE16_MIN_ROWS and joiner.pos_pe.out_dim <= 64
            and joiner.pos_pe.include_input and joiner.dir_pe.include_input          # (include_input=False: two plain calls through _full_input)
            and os.environ.get("NEUMAN_TWO_VIEWS", "1") != "0")


def two_views(joiner, pts, dirs, dirs2):
    """(joiner(pts, dirs), joiner(pts, dirs2) with a zero density column) for the price of one pass through the trunk: the second colour comes from
    the first call's feature vector through the views head alone.  The human trainer's colour-range term asks exactly this (human_nerf_trainer.py:280-290:
    the rays' canonical points seen from random directions); every gradient is the sum the two separate calls would give."""
    _lib.require_gpu()
    shp = pts.shape[:-1]
    p = pts.reshape(-1, 3).to(torch.float32).contiguous()
    d = dirs.reshape(-1, 3).to(torch.float32).contiguous()
    d2 = dirs2.reshape(-1, 3).to(torch.float32).contiguous()
    nerf = joiner.nerf
    raw, feat = _MLP.apply((joiner, 'feature'), p, d, *train_params(nerf))
    v = nerf.views_linears[0]
    raw2 = _ViewsHead.apply(joiner, feat, d2, v.weight, v.bias, nerf.rgb_linear.weight, nerf.rgb_linear.bias)
    return raw.reshape(*shp, 4), raw2.reshape(*shp, 4)


class _Composite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, raw, z_vals, rays_d, white_bkg, noise):
        R, S = z_vals.shape
        dev = raw.device
        rgb = torch.empty((R, 3), device=dev, dtype=torch.float32)
        disp = torch.empty(R, device=dev, dtype=torch.float32)
        acc = torch.empty(R, device=dev, dtype=torch.float32)
        depth = torch.empty(R, device=dev, dtype=torch.float32)
        weights = torch.empty((R, S), device=dev, dtype=torch.float32)
        _lib.check(_lib.lib().nm_composite(_lib.dev_ptr(raw), _lib.dev_ptr(z_vals), _lib.dev_ptr(rays_d), R, S, int(bool(white_bkg)),
                                           _lib.dev_ptr(noise), _lib.dev_ptr(rgb), _lib.dev_ptr(disp), _lib.dev_ptr(acc), _lib.dev_ptr(weights),
                                           _lib.dev_ptr(depth), _lib.stream_ptr()), "nm_composite")
        ctx.save_for_backward(raw, z_vals, rays_d)
        ctx.white_bkg, ctx.noise = bool(white_bkg), noise
        ctx.set_materialize_grads(False)
        return rgb, disp, acc, weights, depth

    @staticmethod
    def backward(ctx, g_rgb, g_disp, g_acc, g_w, g_depth):
        raw, z_vals, rays_d = ctx.saved_tensors
        if g_disp is not None:
            raise _lib.NeumanHipError("the gradient of disp_map is not implemented (no reference loss uses it)")
        if ctx.noise is not None:
            raw = (raw + torch.cat([torch.zeros_like(raw[..., :3]), ctx.noise[..., None]], -1)).contiguous()
        R, S = z_vals.shape
        d_raw = torch.empty_like(raw)
        c = lambda t: None if t is None else t.to(torch.float32).contiguous()
        _lib.check(_lib.lib().nm_composite_backward(_lib.dev_ptr(raw), _lib.dev_ptr(z_vals), _lib.dev_ptr(rays_d), R, S, int(ctx.white_bkg),
                                                    _lib.dev_ptr(c(g_rgb)), _lib.dev_ptr(c(g_acc)), _lib.dev_ptr(c(g_depth)), _lib.dev_ptr(c(g_w)),
                                                    _lib.dev_ptr(d_raw), _lib.stream_ptr()), "nm_composite_backward")
        return d_raw, None, None, None, None


def composite_train(raw, z_vals, rays_d, white_bkg=True, noise=None):
    """raw2outputs with autograd through `raw` (utils/render_utils.py:69-105)."""
    return _Composite.apply(raw.to(torch.float32).contiguous(), z_vals.detach().to(torch.float32).contiguous(),
                            rays_d.detach().to(torch.float32).contiguous(), white_bkg, noise)
