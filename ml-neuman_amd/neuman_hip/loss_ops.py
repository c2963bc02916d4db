"""The human trainer's scalar regularisers as fused value-and-gradient passes (csrc/loss.hip; reference trainers/human_nerf_trainer.py:280-380).

Each function returns the term as a 0-d tensor connected to autograd: the kernel has already written the term's gradient with respect to the network
outputs it reads, `backward` multiplies it by the incoming scalar.  NEUMAN_FUSED_LOSS=0 makes human_trainer spell the terms with torch's elementwise
operations instead (the check of these kernels, not a faster path)."""
import ctypes
import os

import torch

from . import _lib

FUSED = os.environ.get('NEUMAN_FUSED_LOSS', '1') != '0'
_WS = {}


def _ws(dev):
    if dev not in _WS:
        _WS[dev] = torch.empty(int(_lib.lib().nm_loss_workspace_doubles()), device=dev, dtype=torch.float64)
    return _WS[dev]


def _rows(raw):
    r = raw.reshape(-1, 4)
    if r.dtype != torch.float32 or not r.is_contiguous() or (r.data_ptr() & 15):
        r = r.to(torch.float32).contiguous()
    return r


class _Bimodal(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, clamp01, offset):
        xf = x.reshape(-1).to(torch.float32).contiguous()
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        dx = torch.empty_like(xf)
        _lib.check(_lib.lib().nm_loss_bimodal(_lib.dev_ptr(xf), xf.numel(), int(bool(clamp01)), float(offset), _lib.dev_ptr(out), _lib.dev_ptr(dx),
                                              ctypes.c_void_p(_ws(x.device).data_ptr()), _lib.stream_ptr()), "nm_loss_bimodal")
        ctx.dx, ctx.shape = dx, x.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return (ctx.dx * g).reshape(ctx.shape), None, None


def bimodal_prior(x, offset, clamp01=True):
    """mean(-log(e^-|y| + e^-|1 - y|) + offset), y = clamp(x, 0, 1): zero-mean pull of x towards 0 or 1 (:368-379)"""
    return _Bimodal.apply(x, clamp01, offset)


class _PairMse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, mode, scale):
        ar, br = _rows(a), _rows(b)
        out = torch.empty(1, device=a.device, dtype=torch.float32)
        da, db = torch.empty_like(ar), torch.empty_like(br)
        _lib.check(_lib.lib().nm_loss_pair_mse(int(mode), _lib.dev_ptr(ar), _lib.dev_ptr(br), ar.shape[0], float(scale), _lib.dev_ptr(out), _lib.dev_ptr(da), _lib.dev_ptr(db),
                                               ctypes.c_void_p(_ws(a.device).data_ptr()), _lib.stream_ptr()), "nm_loss_pair_mse")
        ctx.da, ctx.db, ctx.sa, ctx.sb = da, db, a.shape, b.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        return ((ctx.da * g).reshape(ctx.sa) if ctx.needs_input_grad[0] else None, (ctx.db * g).reshape(ctx.sb) if ctx.needs_input_grad[1] else None, None, None)


def color_range(other_view, tgts, scale):
    """scale * mse(sigmoid(rgb of other_view), sigmoid(rgb of tgts)) (:280-290)"""
    return _PairMse.apply(other_view, tgts, 0, scale)


def symmetry(mirrored, tgts, scale):
    """scale * mse(tanh(relu(sigma of tgts)), tanh(relu(sigma of mirrored))) (:292-304)"""
    return _PairMse.apply(tgts, mirrored, 1, scale)


class _Shape(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pred, dist_h, dummy, dist_d, w_smpl, w_dummy, factor, exponent):
        pr = _rows(pred)
        dh = dist_h.reshape(-1).to(torch.float32).contiguous()
        dev = pred.device
        out = torch.empty(1, device=dev, dtype=torch.float32)
        norm = torch.empty(3, device=dev, dtype=torch.float32)
        d_pred = torch.empty_like(pr)
        if dummy is not None:
            du, dd = _rows(dummy), dist_d.reshape(-1).to(torch.float32).contiguous()
            d_dummy = torch.empty_like(du)
        else:
            du = dd = d_dummy = None
        _lib.check(_lib.lib().nm_loss_shape(_lib.dev_ptr(pr), _lib.dev_ptr(dh), pr.shape[0], _lib.dev_ptr(du), _lib.dev_ptr(dd), 0 if du is None else du.shape[0],
                                            float(w_smpl), float(w_dummy), float(factor), float(exponent), _lib.dev_ptr(out), _lib.dev_ptr(d_pred), _lib.dev_ptr(d_dummy),
                                            ctypes.c_void_p(_ws(dev).data_ptr()), _lib.dev_ptr(norm), _lib.stream_ptr()), "nm_loss_shape")
        ctx.d_pred, ctx.d_dummy, ctx.sp, ctx.sd = d_pred, d_dummy, pred.shape, None if dummy is None else dummy.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        gp = (ctx.d_pred * g).reshape(ctx.sp) if ctx.needs_input_grad[0] else None
        gd = (ctx.d_dummy * g).reshape(ctx.sd) if (ctx.d_dummy is not None and ctx.needs_input_grad[2]) else None
        return gp, None, gd, None, None, None, None, None


def shape_prior(pred, dist_h, dummy, dist_d, w_smpl, w_dummy, factor, exponent):
    """the SMPL shape prior (:305-343) on the rays' samples (pred, their signed distances) and, when given, the dummy points"""
    return _Shape.apply(pred, dist_h, dummy, dist_d, w_smpl, w_dummy, factor, exponent)
