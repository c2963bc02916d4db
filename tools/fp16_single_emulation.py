"""Would ONE fp16 MFMA per product (operands rounded to fp16, float32 accumulate) be parity grade for the SHADING pass?  Emulated with torch on the device
before any kernel is built: the fine net's hidden products with weights and activations rounded to fp16 (power-of-two scalings as the fp16x3 kernel's:
activations x 32, weights x 2^k), the encodings' products exact (they stay split: small K), the heads like the trunk; composited on the mixed path's own
fine sample positions of a C2 slice and compared with the exact-float32 kernel on the same samples -- beside the committed i8x3 shading kernel.
Variants: 'w' weights only rounded, 'x' activations only, 'wx' both (the one-MFMA form), 'wx_rz' both with truncation instead of RNE.

    python tools/fp16_single_emulation.py [n_rays]
"""
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import ray_utils, render_utils, synthetic  # noqa: E402

torch.backends.cuda.matmul.allow_tf32 = False
dev = torch.device('cuda')
n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
preset = sys.argv[2] if len(sys.argv) > 2 else None
mk = (lambda s: synthetic.make_joiner(s, preset=preset)) if preset else synthetic.make_joiner
coarse, fine = mk(0).to(dev), mk(1).to(dev)
cap = synthetic.SimpleCapture(800, 800)
o, d = ray_utils.shot_all_rays_dev(cap, dev)
sel = torch.arange(300 * 800, 300 * 800 + n_rays, device=dev)
o, d = o[sel].contiguous(), d[sel].contiguous()
near, far = torch.zeros(n_rays, device=dev), torch.full((n_rays,), 3.14, device=dev)
with torch.no_grad():
    raw_mixed, z = render_utils.bkg_pass_rays(coarse, fine, o, d, near, far, 128, 128, True)
    raw32 = fine.forward_rays(o, d, z, precision="fp32")
    rgb32 = render_utils.raw2outputs(raw32, z, d)[0]
    rgb_i8 = render_utils.raw2outputs(raw_mixed, z, d)[0]
    rgb_f16x3 = render_utils.raw2outputs(fine.forward_rays(o, d, z, precision="fp16x3"), z, d)[0]


def r16(x, mode):
    if mode == 'rz':
        h = x.to(torch.float16)
        over = h.float().abs() > x.abs()
        return torch.where(over, torch.nextafter(h, torch.zeros_like(h)), h).float()
    return x.to(torch.float16).float()


def pe(x, n):
    out = [x]
    for k in range(n):
        out += [torch.sin(x * 2.0 ** k), torch.cos(x * 2.0 ** k)]
    return torch.cat(out, -1)


def net_emul(net, pts, dirs, round_w, round_x, mode='rne'):
    P = [p.detach().double() for p in net.nerf.ordered_params()]
    W, B = P[0:16:2], P[1:16:2]
    Wv, bv, Wf, bf, wa, ba, Wr, br = P[16:24]
    rw = (lambda w: r16(w.float(), mode).double()) if round_w else (lambda w: w)
    rx = (lambda x: (r16((x * 32).float(), mode) / 32).double()) if round_x else (lambda x: x)
    e = pe(pts.double(), net.pos_pe.N_freqs)
    ed = pe(dirs.double(), net.dir_pe.N_freqs)
    h = torch.relu(e @ W[0].T + B[0])                                    # stage 0: encodings only (split products: exact here)
    for i in range(1, 8):
        if i == 5:
            h = torch.relu(e @ W[5][:, :63].T + rx(h) @ rw(W[5][:, 63:]).T + B[5])
        else:
            h = torch.relu(rx(h) @ rw(W[i]).T + B[i])
    sigma = rx(h) @ rw(wa).T + ba
    feat = rx(h) @ rw(Wf).T + bf
    hv = torch.relu(rx(feat) @ rw(Wv[:, :256]).T + ed @ Wv[:, 256:].T + bv)
    rgb = rx(hv) @ rw(Wr).T + br
    return torch.cat([rgb, sigma], -1).float()


res = {"rays": n_rays, "preset": preset, "committed": {"i8x3_vs_f32_kernel": float((rgb_i8 - rgb32).abs().max()), "fp16x3_vs_f32_kernel": float((rgb_f16x3 - rgb32).abs().max())}}
with torch.no_grad():
    pts = (o[:, None, :] + d[:, None, :] * z[..., None]).reshape(-1, 3)
    dd = d[:, None, :].expand(-1, z.shape[1], 3).reshape(-1, 3)
    outs = {}
    for name, (rw_, rx_, mode) in {"exact_f64": (False, False, 'rne'), "w": (True, False, 'rne'), "x": (False, True, 'rne'), "wx": (True, True, 'rne'), "wx_rz": (True, True, 'rz')}.items():
        chunks = [net_emul(fine, pts[i:i + 262144], dd[i:i + 262144], rw_, rx_, mode) for i in range(0, pts.shape[0], 262144)]
        raw = torch.cat(chunks).reshape(n_rays, z.shape[1], 4)
        outs[name] = render_utils.raw2outputs(raw.contiguous(), z, d)[0]
    ref = outs["exact_f64"]
    res["f32_kernel_vs_f64"] = float((rgb32 - ref).abs().max())
    res["i8x3_vs_f64"] = float((rgb_i8 - ref).abs().max())
    for name in ("w", "x", "wx", "wx_rz"):
        e = (outs[name] - ref).abs().max(-1).values
        res[name] = {"linf": float(e.max()), "p99": float(e.quantile(0.99)), "median": float(e.median()), "rays_gt_1e-4": int((e > 1e-4).sum()), "rays_gt_5e-5": int((e > 5e-5).sum())}
print(json.dumps(res))
