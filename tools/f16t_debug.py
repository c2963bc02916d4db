"""Stage-by-stage check of the generated stream of nerf_sigma_f16t_kernel (csrc/mlp_f16t_body.h, tools/gen_f16t.py) against nerf_mlp_kernel
<NM_PREC_FP16X3> (mlp.hip): a lane's input registers of stage st + 1 (nm_mlp_sigma_f16t_debug) decoded to activations vs nm_mlp_forward_debug's
activations after stage st, and sigma vs nm_mlp_sigma_rays.

    python tools/f16t_debug.py [n]
"""
import ctypes
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import _lib, synthetic  # noqa: E402


def decode(state):
    """[128 lanes.., 128 dwords] of workgroup 0's tile 0 (4 waves x 64 lanes) -> activations [128 samples, 256] as nerf_mlp_kernel's dump_act prints
    them: (float(hi) + float(lo)) / 32.  Lane (g, s) of wave w holds sample 32 w + s; register 8 t + 4 part + p = k-step t, elements 2p, 2p + 1
    of its 8-slot chunk 2 t + g: feature 32 (t >> 1) + 8 (2 (t & 1) + (e >> 2)) + 4 g + (e & 3)  (mlp_layout.h slot_feature)."""
    st = state.reshape(4, 2, 32, 16, 2, 4).astype(np.uint32)          # w, g, s, t, part, p
    halves = np.stack([(st & 0xffff).astype(np.uint16), (st >> 16).astype(np.uint16)], -1).view(np.float16).astype(np.float32)   # ..., p, 2
    val = (halves[:, :, :, :, 0] + halves[:, :, :, :, 1]) * np.float32(1.0 / 32.0)      # w, g, s, t, p, 2
    act = np.zeros((128, 256), np.float32)
    for g in range(2):
        for t in range(16):
            for p in range(4):
                for h in range(2):
                    e = 2 * p + h
                    f = 32 * (t >> 1) + 8 * (2 * (t & 1) + (e >> 2)) + 4 * g + (e & 3)
                    act[:, f] = val[:, g, :, t, p, h].reshape(128)
    return act


def main(n=1024):
    dev = torch.device('cuda')
    ok = True
    for mapping in ("posenc", "rotate"):
        net = synthetic.make_joiner(1 if mapping == 'posenc' else 2, mapping).to(dev)
        g = torch.Generator(device='cuda').manual_seed(5)
        pts = (torch.rand((n, 3), device=dev, generator=g) * 2 - 1).contiguous()
        dirs = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=-1).contiguous()
        for st in range(8):
            state = torch.zeros((256, 128), device=dev, dtype=torch.int32)
            o = torch.zeros((n, 4), device=dev)
            _lib.check(_lib.lib().nm_mlp_sigma_f16t_debug(net.handle(), _lib.dev_ptr(pts), _lib.dev_ptr(dirs), n, st, ctypes.c_void_p(state.data_ptr()), _lib.dev_ptr(o),
                                                          _lib.stream_ptr()), "nm_mlp_sigma_f16t_debug")
            torch.cuda.synchronize()
            got = decode(state.cpu().numpy())
            ref = net.forward_debug(pts[:128], dirs[:128], st, precision="fp16x3").cpu().numpy()
            bad = got.view(np.uint32) != ref.view(np.uint32)
            line = {"mapping": mapping, "stage": st, "equal": bool(not bad.any()), "differing": int(bad.sum())}
            if bad.any():
                r, f = np.nonzero(bad)
                line.update(rows=sorted(set(int(x) for x in r))[:12], n_rows=len(set(r)), feats=sorted(set(int(x) for x in f))[:24], n_feats=len(set(f)),
                            first=[int(r[0]), int(f[0]), float(got[r[0], f[0]]), float(ref[r[0], f[0]])], maxdiff=float(np.nanmax(np.abs(got - ref))))
                ok = False
            print(json.dumps(line), flush=True)
        os.environ["NEUMAN_SIGMA_KERNEL"] = "t"
        for nn in (n, 1, 100, 3000, 128 * 256 * 2 + 77):
            gg = torch.Generator(device='cuda').manual_seed(nn)
            o3 = (torch.rand((nn, 3), device=dev, generator=gg) * 0.4).contiguous()
            d3 = torch.nn.functional.normalize(torch.randn((nn, 3), device=dev, generator=gg), dim=-1).contiguous()
            z = (torch.rand((nn, 1), device=dev, generator=gg) * 3).contiguous()
            os.environ["NEUMAN_SIGMA_KERNEL"] = "t"
            mine = net.forward_rays(o3, d3, z, precision="fp16x3", sigma_scale=1.3, sigma_only=True)
            os.environ["NEUMAN_SIGMA_KERNEL"] = "w"
            ref = net.forward_rays(o3, d3, z, precision="fp16x3", sigma_scale=1.3, sigma_only=True)
            torch.cuda.synchronize()
            eq = torch.equal(mine, ref)
            badr = torch.nonzero((mine != ref).any(-1).reshape(-1)).reshape(-1)
            print(json.dumps({"mapping": mapping, "n": nn, "sigma_bit_identical": bool(eq), "rows_differing": int(badr.numel()), "first": [int(x) for x in badr[:8]],
                              "tiles": sorted(set(int(x) // 128 for x in badr[:4000]))[:20], "nan": int(torch.isnan(mine).sum()),
                              "maxdiff": float((mine - ref).abs().max())}), flush=True)
            ok = ok and eq
    print("F16T_OK" if ok else "F16T_MISMATCH")


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
