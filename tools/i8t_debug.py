"""Stage-by-stage check of the generated instruction stream of nerf_mlp_i8t_kernel (csrc/mlp_i8t_body.h, tools/gen_i8t.py) against the
plain-HIP form of the same kernel (a variant library built with -DNM_I8T_HIP: python tools/build_variant.py hipref --src mlp_i8t.hip -DNM_I8T_HIP):
the activation state of the first tile after every stage (nm_mlp_forward_i8t_debug) and the outputs.

    python tools/i8t_debug.py            (runs itself once more under NEUMAN_HIP_LIB=.../libneuman_hip_hipref.so)
"""
import ctypes
import json
import os
import subprocess
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import _lib, synthetic  # noqa: E402

REF = os.path.join(ROOT, "ml-neuman_amd", "lib", "exp", "libneuman_hip_hipref.so")


def run(mapping='posenc', n=1024, stages=range(10), tile_round=0):
    dev = torch.device('cuda')
    net = synthetic.make_joiner(1 if mapping == 'posenc' else 2, mapping).to(dev)
    g = torch.Generator(device='cuda').manual_seed(5)
    pts = (torch.rand((n, 3), device=dev, generator=g) * 2 - 1).contiguous()
    dirs = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=-1).contiguous()
    out = {}
    for st in stages:
        state = torch.zeros((256, 130), device=dev, dtype=torch.int32)
        o = torch.zeros((n, 4), device=dev)
        _lib.check(_lib.lib().nm_mlp_forward_i8t_debug(net.handle(), _lib.dev_ptr(pts), _lib.dev_ptr(dirs), n, st + 100 * tile_round, ctypes.c_void_p(state.data_ptr()), _lib.dev_ptr(o),
                                                       _lib.stream_ptr()), "nm_mlp_forward_i8t_debug")
        torch.cuda.synchronize()
        out[f"state{st}"] = state.cpu().numpy()
        out["out"] = o.cpu().numpy()
    return out


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--dump":
        rnd = int(os.environ.get("I8T_ROUND", "0"))
        np.savez(sys.argv[2], **(run(sys.argv[3], int(os.environ["I8T_N"]), [0]) if os.environ.get("I8T_N") else run(sys.argv[3], 256 * 256 * (rnd + 1) if rnd else 1024, range(10), rnd)))
        sys.exit(0)
    for mapping in ("posenc", "rotate"):
        big = bool(os.environ.get("I8T_N"))
        rnd = int(os.environ.get("I8T_ROUND", "0"))
        nn = 256 * 256 * (rnd + 1) if rnd else 1024
        mine = run(mapping, nn, [0] if big else range(10), rnd)
        tmp = f"/tmp/i8t_ref_{mapping}.npz"
        env1 = {k: v for k, v in os.environ.items() if k != "I8T_N"}
        subprocess.run([sys.executable, __file__, "--dump", tmp, mapping], check=True, env=dict(env1, NEUMAN_HIP_LIB=REF))
        ref = dict(np.load(tmp))
        for st in ([] if big else range(10)):
            a, b = mine[f"state{st}"], ref[f"state{st}"]
            bad = a != b
            line = {"mapping": mapping, "stage": st, "equal": bool(not bad.any()), "differing_words": int(bad.sum())}
            if bad.any():
                lanes, words = np.nonzero(bad)
                line.update(lanes=sorted(set(int(x) for x in lanes))[:12], words=sorted(set(int(x) for x in words))[:24], n_lanes=len(set(lanes)), n_words=len(set(words)),
                            first=[int(lanes[0]), int(words[0]), int(a[lanes[0], words[0]]), int(b[lanes[0], words[0]])])
            print(json.dumps(line), flush=True)
        if os.environ.get("I8T_N"):
            n = int(os.environ["I8T_N"])
            mine = run(mapping, n, [0])
            subprocess.run([sys.executable, __file__, "--dump", tmp, mapping], check=True, env=dict(os.environ, NEUMAN_HIP_LIB=REF))
            ref = dict(np.load(tmp))
            bad_rows = np.nonzero((mine["out"] != ref["out"]).any(1))[0]
            tiles = sorted(set(int(r) // 256 for r in bad_rows))
            print(json.dumps({"n": n, "bad_rows": int(bad_rows.size), "bad_tiles": tiles[:40], "n_bad_tiles": len(tiles), "first_rows": [int(r) for r in bad_rows[:16]],
                              "rows_in_tile": sorted(set(int(r) % 256 for r in bad_rows))[:40]}), flush=True)
        eq = np.array_equal(mine["out"], ref["out"])
        d = np.abs(mine["out"] - ref["out"])
        print(json.dumps({"mapping": mapping, "outputs_bit_identical": bool(eq), "max_abs_diff": float(np.nanmax(d)), "nan": int(np.isnan(mine["out"]).sum()),
                          "rows_differing": int((d > 0).any(1).sum())}), flush=True)
