#!/usr/bin/env python
"""How strongly a two-pass frame's colour depends on WHERE its fine samples sit, measured on the CPU oracle by finite differences:
for a subset of rays of each scored workload every fine sample is displaced on its own by h and the oracle's shading pass is
re-evaluated; L_ray = sum over samples of |d rgb| / h (the 1-norm of the gradient: |rgb(z + dz) - rgb(z)| <= L_ray * max_s |dz_s| to first
order, whatever the displacement pattern), L = the largest L_ray.  oracle/attribution.py statement (c) uses 1.5 x L.

    python tools/lipschitz_probe.py   ->  profiles/r03_lipschitz.json        (CPU only, ~2 min)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
sys.path.insert(0, ROOT)

from neuman_hip import synthetic  # noqa: E402
from oracle import attribution, compositing, nerf_mlp, ray_ops  # noqa: E402
from oracle.nerf_mlp import JoinerSpec  # noqa: E402

CASES = {"c1_64x64_32+32": (64, 64, 0, 4096, 32, 32), "smoke_32x32_16+16": (32, 32, 0, 1024, 16, 16),
         "c2_slice_2048_128+128": (800, 800, 400 * 800 + 100, 2048, 128, 128), "bench_first_4096_128+128": (800, 800, 0, 4096, 128, 128)}
H_STEP = 5e-5
N_RAYS = 48


def main():
    nets = [(synthetic.state_numpy(synthetic.make_joiner(s)), JoinerSpec()) for s in (0, 1)]
    out = {}
    for name, (W, H, first, n, S, NI) in CASES.items():
        cap = synthetic.SimpleCapture(W, H)
        o, d = ray_ops.shot_all_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, cap.shape)
        pick = first + np.linspace(0, n - 1, N_RAYS).astype(np.int64)
        o, d = o[pick].astype(np.float32), d[pick].astype(np.float32)
        ora = attribution.oracle_two_pass(nets, o, d, 0.0, 3.14, S, NI)
        z0, rgb0 = ora["z"], ora["rgb"]
        SF = z0.shape[1]
        # one displaced copy of every ray per sample: [N_RAYS * SF, SF]
        zz = np.repeat(z0, SF, axis=0)
        zz[np.arange(N_RAYS * SF), np.tile(np.arange(SF), N_RAYS)] += np.float32(H_STEP)
        zz = np.sort(zz, axis=1)                                  # (a displaced sample that passes its neighbour is re-sorted, as the renderer would)
        oo, dd = np.repeat(o, SF, axis=0), np.repeat(d, SF, axis=0)
        rgb = []
        for i in range(0, zz.shape[0], 2048):
            z = zz[i:i + 2048]
            pts = (oo[i:i + 2048, None, :] + dd[i:i + 2048, None, :] * z[..., None]).astype(np.float32)
            raw = nerf_mlp.joiner_forward(*nets[1], pts, np.broadcast_to(dd[i:i + 2048, None, :], pts.shape))
            rgb.append(compositing.raw2outputs(raw, z, dd[i:i + 2048])[0])
        g = np.abs(np.concatenate(rgb).reshape(N_RAYS, SF, 3) - rgb0[:, None, :]).max(-1) / H_STEP        # [rays, samples]
        L_ray = g.sum(1)
        out[name] = {"samples": [S, NI], "rays_probed": N_RAYS, "step": H_STEP, "L_max": float(L_ray.max()), "L_median": float(np.median(L_ray)),
                     "largest_single_sample_sensitivity": float(g.max())}
        print(name, out[name], flush=True)
    out["what"] = "1-norm over the fine samples of |d rgb / d z_s| of the oracle's shading pass (forward differences, one sample displaced at a time), max over the probed rays"
    with open(os.path.join(ROOT, "profiles", "r03_lipschitz.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
