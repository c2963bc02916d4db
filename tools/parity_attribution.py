#!/usr/bin/env python
"""The two-pass parity attribution (oracle/attribution.py) on the workloads the tests and bench.py use, with the arrays dumped for
offline analysis:  python tools/parity_attribution.py  ->  gpurun_out/attr_<case>.npz + one report line per case."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

from neuman_hip import render_utils, synthetic  # noqa: E402
from oracle import attribution, ray_ops  # noqa: E402
from oracle.nerf_mlp import JoinerSpec  # noqa: E402


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    coarse, fine = synthetic.make_joiner(0), synthetic.make_joiner(1)
    nets = [(synthetic.state_numpy(n), JoinerSpec()) for n in (coarse, fine)]
    coarse, fine = coarse.cuda(), fine.cuda()
    cases = {"c1_64x64_32+32": (64, 64, slice(0, 4096), 32, 32), "smoke_32x32_16+16": (32, 32, slice(0, 1024), 16, 16),
             "c2_slice_2048_128+128": (800, 800, slice(400 * 800 + 100, 400 * 800 + 100 + 2048), 128, 128),
             "bench_first_4096_128+128": (800, 800, slice(0, 4096), 128, 128)}
    for name, (w, h, sl, S, NI) in cases.items():
        cap = synthetic.SimpleCapture(w, h)
        o, d = ray_ops.shot_all_rays(cap.intrinsic_matrix, cap.cam_pose.camera_to_world, cap.shape)
        o, d = o[sl].astype(np.float32), d[sl].astype(np.float32)
        ora = attribution.oracle_two_pass(nets, o, d, 0.0, 3.14, S, NI)
        cu = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
        for prec in ("mixed", "fp16x3"):
            rgb, z, wt, rgb_on = attribution.device_two_pass(render_utils, coarse, fine, cu(o), cu(d), 0.0, 3.14, S, NI, cu(ora["z"]), precision=prec)
            rep, fails = attribution.two_pass(rgb, z, wt, rgb_on, ora["rgb"], ora["z"], ora["w"], ora["fine_on"], case=name, tag=f"{name} {prec}")
            print(json.dumps({"case": name, "precision": prec, "fails": fails}))
            np.savez(os.path.join(ROOT, "gpurun_out", f"attr_{name}_{prec}.npz"), rgb=rgb, z=z, w=wt, rgb_on=rgb_on, o_rgb=ora["rgb"], o_z=ora["z"], o_w=ora["w"])


if __name__ == "__main__":
    main()
