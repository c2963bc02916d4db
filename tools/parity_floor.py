#!/usr/bin/env python
"""The floor for end-to-end two-pass parity, per workload the tests / bench.py / smoke() score: on exactly those rays, how many differ
by more than 1e-4 between the REFERENCE's own render_vanilla (utils/render_utils.py:108-161, imported unmodified) and the CPU oracle
-- two float32 CPU evaluations of the same algorithm.  Build container only (needs /root/reference):

    python tools/parity_floor.py   ->  profiles/r03_parity_floor.json   (oracle/attribution.py reads the counts from there)
"""
import contextlib
import io
import json
import os
import sys
from unittest import mock

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
sys.path.insert(0, ROOT)

from utils import render_utils as R_render  # noqa: E402  (reference)
from models import vanilla as R_vanilla  # noqa: E402
from cameras.pinhole_camera import PinholeCamera  # noqa: E402
from cameras.camera_pose import CameraPose  # noqa: E402
from cameras.captures import BasePinholeCapture  # noqa: E402

from neuman_hip import synthetic  # noqa: E402
from oracle import attribution, ray_ops  # noqa: E402
from oracle.nerf_mlp import JoinerSpec  # noqa: E402

# name -> (W, H, first ray, number of rays, coarse samples, importance samples): the slices tests / bench / smoke use
CASES = {"c1_64x64_32+32": (64, 64, 0, 4096, 32, 32), "smoke_32x32_16+16": (32, 32, 0, 1024, 16, 16),
         "c2_slice_2048_128+128": (800, 800, 400 * 800 + 100, 2048, 128, 128), "bench_first_4096_128+128": (800, 800, 0, 4096, 128, 128)}


def main():
    ours = [synthetic.make_joiner(s) for s in (0, 1)]
    opt = synthetic.default_opt(posenc='posenc')
    ref_nets = []
    for j in ours:
        net, _ = R_vanilla.build_nerf(opt)
        net.load_state_dict(j.state_dict(), strict=True)
        ref_nets.append(net.eval())
    o_nets = [(synthetic.state_numpy(j), JoinerSpec()) for j in ours]
    out = {}
    for name, (W, H, first, n, S, NI) in CASES.items():
        r0, r1 = first // W, -(-(first + n) // W)
        # rows [r0, r1) of the W x H camera as a camera of its own (principal point shifted): the same rays
        cap = BasePinholeCapture(PinholeCamera(W, r1 - r0, 1.25 * W, 1.25 * W, W / 2, H / 2 - r0), CameraPose.from_camera_to_world(np.eye(4)))
        cap.near, cap.far = {'bkg': 0.0}, {'bkg': 3.14}
        with contextlib.redirect_stdout(io.StringIO()):
            ref = R_render.render_vanilla(ref_nets[0], cap, ref_nets[1], rays_per_batch=2048, samples_per_ray=S, importance_samples_per_ray=NI)
        a = first - r0 * W
        ref = ref.reshape(-1, 3)[a:a + n]
        full = synthetic.SimpleCapture(W, H)
        o, d = ray_ops.shot_all_rays(full.intrinsic_matrix, full.cam_pose.camera_to_world, full.shape)
        ora = attribution.oracle_two_pass(o_nets, o[first:first + n].astype(np.float32), d[first:first + n].astype(np.float32), 0.0, 3.14, S, NI)
        err = np.abs(ora["rgb"] - ref).max(-1)
        out[name] = {"rays": n, "oracle_vs_reference_rays_gt_1e-4": int((err > 1e-4).sum()), "oracle_vs_reference_linf": float(err.max())}
        print(name, out[name], flush=True)
    out["what"] = ("utils/render_utils.py render_vanilla (imported unmodified, absent wheels stubbed) vs oracle two-pass render on the same rays: "
                   "the number of rays two float32 CPU evaluations of the reference's algorithm disagree on by more than 1e-4")
    with open(os.path.join(ROOT, "profiles", "r03_parity_floor.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
