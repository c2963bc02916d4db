#!/usr/bin/env python
"""Speed and agreement of the CPU oracle (oracle/, the "port" bench.py times as cpu_baseline) against the REFERENCE's own
render_vanilla on the same rays, weights and thread count.  Build container only (needs /root/reference; the absent wheels the
renderer never touches are stubbed exactly as tests/golden/make_golden.py does):

    python tools/port_vs_reference.py [--rays 4096]  ->  profiles/r02_port_vs_reference.json

bench.py attaches the file's content to `cpu_baseline.port_vs_reference_speed`, so the label "port" comes with the measured
ratio to the real thing.
"""
import argparse
import json
import os
import sys
import time
from unittest import mock

import numpy as np
import torch

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
from oracle import render as O_render  # noqa: E402
from oracle.nerf_mlp import JoinerSpec  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=int, default=4096)
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r02_port_vs_reference.json"))
    args = ap.parse_args()
    W = H = 800
    rows = -(-args.rays // W)
    # a (rows x 800) strip of the 800x800 camera: the same first `rays` rays bench.py's cpu_baseline renders
    def ref_cap():
        cap = BasePinholeCapture(PinholeCamera(W, rows, 1.25 * W, 1.25 * W, W / 2, H / 2), CameraPose.from_camera_to_world(np.eye(4)))
        cap.near, cap.far = {'bkg': 0.0}, {'bkg': 3.14}
        return cap
    ours = [synthetic.make_joiner(s) for s in (0, 1)]
    opt = synthetic.default_opt(posenc='posenc')
    ref_nets = []
    for j in ours:
        net, _ = R_vanilla.build_nerf(opt)
        net.load_state_dict(j.state_dict(), strict=True)
        ref_nets.append(net.eval())
    o_nets = [(synthetic.state_numpy(j), JoinerSpec()) for j in ours]
    o_cap = synthetic.SimpleCapture(W, rows, cy=H / 2)
    threads = torch.get_num_threads()

    def t_ref():
        t0 = time.perf_counter()
        out = R_render.render_vanilla(ref_nets[0], ref_cap(), ref_nets[1], rays_per_batch=2048, samples_per_ray=128,
                                      importance_samples_per_ray=128)
        return time.perf_counter() - t0, out

    def t_port():
        t0 = time.perf_counter()
        out = O_render.render_vanilla(o_nets[0], o_cap, o_nets[1], rays_per_batch=2048, samples_per_ray=128, importance_samples_per_ray=128)
        return time.perf_counter() - t0, out

    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):              # (the reference prints a progress line per batch)
        t_ref()
        t_port()
        tr, a = min((t_ref() for _ in range(2)), key=lambda x: x[0])
        tp, b = min((t_port() for _ in range(2)), key=lambda x: x[0])
    n = rows * W
    err = np.abs(a.reshape(-1, 3) - b.reshape(-1, 3)).max(-1)
    res = {"rays": n, "threads": threads, "host": f"{os.cpu_count()}-thread build container",
           "reference_rays_per_s": n / tr, "port_rays_per_s": n / tp, "port_over_reference": tr / tp,
           "port_vs_reference_rgb_linf": float(err.max()), "port_vs_reference_rays_gt_1e-4": int((err > 1e-4).sum()),
           "what": "utils/render_utils.py render_vanilla (imported unmodified, absent wheels stubbed) vs oracle.render.render_vanilla, "
                   f"first {rows} rows of the 800x800 frame, 128+128 samples/ray, rays_per_batch=2048, best of 2 after a warm-up"}
    print(json.dumps(res, indent=1))
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
