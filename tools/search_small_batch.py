"""The closest-point search at a TRAINING iteration's sizes (the human trainer: 92 k ray samples against the posed body, 184 k canonical + random box points
against the canonical one): time per call, for chunk-size sweeps (NEUMAN_SEARCH_CHUNK).  Prints one JSON line; the checksums pin the outputs."""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import ray_utils, synthetic  # noqa: E402

dev = torch.device("cuda")
verts_c, faces = synthetic.capsule_mesh()
posed, T = synthetic.twist_transforms(verts_c)
cap = synthetic.SimpleCapture(512, 512, fx=1.6 * 512, c2w=synthetic.spherical_c2w(40., 0., 3.0))
coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
o, d = ray_utils.shot_rays(cap, coords)
o = torch.from_numpy(o).to(dev, torch.float32).contiguous()
d = torch.from_numpy(d).to(dev, torch.float32).contiguous()
near, far = ray_utils.geometry_guided_near_far(o, d, torch.from_numpy(posed).to(dev), 0.2)
idx = (near < far).nonzero().flatten()
idx = idx[torch.randperm(idx.numel(), device=dev, generator=torch.Generator(device=dev).manual_seed(0))[:719]]
pts, _, _ = ray_utils.sample_z(o[idx].contiguous(), d[idx].contiguous(), near[idx].contiguous(), far[idx].contiguous(), 128, want_points=True)
ray_pts = pts.reshape(-1, 3).contiguous()                                                    # 92 032 samples along 719 rays
g = torch.Generator(device=dev).manual_seed(1)
box_pts = torch.cat([ray_pts, (torch.rand((ray_pts.shape[0], 3), device=dev, generator=g) - 0.5) * 3], 0).contiguous()
mesh = ray_utils.mesh_to_device(posed, faces, T, dev)


def med(fn, n=9):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3, r


a_ms, a = med(lambda: ray_utils.signed_distance_dev(ray_pts, mesh))
b_ms, b = med(lambda: ray_utils.signed_distance_dev(box_pts, mesh))
print(json.dumps({"chunk": os.environ.get("NEUMAN_SEARCH_CHUNK"), "ray_samples": ray_pts.shape[0], "ray_ms": a_ms, "ray_plus_box_points": box_pts.shape[0], "box_ms": b_ms,
                  "checksums": [float(a[0].double().sum()), float(a[2].double().sum()), float(b[0].double().sum()), int(b[1].long().sum())]}))
