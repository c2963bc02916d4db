"""Warp (K3) time and mesh-build time on the C3-posed workload (tools/bench_configs.py geometry): 512x512 rays on the posed
capsule body, 128 samples per hit ray.  Prints one JSON line; the checksums pin the outputs across kernel versions."""
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
ho, hd, hn, hf = o[idx].contiguous(), d[idx].contiguous(), near[idx].contiguous(), far[idx].contiguous()
pts, _, z = ray_utils.sample_z(ho, hd, hn, hf, 128, want_points=True)


def med(fn, n=5):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[n // 2] * 1e3, r


build_ms, mesh = med(lambda: ray_utils.mesh_to_device(posed, faces, T, dev))
warp_ms, (cp, cd, _) = med(lambda: ray_utils.warp_to_canonical_dev(pts, mesh))
print(json.dumps({"tree": mesh.info(), "hit_rays": int(idx.numel()), "samples": int(pts.shape[0] * pts.shape[1]), "build_ms": build_ms,
                  "warp_ms": warp_ms, "checksum": float(cp.double().sum()), "checksum_d": float(cd.double().sum())}))
