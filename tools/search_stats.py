"""Counters of the closest-point search on the C3-posed workload (a probe build: csrc/warp.hip compiled with -DNM_SEARCH_STATS into the library named by
NEUMAN_HIP_LIB; the shipped library carries no counters).  Prints wave iterations and lane visits per phase, per sample."""
import ctypes
import json
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import _lib, ray_utils, synthetic  # noqa: E402

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
mesh = ray_utils.mesh_to_device(posed, faces, T, dev)
raw = ctypes.CDLL(_lib.LIB_PATH)
if not hasattr(raw, 'nm_debug_search_stats'):
    raise SystemExit("this library carries no search counters: apply profiles/r06_search_disc.patch (or add the NM_STAT lines to csrc/warp.hip), compile with "
                     "-DNM_SEARCH_STATS and point NEUMAN_HIP_LIB at the result (profiles/r06_search_disc_experiment.md)")
out = (ctypes.c_ulonglong * 16)()
raw.nm_debug_search_stats(out)
ray_utils.warp_to_canonical_dev(pts, mesh)
raw.nm_debug_search_stats(out)
n = pts.shape[0] * pts.shape[1]
v = list(out)
print(json.dumps({"samples": n, "wave_iters_per_64_samples": {"walk": v[0] * 64 / n, "leaf": v[1] * 64 / n, "test": v[2] * 64 / n},
                  "lane_visits_per_sample": {"walk": v[3] / n, "leaf": v[4] / n, "test": v[5] / n}, "triangles_pushed_per_sample": v[6] / n,
                  "dropped_stale_per_sample": v[7] / n, "clocks_per_wave_iteration": {"walk": v[8] / max(v[0], 1), "leaf": v[9] / max(v[1], 1), "test": v[10] / max(v[2], 1), "between": v[11] / max(v[0] + v[1] + v[2], 1)},
                  "clock_share": {"walk": v[8] / max(sum(v[8:12]), 1), "leaf": v[9] / max(sum(v[8:12]), 1), "test": v[10] / max(sum(v[8:12]), 1), "between": v[11] / max(sum(v[8:12]), 1)},
                  "lanes_per_iteration": {"walk": v[3] / max(v[0], 1), "leaf": v[4] / max(v[1], 1), "test": v[5] / max(v[2], 1)}}))
