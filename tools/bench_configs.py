"""Throughput of the other BASELINE configs on synthetic scenes (one MI355X): C3 canonical / posed human, C4-like hybrid,
C5-like three-actor composite.  Rays resident in HBM, one warm-up frame, median of 3.  Prints one JSON line per config.

    python tools/bench_configs.py [--small]
"""
import json
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from neuman_hip import ray_utils, render_utils, synthetic  # noqa: E402

small = "--small" in sys.argv
dev = torch.device("cuda")
coarse, fine, human = (synthetic.make_joiner(0).cuda(), synthetic.make_joiner(1).cuda(), synthetic.make_joiner(2, 'rotate').cuda())
verts_c, faces = synthetic.capsule_mesh() if not small else synthetic.capsule_mesh(20, 24)
posed, T = synthetic.twist_transforms(verts_c)


def rays(cap):
    coords = np.argwhere(np.ones(cap.shape))[:, ::-1]
    o, d = ray_utils.shot_rays(cap, coords)
    return torch.from_numpy(o).to(dev, torch.float32).contiguous(), torch.from_numpy(d).to(dev, torch.float32).contiguous()


def timeit(fn, n=3):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]


def report(name, cap, fn, extra=None):
    dt = timeit(fn)
    total = cap.shape[0] * cap.shape[1]
    line = {"config": name, "rays": total, "ms_per_frame": dt * 1e3, "rays_per_s": total / dt}
    line.update(extra or {})
    print(json.dumps(line), flush=True)


with torch.no_grad():
    v_dev = torch.from_numpy(posed).to(dev)
    mesh = ray_utils.mesh_to_device(posed, faces, T, dev)
    print(json.dumps({'mesh_tree': mesh.info()}), flush=True)
    # ---- C3: canonical 360 view of the human net, 512x512, 128 samples, hit rays only
    res = 128 if small else 512
    cap = synthetic.SimpleCapture(res, res, fx=1.6 * res, c2w=synthetic.spherical_c2w(40., 0., 3.0))
    o, d = rays(cap)
    cloud = torch.from_numpy(synthetic.human_vertex_cloud(0)).to(dev)
    near, far = ray_utils.geometry_guided_near_far(o, d, cloud, 0.2)
    hit = float((near < far).float().mean())
    report("C3 canonical human 512x512x128", cap,
           lambda: render_utils.render_smpl_nerf_rays(human, o, d, cloud, None, 128, True, True, 0.2, 1.0), {"hit_fraction": hit})
    # ---- C3 posed: same camera on the twisted capsule mesh, with the obs->canonical warp
    near, far = ray_utils.geometry_guided_near_far(o, d, v_dev, 0.2)
    hit = float((near < far).float().mean())
    report("C3 posed human (warp) 512x512x128", cap,
           lambda: render_utils.render_smpl_nerf_rays(human, o, d, v_dev, mesh, 128, True, False, 0.2, 1.0), {"hit_fraction": hit})
    # ---- C4-like: 1280x720 hybrid, bkg 128+128, human 128
    w, h = (320, 180) if small else (1280, 720)
    cap = synthetic.SimpleCapture(w, h, fx=1.2 * w, c2w=synthetic.spherical_c2w(20., -5., 3.0), near=0.0, far=3.14)
    o, d = rays(cap)
    report("C4-like hybrid 1280x720, bkg 128+128, human 128", cap,
           lambda: render_utils.render_hybrid_rays(coarse, fine, human, o, d, 0.0, 3.14, v_dev, mesh, 128, 128, True, 0.2))
    # ---- C5-like: 1920x1080, 192 + 128, three actors x 192
    w, h = (480, 270) if small else (1920, 1080)
    cap = synthetic.SimpleCapture(w, h, fx=1.2 * w, c2w=synthetic.spherical_c2w(20., -5., 3.5), near=0.0, far=3.14)
    o, d = rays(cap)
    vs, ms = [], []
    for k, dx in enumerate((-0.7, 0.0, 0.7)):
        p2 = (posed + np.array([dx, 0, 0.1 * k], np.float32)).astype(np.float32)
        T2 = T.copy()
        T2[:, :3, 3] += np.array([dx, 0, 0.1 * k])
        vs.append(torch.from_numpy(p2).to(dev))
        ms.append(ray_utils.mesh_to_device(p2, faces, T2, dev))
    report("C5-like 3 actors 1920x1080, bkg 192+128, 3 x 192", cap,
           lambda: render_utils.render_multi_rays(coarse, fine, [human] * 3, o, d, 0.0, 3.14, vs, ms, 192, 128, True, 0.2))
