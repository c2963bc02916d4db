"""Diagnostic: on the 64 x 64 posed golden (tests/golden/posed_big.npz), which rays deviate from the reference's frame by more than 1e-4
when its near / far are replayed, and how far their canonical points are from the oracle's warp of the same sample points."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
for p in (ROOT, os.path.join(ROOT, "ml-neuman_amd"), os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "helpers")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import posed_scene as PS  # noqa: E402
import test_hip_posed_golden as TP  # noqa: E402
from neuman_hip import synthetic  # noqa: E402
from oracle import warp, nerf_mlp, compositing  # noqa: E402

S = TP.build_scene({0: synthetic.make_joiner(0), 1: synthetic.make_joiner(1), 2: synthetic.make_joiner(2, 'rotate')})
B = S['big']
c = PS.cap_big(B)
o, d = PS.frame_rays(c)
cu = TP.cu
given = {'near_far': [(cu(B['posed_near']), cu(B['posed_far']))]}
trc = {}
rgb, depth, acc = B['R'].render_smpl_nerf_rays(B['dev_nets'][2], cu(o), cu(d), cu(B['posed_verts']), B['mesh'], 128, True, False, 0.2, 1.0, given=given, trace=trc)
e = np.abs(rgb.cpu().numpy() - B['posed_rgb'].reshape(-1, 3)).max(-1)
hit = trc['hit'][0].cpu().numpy()
cp, cd, z = trc['can_pts'][0].cpu().numpy(), trc['can_dirs'][0].cpu().numpy(), trc['human_z'][0].cpu().numpy()
pts = (o[hit, None, :] + d[hit, None, :] * z[..., None]).astype(np.float32)
ocp, ocd, _ = warp.warp_samples_to_canonical(pts, B['posed_verts'], B['faces'], B['T'])
dev_p = np.abs(ocp - cp).max((-1, -2))
dev_d = np.abs(ocd - cd).max((-1, -2))
spacing = (z[:, -1] - z[:, 0]) / 127
eh = e[hit]
print("hit rays", hit.size, "e > 1e-4:", (eh > 1e-4).sum())
for name, m in (("pts dev > 1e-5", dev_p > 1e-5), ("dirs dev > 5e-4", dev_d > 5e-4), ("dirs dev > 2e-3", dev_d > 2e-3), ("spacing < 1e-3", spacing < 1e-3), ("spacing < 3e-4", spacing < 3e-4)):
    print(f"{name}: {m.sum()} rays, of which e > 1e-4: {(m & (eh > 1e-4)).sum()}; Linf over the others {eh[~m].max():.2e}")
for tp, td, ts in ((1e-5, 5e-4, 1e-3), (1e-5, 2e-3, 1e-3), (1e-5, 1e9, 1e-3), (1e-5, 1e-3, 5e-4)):
    m = (dev_p > tp) | (dev_d > td) | (spacing < ts)
    print(f"flag pts>{tp} | dirs>{td} | spacing<{ts}: {m.sum()} flagged ({m.mean() * 100:.1f} %), Linf over the rest {eh[~m].max():.2e}, e>5e-5 among the rest {(eh[~m] > 5e-5).sum()}")
print("rays with e > 5e-5: ray e dev_p dev_d spacing")
for k in np.nonzero(eh > 5e-5)[0]:
    print(int(hit[k]), f"{eh[k]:.2e} {dev_p[k]:.2e} {dev_d[k]:.2e} {spacing[k]:.2e}")
