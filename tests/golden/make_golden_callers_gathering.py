"""Frames made by the REFERENCE through the call sequence of render_gathering.py (BASELINE config 5's script; build container only):

    python tests/golden/make_golden_callers_gathering.py   ->  tests/golden/callers_gathering.npz

tests/helpers/caller_bodies.py `gathering` is the loop of render_gathering.py:189-202 written against a namespace of modules; here the namespace is the
reference's own utils.render_utils (imported unmodified; igl = tests/golden/igl_shim.py), the background net and the three actors' nets are the reference's
own HumanNeRF(opt) with synthetic weights, the captures the reference's BasePinholeCapture.  The GPU test runs the same body through neuman_hip.install()."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_posed as MP  # noqa: E402  (igl shim + stubs + the reference's modules)
from make_golden_posed import R_render, synthetic, PinholeCamera, CameraPose, BasePinholeCapture  # noqa: E402
import make_golden_callers as MC  # noqa: E402  (parse_opt)

sys.path.insert(0, os.path.join(MP.ROOT, "tests", "helpers"))
import caller_bodies as CB  # noqa: E402
from models import human_nerf as R_human_nerf  # noqa: E402

ACTOR_SEEDS = (2, 3, 4)


def human_net(opt, human_seed):
    with contextlib.redirect_stdout(io.StringIO()):
        net = R_human_nerf.HumanNeRF(opt)
    for sub, seed, mp in ((net.coarse_bkg_net, 0, 'posenc'), (net.fine_bkg_net, 1, 'posenc'), (net.coarse_human_net, human_seed, 'rotate')):
        sub.load_state_dict(synthetic.make_joiner(seed, mp).state_dict(), strict=True)
    return net.eval()


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    M = type('M', (), {'render_utils': R_render})
    opt = MC.parse_opt(['--rays_per_batch', '512', '--samples_per_ray', str(CB.SG)])
    opt.geo_threshold = 0.2
    bkg_net = human_net(opt, 2)
    nets_list = [human_net(opt, s) for s in ACTOR_SEEDS]                 # read_actor: one HumanNeRF per actor
    g = CB.gathering_inputs()
    caps = []
    for th in (15., -25.):
        cap = BasePinholeCapture(PinholeCamera(CB.WG, CB.HG, 70.0, 70.0, CB.WG / 2, CB.HG / 2), CameraPose.from_camera_to_world(synthetic.spherical_c2w(th, -8., 3.0)))
        cap.near, cap.far = {'bkg': 0.5}, {'bkg': 3.14}
        caps.append(cap)
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        frames = CB.gathering(M, bkg_net, nets_list, lambda i: caps[i], CB.NG, g['verts_list'], g['faces'], g['Ts_list'], opt)
    print(f"gathering {frames.shape} {frames.dtype} in {time.time() - t0:.1f} s; non-background pixels {(frames.min(-1) < 1).mean():.2f}")
    np.savez_compressed(os.path.join(HERE, 'callers_gathering.npz'), frames=frames.astype(np.float32), c2w=np.stack([c.cam_pose.camera_to_world for c in caps]),
                        actor_seeds=np.array(ACTOR_SEEDS))


if __name__ == '__main__':
    main()
