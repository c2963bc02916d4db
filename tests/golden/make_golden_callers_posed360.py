"""Frames made by the REFERENCE through the call sequence of render_360.py's main_posed_360 (:108-126; build container only):

    python tests/golden/make_golden_callers_posed360.py   ->  tests/golden/callers_posed360.npz

tests/helpers/caller_bodies.py `posed_360` on the reference's own utils.render_utils (imported unmodified; igl = tests/golden/igl_shim.py), its HumanNeRF(opt) with
synthetic weights, its ResizedPinholeCapture and its default_360_path around the posed body (camera distance geo_threshold x 36 as the script sets it)."""
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
from make_golden_posed import R_render, PinholeCamera  # noqa: E402
import make_golden_callers as MC  # noqa: E402
import make_golden_callers_gathering as MG  # noqa: E402

sys.path.insert(0, os.path.join(MP.ROOT, "tests", "helpers"))
import caller_bodies as CB  # noqa: E402
from cameras.captures import ResizedPinholeCapture  # noqa: E402  (reference)


def main():
    torch.set_num_threads(os.cpu_count() or 1)
    M = type('M', (), {'render_utils': R_render})
    opt = MC.parse_opt(['--rays_per_batch', '1024', '--samples_per_ray', str(CB.SP)])
    opt.geo_threshold = 0.2
    opt.white_bkg = True
    net = MG.human_net(opt, 2)
    inp = CB.scene_inputs()
    verts, T = inp['verts'][0], inp['Ts'][0]
    center, up = np.asarray(verts, np.float64).mean(0), np.array([0.0, 1.0, 0.0])       # (utils.smpl_verts_to_center_and_up needs the licensed joint regressor)
    with MC.numpy2_array_copy_shim():
        poses = R_render.default_360_path(center, up, opt.geo_threshold * 36, CB.NP)
    base = PinholeCamera(CB.WP * 4, CB.HP * 4, 5.5 * CB.WP * 4, 5.5 * CB.WP * 4, CB.WP * 2.0, CB.HP * 2.0)

    def cap(i):
        return ResizedPinholeCapture(base, poses[i], tgt_size=(CB.HP, CB.WP))
    t0 = time.time()
    with contextlib.redirect_stdout(io.StringIO()):
        frames = CB.posed_360(M, net, cap, CB.NP, verts, inp['faces'], T, opt)
    print(f"posed_360 {frames.shape} {frames.dtype} in {time.time() - t0:.1f} s; hit fraction {(frames.min(-1) < 1).mean():.2f}")
    np.savez_compressed(os.path.join(HERE, 'callers_posed360.npz'), frames=frames.astype(np.float32), c2w=np.stack([p.camera_to_world for p in poses]),
                        K=cap(0).intrinsic_matrix)


if __name__ == '__main__':
    main()
