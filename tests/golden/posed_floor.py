"""The floor for END-TO-END parity on tests/golden/posed.npz: how far two float32 CPU evaluations of the reference's algorithm -- the
oracle (oracle/render.py) and the reference's own run (the goldens) -- sit from each other on the same frames with nothing
replayed.  Prints, per frame, the number of rays beyond 1e-4; tests/test_hip_posed_golden.py holds the device to 1.5 x these.

    python tests/golden/posed_floor.py [--big]    (CPU, ~4 min; needs nothing but this repo; --big: the 64 x 64 frames of posed_big.npz)
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

from oracle import render  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))
import posed_scene as PS  # noqa: E402


def main_big():
    S = PS.load_big()
    nets = PS.oracle_nets()
    out = {}
    c = PS.cap_big(S)
    rgb = render.render_smpl_nerf(nets[2], c, S['posed_verts'], S['faces'], S['T'], rays_per_batch=1024, samples_per_ray=128)
    e = np.abs(rgb - S['posed_rgb']).max(-1)
    out['posed_big'] = {'rays_gt_1e-4': int((e > 1e-4).sum()), 'linf': float(e.max()), 'hit_rays': int((S['posed_near'] < S['posed_far']).sum())}
    rgb = render.render_hybrid_nerf(nets[0], nets[1], nets[2], c, S['posed_verts'], S['faces'], S['T'], rays_per_batch=1024, samples_per_ray=128,
                                    importance_samples_per_ray=128)
    e = np.abs(rgb - S['hybrid_rgb']).max(-1)
    out['hybrid_big'] = {'rays_gt_1e-4': int((e > 1e-4).sum()), 'linf': float(e.max()), 'rays': int(e.size)}
    print(json.dumps(out))


def main():
    S = PS.load()
    nets = PS.oracle_nets()
    out = {}
    c = PS.cap(S, 'posed')
    rgb = render.render_smpl_nerf(nets[2], c, S['posed_verts'], S['faces'], S['T'], rays_per_batch=512, samples_per_ray=128)
    e = np.abs(rgb - S['posed_rgb']).max(-1)
    out['posed'] = {'rays_gt_1e-4': int((e > 1e-4).sum()), 'linf': float(e.max()), 'hit_rays': int((S['posed_near'] < S['posed_far']).sum())}
    c = PS.cap(S, 'hybrid')
    rgb = render.render_hybrid_nerf(nets[0], nets[1], nets[2], c, S['posed_verts'], S['faces'], S['T'], rays_per_batch=512, samples_per_ray=128,
                                    importance_samples_per_ray=128)
    e = np.abs(rgb - S['hybrid_rgb']).max(-1)
    out['hybrid'] = {'rays_gt_1e-4': int((e > 1e-4).sum()), 'linf': float(e.max()), 'rays': int(e.size)}
    c = PS.cap(S, 'multi')
    rgb = render.render_hybrid_nerf_multi_persons(nets[0], nets[1], [nets[2]] * 3, c, S['posed_l'], [S['faces']] * 3, S['T_l'], rays_per_batch=512,
                                                  samples_per_ray=192, importance_samples_per_ray=128)
    e = np.abs(rgb - S['multi_rgb']).max(-1)
    out['multi'] = {'rays_gt_1e-4': int((e > 1e-4).sum()), 'linf': float(e.max()), 'rays': int(e.size)}
    print(json.dumps(out))


if __name__ == '__main__':
    main_big() if '--big' in sys.argv else main()
