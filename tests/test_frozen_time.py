"""CPU: vanilla.frozen_time_joiner -- the time-conditioned net of `--ablate_nerft` (4-D position encoding, ray_utils.py:133-134, 158-159) at one
frame time as a 3-D-encoding net with two other bias vectors -- against the oracle's evaluation of the 4-D net itself (oracle/nerf_mlp.py,
pinned on the reference's outputs in tests/golden/heads.npz).  The fold is exact in real arithmetic; float32 leaves a few 1e-7."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
from neuman_hip import synthetic, vanilla  # noqa: E402
from oracle import nerf_mlp  # noqa: E402
from oracle.nerf_mlp import JoinerSpec  # noqa: E402


def test_time_columns_cover_the_encoding():
    j = synthetic.make_variant_joiner(6, raw_pos_dim=4)
    sp, tc = vanilla.time_columns(j.pos_pe)
    assert len(sp) == 63 and len(tc) == 21 and sorted(sp + tc) == list(range(84))
    x = np.array([[0.3, -0.7, 1.1, 0.35]], np.float32)
    full = nerf_mlp.embed(x, 'posenc', 0, 9, 10)[0]
    np.testing.assert_allclose(full[sp], nerf_mlp.embed(x[:, :3], 'posenc', 0, 9, 10)[0], rtol=0, atol=0)      # the spatial columns ARE the 3-D encoding, in its order
    np.testing.assert_allclose(full[tc], vanilla.time_encoding(j.pos_pe, 0.35), rtol=0, atol=3e-7)


def test_folded_net_equals_the_time_conditioned_net():
    rng = np.random.default_rng(0)
    for seed, t in ((6, 0.35), (7, 0.0), (6, 0.95)):
        j4 = synthetic.make_variant_joiner(seed, raw_pos_dim=4)
        j3 = vanilla.frozen_time_joiner(j4, t)
        assert j3.pos_pe.input_dims == 3 and j3.nerf.pts_linears[0].weight.shape == (256, 63) and j3.nerf.pts_linears[5].weight.shape == (256, 319)
        assert j3.nerf.pts_linears[2].weight is j4.nerf.pts_linears[2].weight and j3.nerf.rgb_linear is j4.nerf.rgb_linear       # shared, not copied
        assert vanilla.frozen_time_joiner(j4, t) is j3                                                                             # cached
        pts = rng.uniform(-1.5, 1.5, size=(500, 3)).astype(np.float32)
        dirs = rng.normal(size=(500, 3)).astype(np.float32)
        dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
        pts4 = np.concatenate([pts, np.full((500, 1), t, np.float32)], 1)
        want = nerf_mlp.joiner_forward(synthetic.state_numpy(j4), JoinerSpec(), pts4, dirs)
        got = nerf_mlp.joiner_forward(synthetic.state_numpy(j3), JoinerSpec(), pts, dirs)
        e = np.abs(got - want).max() / max(1.0, np.abs(want).max())
        assert e < 2e-5, (seed, t, e)
    with torch.no_grad():                                                   # a weight edit invalidates the cached fold
        j4.nerf.pts_linears[0].bias.add_(1.0)
    assert vanilla.frozen_time_joiner(j4, 0.95) is not j3
