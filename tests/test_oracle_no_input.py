"""`--include_input ''` (Embedder(include_input=False), models/vanilla.py:56-58, 63-65, 87-88): the oracle and the host mirror against the reference's
own outputs (tests/golden/no_input.npz, made by make_golden_no_input.py from the reference imported unmodified).  CPU only."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import nerf_mlp, render
from oracle.nerf_mlp import JoinerSpec

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def N():
    return dict(np.load(os.path.join(HERE, "golden", "no_input.npz")))


def checksum(j):
    sd = j.state_dict()
    return np.array([float(sum(v.abs().sum(dtype=torch.float64) for v in sd.values())), float(sd['nerf.pts_linears.0.weight'][0, 0]),
                     float(sd['nerf.pts_linears.7.bias'][5])])


def variant(seed, key, N, **over):
    from neuman_hip import synthetic
    j = synthetic.make_variant_joiner(seed, include_input=False, **over)
    np.testing.assert_allclose(checksum(j), N[key], rtol=1e-6, err_msg="nn.Linear default init changed: regenerate tests/golden/no_input.npz")
    return j, synthetic.state_numpy(j)


def test_host_mirror_shapes_and_errors():
    from neuman_hip import vanilla
    e = vanilla.Embedder(3, 9, 10, True, False)
    assert e.out_dim == 60 and vanilla.Embedder(3, 3, 4, True, False, mapping='rotate').out_dim == 24 and vanilla.Embedder(4, 9, 10, True, False).out_dim == 80
    assert vanilla.Embedder(3, 9, 10).out_dim == 63 and vanilla.Embedder(3, 9, 10, mapping='rotate').out_dim == 63
    with pytest.raises(AssertionError):                                   # the reference's constructor stops at `assert 0` (vanilla.py:70)
        vanilla.Embedder(3, 9, 10, log_sampling=False)
    vanilla.Embedder(3, 9, 10, log_sampling=False, mapping='rotate')       # its rotate encoding never reads the flag
    d = vanilla.Embedder(3, 3, 4, True, False)
    nerf = vanilla.NeRF(input_ch=60, input_ch_views=24, use_viewdirs=True)
    pads = vanilla.absent_input_columns(e, d, nerf)
    assert pads == {0: (0, 3), 10: (0, 3), 16: (256, 3)}
    wide = vanilla.with_absent_columns(nerf.ordered_params(), pads)
    assert wide[0].shape == (256, 63) and wide[10].shape == (256, 319) and wide[16].shape == (128, 283)
    assert float(wide[0][:, :3].detach().abs().max()) == 0 and torch.equal(wide[0][:, 3:], nerf.pts_linears[0].weight)
    assert float(wide[16][:, 256:259].detach().abs().max()) == 0 and torch.equal(wide[16][:, 259:], nerf.views_linears[0].weight[:, 256:])
    assert torch.equal(wide[16][:, :256], nerf.views_linears[0].weight[:, :256]) and torch.equal(wide[10][:, 3:], nerf.pts_linears[5].weight)
    assert vanilla.absent_input_columns(vanilla.Embedder(3, 9, 10), vanilla.Embedder(3, 3, 4), nerf) == {}
    # the 4-D encoding's columns (the time-conditioned net): no leading copy of (x, y, z, t)
    sp, tc = vanilla.time_columns(vanilla.Embedder(4, 9, 10, True, False))
    assert len(sp) == 60 and len(tc) == 20 and sorted(sp + tc) == list(range(80)) and sp[:6] == [0, 1, 2, 4, 5, 6] and tc[:2] == [3, 7]
    assert len(vanilla.time_encoding(vanilla.Embedder(4, 9, 10, True, False), 0.3)) == 20


@pytest.mark.parametrize("mapping", ["posenc", "rotate"])
def test_oracle_forward(N, mapping):
    j, sd = variant(11, f'{mapping}_checksum', N, posenc=mapping)
    assert sd['nerf.pts_linears.0.weight'].shape == (256, 60) and sd['nerf.views_linears.0.weight'].shape == (128, 280)
    out = nerf_mlp.joiner_forward(sd, JoinerSpec(mapping=mapping, include_input=False), N['pts'], N['dirs'])
    s = 30 if mapping == 'rotate' else 1
    assert np.abs(out - N[f'{mapping}_out']).max() < 2e-5 * s


def test_oracle_plain_head_and_frame(N):
    j, sd = variant(12, 'plain_checksum', N, use_viewdirs=False)
    out = nerf_mlp.joiner_forward(sd, JoinerSpec(include_input=False), N['pts'], N['dirs'])
    assert np.abs(out - N['plain_out']).max() < 2e-5
    _, sdc = variant(14, 'frame_coarse_checksum', N)
    _, sdf = variant(15, 'frame_fine_checksum', N)
    cap = types.SimpleNamespace(shape=(18, 24), intrinsic_matrix=np.array([[30., 0, 12.], [0, 30., 9.], [0, 0, 1.]]),
                                cam_pose=types.SimpleNamespace(camera_to_world=np.eye(4)), near={'bkg': 0.0}, far={'bkg': 3.14})
    spec = JoinerSpec(include_input=False)
    rgb1 = render.render_vanilla((sdc, spec), cap, None, rays_per_batch=256, samples_per_ray=16)
    assert np.abs(rgb1 - N['frame_coarse_only_rgb']).max() < 5e-6
    rgb, depth = render.render_vanilla((sdc, spec), cap, (sdf, spec), rays_per_batch=256, samples_per_ray=16, importance_samples_per_ray=16, return_depth=True)
    err = np.abs(rgb - N['frame_rgb']).max(-1)
    assert (err > 1e-4).mean() < 0.02 and err.max() < 2e-2              # two-pass: the inverse CDF's conditioning
