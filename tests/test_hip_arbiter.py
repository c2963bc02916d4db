"""-m gpu: whole 800x800 two-pass frames against the ARBITER -- frames the REFERENCE ITSELF rendered in float64 on identical rays and weights
(tests/golden/arbiter.npz / arbiter_full.npz, made by tests/golden/make_golden_f64.py), beside the reference's own float32 frames.

    fog00      synthetic 'fog' preset as coarse and fine net: density positive everywhere, the inverse CDF well conditioned on every ray.  The
               UNCONDITIONAL statement of the contract: EVERY ray within 1e-4 of the reference (rows 5::10 of the frame: 64 000 rays; and every one of
               the 640 000 rays when arbiter_full.npz is present)
    opaque00   surfaces: the reference's own float32 run leaves 3 of 64 000 rays beyond 1e-4; the device is held to that count + margin
"""
import os

import numpy as np
import pytest
import torch

from oracle import attribution

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frame_rays():
    from neuman_hip import ray_utils, synthetic
    return ray_utils.shot_all_rays_dev(synthetic.SimpleCapture(800, 800), torch.device('cuda'))


def render(preset, precision, frame_rays):
    from neuman_hip import render_utils, synthetic
    net = synthetic.make_joiner(0, preset=preset).cuda()
    o, d = frame_rays
    with torch.no_grad():
        return render_utils.render_vanilla_rays(net, net, o, d, 0.0, 3.14, 128, 128, True, precision=precision)[0].cpu().numpy()


@pytest.mark.parametrize("precision", ["mixed", "fp16x3"])
def test_well_conditioned_frame_every_ray_within_1e_4_of_the_reference(frame_rays, precision):
    rgb = render('fog', precision, frame_rays)
    arb = attribution.load_arbiter("wc_fog00")
    rows = rgb.reshape(800, 800, 3)[arb["rows"]].reshape(-1, 3)
    rep, _ = attribution.against_arbiter(rows, arb, tag=f"fog00 {precision}, rows 5::10")
    assert rep["reference_f32_vs_reference_f64"]["rgb_linf"] < 1e-5          # the workload IS well conditioned: the reference's own float32 run says so
    assert rep["vs_reference_f64"]["rays_gt_1e-4"] == 0 and rep["vs_reference_f64"]["rgb_linf"] < 1e-4
    if os.path.exists(attribution.ARBITER_FULL):
        full = attribution.load_arbiter_full()
        assert full["name"] == "fog00"
        rep, _ = attribution.against_arbiter(rgb, full, tag=f"fog00 {precision}, EVERY ray of the frame")
        assert rep["rays"] == 640000 and rep["vs_reference_f64"]["rays_gt_1e-4"] == 0 and rep["vs_reference_f64"]["rgb_linf"] < 1e-4


@pytest.mark.parametrize("precision", ["mixed", "fp16x3"])
def test_surface_workload_sits_where_the_references_float32_run_sits(frame_rays, precision):
    rgb = render('opaque', precision, frame_rays)
    arb = attribution.load_arbiter("wc_opaque00")
    rows = rgb.reshape(800, 800, 3)[arb["rows"]].reshape(-1, 3)
    rep, fails = attribution.against_arbiter(rows, arb, tag=f"opaque00 {precision}, rows 5::10")
    assert not fails, fails
