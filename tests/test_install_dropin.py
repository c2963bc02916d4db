"""The drop-in boundary, executed: neuman_hip.install() over the REFERENCE's own modules (imported unmodified from
/root/reference with the absent wheels stubbed, as tests/golden/make_golden.py does), in a separate interpreter.
Build container only: skipped where /root/reference does not exist (the GPU box)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def report():
    if not os.path.isdir("/root/reference/utils"):
        pytest.skip("/root/reference is not present on this machine")
    r = subprocess.run([sys.executable, os.path.join(HERE, "helpers", "dropin_check.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_every_hot_path_name_is_rebound_with_the_reference_signature(report):
    sigs = report["signatures"]
    assert len(sigs) == 17 and report["rebound"] == 17
    bad = {k: v for k, v in sigs.items() if not v["ok"]}
    assert not bad, bad


def test_reference_human_nerf_builds_on_the_rebound_modules(report):
    h = report["human_nerf"]
    assert h["state_dict_tensors"] == 90 and h["parameters"] == 2292111          # SURVEY 8c [probe]
    assert h["bkg_is_ours"] and h["human_mapping"] == "rotate"


def test_checkpoints_load_strictly_both_ways(report):
    s = report["state_dict"]
    assert s["keys_equal"] and s["shapes_equal"] and s["values_round_trip"]
    assert s["reference_class"] == "models.vanilla.Joiner"


def test_human_nerf_module_has_the_references_state_dict(report):
    """neuman_hip.human_nerf.HumanNeRF(opt, poses, betas, alignments, scale) against models/human_nerf.py's (its hard-coded SMPL asset
    path redirected to a synthetic model; train.py:103's constructor call): the 101 `hybrid_model_state_dict` keys -- networks, offset
    nets, poses / betas / alignments / da_smpl, body_model.* buffers -- equal in name, shape and dtype, strict loads in both directions,
    vertex_forward equal to float32 round-off"""
    h = report["human_nerf_with_body"]
    assert h["keys_equal"] and h["shapes_equal"] and h["n_keys"] == 101 and not h["only_reference"] and not h["only_ours"], h
    assert h["reference_class_module"] == "models.vanilla"
    assert max(h["vertex_forward_linf"]) < 5e-6, h["vertex_forward_linf"]
