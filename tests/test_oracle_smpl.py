"""CPU: oracle/smpl.py against outputs of the reference itself (tests/golden/smpl.npz, made by make_golden_smpl.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
from neuman_hip import synthetic  # noqa: E402
from oracle import smpl as OS  # noqa: E402


@pytest.fixture(scope="module")
def G():
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "smpl.npz")))
    data = synthetic.smpl_like_model(0)
    chk = np.array([np.abs(v).sum(dtype=np.float64) for k, v in sorted(data.items()) if k != 'f'])
    np.testing.assert_allclose(chk, g['model_checksum'], rtol=1e-12, err_msg="synthetic SMPL-layout model changed: regenerate the golden")
    pose, betas, align = synthetic.smpl_like_frames(int(g['n_frames']), 0)
    return g, OS.Model(data), pose, betas, align


def close(a, b, tol=1e-5):
    """float32 chains through BLAS in both: tolerance relative to the magnitude of the values"""
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(np.asarray(a, np.float64) - b).max())
    assert err <= tol * scale, f"max abs err {err:.3e} > {tol * scale:.3e}"
    return err


def test_rodrigues_matches_reference(G):
    g, model, pose, betas, align = G
    r = OS.batch_rodrigues(pose[0].reshape(-1, 3))
    assert r.dtype == np.float32
    close(r, g['rodrigues'], 2e-6)
    assert np.abs(r[2] - np.eye(3)).max() < 1e-6          # the joint at rest


def test_lbs_matches_reference(G):
    g, model, pose, betas, align = G
    T, v, _, _ = OS.lbs(model, betas[0], pose[0], concat_joints=True)
    rows = g['rows']
    close(T[rows], g['lbs_T0'])
    close(v[rows], g['lbs_v0'])


def test_read_smpls_chain_matches_reference(G):
    g, model, pose, betas, align = G
    rows = g['rows']
    vr = rows[rows < 6890]
    for i in range(int(g['n_frames'])):
        wv, wj, sv, sj, Ts = OS.read_smpl_frame(model, pose[i], betas[i], align[f"{i:05d}.png"], float(g['scale']))
        assert Ts.dtype == np.float64 and wv.dtype == np.float32
        e = [close(Ts[rows], g['Ts'][i], 2e-5), close(wv[vr], g['world_verts'][i], 2e-5), close(sv[vr], g['static_verts'][i]),
             close(wj, g['joints_3d'][i], 2e-5), close(sj, g['static_joints_3d'][i])]
        print(f"[oracle smpl] frame {i}: max abs err Ts {e[0]:.2e} world_verts {e[1]:.2e} static {e[2]:.2e} joints {e[3]:.2e}")


def test_vertex_forward_matches_reference(G):
    g, model, pose, betas, align = G
    vr = g['rows'][g['rows'] < 6890]
    a = np.concatenate([align["00001.png"], np.array([[0.], [0.], [0.], [1.]])], 1).astype(np.float32)
    wv, T = OS.vertex_forward(model, pose[1], betas[1], a, float(g['scale']))
    assert T.dtype == np.float32
    close(T[vr], g['vf_T'], 2e-5)
    close(wv[vr], g['vf_world_verts'], 2e-5)
