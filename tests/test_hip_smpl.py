"""-m gpu: SMPL linear blend skinning on the device (nm_smpl_frames) vs the reference goldens and the CPU oracle."""
import os
import pickle
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))
from oracle import smpl as OS  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G():
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    from neuman_hip import smpl as HS, synthetic
    g = dict(np.load(os.path.join(ROOT, "tests", "golden", "smpl.npz")))
    data = synthetic.smpl_like_model(0)
    pose, betas, align = synthetic.smpl_like_frames(int(g['n_frames']), 0)
    return types.SimpleNamespace(g=g, data=data, pose=pose, betas=betas, align=align, HS=HS, syn=synthetic, body=HS.SMPL(data), omodel=OS.Model(data))


def close(a, b, tol=2e-5):
    scale = max(1.0, float(np.abs(b).max()))
    err = float(np.abs(np.asarray(a, np.float64) - b).max())
    assert err <= tol * scale, f"max abs err {err:.3e} > {tol * scale:.3e}"
    return err


def align44(G, i):
    a = np.eye(4)
    a[:, :3] = G.align[f"{i:05d}.png"]
    return a


def test_frames_match_reference_read_smpls(G):
    """the render scripts' chain (neuman_helper.py:288-330), all frames in one batch"""
    g, n = G.g, int(G.g['n_frames'])
    T, world, static = G.body.frames(G.pose, G.betas, np.stack([align44(G, i) for i in range(n)]), float(g['scale']), True)
    assert T.dtype == torch.float64 and T.shape == (n, 6914, 4, 4) and world.shape == (n, 6914, 3)
    T, world, static = T.cpu().numpy(), world.cpu().numpy(), static.cpu().numpy()
    rows = g['rows']
    vr = rows[rows < 6890]
    for i in range(n):
        e = [close(T[i][rows], g['Ts'][i]), close(world[i][vr], g['world_verts'][i]), close(static[i][vr], g['static_verts'][i]),
             close(world[i][6890:], g['joints_3d'][i]), close(static[i][6890:], g['static_joints_3d'][i])]
        print(f"[smpl] frame {i} vs reference: max abs err Ts {e[0]:.2e} world {e[1]:.2e} static {e[2]:.2e} joints {e[3]:.2e} / {e[4]:.2e}")
        # and against the oracle on EVERY row (the fixture keeps 184 of 6914)
        wv, wj, sv, sj, Ts = OS.read_smpl_frame(G.omodel, G.pose[i], G.betas[i], G.align[f"{i:05d}.png"], float(g['scale']))
        close(T[i], Ts)
        close(world[i], np.concatenate([wv, wj]))
        close(static[i], np.concatenate([sv, sj]))


def test_vertex_forward_matches_reference(G):
    g = G.g
    vr = g['rows'][g['rows'] < 6890]
    a = np.concatenate([G.align["00001.png"], np.array([[0.], [0.], [0.], [1.]])], 1).astype(np.float32)
    wv, T = G.HS.vertex_forward(G.body, G.pose[1:2], G.betas[1:2], a, float(g['scale']))
    assert wv.shape == (1, 6890, 3) and T.shape == (1, 6890, 4, 4) and T.dtype == torch.float32 and T.is_cuda
    close(T[0].cpu().numpy()[vr], g['vf_T'])
    close(wv[0].cpu().numpy()[vr], g['vf_world_verts'])


def test_read_smpls_reads_the_reference_files(G, tmp_path):
    """same files on disk as NeuManReader.read_smpls: SMPL_NEUTRAL.pkl, smpl_output_romp.pkl (joblib), alignments.npy"""
    import joblib
    g, n = G.g, int(G.g['n_frames'])
    with open(tmp_path / 'SMPL_NEUTRAL.pkl', 'wb') as f:
        pickle.dump(G.data, f, protocol=2)
    joblib.dump({1: {'pose': G.pose, 'betas': G.betas}}, tmp_path / 'smpl_output_romp.pkl')
    np.save(tmp_path / 'alignments.npy', G.align, allow_pickle=True)
    caps = [types.SimpleNamespace(image_path=str(tmp_path / 'images' / f'{i:05d}.png')) for i in (2, 0)]      # any order, any subset
    smpls, world_verts, static_verts, Ts = G.HS.read_smpls(str(tmp_path), caps, scale=float(g['scale']), model_dir=str(tmp_path))
    rows = g['rows']
    vr = rows[rows < 6890]
    for k, i in enumerate((2, 0)):
        assert world_verts[k].shape == (6890, 3) and world_verts[k].dtype == np.float32 and Ts[k].shape == (6914, 4, 4) and Ts[k].dtype == np.float64
        close(Ts[k][rows], g['Ts'][i])
        close(world_verts[k][vr], g['world_verts'][i])
        close(static_verts[k][vr], g['static_verts'][i])
        close(smpls[k]['joints_3d'], g['joints_3d'][i])
        np.testing.assert_array_equal(smpls[k]['pose'], G.pose[i])


def test_identity_pose_and_errors(G):
    """pose == da pose, identity alignment, scale 1: T_da2scene is the identity and the world vertices are the da-pose ones"""
    da = G.HS.da_pose()
    T, world, static = G.body.frames(da[None], G.betas[:1], np.eye(4)[None], 1.0, True)
    np.testing.assert_allclose(T.cpu().numpy()[0], np.tile(np.eye(4), (6914, 1, 1)), atol=5e-6)
    np.testing.assert_allclose(world.cpu().numpy(), static.cpu().numpy(), atol=5e-6)
    from neuman_hip import _lib
    with pytest.raises(_lib.NeumanHipError):
        G.body.frames(G.pose, G.betas[:1], np.eye(4)[None], 1.0)
    bad = dict(G.data)
    bad['kintree_table'] = np.stack([np.arange(24)[::-1].copy(), np.arange(24)])
    with pytest.raises(_lib.NeumanHipError):
        G.HS.SMPL(bad)


def test_posed_mesh_feeds_the_warp(G):
    """end of the row: the skinned mesh and its transforms drive nm_warp_to_canonical; a posed vertex maps back to its
    da-pose position (T_da2scene is canonical -> scene, the warp inverts it)"""
    from neuman_hip import ray_utils
    T, world, static = G.body.frames(G.pose[:1], G.betas[:1], align44(G, 0)[None], float(G.g['scale']), True)
    faces = np.ascontiguousarray(G.data['f'].astype(np.int32))
    mesh = ray_utils.Mesh(world[0, :6890], faces, T[0], 'cuda')
    idx = torch.arange(0, 6890, 53, device='cuda')
    pts = world[0, idx].reshape(-1, 2, 3).contiguous()                 # [R, S = 2, 3]
    can, _, _ = ray_utils.warp_to_canonical_dev(pts, mesh)
    np.testing.assert_allclose(can.reshape(-1, 3).cpu().numpy(), static[0, idx].cpu().numpy(), atol=2e-4)
