"""Golden vectors for SURVEY row a12 (SMPL linear blend skinning -> per-frame verts / Ts), from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_smpl.py

What runs is the reference's own code, unmodified: models/smpl.py (SMPL, lbs, batch_rodrigues, batch_rigid_transform),
data_io/neuman_helper.py:NeuManReader.read_smpls (the numpy chain the render scripts use, :258-331) and
models/human_nerf.py:HumanNeRF.vertex_forward (:92-122).  The licensed SMPL asset is absent, so the harness writes a
synthetic model with the same file layout (neuman_hip.synthetic.smpl_like_model) to a temporary SMPL_NEUTRAL.pkl, the
per-frame parameters to smpl_output_romp.pkl (joblib) and alignments.npy -- the on-disk formats of SURVEY 8f-3 -- and
redirects the reader's hard-coded asset directory (data/smplx/smpl under the read-only reference) to it.
The outputs are large ([6914,4,4] f64 per frame), so the fixture keeps every joint row and a fixed sample of vertex rows.
"""
import os
import pickle
import sys
import tempfile
import types
from unittest import mock

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "ml-neuman_amd"))

import joblib  # noqa: E402
from models import smpl as R_smpl, human_nerf as R_hn  # noqa: E402  (reference)
from data_io import neuman_helper as R_nh  # noqa: E402
from neuman_hip import synthetic  # noqa: E402  (ours: workload definitions only)

N_FRAMES, SCALE, SEED = 3, 1.37, 0


def main():
    model = synthetic.smpl_like_model(SEED)
    pose, betas, align = synthetic.smpl_like_frames(N_FRAMES, SEED)
    rng = np.random.default_rng(5)
    rows = np.sort(rng.choice(6890, 160, replace=False))
    rows = np.concatenate([rows, np.arange(6890, 6890 + 24)])          # + every joint row
    out = {'rows': rows, 'scale': np.float64(SCALE), 'n_frames': np.int64(N_FRAMES),
           'model_checksum': np.array([np.abs(v).sum(dtype=np.float64) for k, v in sorted(model.items()) if k != 'f'])}
    with tempfile.TemporaryDirectory() as tmp:
        with open(os.path.join(tmp, 'SMPL_NEUTRAL.pkl'), 'wb') as f:
            pickle.dump(model, f, protocol=2)
        joblib.dump({1: {'pose': pose, 'betas': betas}}, os.path.join(tmp, 'smpl_output_romp.pkl'))
        np.save(os.path.join(tmp, 'alignments.npy'), align, allow_pickle=True)
        real_smpl = R_smpl.SMPL
        R_nh.SMPL = lambda path, gender='neutral', device=None: real_smpl(tmp, gender=gender, device=device)
        caps = [types.SimpleNamespace(image_path=os.path.join(tmp, 'images', f'{i:05d}.png')) for i in range(N_FRAMES)]
        smpls, world_verts, static_verts, Ts = R_nh.NeuManReader.read_smpls(tmp, caps, scale=SCALE, smpl_type='romp')
        body = real_smpl(tmp, gender='neutral', device=torch.device('cpu'))
    out['world_verts'] = np.stack(world_verts)[:, rows[rows < 6890]]                 # [n, 160, 3] f32
    out['static_verts'] = np.stack(static_verts)[:, rows[rows < 6890]]               # [n, 160, 3] f32 (da-pose vertices)
    out['Ts'] = np.stack(Ts)[:, rows]                                                # [n, 184, 4, 4] f64
    out['joints_3d'] = np.stack([s['joints_3d'] for s in smpls])                     # [n, 24, 3] f32
    out['static_joints_3d'] = np.stack([s['static_joints_3d'] for s in smpls])       # [n, 24, 3] f32
    # the building blocks, frame 0: what lbs itself returns
    v0, T0 = body.verts_transformations(poses=pose[:1], betas=betas[:1], return_tensor=False, concat_joints=True)
    out['lbs_T0'] = T0[rows]                                                         # [184, 4, 4] f32
    out['lbs_v0'] = v0[rows]                                                         # v_shaped rows + rest joints
    out['rodrigues'] = R_smpl.batch_rodrigues(torch.from_numpy(pose[0].reshape(-1, 3))).numpy()
    # HumanNeRF.vertex_forward (torch f32 chain) on a stand-in `self` with exactly the attributes the method reads
    da = np.zeros((24, 3), np.float32)
    da[1], da[2] = (0, 0, 1.0), (0, 0, -1.0)
    fake = types.SimpleNamespace(
        poses=torch.from_numpy(pose), betas=torch.from_numpy(betas), body_model=body, da_smpl=torch.from_numpy(da.reshape(1, 72)),
        alignments=torch.from_numpy(np.stack([np.concatenate([align[f'{i:05d}.png'], np.array([[0.], [0.], [0.], [1.]])], 1)
                                              for i in range(N_FRAMES)]).astype(np.float32)), scale=SCALE)
    with torch.no_grad():
        wv, T = R_hn.HumanNeRF.vertex_forward(fake, 1)
    out['vf_world_verts'] = wv[0].numpy()[rows[rows < 6890]]
    out['vf_T'] = T[0].numpy()[rows[rows < 6890]]
    np.savez_compressed(os.path.join(HERE, 'smpl.npz'), **out)
    for k, v in out.items():
        print(k, getattr(v, 'shape', v), getattr(v, 'dtype', ''))


if __name__ == '__main__':
    main()
