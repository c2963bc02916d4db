"""neuman_hip.data_io (COLMAP ASCII, poses, near / far, normalisation, splits, checkpoints, params.json) against what the
REFERENCE's own readers make of the same files (tests/golden/scene.npz, make_golden_scene.py)."""
import os
import types

import numpy as np
import pytest
import torch

from neuman_hip import data_io

HERE = os.path.dirname(os.path.abspath(__file__))
SCENE = os.path.join(HERE, "golden", "colmap")


@pytest.fixture(scope="module")
def G():
    return dict(np.load(os.path.join(HERE, "golden", "scene.npz")))


def test_colmap_model_is_read_like_the_reference(G):
    caps, pcd = data_io.ColmapAsciiReader.read_scene(os.path.join(SCENE, "sparse"), os.path.join(SCENE, "images"), None, 'video',
                                                     check_files=False)
    assert [os.path.basename(c.image_path) for c in caps] == list(G['names'])          # video order: sorted by file name
    np.testing.assert_array_equal(pcd, G['pcd'])
    np.testing.assert_array_equal(np.stack([c.intrinsic_matrix for c in caps]), G['K'])
    np.testing.assert_array_equal(np.array([c.shape for c in caps]), G['shape'])
    np.testing.assert_array_equal(np.array([c.frame_id['frame_id'] for c in caps]), G['frame_id'])
    assert all(c.frame_id['total_frames'] == len(caps) for c in caps)
    # poses: the same float32 quaternion / translation, the same float32 rotation matrix, the same f64 inverse -> bit equal
    np.testing.assert_array_equal(np.stack([c.cam_pose.world_to_camera for c in caps]), G['w2c'])
    np.testing.assert_array_equal(np.stack([c.cam_pose.camera_to_world for c in caps]), G['c2w'])
    small = data_io.ColmapAsciiReader.read_captures(os.path.join(SCENE, "sparse", "images.txt"), os.path.join(SCENE, "sparse", "cameras.txt"),
                                                    "x", (90, 160), 'video', check_files=False)
    np.testing.assert_allclose(np.stack([c.intrinsic_matrix for c in small]), G['K_small'], rtol=1e-15)
    np.testing.assert_array_equal(np.array([c.shape for c in small]), G['shape_small'])
    with pytest.raises(AssertionError):                                                  # the reference insists on the image files
        data_io.ColmapAsciiReader.read_images_meta(os.path.join(SCENE, "sparse", "images.txt"), "/nonexistent")


def test_near_far_and_normalisation(G):
    caps, pcd, scale = data_io.read_scene(SCENE, normalize=False, check_files=False)
    assert scale == 1
    np.testing.assert_allclose([c.near['bkg'] for c in caps], G['near_bkg'], rtol=1e-12)
    np.testing.assert_allclose([c.far['bkg'] for c in caps], G['far_bkg'], rtol=1e-12)
    caps, pcd, scale = data_io.read_scene(SCENE, normalize=True, check_files=False)
    np.testing.assert_allclose(scale, G['scale'], rtol=1e-12)
    np.testing.assert_allclose(np.percentile([c.far['bkg'] for c in caps], 95), 3.14, rtol=1e-12)
    np.testing.assert_allclose(pcd[:, :3], G['pcd'][:, :3] * np.float32(G['scale']), rtol=1e-6)
    # the reference moves the camera centres through a float32 translation: same matrices to the last bit it keeps
    np.testing.assert_allclose(np.stack([c.cam_pose.camera_to_world for c in caps]), G['c2w_normalized'], rtol=0, atol=1e-7)
    # and the rays shot from these captures are the hot path's input: one end-to-end check against the oracle's ray generator
    from oracle import ray_ops
    o, d = ray_ops.shot_all_rays(caps[3].intrinsic_matrix, caps[3].cam_pose.camera_to_world, caps[3].shape)
    assert o.shape == (caps[3].shape[0] * caps[3].shape[1], 3) and np.allclose(np.linalg.norm(d, axis=1), 1.0)
    np.testing.assert_allclose(o[0], caps[3].cam_pose.camera_center_in_world, atol=1e-12)


def test_splits(tmp_path):
    for n in (10, 23, 104, 1000):
        tr, va, te = data_io.split_indices(n)
        assert sorted(tr + va + te) == list(range(n)) and not (set(va) & set(te))
        assert abs(len(va) + len(te) - n // 5) <= 1 and len(te) == (len(va) + len(te)) // 2
    caps = [types.SimpleNamespace(image_path=f"/x/images/{i:05d}.png") for i in range(23)]
    paths = data_io.create_split_files(str(tmp_path), caps)
    names = [data_io.read_text(p) for p in paths]
    assert sum(len(x) for x in names) == 23 and names[2] == ['00002.png', '00007.png']
    assert [os.path.basename(c.image_path) for c in data_io.captures_of_split(caps, paths[2])] == names[2]


def test_checkpoints_and_params(tmp_path):
    from neuman_hip import synthetic
    coarse, fine = synthetic.make_joiner(0), synthetic.make_joiner(1)
    ckpt = {'coarse_model_state_dict': {'module.' + k: v for k, v in coarse.state_dict().items()},      # saved from nn.DataParallel (train.py:26-28)
            'fine_model_state_dict': fine.state_dict()}
    p = str(tmp_path / "checkpoint.pth.tar")
    torch.save(ckpt, p)
    a, b = synthetic.make_joiner(7), synthetic.make_joiner(8)
    data_io.load_background_checkpoint(p, a, b)
    assert all(torch.equal(x, y) for x, y in zip(a.state_dict().values(), coarse.state_dict().values()))
    assert all(torch.equal(x, y) for x, y in zip(b.state_dict().values(), fine.state_dict().values()))
    # a hybrid checkpoint: only the canonical human net is pulled out (models/human_nerf.py:63-74)
    human = synthetic.make_joiner(2, 'rotate')
    hyb = {'hybrid_model_state_dict': {**{'coarse_human_net.' + k: v for k, v in human.state_dict().items()},
                                       **{'coarse_bkg_net.' + k: v for k, v in coarse.state_dict().items()}, 'poses': torch.zeros(3, 72)}}
    p2 = str(tmp_path / "hybrid.pth.tar")
    torch.save(hyb, p2)
    h2 = synthetic.make_joiner(9, 'rotate')
    data_io.load_canonical_human(p2, h2)
    assert all(torch.equal(x, y) for x, y in zip(h2.state_dict().values(), human.state_dict().values()))
    # partial load: tensors that match by name and shape, the rest reported
    plain = synthetic.make_variant_joiner(5, use_viewdirs=False)
    missing = data_io.safe_load_weights(plain, coarse.state_dict())
    assert missing == {'nerf.output_linear.weight', 'nerf.output_linear.bias'}
    assert torch.equal(plain.nerf.pts_linears[3].weight, coarse.nerf.pts_linears[3].weight)
    with pytest.raises(RuntimeError):
        data_io.safe_load_weights(plain, {'nothing': torch.zeros(1)})
    opt = types.SimpleNamespace(out=str(tmp_path / "run"), samples_per_ray=128, white_bkg=True, name="x")
    path = data_io.save_opt(opt)
    back = data_io.read_params(path)
    assert vars(back) == vars(opt)
