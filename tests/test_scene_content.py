"""neuman_hip/scene_content.py against what the REFERENCE's NeuManCapture / ResizedNeuManCapture read from the committed scene files
(tests/golden/scene_files, tests/golden/scene_content.npz made by make_golden_scene_content.py).  No GPU."""
import os

import numpy as np
import pytest

from neuman_hip import data_io, scene_content as sc

HERE = os.path.dirname(os.path.abspath(__file__))
FILES = os.path.join(HERE, "golden", "scene_files")


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(HERE, "golden", "scene_content.npz")))


def base_capture(i, h=24, w=32):
    cam = data_io.PinholeCamera(w, h, 40.0 * w / 32, 41.0 * h / 24, 16.0 * w / 32, 12.0 * h / 24)
    pose = data_io.CameraPose(np.zeros(3, np.float32), np.array([1, 0, 0, 0], np.float32))
    return data_io.Capture(os.path.join(FILES, 'images', f'{i:05d}.png'), cam, pose, frame_id={'frame_id': i, 'total_frames': 2})


def same(a, b, exact=True):
    assert a.shape == b.shape and a.dtype == b.dtype, (a.shape, a.dtype, b.shape, b.dtype)
    if exact:
        assert np.array_equal(a, b)
    else:
        assert np.allclose(a, b, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("scale", [1, np.float64(0.37)])
def test_full_size_contents(g, scale):
    for i in range(2):
        cap = sc.attach_contents(FILES, [base_capture(i)], scale=scale)[0]
        p = f'full/{i}/{float(scale):g}'
        same(cap.image, g[f'{p}/image'])
        same(cap.mask, g[f'{p}/mask'])
        same(cap.binary_mask, g[f'{p}/binary_mask'])
        same(cap.depth_map, g[f'{p}/depth_map'])
        same(cap.mono_depth_map, g[f'{p}/mono_depth_map'])
        same(cap.fused_depth_map, g[f'{p}/fused_depth_map'], exact=False)          # scipy's linregress vs explicit sums
        assert cap.mask.min() == 0 and cap.mask.max() == 1 and 0 < cap.mask.sum() < cap.mask.size
        assert (cap.depth_map == 0).mean() > 0.2 and cap.depth_map.max() < 400 * float(scale)     # holes and the outlier are gone
        assert cap.image is cap.image                                                  # read once, kept


def test_resized_contents(g):
    for i in range(2):
        cap = sc.attach_contents(FILES, [base_capture(i, 12, 16)], tgt_size=(12, 16))[0]
        p = f'resized/{i}'
        assert tuple(cap.shape) == tuple(g[f'{p}/shape'])
        same(cap.image, g[f'{p}/image'])
        same(cap.mask, g[f'{p}/mask'])
        same(cap.depth_map, g[f'{p}/depth_map'])
        same(cap.mono_depth_map, g[f'{p}/mono_depth_map'])


def test_missing_depth_files_give_zeros(g, tmp_path):
    cap = sc.ContentCapture(base_capture(1), os.path.join(FILES, 'segmentations', '00001.png'), None, str(tmp_path / 'absent.png'))
    same(cap.depth_map, g['dummy/depth_map'])
    same(cap.mono_depth_map, g['dummy/mono_depth_map'])


def test_colmap_array_reader_round_trip(tmp_path):
    rng = np.random.default_rng(0)
    for shape in [(5, 7), (1, 9), (6, 1)]:
        arr = rng.normal(size=shape).astype(np.float32)
        path = tmp_path / 'a.bin'
        with open(path, 'wb') as f:
            f.write(f'{shape[1]}&{shape[0]}&1&'.encode())
            f.write(np.asfortranarray(arr.T).tobytes(order='F'))
        got = sc.read_colmap_array(str(path))
        assert np.array_equal(got.reshape(shape), arr)


def test_not_a_detectron_mask_is_rejected():
    with pytest.raises(ValueError):
        sc.human_mask(np.zeros((4, 4), np.uint8))
