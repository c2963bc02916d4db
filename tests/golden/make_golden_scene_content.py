"""A tiny on-disk NeuMan scene (tests/golden/scene_files/: images, segmentations, depth_maps, mono_depth) and what the REFERENCE's
capture classes read from it (tests/golden/scene_content.npz).  Build container only:

    python tests/golden/make_golden_scene_content.py

cameras/contents.py and data_io/neuman_helper.py (NeuManCapture, ResizedNeuManCapture) are imported unmodified.  imageio is absent
offline; its `imread` is stood in for by Pillow's decoder (imageio's own PNG plugin is Pillow), everything downstream -- COLMAP
array parsing, depth clipping, mask inversion, resizing, the scene scale, the linregress fusion -- is the reference's code.
"""
import os
import sys
import types
from unittest import mock

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "lpips", "tensorboardX", "skimage", "skimage.metrics",
          "torchvision", "torchvision.utils", "cv2", "matplotlib", "matplotlib.pyplot"]:
    sys.modules.setdefault(m, mock.MagicMock(name=m))
sys.modules['imageio'] = types.SimpleNamespace(imread=lambda path: np.array(Image.open(path)))
sys.path.insert(0, REF)

from cameras import pinhole_camera, camera_pose  # noqa: E402  (reference)
from data_io import neuman_helper  # noqa: E402
from geometry.basics import Rotation, Translation  # noqa: E402

H, W = 24, 32
OUT = os.path.join(HERE, 'scene_files')


def write_colmap_array(path, arr):
    """COLMAP's dense map format: `w&h&c&` then float32, column-major"""
    h, w = arr.shape
    with open(path, 'wb') as f:
        f.write(f'{w}&{h}&1&'.encode())
        f.write(np.asfortranarray(arr.T.astype(np.float32)).tobytes(order='F'))


def write_files():
    rng = np.random.default_rng(21)
    for d in ('images', 'segmentations', 'depth_maps', 'mono_depth'):
        os.makedirs(os.path.join(OUT, d), exist_ok=True)
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(2):
        name = f'{i:05d}.png'
        Image.fromarray(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8)).save(os.path.join(OUT, 'images', name))
        seg = np.where(((yy - 12 - i) / 7.0) ** 2 + ((xx - 15 + i) / 5.0) ** 2 < 1, 255, 0).astype(np.uint8)     # Detectron2: 255 = person
        Image.fromarray(seg).save(os.path.join(OUT, 'segmentations', name))
        mono = rng.uniform(0.3, 3.0, size=(H, W))
        depth = 1.6 * mono + 0.2 + rng.normal(size=(H, W)) * 0.03
        depth[rng.uniform(size=(H, W)) < 0.25] = -1.0                                  # COLMAP marks missing depth negative
        depth[3, 4] = 400.0                                                            # an outlier beyond the 95th percentile
        write_colmap_array(os.path.join(OUT, 'depth_maps', name + '.geometric.bin'), depth)
        Image.fromarray(np.round(mono * 10000).astype(np.uint16)).save(os.path.join(OUT, 'mono_depth', name))


def main():
    write_files()
    out = {}
    cam = pinhole_camera.PinholeCamera(W, H, 40.0, 41.0, 16.0, 12.0)
    pose = camera_pose.CameraPose(Translation(np.zeros(3, np.float32)), Rotation(np.array([1, 0, 0, 0], np.float32)))
    for i in range(2):
        name = f'{i:05d}.png'
        paths = dict(image=os.path.join(OUT, 'images', name), depth=os.path.join(OUT, 'depth_maps', name + '.geometric.bin'),
                     mask=os.path.join(OUT, 'segmentations', name), mono=os.path.join(OUT, 'mono_depth', name))
        for scale in (1, np.float64(0.37)):
            cap = neuman_helper.NeuManCapture(paths['image'], paths['depth'], paths['mask'], cam, pose, i, 0, mono_depth_path=paths['mono'])
            cap.captured_depth.scale = scale                                           # neuman_helper.py:236-238
            cap.captured_mono_depth.scale = scale
            p = f'full/{i}/{float(scale):g}'
            out[f'{p}/image'], out[f'{p}/mask'], out[f'{p}/binary_mask'] = cap.image, cap.mask, cap.binary_mask
            out[f'{p}/depth_map'], out[f'{p}/mono_depth_map'], out[f'{p}/fused_depth_map'] = cap.depth_map, cap.mono_depth_map, cap.fused_depth_map
        small = neuman_helper.ResizedNeuManCapture(paths['image'], paths['depth'], paths['mask'], cam, pose, (12, 16), i, 0, mono_depth_path=paths['mono'])
        p = f'resized/{i}'
        out[f'{p}/image'], out[f'{p}/mask'], out[f'{p}/depth_map'], out[f'{p}/mono_depth_map'] = small.image, small.mask, small.depth_map, small.mono_depth_map
        out[f'{p}/shape'] = np.array(small.shape)
    # a frame with no MVS / mono depth on disk: the '...dummy' convention of read_captures (:350-355)
    cap = neuman_helper.NeuManCapture(paths['image'], paths['image'] + 'dummy', paths['mask'], cam, pose, 1, 0, mono_depth_path=paths['image'] + 'dummy')
    out['dummy/depth_map'], out['dummy/mono_depth_map'] = cap.depth_map, cap.mono_depth_map
    np.savez_compressed(os.path.join(HERE, 'scene_content.npz'), **out)
    print('wrote', len(out), 'arrays;', {k: (v.dtype, v.shape) for k, v in list(out.items())[:8]})


if __name__ == '__main__':
    main()
