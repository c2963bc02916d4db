"""A small synthetic COLMAP ASCII model (tests/golden/colmap/sparse/*.txt) and what the REFERENCE's own readers make of it
(tests/golden/scene.npz).  Build container only:

    python tests/golden/make_golden_scene.py

data_io/colmap_helper.py ColmapAsciiReader (imported unmodified; imageio is stubbed -- no pixel is read), cameras/camera_pose.py,
geometry/pcd_projector.py project_point_cloud_at_capture and the near / far + normalisation lines of
data_io/neuman_helper.py:199-242 applied to the reference's capture objects.
"""
import os
import sys
import tempfile
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX",
          "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2"]:
    sys.modules[m] = mock.MagicMock(name=m)
sys.path.insert(0, REF)

from data_io import colmap_helper  # noqa: E402  (reference)
from geometry import pcd_projector  # noqa: E402


def write_model(sparse, n_img=12, n_pts=400, seed=3):
    rng = np.random.default_rng(seed)
    os.makedirs(sparse, exist_ok=True)
    with open(os.path.join(sparse, 'cameras.txt'), 'w') as f:
        f.write('# Camera list with one line of data per camera:\n#   CAMERA_ID, MODEL, WIDTH, HEIGHT, PARAMS[]\n# Number of cameras: 3\n')
        f.write('1 SIMPLE_RADIAL 640 360 512.25 321.5 182.25 0.013\n')
        f.write('2 PINHOLE 640 360 500.5 498.75 320 180\n')
        f.write('3 OPENCV 320 240 250.5 251.5 160.25 119.75 0.01 -0.02 0.001 0.0005\n')
    pts = rng.normal(size=(n_pts, 3)) * np.array([2.0, 1.0, 2.0]) + np.array([0, 0, 6.0])
    with open(os.path.join(sparse, 'points3D.txt'), 'w') as f:
        f.write('# 3D point list with one line of data per point:\n#   POINT3D_ID, X, Y, Z, R, G, B, ERROR, TRACK[] as (IMAGE_ID, POINT2D_IDX)\n')
        f.write(f'# Number of points: {n_pts}, mean track length: 3.25\n')
        for i, p in enumerate(pts):
            c = rng.integers(0, 256, size=3)
            f.write(f'{i + 1} {p[0]:.6f} {p[1]:.6f} {p[2]:.6f} {c[0]} {c[1]} {c[2]} {rng.uniform(0.1, 1.0):.4f} 1 2 3 4\n')
    with open(os.path.join(sparse, 'images.txt'), 'w') as f:
        f.write('# Image list with two lines of data per image:\n#   IMAGE_ID, QW, QX, QY, QZ, TX, TY, TZ, CAMERA_ID, NAME\n#   POINTS2D[] as (X, Y, POINT3D_ID)\n')
        f.write(f'# Number of images: {n_img}, mean observations per image: 120.5\n')
        order = rng.permutation(n_img)                      # ids out of file-name order: 'video' order must sort by name
        for k in order:
            ang = 0.08 * (k - n_img / 2)
            axis = np.array([0.1, 1.0, 0.05])
            axis /= np.linalg.norm(axis)
            q = np.concatenate([[np.cos(ang / 2)], np.sin(ang / 2) * axis])
            t = np.array([0.3 * np.sin(ang * 3), 0.05 * k, 0.2 * np.cos(ang)])
            f.write(f'{int(k) + 1} {q[0]:.9f} {q[1]:.9f} {q[2]:.9f} {q[3]:.9f} {t[0]:.6f} {t[1]:.6f} {t[2]:.6f} {1 + int(k) % 3} {int(k):05d}.png\n')
            f.write('10.5 20.5 1 30.25 40.75 -1\n')
    return n_img


def main():
    sparse = os.path.join(HERE, 'colmap', 'sparse')
    n_img = write_model(sparse)
    with tempfile.TemporaryDirectory() as images:
        for k in range(n_img):
            open(os.path.join(images, f'{k:05d}.png'), 'w').close()        # the reader only checks that the files exist
        caps = colmap_helper.ColmapAsciiReader.read_captures(os.path.join(sparse, 'images.txt'), os.path.join(sparse, 'cameras.txt'), images, None,
                                                             'video')
        small = colmap_helper.ColmapAsciiReader.read_captures(os.path.join(sparse, 'images.txt'), os.path.join(sparse, 'cameras.txt'), images,
                                                              (90, 160), 'video')
    pcd = colmap_helper.ColmapAsciiReader.read_point_cloud(os.path.join(sparse, 'points3D.txt'))
    out = {'names': np.array([os.path.basename(c.image_path) for c in caps]), 'pcd': pcd,
           'K': np.stack([c.intrinsic_matrix for c in caps]), 'shape': np.array([c.shape for c in caps]),
           'c2w': np.stack([c.cam_pose.camera_to_world for c in caps]), 'w2c': np.stack([c.cam_pose.world_to_camera for c in caps]),
           'frame_id': np.array([c.frame_id['frame_id'] for c in caps]), 'K_small': np.stack([c.intrinsic_matrix for c in small]),
           'shape_small': np.array([c.shape for c in small])}
    # neuman_helper.py:199-226 on the reference's captures (bkg), then :229-242
    near, far, depths95 = [], [], []
    for c in caps:
        z = pcd_projector.project_point_cloud_at_capture(pcd, c, render_type='pcd')[:, 2]
        n_, f_ = 0, np.percentile(z, 95)
        depths95.append(f_)
        center, length = (n_ + f_) / 2, (f_ - n_) * 1.1
        near.append(max(0.0, float(center - length / 2)))
        far.append(float(center + length / 2))
    out['near_bkg'], out['far_bkg'], out['p95'] = np.array(near), np.array(far), np.array(depths95)
    scale = 3.14 / np.percentile(np.array(far), 95)
    for c in caps:
        c.cam_pose.camera_center_in_world *= scale
    out['scale'] = np.array(scale)
    out['c2w_normalized'] = np.stack([c.cam_pose.camera_to_world for c in caps])
    np.savez_compressed(os.path.join(HERE, 'scene.npz'), **out)
    print({k: v.shape for k, v in out.items()}, float(scale))


if __name__ == "__main__":
    main()
