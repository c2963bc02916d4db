"""Pose-gradient masks the REFERENCE derives from DensePose label maps (tests/golden/densepose_mask.npz).  Build container only:
    python tests/golden/make_golden_densepose_mask.py
trainers/human_nerf_trainer.py turn_smpl_gradient_off, imported unmodified (absent wheels stubbed), on 203 random label maps."""
import os
import sys
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "imageio", "lpips", "tensorboardX", "skimage",
          "skimage.metrics", "torchvision", "torchvision.utils", "cv2", "matplotlib", "matplotlib.pyplot"]:
    sys.modules.setdefault(m, mock.MagicMock(name=m))
sys.path.insert(0, "/root/reference")
from trainers import human_nerf_trainer as R  # noqa: E402

rng = np.random.default_rng(0)
label_sets = [rng.choice(25, size=rng.integers(1, 25), replace=False) for _ in range(200)] + [np.arange(25), np.array([0]), np.array([1, 2])]
maps = np.stack([rng.choice(s, size=(9, 7)) for s in label_sets])
masks = np.stack([R.turn_smpl_gradient_off(m) for m in maps])
np.savez_compressed(os.path.join(HERE, 'densepose_mask.npz'), maps=maps, masks=masks)
print(maps.shape, masks.shape, masks.mean())
