"""Training ray batches as the REFERENCE's datasets produce them (tests/golden/ray_batches.npz).  Build container only:

    python tests/golden/make_golden_ray_batches.py

A synthetic 3-capture scene (40 x 56 pixels; images, masks, depth maps, posed vertex clouds from tests/helpers/batch_scene.py)
wrapped in the reference's own camera, capture and scene classes, then -- all imported unmodified, absent wheels stubbed --
    utils/utils.py add_border_mask                              -> border masks
    data_io/cache_helper.py export_ / load_near_far_cache       -> the SMPL-guided near/far of every pixel
    datasets/background_rays.py BackgroundRayDataset[0]         -> batches under np.random.seed(s): border-aware, plain mask, NeRF-T
    datasets/human_rays.py HumanRayDataset[0]                   -> batches under random.seed(s) / np.random.seed(s): with and without
                                                                   the LPIPS patch
The device batchers of neuman_hip/ray_batches.py replay the same seeds (draws='numpy') in tests/test_hip_ray_batches.py.
"""
import os
import random
import sys
import tempfile
import types
from unittest import mock

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
for m in ["igl", "open3d", "pytorch3d", "pytorch3d.structures", "pytorch3d.renderer", "pytorch3d.ops", "pytorch3d.ops.knn", "imageio", "lpips",
          "tensorboardX", "skimage", "skimage.metrics", "torchvision", "torchvision.utils", "cv2", "matplotlib", "matplotlib.pyplot"]:
    sys.modules.setdefault(m, mock.MagicMock(name=m))
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "tests", "helpers"))

import torch  # noqa: E402
from cameras import camera_pose, captures as captures_module, pinhole_camera  # noqa: E402  (reference)
from data_io import cache_helper  # noqa: E402
from geometry.basics import Rotation, Translation  # noqa: E402
import importlib.util  # noqa: E402


def _ref_module(name, path):
    """(the reference's datasets/ has no __init__.py and loses to the installed `datasets` wheel: load its files by path)"""
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


background_rays = _ref_module('ref_background_rays', os.path.join(REF, 'datasets', 'background_rays.py'))
human_rays = _ref_module('ref_human_rays', os.path.join(REF, 'datasets', 'human_rays.py'))
from scenes import scene as scene_module  # noqa: E402
from utils import utils as R_utils  # noqa: E402

import batch_scene  # noqa: E402  (tests/helpers)


class Cap(captures_module.RigPinholeCapture):
    """the attributes the datasets read from a NeuManCapture, filled from arrays instead of files"""

    def __init__(self, content, cam, pose, view_id, image_path):
        super().__init__(cam, pose, view_id, 0)
        self.image, self.mask, self.depth_map = content['image'], content['mask'], content['depth']
        self.fused_depth_map = content['depth']
        self.image_path = image_path
        self.near, self.far = dict(content['near']), dict(content['far'])
        self.frame_id = {'frame_id': view_id, 'total_frames': content['total_frames']}
        self.posed_mesh = types.SimpleNamespace(device='cpu')

    @property
    def binary_mask(self):
        m = self.mask.copy()
        m[m > 0] = 1
        return m


def pack(out, prefix, batch):
    for k, v in batch.items():
        if isinstance(v, torch.Tensor):
            out[f'{prefix}/{k}'] = v.numpy()
        else:
            out[f'{prefix}/{k}'] = np.asarray(v)


def main():
    spec = batch_scene.make(seed=11)
    out = {}
    tmp = tempfile.mkdtemp()
    os.makedirs(os.path.join(tmp, 'images'))
    caps = []
    for i, c in enumerate(spec['captures']):
        cam = pinhole_camera.PinholeCamera(spec['w'], spec['h'], *c['intrinsics'])
        pose = camera_pose.CameraPose(Translation(c['t'].astype(np.float32)), Rotation(c['q'].astype(np.float32)))   # as colmap_helper.py:143-145
        cap = Cap(c, cam, pose, i, os.path.join(tmp, 'images', c['name']))
        caps.append(cap)
        out[f'cam/{i}/intrinsic'] = np.asarray(cap.intrinsic_matrix)
        out[f'cam/{i}/c2w'] = np.asarray(cap.cam_pose.camera_to_world)
    scene = scene_module.RigCameraScene(caps, len(caps), 1)
    scene.verts = [torch.from_numpy(c['verts']) for c in spec['captures']]

    opt = types.SimpleNamespace(rays_per_batch=512, ablate_nerft=False, use_fused_depth=False, white_bkg=True, geo_threshold=0.2, chunk=700,
                                normalize=True, penalize_lpips=0.0, body_rays_ratio=0.6, border_rays_ratio=0.15, bkg_rays_ratio=0.25,
                                dilation=spec['dilation'])
    split = os.path.join(tmp, 'train_split.txt')
    with open(split, 'w') as f:
        f.write('\n'.join(c['name'] for c in spec['captures']))

    # ---- near/far cache of every pixel
    cache_helper.export_near_far_cache(opt, scene, opt.geo_threshold, opt.chunk, 'cpu')
    cache = cache_helper.load_near_far_cache(opt, scene, opt.geo_threshold)
    out['cache'] = np.stack([cache[c['name']] for c in spec['captures']])
    print('near/far cache: hit pixels per capture', [(int((v[..., 0] < v[..., 1]).sum())) for v in out['cache']])

    # ---- background batches: plain mask, then border-aware, then NeRF-T (whole image)
    ds = background_rays.BackgroundRayDataset(opt, scene, 'train', split)
    np.random.seed(101)
    pack(out, 'bkg/plain', ds[0])
    R_utils.add_border_mask(scene, iterations=opt.dilation)
    out['border'] = np.stack([c.border_mask for c in caps])
    np.random.seed(102)
    pack(out, 'bkg/border', ds[0])
    opt_t = types.SimpleNamespace(**{**vars(opt), 'ablate_nerft': True})
    np.random.seed(103)
    pack(out, 'bkg/nerft', background_rays.BackgroundRayDataset(opt_t, scene, 'train', split)[0])
    for it in (0, 3):                                               # and border masks for other dilation depths (0 = none)
        R_utils.add_border_mask(scene, iterations=it)
        out[f'border_it{it}'] = np.stack([c.border_mask for c in caps])
    R_utils.add_border_mask(scene, iterations=opt.dilation)

    # ---- human batches
    hs = human_rays.HumanRayDataset(opt, scene, 'train', split, near_far_cache=cache)
    random.seed(201)
    np.random.seed(201)
    pack(out, 'human/plain', hs[0])
    opt_p = types.SimpleNamespace(**{**vars(opt), 'penalize_lpips': 0.01, 'rays_per_batch': 1024 + 300})
    hp = human_rays.HumanRayDataset(opt_p, scene, 'train', split, near_far_cache=cache)
    got = set()
    for seed in range(300, 340):                                    # one batch led by a patch, one where the draw said no patch
        random.seed(seed)
        np.random.seed(seed)
        b = hp[0]
        kind = int(b['patch_counter'])
        if kind not in got:
            got.add(kind)
            pack(out, f'human/patch{kind}', b)
            out[f'human/patch{kind}/seed'] = np.array(seed)
        if len(got) == 2:
            break
    assert got == {0, 1}
    hp.cap_id = 1                                                   # the trainers' fixed-capture mode
    random.seed(401)
    np.random.seed(401)
    pack(out, 'human/fixed', hp[0])
    np.savez_compressed(os.path.join(HERE, 'ray_batches.npz'), **out)
    print('wrote ray_batches.npz:', len(out), 'arrays,', os.path.getsize(os.path.join(HERE, 'ray_batches.npz')), 'bytes')


if __name__ == '__main__':
    main()
