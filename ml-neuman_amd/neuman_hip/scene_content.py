"""The pixel side of scene ingestion: what cameras/contents.py and the NeuManCapture classes of
data_io/neuman_helper.py:21-146 read for every frame -- the image, the human segmentation, the MVS depth map (COLMAP's dense
.geometric.bin), the monocular depth (16-bit PNG) and the two fused into one depth map -- without imageio (PNG decoding is Pillow's,
which is what imageio's PNG plugin calls).  The arrays are what ray_batches.FrameStore uploads once.

    read_colmap_array        contents.py:18-38      COLMAP's `w&h&c&` + float32 column-major blob
    read_mvs_depth           contents.py:101-111    negatives -> 0, beyond the 95th percentile of the positive depths -> 0
    read_mono_depth          contents.py:94-99      uint16 PNG / 10000
    human_mask               neuman_helper.py:54-66 the Detectron2 segmentation (255 / 0) -> the 0 / 1 mask the datasets sample by
    ContentCapture           NeuManCapture / ResizedNeuManCapture: lazy, optionally resized (bilinear image, nearest mask / depths),
                             depth maps scaled by the scene normalisation (neuman_helper.py:236-238)
    attach_contents          read_captures' path conventions (:334-366) on the captures data_io.read_scene returns
"""
import os

import numpy as np
from PIL import Image

from . import ray_batches
from .data_io import Capture


def read_colmap_array(path):
    """COLMAP dense output (depth / normal maps): an ASCII header `width&height&channels&` followed by float32 in column-major
    (Fortran) order -> [height, width] (or [height, width, channels])."""
    with open(path, 'rb') as f:
        blob = f.read()
    cut, seen = 0, 0
    while seen < 3:
        cut = blob.index(b'&', cut) + 1
        seen += 1
    width, height, channels = (int(v) for v in blob[:cut].split(b'&')[:3])
    data = np.frombuffer(blob, dtype=np.float32, offset=cut, count=width * height * channels)
    return np.transpose(data.reshape((width, height, channels), order='F'), (1, 0, 2)).squeeze().copy()


def read_mvs_depth(path):
    """the .geometric.bin depth of one frame as the reference uses it: invalid (negative) depths zeroed, and everything beyond the
    95th percentile of the valid ones too (sky / outliers)"""
    depth = read_colmap_array(path)
    depth[depth < 0] = 0
    valid = depth[depth > 0]
    limit = np.percentile(valid, [0, 95])[1] if valid.size else 0
    depth[depth > limit] = 0
    return depth


def read_mono_depth(path):
    """monocular depth prediction stored as a 16-bit PNG in units of 1e-4"""
    depth = np.array(Image.open(path)) / 10000.0
    assert (depth >= 0).all()
    return depth


def read_image(path):
    return np.array(Image.open(path))


def human_mask(segmentation):
    """NeuManCapture.mask: the stored segmentation (Detectron2 convention, its maximum must be 255) with 255 mapped to 1 and then
    inverted, `1 - m`, in the image's own integer type -- the 0 = 'sample the background here' convention every dataset relies on"""
    m = np.array(segmentation).copy()
    if m.max() != 255:
        raise ValueError("segmentation is not a Detectron2 mask (max != 255)")
    m[m == 255] = 1
    m = 1 - m
    assert m.sum() > 0
    return m


def _resize(arr, tgt_hw, sampling):
    return np.array(Image.fromarray(arr).resize(tgt_hw[::-1], sampling))


class ContentCapture(Capture):
    """A data_io.Capture with its files: .image, .mask, .binary_mask, .depth_map, .mono_depth_map, .fused_depth_map, read on first use
    and kept.  `tgt_size` (h, w): the capture's camera is already the resized one (data_io.read_scene(tgt_size=...)); contents are
    resized to it as ResizedNeuManCapture does.  `depth_scale`: the scene normalisation factor."""

    def __init__(self, base, mask_path=None, depth_path=None, mono_depth_path=None, tgt_size=None, depth_scale=1.0):
        super().__init__(base.image_path, base.pinhole_cam, base.cam_pose, getattr(base, 'frame_id', None))
        self.near, self.far = base.near, base.far
        self.mask_path, self.depth_path, self.mono_depth_path = mask_path, depth_path, mono_depth_path
        self.tgt_size, self.depth_scale = tgt_size, depth_scale
        self._cache = {}

    def _once(self, key, make):
        if key not in self._cache:
            self._cache[key] = make()
        return self._cache[key]

    @property
    def image(self):
        def make():
            img = read_image(self.image_path)
            if self.tgt_size is not None:
                img = _resize(img, self.tgt_size, Image.BILINEAR)
            assert img.shape[:2] == tuple(self.shape), f'image {img.shape} does not match the camera {self.shape}'
            return img
        return self._once('image', make)

    @property
    def mask(self):
        def make():
            seg = np.load(self.mask_path) if self.mask_path.endswith('.npy') else read_image(self.mask_path)
            if self.tgt_size is not None:
                seg = _resize(seg, self.tgt_size, Image.NEAREST)
            m = human_mask(seg)
            assert m.shape[:2] == tuple(self.shape)
            return m
        return self._once('mask', make)

    @property
    def binary_mask(self):
        m = self.mask.copy()
        m[m > 0] = 1
        return m

    def _depth(self, key, path, reader):
        def make():
            if path is None or not os.path.isfile(path):             # the reference's '...dummy' path: all zeros (contents.py:114-117)
                w, h = Image.open(self.image_path).size
                d = np.zeros((h, w), dtype=np.float32)
            else:
                d = reader(path)
            d = d * self.depth_scale
            if self.tgt_size is not None:
                d = _resize(d, self.tgt_size, Image.NEAREST)
            return d
        return self._once(key, make)

    @property
    def depth_map(self):
        d = self._depth('depth', self.depth_path, read_mvs_depth)
        assert (d >= 0).all()
        return d

    @property
    def mono_depth_map(self):
        return self._depth('mono', self.mono_depth_path, read_mono_depth)

    @property
    def fused_depth_map(self):
        return self._once('fused', lambda: ray_batches.fused_depth(self.depth_map, self.mono_depth_map, self.mask))


def attach_contents(scene_dir, captures, tgt_size=None, scale=1.0, mask_dir='segmentations'):
    """read_captures' file layout (neuman_helper.py:346-358) for the captures of data_io.read_scene: images/<name>,
    depth_maps/<name>.geometric.bin, mono_depth/<name>, <mask_dir>/<name>[.npy] -> ContentCapture list (same order)."""
    out = []
    for cap in captures:
        name = os.path.basename(cap.image_path)
        depth = cap.image_path.replace('/images/', '/depth_maps/') + '.geometric.bin'
        mono = cap.image_path.replace('/images/', '/mono_depth/')
        mask = os.path.join(scene_dir, mask_dir, name + '.npy')
        if not os.path.isfile(mask):
            mask = os.path.join(scene_dir, mask_dir, name)
        out.append(ContentCapture(cap, mask, depth if os.path.isfile(depth) else None, mono if os.path.isfile(mono) else None, tgt_size, scale))
    return out


def read_neuman_scene(scene_dir, tgt_size=None, normalize=False, bkg_range_scale=1.1, mask_dir='segmentations'):
    """NeuManReader.read_scene (neuman_helper.py:198-247) up to the SMPL part: cameras and poses from `sparse/`, background near / far,
    optional normalisation (which also scales the depth maps), and every frame's files attached -> (captures, point_cloud, scale)."""
    from .data_io import read_scene
    caps, pcd, scale = read_scene(scene_dir, tgt_size, normalize, bkg_range_scale)
    return attach_contents(scene_dir, caps, tgt_size, scale, mask_dir), pcd, scale
