"""Training ray batches assembled on the device (the callers in front of the path).

The reference builds every training batch on the host inside DataLoader workers (datasets/background_rays.py:42-123,
datasets/human_rays.py:109-248): per iteration and per capture an `np.argwhere` over the whole mask, fancy-indexed gathers from the
image / depth map / near-far cache, `shot_rays` in float64, then eight arrays travel to the GPU.  Here the scene's content lives in
HBM once (`FrameStore`: images uint8, masks, depth maps, the SMPL-guided near/far cache, one table of camera matrices, the candidate
pixel lists of every sampling class as flat index pools) and a batch is a handful of gathers plus one `nm_shot_rays_cams` launch,
whatever the number of captures it touches.  At 288 GB a 100-frame 1280x720 scene (0.9 GB with caches) is nothing.

Random draws.  `draws='numpy'` consumes `np.random` / `random` in exactly the order the reference's `__getitem__` does, so under the
same seeds the batch equals the reference's ray for ray (tests/test_hip_ray_batches.py, against batches the reference's datasets
produced); only the draw indices are uploaded.  `draws='device'` (default) draws on the device with a torch generator: the same
distribution (a multinomial split over captures is a uniform capture index per ray), no host traffic at all.

Batches are dicts with the reference's keys, shapes and dtypes (without the leading axis of 1 its DataLoader adds and its
trainers strip again, utils/utils.py:89-93), every tensor on the device.
"""
import os
import random

import numpy as np
import torch
import torch.nn.functional as F
import torch.utils.data

from . import _lib, ray_utils

PATCH_SIZE = 32                                   # utils/constant.py:8
PATCH_SIZE_SQUARED = PATCH_SIZE ** 2
NEAR_INDEX, FAR_INDEX = 0, 1                      # utils/constant.py:5-6
TRAIN_SET_LENGTH = 1000000                        # utils/constant.py:10
VALIDATION_SET_LENGTH = 10


def border_mask(binary_mask, iterations):
    """utils/utils.py:257-262: scipy.ndimage.binary_dilation(binary_mask, iterations=n) - binary_mask with scipy's default
    structuring element (the 4-neighbour cross; outside the image counts as 0), n <= 0 -> all zeros.  binary_mask: device
    uint8 [H,W] (or [C,H,W]) of 0/1; returns the same shape and dtype.  n passes of a 5-point maximum."""
    m = binary_mask
    if iterations <= 0:
        return torch.zeros_like(m)
    x = m.reshape(-1, 1, *m.shape[-2:]).to(torch.float32)
    cross = torch.tensor([[0., 1., 0.], [1., 1., 1.], [0., 1., 0.]], device=m.device).reshape(1, 1, 3, 3)
    for _ in range(int(iterations)):
        x = (F.conv2d(x, cross, padding=1) > 0).to(torch.float32)
    return (x.reshape(m.shape).to(m.dtype) - m)


def fused_depth(depth, mono_depth, mask):
    """data_io/neuman_helper.py:71-79: where the MVS depth is missing or on the human, the monocular depth mapped through the
    least-squares line fitted (scipy.stats.linregress) on the valid background pixels.  Host numpy, float64 sums; returns an
    array of depth's dtype."""
    depth, mono = np.asarray(depth), np.asarray(mono_depth)
    valid = (depth > 0) & (np.asarray(mask) == 0)
    x, y = mono[valid].astype(np.float64), depth[valid].astype(np.float64)
    xm, ym = x.mean(), y.mean()
    slope = ((x - xm) * (y - ym)).sum() / ((x - xm) ** 2).sum()
    out = depth.copy()
    out[~valid] = mono[~valid] * slope + (ym - slope * xm)
    return out


def num_rays_per_class(opt, num):
    """datasets/human_rays.py:81-95: body / border / background ray counts of `num` rays (Python's round, the leftover to the
    largest class)."""
    counts = [int(round(num * opt.body_rays_ratio)),
              int(round(num * opt.border_rays_ratio)) if opt.dilation > 0 else 0,
              int(round(num * opt.bkg_rays_ratio))]
    counts[int(np.argmax(counts))] += num - sum(counts)
    assert min(counts) >= 0 and sum(counts) == num, counts
    return {'num_body_rays': counts[0], 'num_border_rays': counts[1], 'num_bkg_rays': counts[2]}


def patch_corner(shape_hw, pos_xy, size=PATCH_SIZE):
    """datasets/human_rays.py:17-33: upper-left (x, y) of a size x size patch centred on pos as far as the image allows."""
    h, w = shape_hw
    lu_x, lu_y = int(pos_xy[0] - size // 2), int(pos_xy[1] - size // 2)
    lu_x, lu_y = max(lu_x, 0), max(lu_y, 0)
    if lu_y + size > h:
        lu_y = h - size
    if lu_x + size > w:
        lu_x = w - size
    return lu_x, lu_y


class FrameStore:
    """A scene's training content, resident in HBM.

    captures: objects with .image uint8 [H,W,3], .intrinsic_matrix, .cam_pose (camera_to_world), .near / .far dicts, and
    optionally .mask (0 = background, as NeuManCapture.mask, data_io/neuman_helper.py:54-66), .depth_map, .fused_depth_map,
    .frame_id {'frame_id', 'total_frames'}, .image_path.  All captures share one image size (the reference's cache export
    assumes it too, data_io/cache_helper.py:17).
    dilation: iterations of the border mask (train.py:38,106: opt.dilation); None = no border masks (plain background sets).
    verts: per-capture posed SMPL vertices [C][V,3] -> the near/far cache of data_io/cache_helper.py:16-36 for every pixel, computed
    on the device (near_far [C,H,W,2] f32; near > far where the ray misses the body).
    """

    def __init__(self, captures, device, dilation=None, verts=None, geo_threshold=ray_utils.DEFAULT_GEO_THRESH, use_fused_depth=False,
                 near_far_cache=None):
        _lib.require_gpu()
        self.device = torch.device(device)
        self.captures = list(captures)
        assert len(self.captures) > 0, "no captures"
        self.h, self.w = (int(v) for v in self.captures[0].shape)
        dev = self.device

        def stack(get, dtype):
            arrs = []
            for c in self.captures:
                a = np.asarray(get(c))
                assert a.shape[:2] == (self.h, self.w), f"capture content {a.shape} does not match the camera {self.h}x{self.w}"
                arrs.append(a)
            return torch.from_numpy(np.stack(arrs).astype(dtype, copy=False)).to(dev)

        self.images = stack(lambda c: c.image, np.uint8)                                      # [C,H,W,3]
        # (img / 255).astype(float32) of the reference, as a table: one rounding, the same one (f64 quotient, then f32)
        self.color_lut = torch.from_numpy((np.arange(256) / 255).astype(np.float32)).to(dev)
        has = lambda name: all(hasattr(c, name) for c in self.captures)                       # noqa: E731
        self.masks = stack(lambda c: c.mask, np.uint8) if has('mask') else None              # 0 = background
        self.binary = (self.masks > 0).to(torch.uint8) if self.masks is not None else None
        self.border = border_mask(self.binary, dilation) if (dilation is not None and self.binary is not None) else None
        depth_attr = 'fused_depth_map' if use_fused_depth else 'depth_map'
        self.depth = stack(lambda c: getattr(c, depth_attr), np.float32) if has(depth_attr) else None

        cams = np.zeros((len(self.captures), 25), dtype=np.float64)
        f32_pose = []
        for i, c in enumerate(self.captures):
            cams[i, :9] = np.linalg.inv(np.asarray(c.intrinsic_matrix, dtype=np.float64)).reshape(-1)
            c2w = np.asarray(c.cam_pose.camera_to_world)
            f32_pose.append(c2w.dtype == np.float32)
            cams[i, 9:] = c2w.astype(np.float64).reshape(-1)
        assert all(f32_pose) or not any(f32_pose), "captures mix float32 and float64 poses"
        self.ray_mode = 2 if f32_pose[0] else 1                                               # see nm_shot_rays
        self.cams = torch.from_numpy(cams).to(dev)

        def per_cap(fn):                                          # python floats -> float32 as torch's .float() of the reference
            return torch.tensor([float(fn(c)) for c in self.captures], dtype=torch.float64).to(torch.float32).to(dev)
        self.near_bkg = per_cap(lambda c: c.near['bkg'])
        self.far_bkg = per_cap(lambda c: c.far['bkg'])
        if all('human' in getattr(c, 'near', {}) for c in self.captures):
            self.near_human = per_cap(lambda c: c.near['human'])
            self.far_human = per_cap(lambda c: c.far['human'])
        else:
            self.near_human = self.far_human = None
        if has('frame_id'):
            self.view_f64 = [c.frame_id['frame_id'] / c.frame_id['total_frames'] for c in self.captures]
            self.view_f = torch.tensor(self.view_f64, dtype=torch.float64).to(torch.float32).to(dev)
        else:
            self.view_f64, self.view_f = None, None
        self.fnames = [os.path.basename(getattr(c, 'image_path', str(i))) for i, c in enumerate(self.captures)]
        self.fname_to_index = {f: i for i, f in enumerate(self.fnames)}

        self._pools = {}
        self.near_far = None
        if near_far_cache is not None:                            # the reference's dict: file name -> [H,W,3] (near, far, 1)
            nf = np.stack([np.asarray(near_far_cache[f])[..., :2] for f in self.fnames]).astype(np.float32)
            self.near_far = torch.from_numpy(nf).to(dev)
        elif verts is not None:
            self.near_far = torch.stack([self.near_far_of_capture(i, verts[i], geo_threshold) for i in range(len(self.captures))])

    # ---- the SMPL-guided near/far of every pixel of one capture: data_io/cache_helper.py:24-32 on the device
    def near_far_of_capture(self, i, verts, geo_threshold):
        cap = self.captures[i]
        ys, xs = torch.meshgrid(torch.arange(self.h, device=self.device, dtype=torch.int32),
                                torch.arange(self.w, device=self.device, dtype=torch.int32), indexing='ij')
        o, d = ray_utils.shot_rays_dev(cap, torch.stack([xs.reshape(-1), ys.reshape(-1)], dim=1))
        v = torch.as_tensor(np.asarray(verts, dtype=np.float32) if not isinstance(verts, torch.Tensor) else verts,
                            dtype=torch.float32, device=self.device).reshape(-1, 3)
        near, far = ray_utils.geometry_guided_near_far(o, d, v, geo_threshold)
        return torch.stack([near, far], dim=-1).reshape(self.h, self.w, 2)

    # ---- candidate pixels of a sampling class as one flat pool over all captures: flat [n] int32 (y * W + x, ascending per
    #      capture = np.argwhere's order) and offsets [C+1]
    def pool(self, name):
        hit = self._pools.get(name)
        if hit is not None:
            return hit
        if name == 'bkg':                                         # background_rays.py:75 / human_rays.py:161: mask == 0
            sel = self.masks == 0
        elif name == 'bkg_outside_border':                        # background_rays.py:72: (border_mask | mask) == 0
            sel = (self.border | self.masks) == 0
        elif name == 'body':                                      # human_rays.py:157: mask != 0
            sel = self.masks != 0
        elif name == 'border':                                    # human_rays.py:159: border_mask == 1
            sel = self.border == 1
        else:
            raise KeyError(name)
        sel = sel.reshape(len(self.captures), -1)
        counts = sel.sum(dim=1)
        flat = torch.nonzero(sel)[:, 1].to(torch.int32)
        offsets = torch.zeros(len(self.captures) + 1, dtype=torch.int64, device=self.device)
        offsets[1:] = torch.cumsum(counts, 0)
        out = (flat, offsets, [int(v) for v in offsets.cpu()])
        self._pools[name] = out
        return out

    # ---- what every batch needs from (capture, pixel) pairs
    def colors(self, cap_id, flat):
        return self.color_lut[self.images.reshape(len(self.captures), -1, 3)[cap_id, flat].long()]

    def rays(self, cap_id, flat):
        xy = torch.stack([flat % self.w, flat // self.w], dim=1).to(torch.int32).contiguous()
        cid = cap_id.to(torch.int32).contiguous()
        o = torch.empty((xy.shape[0], 3), device=self.device, dtype=torch.float32)
        d = torch.empty_like(o)
        _lib.check(_lib.lib().nm_shot_rays_cams(_lib.dev_ptr(xy, torch.int32), _lib.dev_ptr(cid, torch.int32), xy.shape[0], self.ray_mode,
                                                self.cams.data_ptr(), len(self.captures), _lib.dev_ptr(o), _lib.dev_ptr(d),
                                                _lib.stream_ptr()), "nm_shot_rays_cams")
        return o, d


class BackgroundRayBatcher:
    """datasets/background_rays.py BackgroundRayDataset.__getitem__ on a FrameStore: `rays_per_batch` random background pixels
    spread over the captures of the split."""

    def __init__(self, opt, store, inclusions=None, dset_type='train', draws='device', seed=0, rank=0, world=1):
        self.opt, self.store, self.dset_type = opt, store, dset_type
        self.batch_size = opt.rays_per_batch
        self.caps = list(range(len(store.captures))) if inclusions is None else [store.fname_to_index[f] for f in inclusions]
        self.draws = draws
        self.gen = torch.Generator(device=store.device).manual_seed(seed)
        # data-parallel training (neuman_hip/dp.py; reference train.py:26-28): every rank draws the SAME batch (same seed) and builds
        # the rays rank, rank + world, ... of it -- the union over the ranks is the single-process batch
        self.rank, self.world = int(rank), int(world)
        self.ablate = bool(getattr(opt, 'ablate_nerft', False))
        if not self.ablate:
            if store.masks is None:
                raise ValueError("captures carry no mask (background_rays.py:77-78)")
            self.pool_name = 'bkg_outside_border' if store.border is not None else 'bkg'
            _, _, self.offsets_host = store.pool(self.pool_name)
        self.cap_ids_dev = torch.tensor(self.caps, device=store.device, dtype=torch.int64)

    def __len__(self):
        return {'train': TRAIN_SET_LENGTH, 'val': VALIDATION_SET_LENGTH}[self.dset_type]

    def _draw_numpy(self):
        """the reference's np.random calls in its order: one multinomial over the captures, then per capture with rays either
        randint over its candidate list or (NeRF-T ablation) a y and an x over the whole image"""
        st, n = self.store, self.batch_size
        bins = np.random.multinomial(n, np.ones(len(self.caps)) / float(len(self.caps)))
        cap_id, pick = [], []
        for c, num in zip(self.caps, bins):
            if num == 0:
                continue
            if self.ablate:
                y = np.random.randint(0, st.h, num)
                x = np.random.randint(0, st.w, num)
                pick.append(y.astype(np.int64) * st.w + x)
            else:
                size = self.offsets_host[c + 1] - self.offsets_host[c]
                pick.append(self.offsets_host[c] + np.random.randint(0, size, num).astype(np.int64))
            cap_id.append(np.full(num, c, dtype=np.int64))
        dev = st.device
        return torch.from_numpy(np.concatenate(cap_id)).to(dev), torch.from_numpy(np.concatenate(pick)).to(dev)

    def _draw_device(self):
        st, n = self.store, self.batch_size
        cap_id = self.cap_ids_dev[torch.randint(0, len(self.caps), (n,), device=st.device, generator=self.gen)]
        cap_id = torch.sort(cap_id).values                        # grouped by capture like the reference's concatenation
        u = torch.rand((n,), device=st.device, generator=self.gen, dtype=torch.float64)
        if self.ablate:
            return cap_id, torch.clamp((u * (st.h * st.w)).long(), max=st.h * st.w - 1)
        _, offsets, _ = st.pool(self.pool_name)
        size = offsets[cap_id + 1] - offsets[cap_id]
        return cap_id, offsets[cap_id] + torch.minimum((u * size).long(), size - 1)

    def next_batch(self):
        st = self.store
        cap_id, pick = self._draw_numpy() if self.draws == 'numpy' else self._draw_device()
        if self.world > 1:
            cap_id, pick = cap_id[self.rank::self.world].contiguous(), pick[self.rank::self.world].contiguous()
        flat = pick if self.ablate else st.pool(self.pool_name)[0][pick].long()
        o, d = st.rays(cap_id, flat)
        n = cap_id.shape[0]
        out = {
            'color': st.colors(cap_id, flat),
            'origin': o,
            'direction': d,
            'near': st.near_bkg[cap_id][:, None],
            'far': st.far_bkg[cap_id][:, None],
            'is_bkg': torch.ones((n, 1), dtype=torch.int64, device=st.device),
        }
        if st.depth is not None:
            out['depth'] = st.depth.reshape(len(st.captures), -1)[cap_id, flat]
        if st.view_f is not None:
            out['viewf_list'] = st.view_f[cap_id][:, None]
        return out

    __call__ = next_batch


class HumanRayBatcher:
    """datasets/human_rays.py HumanRayDataset.__getitem__ on a FrameStore: the rays of ONE capture per batch, split into body /
    border / background classes, optionally led by a 32 x 32 patch on the body (for the LPIPS term)."""

    def __init__(self, opt, store, inclusions=None, dset_type='train', draws='device', seed=0):
        assert store.near_far is not None and store.near_human is not None, "the store needs the near/far cache and human bounds"
        self.opt, self.store, self.dset_type = opt, store, dset_type
        self.batch_size = opt.rays_per_batch
        self.inclusions = list(store.fnames) if inclusions is None else list(inclusions)
        self.num_patch = 1 if opt.penalize_lpips > 0 else 0
        self.cap_id = None
        self.draws = draws
        self.gen = torch.Generator(device=store.device).manual_seed(seed)
        self.host_rng = random.Random(seed)

    def __len__(self):
        return {'train': TRAIN_SET_LENGTH, 'val': VALIDATION_SET_LENGTH}[self.dset_type]

    def _randbelow(self, n):
        """index of random.choice(seq) for len(seq) == n: the module-level generator in 'numpy' mode (the reference's), ours else"""
        return random.choice(range(n)) if self.draws == 'numpy' else self.host_rng.randrange(n)

    def _pick(self, pool_name, cap, num):
        """`num` flat pixel indices drawn with replacement from a class's candidates in capture `cap` (human_rays.py:188)"""
        st = self.store
        flat, offsets, offsets_host = st.pool(pool_name)
        lo, size = offsets_host[cap], offsets_host[cap + 1] - offsets_host[cap]
        if size == 0:
            raise ValueError(f"capture {st.fnames[cap]} has no '{pool_name}' pixels to sample")
        if self.draws == 'numpy':
            k = torch.from_numpy(np.random.randint(0, size, num).astype(np.int64)).to(st.device)
        else:
            k = torch.randint(0, size, (num,), device=st.device, generator=self.gen)
        return flat[lo + k].long()

    def _patch(self, cap):
        """human_rays.py:162-181: a body pixel picked at random seeds a patch, returned in row-major order"""
        st = self.store
        flat, _, offsets_host = st.pool('body')
        lo, size = offsets_host[cap], offsets_host[cap + 1] - offsets_host[cap]
        seed = int(flat[lo + self._randbelow(size)])              # (one scalar read back: the patch corner is clamped on the host)
        x0, y0 = patch_corner((st.h, st.w), (seed % st.w, seed // st.w))
        ys = torch.arange(y0, y0 + PATCH_SIZE, device=st.device)
        xs = torch.arange(x0, x0 + PATCH_SIZE, device=st.device)
        return (ys[:, None] * st.w + xs[None, :]).reshape(-1)

    def next_batch(self):
        st, opt = self.store, self.opt
        if self.cap_id is None:
            name = random.choice(self.inclusions) if self.draws == 'numpy' else self.host_rng.choice(self.inclusions)
            cap = st.fname_to_index[name]
        else:
            cap = self.cap_id
        assert 0 <= cap < len(st.captures)
        if self.num_patch == 0:
            bins = [self.batch_size]
        else:
            assert self.batch_size > PATCH_SIZE_SQUARED
            bins = [PATCH_SIZE_SQUARED, self.batch_size - PATCH_SIZE_SQUARED]
        need_patch = (random.random() if self.draws == 'numpy' else self.host_rng.random()) < opt.body_rays_ratio
        patch_counter = 0
        picks = []
        for num in bins:
            if num == 0:
                continue
            if self.num_patch == 1 and need_patch and patch_counter == 0:
                assert num == PATCH_SIZE_SQUARED
                picks.append(self._patch(cap))
                patch_counter += 1
                continue
            for key, n_rays in num_rays_per_class(opt, num).items():
                if n_rays:
                    picks.append(self._pick({'num_body_rays': 'body', 'num_border_rays': 'border', 'num_bkg_rays': 'bkg'}[key], cap, n_rays))
        flat = torch.cat(picks)
        n = flat.shape[0]
        assert n == self.batch_size
        cap_id = torch.full((n,), cap, device=st.device, dtype=torch.int64)
        o, d = st.rays(cap_id, flat)
        cache = st.near_far[cap].reshape(-1, 2)[flat]
        valid = cache[:, NEAR_INDEX] < cache[:, FAR_INDEX]                                   # human_rays.py:197
        human_near = torch.where(valid, cache[:, NEAR_INDEX], st.near_human[cap])[:, None]
        human_far = torch.where(valid, cache[:, FAR_INDEX], st.far_human[cap])[:, None]
        out = {
            'color': st.colors(cap_id, flat),
            'origin': o,
            'direction': d,
            'human_near': human_near,
            'human_far': human_far,
            'bkg_near': st.near_bkg[cap].expand(n)[:, None],
            'bkg_far': st.far_bkg[cap].expand(n)[:, None],
            'is_bkg': (1 - st.binary[cap].reshape(-1)[flat]).long(),
            'is_hit': valid.long(),
            'cur_view_f': st.view_f64[cap] if st.view_f64 is not None else None,
            'cur_view': st.captures[cap].frame_id['frame_id'] if st.view_f64 is not None else None,
            'cap_id': cap,
            'patch_counter': torch.tensor(patch_counter),
        }
        return out

    __call__ = next_batch


# ------------------------------------------------------------------------------------------------
# the near/far cache files of data_io/cache_helper.py (interchange with a reference checkout)
# ------------------------------------------------------------------------------------------------
def _cache_path(captures, cap, geo_threshold, normalize):
    h, w = captures[0].shape
    return os.path.abspath(os.path.join(captures[0].image_path,
                                        f'../../cache/near_far_cache_{os.path.basename(cap.image_path)}_{h}_{w}_{geo_threshold}_{normalize}.npy'))


def export_near_far_cache(opt, scene, geo_threshold, chunk=None, device='cuda', store=None):
    """data_io/cache_helper.py:16-36: one [H,W,3] float64 .npy per capture (near, far, 1) beside the images, skipped when the file
    exists.  `chunk` is accepted and ignored (the device kernel takes the whole image at once)."""
    caps = scene.captures
    store = store or FrameStore(caps, device)
    for i, cap in enumerate(caps):
        path = _cache_path(caps, cap, geo_threshold, opt.normalize)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        if os.path.isfile(path):
            continue
        nf = store.near_far_of_capture(i, scene.verts[scene.image_path_to_index[cap.image_path]], geo_threshold)
        full = np.ones((store.h, store.w, 3))
        full[..., :2] = nf.cpu().numpy()
        np.save(path, full)


def load_near_far_cache(opt, scene, geo_threshold):
    """data_io/cache_helper.py:39-48: file name -> [H,W,3]"""
    book = {}
    for cap in scene.captures:
        path = _cache_path(scene.captures, cap, geo_threshold, opt.normalize)
        assert os.path.isfile(path), f'{path} not exist'
        book[os.path.basename(cap.image_path)] = np.load(path)
    return book


# ------------------------------------------------------------------------------------------------
# the reference's dataset classes by name and constructor, for its train.py (train.py:40-52, 108-119): a torch Dataset whose items are
# device batches.  Use with DataLoader(num_workers=0) -- the default collate adds the leading axis of 1 the reference's trainers strip
# (utils/utils.py:89-93), and nothing has to cross from worker processes because nothing is built on the host.
# ------------------------------------------------------------------------------------------------
class BackgroundRayDataset(torch.utils.data.Dataset):
    """datasets/background_rays.py:15-41 (opt, scene, dset_type, split) over a FrameStore built from scene.captures"""

    def __init__(self, opt, scene, dset_type, split, device='cuda', store=None, draws='device'):
        from .data_io import read_text
        self.opt, self.scene, self.dset_type, self.split = opt, scene, dset_type, split
        self.inclusions = read_text(split)
        has_border = all(hasattr(c, 'border_mask') for c in scene.captures)
        self.store = store or FrameStore(scene.captures, device, dilation=(opt.dilation if has_border else None),
                                         use_fused_depth=bool(getattr(opt, 'use_fused_depth', False)))
        self.batcher = BackgroundRayBatcher(opt, self.store, self.inclusions, dset_type, draws=draws)

    def __len__(self):
        return len(self.batcher)

    def __getitem__(self, index):
        return self.batcher.next_batch()


class HumanRayDataset(torch.utils.data.Dataset):
    """datasets/human_rays.py:36-79 (opt, scene, dset_type, split, near_far_cache=None); the cache is computed on the device from
    scene.verts when none is passed (the reference exports and reloads .npy files at this point, :62-64)"""

    def __init__(self, opt, scene, dset_type, split, near_far_cache=None, device='cuda', store=None, draws='device'):
        from .data_io import read_text
        self.opt, self.scene, self.dset_type, self.split = opt, scene, dset_type, split
        self.inclusions = read_text(split)
        self.store = store or FrameStore(scene.captures, device, dilation=opt.dilation, near_far_cache=near_far_cache,
                                         verts=None if near_far_cache is not None else scene.verts, geo_threshold=opt.geo_threshold)
        self.batcher = HumanRayBatcher(opt, self.store, self.inclusions, dset_type, draws=draws)
        self.near_far_cache = near_far_cache

    @property
    def cap_id(self):
        return self.batcher.cap_id

    @cap_id.setter
    def cap_id(self, value):
        self.batcher.cap_id = value

    def __len__(self):
        return len(self.batcher)

    def __getitem__(self, index):
        return self.batcher.next_batch()
