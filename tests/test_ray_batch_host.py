"""Host-side pieces of neuman_hip/ray_batches.py (no GPU): class counts, patch clamping, border masks, fused depth."""
import os
import types

import numpy as np
import torch

from neuman_hip import ray_batches as rb

HERE = os.path.dirname(os.path.abspath(__file__))


def test_border_mask_on_cpu_equals_scipy():
    from scipy import ndimage
    rng = np.random.default_rng(0)
    m = (rng.uniform(size=(3, 37, 41)) > 0.93).astype(np.uint8)
    for it in (0, 1, 2, 7, 30):
        want = np.stack([(ndimage.binary_dilation(x, iterations=it).astype(x.dtype) - x) if it > 0 else x - x for x in m])
        assert np.array_equal(rb.border_mask(torch.from_numpy(m), it).numpy(), want), it
    g = dict(np.load(os.path.join(HERE, "golden", "ray_batches.npz")))
    import sys
    sys.path.insert(0, os.path.join(HERE, "helpers"))
    import batch_scene
    masks = np.stack([c['mask'] for c in batch_scene.make(seed=11)['captures']])
    assert np.array_equal(rb.border_mask(torch.from_numpy(masks), 4).numpy(), g['border'])     # the reference's add_border_mask


def test_ray_class_counts():
    """datasets/human_rays.py:81-95, including Python's round-half-even and the leftover rule"""
    o = types.SimpleNamespace(body_rays_ratio=0.95, border_rays_ratio=0.05, bkg_rays_ratio=0.0, dilation=30)
    assert rb.num_rays_per_class(o, 4096) == {'num_body_rays': 3891, 'num_border_rays': 205, 'num_bkg_rays': 0}
    assert rb.num_rays_per_class(o, 10) == {'num_body_rays': 10, 'num_border_rays': 0, 'num_bkg_rays': 0}      # round(9.5) = 10, round(0.5) = 0
    o0 = types.SimpleNamespace(body_rays_ratio=0.6, border_rays_ratio=0.15, bkg_rays_ratio=0.25, dilation=0)
    assert rb.num_rays_per_class(o0, 300) == {'num_body_rays': 225, 'num_border_rays': 0, 'num_bkg_rays': 75}  # no border class without dilation
    for n in range(1, 200):
        assert sum(rb.num_rays_per_class(o0, n).values()) == n


def test_patch_corner_clamps_to_the_image():
    assert rb.patch_corner((40, 56), (28, 20)) == (12, 4)
    assert rb.patch_corner((40, 56), (3, 2)) == (0, 0)
    assert rb.patch_corner((40, 56), (55, 39)) == (24, 8)
    assert rb.patch_corner((100, 100), (50, 50), size=32) == (34, 34)


def test_fused_depth_is_the_linregress_map():
    from scipy import stats
    rng = np.random.default_rng(2)
    mono = rng.uniform(0.2, 4.0, size=(30, 44))
    depth = (1.7 * mono + 0.3 + rng.normal(size=mono.shape) * 0.05).astype(np.float32)
    depth[rng.uniform(size=depth.shape) < 0.3] = 0                       # holes of the MVS depth
    mask = np.zeros(depth.shape, np.uint8)
    mask[8:20, 10:30] = 1
    valid = (depth > 0) & (mask == 0)
    res = stats.linregress(mono[valid], depth[valid])
    want = depth.copy()
    want[~valid] = mono[~valid] * res.slope + res.intercept
    got = rb.fused_depth(depth, mono, mask)
    assert got.dtype == depth.dtype and np.allclose(got, want, rtol=1e-6, atol=1e-6)
    assert np.array_equal(got[valid], depth[valid])
