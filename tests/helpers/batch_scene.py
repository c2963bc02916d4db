"""The synthetic training scene behind tests/golden/ray_batches.npz: three 40 x 56 captures with random images, an elliptic body
mask, depth maps and a posed vertex cloud in front of each camera.  Pure numpy, seeded: the golden generator (which wraps it in the
reference's classes) and the device tests (which wrap it in neuman_hip.data_io's) build the identical arrays."""
import numpy as np

H, W, DILATION = 40, 56, 4


def _rot(q):
    w, x, y, z = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def make(seed=11, n_caps=3, n_verts=240):
    rng = np.random.default_rng(seed)
    caps = []
    yy, xx = np.mgrid[0:H, 0:W]
    for i in range(n_caps):
        ang = 0.15 * (i - 1)
        q = np.array([np.cos(ang / 2), 0.02 * i, np.sin(ang / 2), -0.01 * i])
        q = q / np.linalg.norm(q)
        t = np.array([0.1 * i - 0.1, 0.05 * i, 0.02 * i])
        R = _rot(q)                                               # world -> camera
        body_cam = np.array([0.05 * (i - 1), -0.03 * i, 2.4 + 0.2 * i])
        pts_cam = body_cam + rng.normal(size=(n_verts, 3)) * np.array([0.16, 0.30, 0.10])
        verts = ((pts_cam - t) @ R).astype(np.float32)             # R^T (p - t), row form
        fx, fy, cx, cy = 60.0 + 2 * i, 61.0 + i, W / 2 - 0.5 + i, H / 2 + 0.25 * i
        cy0, cx0 = H / 2 + 1.5 * i, W / 2 - 2.0 * i
        mask = ((((yy - cy0) / 11.0) ** 2 + ((xx - cx0) / 7.0) ** 2) < 1.0).astype(np.uint8)
        caps.append({
            'name': f'{i:05d}.png', 'q': q, 't': t, 'intrinsics': (fx, fy, cx, cy),
            'image': rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8),
            'mask': mask,
            'depth': rng.uniform(0.4, 3.0, size=(H, W)).astype(np.float32),
            'near': {'bkg': 0.0, 'human': 1.7 + 0.1 * i}, 'far': {'bkg': 3.14 + 0.01 * i, 'human': 3.3 + 0.1 * i},
            'total_frames': n_caps, 'verts': verts,
        })
    return {'h': H, 'w': W, 'dilation': DILATION, 'captures': caps}
