"""CPU: the egress restatement (oracle/frame.py) against hand-computed known answers of the published rules."""
import numpy as np

from oracle import frame as OF


def test_to_uint8_known_answers():
    x = np.array([0.0, 1.0, 0.5, 0.25, 1 / 255, 0.5 / 255, 0.49 / 255, 254.5 / 255, 254.6 / 255, -0.1, 1.1], dtype=np.float32)
    #            0    255  127.5+.49 -> 127;  63.75+.49 -> 64; 1.49 -> 1; 0.99.. -> 0 or 1 by the f32 value of 0.5/255
    got = OF.to_uint8(x)
    assert got.dtype == np.uint8
    assert list(got[[0, 1, 2, 3, 4, 6, 8, 9, 10]]) == [0, 255, 127, 64, 1, 0, 255, 0, 255]
    for i in (5, 7):                                            # the two half-way inputs: decided by the f32 rounding of x
        assert got[i] == int(np.float64(x[i]) * 255 + 0.499999999)


def test_psnr_known_answers():
    a = np.zeros((4, 4, 3), np.uint8)
    b = a.copy()
    assert OF.psnr_uint8(a, b) == np.inf
    b[...] = 1                                                   # mse = 1 -> 20 log10(255)
    assert abs(OF.psnr_uint8(a, b) - 20 * np.log10(255.0)) < 1e-12
    b[...] = 255
    assert abs(OF.psnr_uint8(a, b)) < 1e-12
    c = a.copy()
    c[0, 0, 0] = 200                                             # uint8 difference must not wrap
    assert abs(OF.psnr_uint8(c, a) - 10 * np.log10(255.0 ** 2 / (200.0 ** 2 / 48))) < 1e-12


def test_signed_distance_sign_is_the_winding_number():
    """oracle/warp.py:signed_distance (pseudonormal sign, what igl.signed_distance computes) against an independent
    inside/outside test, the generalised winding number, on a closed mesh"""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "ml-neuman_amd"))
    from neuman_hip import synthetic
    from oracle import warp as OW
    verts, faces = synthetic.capsule_mesh(n_rings=8, n_seg=10)
    posed, _ = synthetic.twist_transforms(verts)
    rng = np.random.default_rng(1)
    pts = (posed[rng.integers(0, len(posed), 200)] + rng.normal(size=(200, 3)) * 0.06).astype(np.float32)
    S, I, C = OW.signed_distance(pts, posed, faces)
    wn = OW.winding_number(pts, posed, faces)
    assert ((S < 0) == (wn > 0.5)).all() and 0.1 < (S < 0).mean() < 0.9
    np.testing.assert_allclose(np.abs(S), np.linalg.norm(C - pts, axis=1), atol=1e-6)


def test_ssim_known_properties():
    from oracle import frame as OF
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (40, 50, 3), dtype=np.uint8)
    assert OF.ssim_uint8(a, a) == 1.0
    b = (a.astype(np.int32) + rng.integers(-20, 21, a.shape)).clip(0, 255).astype(np.uint8)
    s = OF.ssim_uint8(a, b)
    assert 0.5 < s < 1.0 and abs(s - OF.ssim_uint8(b, a)) < 1e-15
    # a constant image against another constant: only the luminance term is left: (2 u v + C1) / (u^2 + v^2 + C1)
    u, v = 100.0, 140.0
    c1 = (0.01 * 255) ** 2
    s = OF.ssim_uint8(np.full((9, 9, 1), 100, np.uint8), np.full((9, 9, 1), 140, np.uint8))
    assert abs(s - (2 * u * v + c1) / (u * u + v * v + c1)) < 1e-12


def test_ssim_is_the_windowed_definition():
    """The oracle's SSIM (scipy's uniform_filter) against the definition written out window by window (Wang et al. 2004 with the
    sample covariance and the 7 x 7 uniform window scikit-image defaults to): every interior pixel's own 49-pixel window, no filter
    call, no boundary convention -- pins the filter's alignment and the normalisations on the cropped region the score is taken over."""
    from oracle import frame as OF
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (15, 17, 2), dtype=np.uint8)
    b = (a.astype(np.int32) + rng.integers(-30, 31, a.shape)).clip(0, 255).astype(np.uint8)
    c1, c2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    per_channel = []
    for ch in range(a.shape[2]):
        X, Y = a[..., ch].astype(np.float64), b[..., ch].astype(np.float64)
        vals = []
        for y in range(3, a.shape[0] - 3):
            for x in range(3, a.shape[1] - 3):
                wx, wy = X[y - 3:y + 4, x - 3:x + 4].ravel(), Y[y - 3:y + 4, x - 3:x + 4].ravel()
                ux, uy = wx.mean(), wy.mean()
                vx, vy = ((wx - ux) ** 2).sum() / 48.0, ((wy - uy) ** 2).sum() / 48.0
                vxy = ((wx - ux) * (wy - uy)).sum() / 48.0
                vals.append(((2 * ux * uy + c1) * (2 * vxy + c2)) / ((ux * ux + uy * uy + c1) * (vx + vy + c2)))
        per_channel.append(np.mean(vals))
    assert abs(OF.ssim_uint8(a, b) - float(np.mean(per_channel))) < 1e-12
