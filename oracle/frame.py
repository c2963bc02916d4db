"""CPU restatement of the frame egress that follows the ray-march path -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

reference render_test_views.py:83-92 / render_360.py:77-81 hand the renderer's float32 [H,W,3] frame to
`imageio.imsave`, read the PNG back and score it with `skimage.metrics.peak_signal_noise_ratio` (:35).  Both
packages are conda dependencies of the reference (environment.yml:22, :30, unpinned) and are absent here and from
/root/reference, so this file restates their published rules -- PARITY UNPINNED for these two functions:

* imageio 2.x `core/util.py: image_as_uint(im, bitdepth=8)` for a float image whose values lie in [0, 1]:
  `im.astype(float64) * 255 + 0.499999999`, then `.astype(uint8)` (truncation).  (Outside [0, 1] imageio rescales the
  whole image by its min / max; the renderers' frames are convex combinations of sigmoids plus `1 - acc`, so they stay
  inside up to an ulp; the device kernel clips instead.)
* scikit-image `peak_signal_noise_ratio(image_true, image_test)` with uint8 inputs: data_range = 255,
  `10 * log10(255**2 / mean((true - test)**2))` with the difference taken in float64.
"""
import numpy as np


def to_uint8(frame):
    im = np.clip(np.asarray(frame, dtype=np.float64), 0.0, 1.0)
    return (im * 255.0 + 0.499999999).astype(np.uint8)


def psnr_uint8(gt, pred):
    err = np.mean((gt.astype(np.float64) - pred.astype(np.float64)) ** 2)
    return np.inf if err == 0 else 10.0 * np.log10(255.0 ** 2 / err)


def ssim_uint8(pred, gt):
    """skimage.metrics.structural_similarity(pred, gt, multichannel=True) for uint8 [H,W,C] images, restated from the published
    implementation (scikit-image 0.18 metrics/_structural_similarity.py): per channel, float64 images, scipy's uniform_filter
    with win_size 7, sample covariance, K1 0.01 / K2 0.03, data_range 255 (the dtype's range), the mean of the map cropped by
    (win_size - 1) // 2 pixels; then the mean over channels.  **parity unpinned** (scikit-image absent)."""
    from scipy.ndimage import uniform_filter
    win, pad = 7, 3
    NP = win * win
    cov_norm = NP / (NP - 1.0)
    C1, C2 = (0.01 * 255.0) ** 2, (0.03 * 255.0) ** 2
    out = []
    for c in range(pred.shape[2]):
        X, Y = pred[..., c].astype(np.float64), gt[..., c].astype(np.float64)
        ux, uy = uniform_filter(X, size=win), uniform_filter(Y, size=win)
        uxx, uyy, uxy = uniform_filter(X * X, size=win), uniform_filter(Y * Y, size=win), uniform_filter(X * Y, size=win)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        S = ((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux ** 2 + uy ** 2 + C1) * (vx + vy + C2))
        out.append(S[pad:-pad, pad:-pad].mean())
    return float(np.mean(out))
