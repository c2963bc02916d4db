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
