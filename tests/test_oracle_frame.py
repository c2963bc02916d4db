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
