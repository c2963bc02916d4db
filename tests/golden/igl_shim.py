"""A stand-in for the three libigl calls the reference makes on the hot path, so that the reference's OWN code behind the warp
can execute in the build container (libigl 2.2.1, environment.yml:13, is not installable offline):

    igl.point_mesh_squared_distance(P, V, F)          utils/ray_utils.py:53
    igl.barycentric_coordinates_tri(P, A, B, C)       utils/ray_utils.py:55
    igl.signed_distance(P, V, F)                      utils/ray_utils.py:70, trainers/human_nerf_trainer.py:310, 326

Installed as sys.modules['igl'] by make_golden_posed.py / make_golden_human_loss.py BEFORE the reference is imported; the
reference's modules are imported unmodified.  The arithmetic is oracle/warp.py's (the published definitions: exact closest point on
every triangle, global arg-min; barycentrics of that point; angle-weighted pseudonormal sign); the return conventions are the
python bindings' of igl 2.2.1: tuples in the order (sqrD | S, I, C), floating outputs in the dtype of `P` (the bindings
are dtype-matched: float32 queries give float32 results), indices as an integer vector, closest points [N, 3].

What this does and does not pin: everything the reference computes AROUND these primitives (the transform blend, the 4x4
inverses, the finite-difference directions, the hit / miss bookkeeping, the merges, the loss terms) is executed by the reference
itself; the primitives themselves remain pinned on their mathematical definition only (tests/test_oracle_warp_independent.py).
"""
import os
import sys

import numpy as np

_ROOT = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from oracle import warp as _w  # noqa: E402

SIGNED_DISTANCE_PSEUDONORMAL = 0          # igl.SIGNED_DISTANCE_TYPE_PSEUDONORMAL (the default for 3-D meshes)


def _like(p, x):
    dt = np.asarray(p).dtype
    return np.ascontiguousarray(x, dtype=dt if dt.kind == 'f' else np.float64)


def point_mesh_squared_distance(p, v, f):
    sqr, fid, closest = _w.closest_point_on_mesh(np.asarray(p), np.asarray(v), np.asarray(f)[:, :3])
    return _like(p, sqr), fid.astype(np.int32), _like(p, closest)


def barycentric_coordinates_tri(p, a, b, c):
    f8 = np.float64
    return _like(p, _w.barycentric_coordinates_tri(np.asarray(p, f8), np.asarray(a, f8), np.asarray(b, f8), np.asarray(c, f8)))


def signed_distance(p, v, f, sign_type=SIGNED_DISTANCE_PSEUDONORMAL, return_normals=False):
    assert sign_type == SIGNED_DISTANCE_PSEUDONORMAL and not return_normals
    s, fid, closest = _w.signed_distance(np.asarray(p), np.asarray(v), np.asarray(f)[:, :3])
    return _like(p, s), fid.astype(np.int32), _like(p, closest)
