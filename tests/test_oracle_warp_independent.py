"""oracle/warp.py's closest-point query is "parity unpinned" (libigl, the reference's dependency for it, is not installable here and
the reference holds no golden for it).  This file pins it on the MATHEMATICAL definition with an implementation that shares nothing
with it: for every (point, triangle) pair the closest point is the solution of a convex quadratic programme
    min |a + s (b - a) + t (c - a) - p|^2   subject to   s >= 0, t >= 0, s + t <= 1,
solved by scipy's general-purpose SLSQP; the mesh answer is the minimum over the triangles.  The oracle's Voronoi-region formulas
(Ericson 5.1.5) must give the same distances and points, and its barycentric weights must reproduce the point.  No GPU."""
import numpy as np
import pytest
from scipy.optimize import minimize

from neuman_hip import synthetic
from oracle import warp as OW


def qp_closest(p, a, b, c):
    ab, ac = b - a, c - a

    def f(x):
        r = a + x[0] * ab + x[1] * ac - p
        return float(r @ r)

    def g(x):
        r = a + x[0] * ab + x[1] * ac - p
        return np.array([2 * (r @ ab), 2 * (r @ ac)])
    cons = [{'type': 'ineq', 'fun': lambda x: x[0], 'jac': lambda x: np.array([1.0, 0.0])},
            {'type': 'ineq', 'fun': lambda x: x[1], 'jac': lambda x: np.array([0.0, 1.0])},
            {'type': 'ineq', 'fun': lambda x: 1.0 - x[0] - x[1], 'jac': lambda x: np.array([-1.0, -1.0])}]
    best = None
    for x0 in ([1 / 3, 1 / 3], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0]):            # a convex problem: the starts only guard the solver's tolerance
        res = minimize(f, np.array(x0), jac=g, constraints=cons, method='SLSQP', options={'ftol': 1e-16, 'maxiter': 200})
        x = np.clip(res.x, 0.0, 1.0)
        if x.sum() > 1.0:
            x = x / x.sum()
        val = f(x)
        if best is None or val < best[0]:
            best = (val, a + x[0] * ab + x[1] * ac)
    return best


@pytest.fixture(scope="module")
def mesh():
    verts, faces = synthetic.capsule_mesh(n_rings=5, n_seg=7)
    return verts.astype(np.float64), np.ascontiguousarray(faces[:, :3], np.int64)


def test_per_triangle_closest_point_is_the_qp_solution(mesh):
    verts, faces = mesh
    rng = np.random.default_rng(0)
    a, b, c = (verts[faces[:24, k]] for k in range(3))
    pts = np.concatenate([rng.normal(size=(10, 3)) * 0.6,                                        # anywhere
                          a[:3] + 0.2 * (a[:3] - b[:3]),                                          # beyond a vertex
                          0.5 * (a[3:6] + b[3:6]) + 0.1 * np.cross(b[3:6] - a[3:6], c[3:6] - a[3:6]),   # over an edge
                          (a[6:9] + b[6:9] + c[6:9]) / 3 + 0.3 * rng.normal(size=(3, 3))])         # around a face
    got = OW.closest_point_on_triangles(pts[:, None, :], a[None], b[None], c[None])               # [N, F, 3]
    worst_d, worst_q = 0.0, 0.0
    for i, p in enumerate(pts):
        for f in range(a.shape[0]):
            d2, q = qp_closest(p, a[f], b[f], c[f])
            o = got[i, f]
            worst_d = max(worst_d, abs(np.sqrt(d2) - np.linalg.norm(o - p)))
            worst_q = max(worst_q, np.linalg.norm(o - q))
    print(f"[warp oracle] {pts.shape[0] * a.shape[0]} point-triangle pairs vs SLSQP: distance {worst_d:.2e}, closest point {worst_q:.2e}")
    assert worst_d < 1e-7 and worst_q < 2e-5          # (the point is flat-conditioned along the triangle near an optimum: sqrt of the solver's tolerance)


def test_mesh_query_is_the_minimum_over_triangles_and_barycentrics_reproduce_it(mesh):
    verts, faces = mesh
    rng = np.random.default_rng(1)
    pts = np.concatenate([rng.normal(size=(12, 3)) * np.array([0.3, 0.7, 0.2]), verts[:2] * 1.3, verts[5:7] * 0.5])
    sqr, fid, q = OW.closest_point_on_mesh(pts, verts, faces)
    a, b, c = (verts[faces[:, k]] for k in range(3))
    for i, p in enumerate(pts):
        best = min(qp_closest(p, a[f], b[f], c[f])[0] for f in range(faces.shape[0]))
        assert abs(np.sqrt(best) - np.sqrt(sqr[i])) < 1e-7, (i, best, sqr[i])
        assert abs(np.linalg.norm(q[i] - p) ** 2 - sqr[i]) < 1e-12
    w = OW.barycentric_coordinates_tri(q, a[fid], b[fid], c[fid])
    assert np.abs(w.sum(1) - 1).max() < 1e-12 and w.min() > -1e-9
    assert np.abs((w[:, :, None] * np.stack([a[fid], b[fid], c[fid]], 1)).sum(1) - q).max() < 1e-12
    assert np.abs(w - OW.barycentric_reference_diff_formula(q, a[fid], b[fid], c[fid])).max() < 1e-9   # the reference's own in-repo formula


def test_culled_search_equals_the_all_pairs_loop():
    """closest_point_on_mesh_culled (what large queries take) returns the all-pairs loop's arrays BIT FOR BIT: same distances, same
    face ids (ties to the smallest id), same points -- on an SMPL-sized mesh with interior, shell, far, on-vertex (tied) and
    non-finite queries"""
    verts_c, faces = synthetic.capsule_mesh()
    posed, _ = synthetic.twist_transforms(verts_c)
    rng = np.random.default_rng(5)
    pts = (rng.normal(size=(700, 3)) * np.array([0.4, 0.8, 0.3])).astype(np.float32)
    pts[:40] = posed[rng.integers(0, posed.shape[0], 40)]                  # on vertices: several faces tie exactly
    pts[40:60] += np.array([4.0, -2.0, 3.0], np.float32)                   # far away: every triangle is a candidate
    pts[60] = np.nan
    with np.errstate(invalid='ignore'):
        a = OW.closest_point_on_mesh(pts, posed, faces, culled=False)
        b = OW.closest_point_on_mesh_culled(pts, posed, faces)
    for x, y, name in zip(a, b, ("sqrD", "face id", "closest")):
        fin = np.isfinite(pts).all(1)
        assert np.array_equal(x[fin], y[fin]), name
