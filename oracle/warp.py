"""Oracle: observation -> canonical warp of ray samples.  **parity unpinned**.

Restates reference utils/ray_utils.py:48-66 (warp_samples_to_canonical).  The
reference calls libigl 2.2.1 (environment.yml:13, not vendored, not installable
offline) for the closest-point query (:53) and the barycentrics (:55) and has
no test that pins either, so this file follows the published definitions:

* ``closest_point_on_mesh``: exact Euclidean closest point on every triangle
  (Voronoi-region test, Ericson "Real-Time Collision Detection" 5.1.5), global
  arg-min of the squared distance over the faces -- what
  ``igl.point_mesh_squared_distance`` returns (sqrD, face id, closest point);
* ``barycentric_coordinates_tri``: barycentrics of that point in the winning
  triangle, ordered (v0, v1, v2), cross-checked against the reference's own
  differentiable formula at utils/ray_utils.py:73-88.

On shared edges/vertices the winning face id is implementation-defined, but the
blended transform is continuous across faces, so parity is defined on
(closest point, can_pts, can_dirs), never on the face id.  Test infrastructure only.

What does pin it: tests/test_oracle_warp_independent.py solves the defining quadratic programme of every (point, triangle) pair with
scipy's general-purpose SLSQP -- an implementation that shares nothing with the region formulas below -- and finds the same
distances (2e-16) and points (4e-13), and the mesh query equal to the minimum over the triangles.
"""
import numpy as np

F32 = np.float32


def _dot(a, b):
    return np.sum(a * b, axis=-1)


def closest_point_on_triangles(p, a, b, c):
    """p [N,1,3], a/b/c [1,F,3] (float64) -> closest point [N,F,3]."""
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = _dot(ab, ap), _dot(ac, ap)
    bp = p - b
    d3, d4 = _dot(ab, bp), _dot(ac, bp)
    cp = p - c
    d5, d6 = _dot(ab, cp), _dot(ac, cp)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide='ignore', invalid='ignore'):
        v_ab = d1 / (d1 - d3)
        w_ac = d2 / (d2 - d6)
        w_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
        denom = 1.0 / (va + vb + vc)
    v_in, w_in = vb * denom, vc * denom
    conds = [
        (d1 <= 0) & (d2 <= 0),                         # vertex A
        (d3 >= 0) & (d4 <= d3),                        # vertex B
        (vc <= 0) & (d1 >= 0) & (d3 <= 0),             # edge AB
        (d6 >= 0) & (d5 <= d6),                        # vertex C
        (vb <= 0) & (d2 >= 0) & (d6 <= 0),             # edge AC
        (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0),   # edge BC
    ]
    zeros = np.zeros_like(d1)
    # barycentric (v, w) of the closest point: q = a + v*ab + w*ac
    v_sel = np.select(conds, [zeros, zeros + 1, v_ab, zeros, zeros, 1 - w_bc], default=v_in)
    w_sel = np.select(conds, [zeros, zeros, zeros, zeros + 1, w_ac, w_bc], default=w_in)
    return a + ab * v_sel[..., None] + ac * w_sel[..., None]


def closest_point_on_mesh(pts, verts, faces, chunk=32, culled=None):
    """igl.point_mesh_squared_distance(P, V, F) -> (sqrD [N], face id [N], closest [N,3]).  float64 math.

    The definition is the loop below: every point against every triangle, arg-min (first minimum = smallest face id).  For
    more than a few thousand points `culled` (default: N * F > 2e8) takes `closest_point_on_mesh_culled`, which evaluates the
    same per-pair arithmetic on a superset of the triangles that can win and returns the same arrays bit for bit
    (tests/test_oracle_warp_independent.py::test_culled_search_equals_the_all_pairs_loop)."""
    if culled is None:
        culled = pts.shape[0] * faces.shape[0] > 2e8
    if culled:
        return closest_point_on_mesh_culled(pts, verts, faces)
    p = pts.astype(np.float64)
    v = verts.astype(np.float64)
    a, b, c = v[faces[:, 0]][None], v[faces[:, 1]][None], v[faces[:, 2]][None]
    n = p.shape[0]
    sqr = np.empty(n)
    fid = np.empty(n, np.int64)
    closest = np.empty((n, 3))
    for s in range(0, n, chunk):
        q = closest_point_on_triangles(p[s:s + chunk, None, :], a, b, c)
        d2 = _dot(q - p[s:s + chunk, None, :], q - p[s:s + chunk, None, :])
        i = np.argmin(d2, axis=1)
        r = np.arange(i.shape[0])
        fid[s:s + chunk] = i
        sqr[s:s + chunk] = d2[r, i]
        closest[s:s + chunk] = q[r, i]
    return sqr, fid, closest


def closest_point_on_mesh_culled(pts, verts, faces, pair_budget=4_000_000):
    """The all-pairs search restricted, per point, to the triangles that can hold its closest point -- exact, not approximate:
    with d_ub the distance from p to the nearest mesh VERTEX (a point of the surface, so an upper bound of the answer), a
    triangle whose centroid is further than d_ub + (its circumscribed radius about the centroid) from p cannot beat it.  The
    candidates come from a k-d tree over the centroids with the largest such radius (a superset); the surviving (point, triangle)
    pairs go through `closest_point_on_triangles` in flat arrays -- the same elementwise float64 operations as the loop of
    `closest_point_on_mesh` -- and the minimum per point is taken with ties to the smallest face id, as np.argmin does there."""
    from scipy.spatial import cKDTree
    p = pts.astype(np.float64)
    v = verts.astype(np.float64)
    tri = faces[:, :3]
    a, b, c = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
    cen = (a + b + c) / 3.0
    rad = np.sqrt(np.maximum(np.maximum(_dot(a - cen, a - cen), _dot(b - cen, b - cen)), _dot(c - cen, c - cen)))
    rmax = float(rad.max()) * (1 + 1e-9) + 1e-12
    used = np.unique(tri)
    n = p.shape[0]
    finite = np.isfinite(p).all(1)
    d_ub = np.full(n, np.inf)
    d_ub[finite] = cKDTree(v[used]).query(p[finite])[0]
    ctree = cKDTree(cen)
    sqr = np.full(n, np.inf)
    fid = np.zeros(n, np.int64)
    closest = np.zeros((n, 3))
    start = 0
    step = 4096
    while start < n:
        stop = min(n, start + step)
        rows = np.arange(start, stop)[finite[start:stop]]
        if rows.size:
            lists = ctree.query_ball_point(p[rows], d_ub[rows] * (1 + 1e-9) + rmax)
            cnt = np.fromiter((len(l) for l in lists), np.int64, rows.size)
            if cnt.sum() > pair_budget and rows.size > 64:               # (far points see the whole mesh: shrink the batch)
                step = max(64, step // 4)
                continue
            pi = np.repeat(rows, cnt)
            fi = np.concatenate([np.asarray(l, np.int64) for l in lists]) if cnt.sum() else np.zeros(0, np.int64)
            keep = np.sqrt(_dot(cen[fi] - p[pi], cen[fi] - p[pi])) <= d_ub[pi] * (1 + 1e-9) + rad[fi] * (1 + 1e-9) + 1e-12
            pi, fi = pi[keep], fi[keep]
            q = closest_point_on_triangles(p[pi], a[fi], b[fi], c[fi])
            d2 = _dot(q - p[pi], q - p[pi])
            order = np.lexsort((fi, d2, pi))                              # per point: smallest d2, then smallest face id
            first = np.ones(order.size, bool)
            first[1:] = pi[order][1:] != pi[order][:-1]
            w = order[first]
            sqr[pi[w]], fid[pi[w]], closest[pi[w]] = d2[w], fi[w], q[w]
        bad = np.arange(start, stop)[~finite[start:stop]]
        if bad.size:                                                      # non-finite queries: whatever the loop's arithmetic gives
            s_, f_, c_ = closest_point_on_mesh(pts[bad], verts, faces, culled=False)
            sqr[bad], fid[bad], closest[bad] = s_, f_, c_
        start = stop
    return sqr, fid, closest


def barycentric_coordinates_tri(p, a, b, c):
    """igl.barycentric_coordinates_tri(P, A, B, C) -> [N,3] weights of (A,B,C)."""
    v0, v1, v2 = b - a, c - a, p - a
    d00, d01, d11 = _dot(v0, v0), _dot(v0, v1), _dot(v1, v1)
    d20, d21 = _dot(v2, v0), _dot(v2, v1)
    denom = d00 * d11 - d01 * d01
    v = (d11 * d20 - d01 * d21) / denom
    w = (d00 * d21 - d01 * d20) / denom
    return np.stack([1.0 - v - w, v, w], axis=1)


def barycentric_reference_diff_formula(p, a, b, c):
    """The reference's own in-repo barycentric formula, utils/ray_utils.py:73-88 (ordering cross-check)."""
    n = np.cross(b - a, c - a)
    denom = _dot(n, n)
    u = _dot(n, np.cross(c - b, p - b)) / denom
    v = _dot(n, np.cross(a - c, p - c)) / denom
    return np.stack([u, v, 1 - u - v], axis=1)


def warp_samples_to_canonical(pts, verts, faces, T):
    """reference utils/ray_utils.py:48-66.

    pts [R,S,3] f32, verts [V,3] f32, faces [F,>=3] int, T [>=V,4,4] f64
    -> can_pts [R,S,3], can_dirs [R,S,3], closest [R,S,3] (float64, callers cast
    to f32 as utils/render_utils.py:226-227 does).
    """
    assert pts.ndim == 3 and pts.shape[-1] == 3
    num_rays, num_samples, _ = pts.shape
    flat = pts.reshape(-1, 3)
    tri = faces[:, :3]
    _, f_id, closest = closest_point_on_mesh(flat, verts, tri)
    ctri = verts[tri[f_id]].astype(np.float64)
    bary = barycentric_coordinates_tri(closest, ctri[:, 0], ctri[:, 1], ctri[:, 2])
    T_interp = (T[tri[f_id]] * bary[..., None, None]).sum(axis=1)
    T_inv = np.linalg.inv(T_interp)
    hom = np.concatenate([flat.astype(np.float64), np.ones((flat.shape[0], 1))], -1)
    can_pts = (T_inv @ hom[..., None])[:, :3, 0].reshape(num_rays, num_samples, 3)
    closest = closest.reshape(num_rays, num_samples, 3)
    can_dirs = can_pts[:, 1:] - can_pts[:, :-1]
    can_dirs = np.concatenate([can_dirs, can_dirs[:, -1:]], axis=1)
    can_dirs = can_dirs / np.linalg.norm(can_dirs, axis=2, keepdims=True)
    return can_pts, can_dirs, closest


# ---------------------------------------------------------------------------------------------------------------------
# signed distance (reference: igl.signed_distance at utils/ray_utils.py:70, trainers/human_nerf_trainer.py:310, 326) --
# **parity unpinned** like the rest of this file (libigl absent).  libigl's default sign for a 3-D triangle mesh is the
# angle-weighted pseudonormal test (Baerentzen & Aanaes 2005): the sign of (p - q) . n(q), n the face normal in the
# triangle's interior, the sum of the two unit face normals on an edge, the incident-angle-weighted sum of unit face normals
# at a vertex.  For a closed, consistently oriented mesh that sign is the inside / outside of the solid; `winding_number`
# below computes the latter independently (van Oosterom & Strackee solid angles) and the tests compare the two.
# ---------------------------------------------------------------------------------------------------------------------
def _unit(x):
    return x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), 1e-300)


def pseudonormals(verts, faces):
    """-> face normals [F,3], vertex pseudonormals [V,3], edge pseudonormals [F,3,3] (edge e of face f joins its
    vertices e and (e+1)%3), float64, unnormalised sums of UNIT face normals."""
    v = verts.astype(np.float64)
    f = faces[:, :3].astype(np.int64)
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    fn = _unit(np.cross(b - a, c - a))
    vn = np.zeros_like(v)
    corners = [(a, b, c), (b, c, a), (c, a, b)]
    for k, (p0, p1, p2) in enumerate(corners):
        e1, e2 = _unit(p1 - p0), _unit(p2 - p0)
        ang = np.arccos(np.clip(_dot(e1, e2), -1.0, 1.0))
        np.add.at(vn, f[:, k], fn * ang[:, None])
    edge = {}
    for i in range(f.shape[0]):
        for e in range(3):
            key = (min(f[i, e], f[i, (e + 1) % 3]), max(f[i, e], f[i, (e + 1) % 3]))
            edge.setdefault(key, []).append(i)
    en = np.zeros((f.shape[0], 3, 3))
    for i in range(f.shape[0]):
        for e in range(3):
            key = (min(f[i, e], f[i, (e + 1) % 3]), max(f[i, e], f[i, (e + 1) % 3]))
            en[i, e] = fn[edge[key]].sum(0)
    return fn, vn, en


def signed_distance(pts, verts, faces):
    """igl.signed_distance(P, V, F) -> (S [N], face id [N], closest [N,3]) with the pseudonormal sign."""
    sqr, fid, q = closest_point_on_mesh(pts, verts, faces)
    v = verts.astype(np.float64)
    f = faces[:, :3].astype(np.int64)
    fn, vn, en = pseudonormals(verts, faces)
    tri = v[f[fid]]                                                      # [N,3,3]
    # which feature of the winning triangle holds q: barycentrics of q (exact up to rounding: q was built from them)
    bary = barycentric_coordinates_tri(q, tri[:, 0], tri[:, 1], tri[:, 2])
    on = bary > 1e-9                                                     # vertices with weight
    n = np.empty_like(q)
    for i in range(q.shape[0]):
        k = np.flatnonzero(on[i])
        if len(k) == 1:
            n[i] = vn[f[fid[i], k[0]]]
        elif len(k) == 2:
            e = {(0, 1): 0, (1, 2): 1, (0, 2): 2}[tuple(k)]
            n[i] = en[fid[i], e]
        else:
            n[i] = fn[fid[i]]
    d = pts.astype(np.float64) - q
    sign = np.where(_dot(d, n) < 0, -1.0, 1.0)
    return sign * np.sqrt(sqr), fid, q


def winding_number(pts, verts, faces):
    """generalised winding number of a closed oriented mesh at each point (1 inside, 0 outside), float64"""
    v = verts.astype(np.float64)
    f = faces[:, :3].astype(np.int64)
    out = np.zeros(pts.shape[0])
    for i, p in enumerate(pts.astype(np.float64)):
        a, b, c = v[f[:, 0]] - p, v[f[:, 1]] - p, v[f[:, 2]] - p
        la, lb, lc = (np.linalg.norm(x, axis=1) for x in (a, b, c))
        num = _dot(a, np.cross(b, c))
        den = la * lb * lc + _dot(a, b) * lc + _dot(b, c) * la + _dot(c, a) * lb
        out[i] = np.sum(2.0 * np.arctan2(num, den)) / (4.0 * np.pi)
    return out
