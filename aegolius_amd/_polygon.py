"""Host side of `interior_polygon` (reference cores/triangulation_functions.py:390-430).

The reference decides, per polygon, between three evaluators: the polygon itself when it is convex
(:394-401), its ear-clipped triangles when it is concave and simple (:423-428), and a debug-print
laden decomposition for self-intersecting outlines (:403-421). The first two reduce to "is the point
inside ANY of these convex pieces", a pure per-point test that the device evaluates
(prim_polysign); this module produces the pieces (topology only — no per-point work).
"""
import numpy as np


def _cross2(a, b):
    return a[0] * b[1] - a[1] * b[0]


def _convexity(vs):
    """check_convex_all :22-37 — note that the last vertex is (deliberately mirrored) not tested."""
    m = vs.shape[1]
    c = [_cross2(vs[:2, i] - vs[:2, i - 1], vs[:2, i + 1] - vs[:2, i]) for i in range(1, m - 1)]
    c.append(_cross2(vs[:2, 0] - vs[:2, -1], vs[:2, 1] - vs[:2, 0]))
    return np.asarray(c)


def _inside_triangle(vs, t):
    """is_inside_triangle :40-60 (barycentric test of every vertex in vs against triangle t)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        d = (t[1, 1] - t[1, 2]) * (t[0, 0] - t[0, 2]) + (t[0, 2] - t[0, 1]) * (t[1, 0] - t[1, 2])
        l1 = ((t[1, 1] - t[1, 2]) * (vs[0, :] - t[0, 2]) + (t[0, 2] - t[0, 1]) * (vs[1, :] - t[1, 2])) / d
        l2 = ((t[1, 2] - t[1, 0]) * (vs[0, :] - t[0, 2]) + (t[0, 0] - t[0, 2]) * (vs[1, :] - t[1, 2])) / d
    l3 = 1 - l1 - l2
    c = l1 * l2 * l3
    return (c >= 0) * (l1 < 1) * (l2 < 1) * (l3 < 1)


def _is_ear(points, t):
    convex = _cross2(t[:2, 1] - t[:2, 0], t[:2, 2] - t[:2, 1]) > 0      # check_convex :10-19
    return bool(convex) and not bool(np.any(_inside_triangle(points, t)))


def ear_clip(vs):
    """triangulate :81-105 -> list of (3, 3) vertex triples (columns = vertices)."""
    points = np.array(vs, dtype=np.float64)
    out = []
    i = 0
    while points.shape[1] > 3:
        n = points.shape[1]
        if i >= n:
            raise ValueError("polygon cannot be ear-clipped (no ear found)")
        ix = [i - 1, i, (i + 1) % n]
        tri = points[:, ix]
        if _is_ear(points, tri):
            out.append(tri.copy())
            points = np.delete(points, i, axis=1)
            i = 0
        else:
            i += 1
    out.append(points.copy())
    return out


def _self_intersects(vs):
    m = vs.shape[1]
    p = vs[:2]
    for i in range(m):
        a0, a1 = p[:, i], p[:, (i + 1) % m]
        for j in range(i + 1, m):
            if j == i or (j + 1) % m == i or (i + 1) % m == j:
                continue
            b0, b1 = p[:, j], p[:, (j + 1) % m]
            den = _cross2(a1 - a0, b1 - b0)
            if den == 0:
                continue
            t1 = _cross2(b0 - a0, b1 - b0) / den
            t2 = _cross2(b0 - a0, a1 - a0) / den
            if 0 < t1 < 1 and 0 < t2 < 1:
                return True
    return False


def _half_planes(points):
    """interior_convex :355-387: per edge k the pair (points[:, k], inward-flipped edge normal)."""
    m = points.shape[1]
    zero = np.average(points, axis=1)
    rows = []
    for i in range(m):
        k = (i + 1) % m
        ci = points[:, i] / 2 + points[:, k] / 2 - zero
        vi = points[:, k] - points[:, i]
        with np.errstate(divide="ignore", invalid="ignore"):
            vin = vi / np.linalg.norm(vi)
        ni = vin.copy()
        ni[0] = -vin[1]
        ni[1] = vin[0]
        ni = ni * (1 - 2 * (ni.dot(ci) < 0))
        if ni.size > 2 and ni[2] != 0:
            raise NotImplementedError("polygon vertices must share one z value (planar outline)")
        rows.append((points[0, i], points[1, i], ni[0], ni[1]))
    return rows


def convex_pieces(points):
    """Convex pieces whose union is the polygon interior, chosen exactly as interior_polygon does."""
    pts = np.array(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[0] < 2 or pts.shape[1] < 3:
        raise ValueError("polygon vertices must have shape (3, M) with M >= 3; got %r" % (pts.shape,))
    if pts.shape[0] == 2:
        pts = np.concatenate([pts, np.zeros((1, pts.shape[1]))])
    conv = _convexity(pts)
    if np.all(conv >= 0):
        return [_half_planes(pts)]
    if np.all(conv <= 0):
        return [_half_planes(pts[:, ::-1])]
    if _self_intersects(pts):
        raise NotImplementedError("self-intersecting polygon outlines are not supported")
    if np.count_nonzero(conv >= 0) < pts.shape[0] // 2:
        pts = pts[:, ::-1]
    return [_half_planes(t) for t in ear_clip(pts)]


def emit_polygon_sign(L, vdst, creg, points):
    pieces = convex_pieces(points)
    flat = []
    for rows in pieces:
        flat.append(float(len(rows)))
        for r in rows:
            flat.extend(r)
    off = L.add_table(flat)
    L.emit("P_POLYSIGN", vdst, creg, params=[len(pieces), off])
