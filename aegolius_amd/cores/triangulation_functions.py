"""Polygon helpers by their reference names (reference cores/triangulation_functions.py).

The functions that look at a polygon's VERTICES (convexity, ear clipping, crossings of the outline with itself)
are host code on a handful of points, as in the reference — minus its debug prints. The three `interior_*`
functions classify every point of a cloud and run on the GPU (csrc/sdfk_device.h prim_polysign), like
`sdf_polygon_2d` does inside a tree.
"""
import numpy as np

from .. import _polygon


def _cross_z(a, b):
    return a[..., 0] * b[..., 1] - a[..., 1] * b[..., 0]


def check_convex(v3s):
    """True if the three vertices (columns of a (D >= 2, 3) array) turn left (:10-19)."""
    v3s = np.asarray(v3s)
    return _cross_z(v3s[:2, 1] - v3s[:2, 0], v3s[:2, 2] - v3s[:2, 1]) > 0


def check_convex_all(vs):
    """Cross products of consecutive edges of a polygon; the turn at the last vertex is not tested, the one at the
    first vertex comes last (:22-37)."""
    return _polygon._convexity(np.asarray(vs, dtype=np.float64))


def is_inside_triangle(vs, v3s):
    """Barycentric test of the points `vs` against the triangle `v3s`; its own vertices do not count (:40-60)."""
    return _polygon._inside_triangle(np.asarray(vs, dtype=np.float64), np.asarray(v3s, dtype=np.float64))


def is_ear(vs, v3s):
    """A convex corner that holds no other vertex of the polygon (:63-78)."""
    return _polygon._is_ear(np.asarray(vs, dtype=np.float64), np.asarray(v3s, dtype=np.float64))


def triangulate(vs):
    """Ear clipping: (3, 3, N - 2) array of triangles (:81-105)."""
    tris = _polygon.ear_clip(np.asarray(vs, dtype=np.float64))
    out = np.zeros((3, 3, len(tris)))
    for i, t in enumerate(tris):
        out[:, :, i] = t
    return out


def check_intersection(v1, v2, v3, v4):
    """Do the segments v1 v2 and v3 v4 cross properly, and where (:108-125)."""
    v1, v2, v3, v4 = (np.asarray(v, dtype=np.float64) for v in (v1, v2, v3, v4))
    with np.errstate(divide="ignore", invalid="ignore"):
        b = (v2[0] - v1[0]) * (v4[1] - v3[1]) - (v2[1] - v1[1]) * (v4[0] - v3[0])
        t1 = ((v3[0] - v1[0]) * (v4[1] - v3[1]) - (v3[1] - v1[1]) * (v4[0] - v3[0])) / b
        t2 = ((v3[0] - v1[0]) * (v2[1] - v1[1]) - (v3[1] - v1[1]) * (v2[0] - v1[0])) / b
    return (t1 > 0) * (t1 < 1) * (t2 > 0) * (t2 < 1), np.asarray((v1[0] + (v2[0] - v1[0]) * t1, v1[1] + (v2[1] - v1[1]) * t1))


def check_intersection_all(vs):
    """Edges of the outline that cross later edges: (leading edges, crossed edges, crossing points) (:128-183)."""
    return _polygon.segment_crossings(np.asarray(vs, dtype=np.float64))


def create_points_sets(vs, idata):
    """The loops a self-intersecting outline falls into when cut at its crossing points (:186-302)."""
    return _polygon.split_at_crossings(np.asarray(vs, dtype=np.float64), idata)


def _classify(co, table, count):
    from .. import _eval
    from .._lower import Lowerer
    L = Lowerer()
    v = L.new_v()
    L.emit("P_POLYSIGN", v, 0, params=[count, L.add_table(table)])
    co = np.asarray(co)
    if co.ndim == 2 and co.shape[0] == 2:                       # the reference takes (D >= 2, N) clouds
        co = np.concatenate([co, np.zeros((1, co.shape[1]), dtype=co.dtype)])
    return _eval._run(L.finish(v), co)


def interior_convex(co, points):
    """Max over the edges of sign(dot(p - p_k, inward-flipped normal_k)): -1 inside the convex polygon, 0 on an edge
    line, +1 outside (:355-387)."""
    rows = _polygon._half_planes(_polygon._as_vertices(points))
    return _classify(co, [float(len(rows))] + [x for r in rows for x in r], -1.0)


def interior_triangle(co, points):
    """The same for a triangle (:305-352)."""
    return interior_convex(co, np.asarray(points)[:, :3])


def interior_polygon(co, points):
    """-1 inside the polygon, +1 outside: convex outlines by their half planes, concave ones by ear clipping,
    self-intersecting ones as the union of their loops (:390-430). Like the reference it turns a clockwise convex
    outline around IN PLACE."""
    pts = _polygon._as_vertices(points)
    conv = _polygon._convexity(pts)
    pieces = _polygon.convex_pieces(pts)
    if not np.all(conv >= 0) and np.all(conv <= 0) and isinstance(points, np.ndarray):
        points[:, :] = points[:, ::-1]
    flat = []
    for rows in pieces:
        flat.append(float(len(rows)))
        for r in rows:
            flat.extend(r)
    return _classify(co, flat, float(len(pieces)))
