"""Host side of `interior_polygon` (reference cores/triangulation_functions.py:390-430).

The reference decides, per polygon, between three evaluators: the polygon itself when it is convex
(:394-401), its ear-clipped triangles when it is concave and simple (:423-428), and — for
self-intersecting outlines (:403-421) — the loops the outline falls into when cut at its crossing
points, each evaluated recursively. All three reduce to "is the point inside ANY of these convex
pieces", a pure per-point test that the device evaluates (prim_polysign); this module produces the
pieces (topology only — no per-point work).
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
        if i >= n:                                               # no ear left: the reference's index runs off the vertex list
            raise IndexError("index %d is out of bounds for axis 1 with size %d (the outline cannot be ear-clipped)" % (i, n))
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


def segment_crossings(vs):
    """check_intersection_all :128-183 — for every edge c (but the closing one) the later edges it properly crosses:
    (leading edge indices, per leading edge the start indices of the crossed edges, per leading edge the (2, k)
    crossing points). Exactly parallel candidates are dropped before the test, touching end points do not count."""
    m = vs.shape[1]
    loop = np.concatenate([np.arange(m), [0]])
    p = np.asarray(vs, dtype=np.float64)[:2][:, loop]
    lead, crossed, where = [], [], []
    for c in range(m - 1):
        starts = np.arange(c + 1, m)
        x, y = p[:, starts], p[:, starts + 1]
        with np.errstate(divide="ignore", invalid="ignore"):
            v = (y - x) / np.linalg.norm(y - x, axis=0)
            u = (p[:, c + 1] - p[:, c]) / np.linalg.norm(p[:, c + 1] - p[:, c])
            keep = np.abs(v[0] * u[0] + v[1] * u[1]) != 1
        x, y, starts = x[:, keep], y[:, keep], starts[keep]
        if starts.size == 0:
            continue
        a0, a1 = p[:, c], p[:, c + 1]
        with np.errstate(divide="ignore", invalid="ignore"):            # check_intersection :108-125
            den = (a1[0] - a0[0]) * (y[1] - x[1]) - (a1[1] - a0[1]) * (y[0] - x[0])
            t1 = ((x[0] - a0[0]) * (y[1] - x[1]) - (x[1] - a0[1]) * (y[0] - x[0])) / den
            t2 = ((x[0] - a0[0]) * (a1[1] - a0[1]) - (x[1] - a0[1]) * (a1[0] - a0[0])) / den
        hit = (t1 > 0) & (t1 < 1) & (t2 > 0) & (t2 < 1)
        if np.any(hit):
            lead.append(c)
            crossed.append(loop[starts][hit])
            where.append(np.stack([a0[0] + (a1[0] - a0[0]) * t1, a0[1] + (a1[1] - a0[1]) * t1])[:, hit])
    return lead, crossed, where


def split_at_crossings(vs, crossings):
    """create_points_sets :186-302 — the loops a self-intersecting outline falls into when it is cut at its crossing
    points: list of (3, k) vertex arrays (original vertices and crossing points, z = 0). The grouping rules are the
    reference's own (they decide which input outlines are accepted at all): an outline whose edges cross different
    numbers of other edges raises the ValueError NumPy raises there for the ragged index array, and only the first
    `crossings + 1` loops are used."""
    lead, crossed_list, where_list = crossings
    m = vs.shape[1]
    crossed = np.asarray(crossed_list)                       # ragged -> ValueError, as in the reference
    where = np.asarray(where_list)
    n_x = crossed.size
    xid = [m + i for i in range(n_x)]                        # vertex numbers of the crossing points
    all_crossed = np.concatenate(crossed, axis=0)
    groups, c = [], 0
    for i, first in enumerate(lead):
        for other in crossed[i]:
            groups.append([first, xid[c], int((other + 1) % m)])
            groups.append([int(other), xid[c], int((first + 1) % m)])
            c += 1
    i = 0
    for _ in range(len(groups)):
        grp = [int(q) for q in np.ravel(groups[i])]
        merged_with = None
        for g, oth in enumerate(groups):
            if g == i:
                continue
            if grp[0] == oth[-1] and grp[-1] == oth[0]:      # two half loops that close each other
                grp = grp + [int(q) for q in oth[1:-1]]
                merged_with = g
                break
            plain = len(grp) == 3 and len(oth) == 3
            consecutive = grp[1] - oth[1] == -1
            next_is_end = (grp[0] + 1) % m == oth[2]
            forward = (grp[0] < grp[2]) or (oth[0] < oth[2])
            if consecutive and next_is_end and plain and forward:
                grp = [oth[0], oth[1], grp[1], grp[-1]]
                if grp[0] == grp[-1]:
                    grp = grp[:-1]
                merged_with = g
                break
            wraps = grp[1] == xid[0] and oth[1] == xid[-1]
            if (wraps and (oth[0] + 1) % m == grp[2] and plain and grp[0] in lead and grp[2] not in xid
                    and grp[0] not in xid and grp[1] != oth[1]):
                grp = [grp[0], grp[1], oth[1], oth[2]]
                if grp[0] == grp[-1]:
                    grp = grp[:-1]
                merged_with = g
                break
        if merged_with is not None:
            groups[i] = grp
            groups = [q for k, q in enumerate(groups) if k != merged_with]
            continue
        for _j in range(m):                                   # walk along the outline until the loop closes
            if grp[-1] not in all_crossed and grp[-1] not in lead:
                if (grp[-1] + 1) % m == grp[0]:
                    break
                if grp[-1] < m - 1:
                    grp.append(grp[-1] + 1)
            if grp[0] not in all_crossed and grp[0] - 1 not in lead:
                if (grp[-1] + 1) % m == grp[0]:
                    break
                if 0 < grp[0] < m:
                    grp.append(grp[0] - 1)
        groups[i] = grp
        i += 1
        if i == len(groups):
            break
    pts = np.concatenate([np.asarray(vs, dtype=np.float64)[:2], np.moveaxis(where, 1, 0).reshape(2, n_x)], axis=1)
    pts = np.concatenate([pts, np.zeros((1, pts.shape[1]))], axis=0)
    return [pts[:, groups[k]] for k in range(n_x + 1)]


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


def _as_vertices(points):
    pts = np.array(points, dtype=np.float64)
    if pts.ndim != 2 or pts.shape[0] < 2 or pts.shape[1] < 3:
        raise ValueError("polygon vertices must have shape (3, M) with M >= 3; got %r" % (pts.shape,))
    if pts.shape[0] == 2:
        pts = np.concatenate([pts, np.zeros((1, pts.shape[1]))])
    return pts


def convex_pieces(points):
    """Convex pieces whose union is the polygon interior, chosen exactly as interior_polygon does."""
    pts = _as_vertices(points)
    conv = _convexity(pts)
    if np.all(conv >= 0):
        return [_half_planes(pts)]
    if np.all(conv <= 0):
        return [_half_planes(pts[:, ::-1])]
    crossings = segment_crossings(pts)
    if crossings[0]:                                         # self-intersecting outline: the union of its loops (:403-412)
        pieces = []
        for loop in split_at_crossings(pts, crossings):
            pieces.extend(convex_pieces(loop))
        return pieces
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
