"""CPU oracle for the SDF grid-evaluation path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the reference's algorithm (peterropac/Aegolius = SPOMSO 1.4.0,
`Code/spomso/spomso/cores/`, abbreviated C/ below), walking the same symbolic tree that
`aegolius_amd` records but never touching its lowering, its bytecode or libsdfk.so. It exists so that
the HIP kernels can be checked on the GPU box, where the reference cannot travel.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Pinning: tests/test_oracle_golden.py checks every function here against golden vectors produced by
the REAL reference in the build container (tests/golden/generate_golden.py imports it from
/root/reference; fixtures under tests/golden/*.npz) to <= 1e-12.

Semantics mirrored on purpose:
  * arrays are passed between "closures" by reference and `symmetry`, `rotational_symmetry`,
    `axis_revolution` overwrite their input in place (C/modifications.py:951, 1022-1028, 459), which
    later siblings inside displacement / recover_volume / define_volume observe;
  * the Euclidean transform always builds fresh arrays (C/transformations.py:238-240);
  * np.mod / np.sign / comparison conventions of NumPy.
"""
import numpy as np

from aegolius_amd._ir import CombineSDF, ModSDF, NodeSDF, PrimSDF, UnsupportedSDF

norm = np.linalg.norm


# =================================================================================================
# tree walk
# =================================================================================================
def evaluate(node, co):
    """node.create(co) of the reference: float64 field of shape (N,). `co` is not modified."""
    co = np.asarray(co, dtype=np.float64)
    return eval_node(node, co)


_MAGNITUDE = None      # per-point running max of |intermediate|, only inside evaluate_with_magnitude


def _note(values):
    if _MAGNITUDE is not None:
        v = np.abs(np.asarray(values, dtype=np.float64))
        if v.ndim == 2:
            v = v.max(axis=0)
        if v.shape == _MAGNITUDE.shape:
            np.maximum(_MAGNITUDE, np.where(np.isfinite(v), v, 0.0), out=_MAGNITUDE)


def evaluate_with_magnitude(node, co):
    """(field, magnitude): `magnitude[i]` is the largest absolute value among the node-local coordinates
    and the node / operand fields met while evaluating point i — the scale against which a rounding
    error of an fp32 evaluation of the same tree has to be judged when the tree adds and subtracts
    fields (SUM, DIFFERENCE, displacement), where the result can be far smaller than its operands."""
    global _MAGNITUDE
    co = np.asarray(co, dtype=np.float64)
    _MAGNITUDE = np.zeros(co.shape[1])
    try:
        field = eval_node(node, co)
        return field, _MAGNITUDE
    finally:
        _MAGNITUDE = None


def eval_node(node, co):
    # C/transformations.py:232-242
    rm = np.asarray(node.rotation_matrix, dtype=np.float64).T
    sm = node.scale
    c = rm.dot(co)
    c = c / sm
    c = np.subtract(c.T, rm.dot(np.asarray(node.center, dtype=np.float64))).T
    _note(c)
    out = sm * eval_expr(node.modified_object, c, node._geo_parameters)
    _note(out)
    return out


def eval_expr(expr, co, params):
    if isinstance(expr, PrimSDF):
        return PRIMS[expr.name](co, *params)
    if isinstance(expr, ModSDF):
        return MODS[expr.name](expr, co, params)
    if isinstance(expr, CombineSDF):
        return _combine(expr, co)
    if isinstance(expr, NodeSDF):
        return eval_node(expr.obj, co)
    if isinstance(expr, UnsupportedSDF):            # an opaque user callable used as SDF function: just call it
        return expr.fn(co, *params)
    raise TypeError("oracle: cannot evaluate %r" % (expr,))


def _call(fn, co, params):
    """A user-facing callable: one of our expressions or `obj.propagate`."""
    from aegolius_amd._lower import as_expr
    return eval_expr(as_expr(fn), co, tuple(params))


# =================================================================================================
# combiners  C/combine.py:12-78
# =================================================================================================
def smin_poly(x, y, a, power):
    if a == 0.0:
        return np.minimum(x, y)
    h = np.maximum(a - np.abs(x - y), 0.0) / a
    if power == 2:
        return np.minimum(x, y) - h * h * a / 4.0
    return np.minimum(x, y) - h * h * h * a / 6.0


def smax_boltz(x, y, a):
    e1, e2 = np.exp(x / a), np.exp(y / a)
    return (x * e1 + y * e2) / (e1 + e2)


PLAIN = {
    "UNION2": lambda a, b: np.minimum(a, b),
    "UNION": lambda *f: np.amin(f, axis=0),
    "SUBTRACT2": lambda a, b: np.maximum(a, -b),
    "INTERSECT2": lambda a, b: np.maximum(a, b),
    "INTERSECT": lambda *f: np.amax(f, axis=0),
    "SUM": lambda a, b: a + b,
    "DIFFERENCE": lambda a, b: a - b,
}
PARAMETRIC = {
    "SMOOTH_UNION2_2": lambda a, b, w: smin_poly(a, b, w, 2),
    "SMOOTH_UNION2": lambda a, b, w: smin_poly(a, b, w, 3),
    "SMOOTH_INTERSECT2": lambda a, b, w: -smin_poly(-a, -b, w, 3),
    "SMOOTH_INTERSECT2_BOLTZMANN": lambda a, b, w: smax_boltz(a, b, w),
    "SMOOTH_SUBTRACT2": lambda a, b, w: -smin_poly(-a, b, w, 3),
    "SMOOTH_SUBTRACT2_BOLTZMANN": lambda a, b, w: smax_boltz(a, -b, w),
}


def _combine(expr, co):
    fields = [eval_node(kid, co) for kid in expr.children]      # C/combine.py:130-133
    op = expr.owner.operation_type
    if expr.parametric:
        return PARAMETRIC[op](*fields, expr.parameters)
    return PLAIN[op](*fields)


# =================================================================================================
# primitives  C/sdf_3D.py, C/sdf_2D.py
# =================================================================================================
def _rot2(angle):
    return np.asarray([[np.cos(angle), np.sin(angle)], [-np.sin(angle), np.cos(angle)]])


def p_sphere(co, radius):                       # sdf_3D.py:25-27
    return norm(co, axis=0) - radius


def p_cylinder(co, radius, height):             # :30-37
    d0 = norm(co[:2], axis=0) - radius
    d1 = np.abs(co[2]) - height / 2
    return np.minimum(np.maximum(d0, d1), 0) + norm([np.maximum(d0, 0), np.maximum(d1, 0)], axis=0)


def p_box(co, size):                            # :40-47
    q = np.abs(co).T - np.asarray(size) / 2
    return norm(np.maximum(q, 0.0), axis=1) + np.minimum(np.maximum(q[:, 0], np.maximum(q[:, 1], q[:, 2])), 0.0)


def p_torus(co, R, r):                          # :50-53
    return norm([norm(co[:2], axis=0) - R, co[2]], axis=0) - r


def p_chainlink(co, R, r, length):              # :56-61
    x = co[0] - np.clip(co[0], -length / 2, length / 2)
    return norm([norm([x, co[1]], axis=0) - R, co[2]], axis=0) - r


def p_braid(co, length, R, r, pitch):           # :64-75
    c, s = np.cos(pitch * co[2]), np.sin(pitch * co[2])
    x = c * co[0] - s * co[1]
    y = s * co[0] + c * co[1]
    z = co[2] - np.clip(co[2], -length / 2, length / 2)
    return norm([norm([x, z], axis=0) - R, y], axis=0) - r


def _arc_fold(co2, start_angle, end_angle):
    mid = (start_angle + end_angle) / 2
    xy = _rot2(mid).dot(co2)
    xy[1] = np.abs(xy[1])
    psi = np.clip(np.arctan2(xy[1], xy[0]), 0, np.abs(end_angle - mid))
    return xy, psi


def p_arc3d(co, R, r, start_angle, end_angle):  # :78-96
    xy, psi = _arc_fold(co[:2], start_angle, end_angle)
    d = xy - np.asarray((R * np.cos(psi), R * np.sin(psi)))
    return norm([d[0], d[1], co[2]], axis=0) - r


def p_plane(co, normal, offset):                # :99-102
    n = np.asarray(normal) / norm(normal)
    return np.dot(co.T, n) - offset


def p_uplane(co, normal, thickness):            # :105-108
    n = np.asarray(normal) / norm(normal)
    return np.abs(np.dot(co.T, n)) - thickness / 2


def _segment(co, a, b):                         # :111-118 / sdf_2D.py:31-38
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    pa = (co.T - a).T
    ba = b - a
    h = np.clip(np.sum((pa.T * ba).T, axis=0) / np.dot(ba, ba), 0, 1)
    return norm(pa - np.outer(ba, h), axis=0)


def p_segment3(co, a, b):
    return _segment(co, a, b)


def p_cone(co, height, angle):                  # :121-136
    q = np.asarray((np.tan(angle), -1)) * height
    z = co[2] - height * (0.5 ** (1 / 3))
    w = np.asarray((norm(co[:2], axis=0), z))
    a = w - np.outer(q, np.clip(np.dot(w.T, q) / np.dot(q, q), 0.0, 1.0))
    b = w.T - q * np.asarray((np.clip(w[0] / q[0], 0.0, 1.0), np.ones(w.shape[1]))).T
    d = np.minimum(np.sum(a * a, axis=0), np.sum(b * b, axis=1))
    s = np.maximum(-(w[0] * q[1] - w[1] * q[0]), -(w[1] - q[1]))
    return np.sqrt(d) * np.sign(s)


def _infcone(co, angle):
    v = np.asarray([np.sin(angle), np.cos(angle)])
    q = np.asarray((norm(co[:2], axis=0), -co[2]))
    d = norm(q - np.outer(v, np.maximum(np.dot(q.T, v), 0.0)), axis=0)
    return d, q, v


def p_infcone(co, angle):                       # :151-157
    return _infcone(co, angle)[0]


def p_oriented_infcone(co, angle):              # :139-148
    d, q, v = _infcone(co, angle)
    return d * (-2 * (q[0] * v[1] - q[1] * v[0] < 0.0) + 1)


def _sector_tail(xy, radius, half):             # shared by :168-183 and sdf_2D.py:114-129
    phi = np.arctan2(xy[1], xy[0])
    psi = np.clip(phi, 0, half)
    length = norm(xy - np.asarray((radius * np.cos(psi), radius * np.sin(psi))), axis=0)
    c = np.asarray([np.cos(half), np.sin(half)])
    m = norm(xy - np.outer(c, np.clip(np.dot(xy.T, c), 0, radius)), axis=0)
    out = np.minimum(m, length)
    inside = (norm(xy, axis=0) <= radius) * (phi <= half)
    out[inside] = -out[inside]
    return out


def p_solidangle(co, radius, angle_1, angle_2):  # :160-183
    xy = _rot2((angle_2 + angle_1) / 2).dot(co[:2])
    xy[1] = norm([xy[1], co[2]], axis=0)
    return _sector_tail(xy, radius, np.abs(angle_2 - angle_1) / 2)


def _flat_polygon(co, verts, threshold):        # :186-250
    verts = [np.asarray(v, dtype=float) for v in verts]
    k = len(verts)
    edges = [verts[(i + 1) % k] - verts[i] for i in range(k)]
    rel = [(co.T - v).T for v in verts]
    normal = np.cross(edges[0], edges[-1])
    side = sum(np.sign(np.dot(np.cross(e, normal), r)) for e, r in zip(edges, rel))
    near_edge = side < threshold
    out = np.zeros(co.shape[1])
    best = None
    for e, r in zip(edges, rel):
        rr = r[:, near_edge]
        t = np.outer(e, np.clip(np.dot(e, rr) / np.dot(e, e), 0, 1)) - rr
        dd = np.sum(t * t, axis=0)
        best = dd if best is None else np.minimum(best, dd)
    out[near_edge] = np.sqrt(best)
    out[~near_edge] = np.sqrt(np.dot(normal, rel[0][:, ~near_edge]) ** 2 / np.dot(normal, normal))
    return out


def p_triangle3(co, a, b, c):
    return _flat_polygon(co, (a, b, c), 2.0)


def p_quad3(co, a, b, c, d):
    return _flat_polygon(co, (a, b, c, d), 3.0)


def _nearest(co, pts):
    """Exact nearest-neighbour distance (what scipy.spatial.KDTree.query returns), in blocks."""
    pts = np.asarray(pts, dtype=float)
    out = np.empty(co.shape[1])
    step = max(1, (1 << 22) // max(pts.shape[1], 1))
    for s in range(0, co.shape[1], step):
        d = co[:, s:s + step, None] - pts[:, None, :]
        out[s:s + step] = np.sqrt(np.min(np.sum(d * d, axis=0), axis=1))
    return out


def _resample(points, t):                       # :255-257
    v = np.floor(t).astype(int)
    u = t - v
    return points[:, v + 1] * u + points[:, v] * (1 - u)


def p_segcurve3(co, points, t):                 # :253-261
    return _nearest(co[:3], _resample(np.asarray(points, dtype=float)[:3], t))


def p_segline3(co, points):                     # :264-271
    points = np.asarray(points, dtype=float)
    out = np.ones(co.shape[1]) * 1e16
    for i in range(points.shape[1] - 1):
        out = np.minimum(out, _segment(co, points[:3, i], points[:3, i + 1]))
    return out


def p_paramcurve3(co, f, fp, t):                # :274-280
    return _nearest(co[:3], np.asarray(f(t, *fp), dtype=float))


def p_cloud3(co, points):                       # :283-286
    return _nearest(co[:3], np.asarray(points, dtype=float)[:3])


# ---- 2-D ----
def p_circle(co, radius):                       # sdf_2D.py:12-14
    return norm(co[:2], axis=0) - radius


def p_neucircle(co, radius, order):             # :17-19
    return norm(co[:2], axis=0, ord=order) - radius


def p_box2(co, size):                           # :22-28
    d = (np.abs(co[:2]).T - np.asarray(size) / 2).T
    return norm(np.maximum(d, 0), axis=0) + np.minimum(np.maximum(d[0], d[1]), 0)


def p_segment2(co, a, b):
    return _segment(co[:2], np.asarray(a)[:2], np.asarray(b)[:2])


def p_rbox2(co, size, rounding):                # :41-57
    r = rounding[0] * np.ones(co.shape[1])
    r[co[0] > 0] = rounding[1]
    r[co[1] > 0] = rounding[2]
    r[(co[0] < 0) * (co[1] > 0)] = rounding[3]
    d = (np.abs(co[:2]).T - np.asarray(size) / 2).T + r
    return norm(np.maximum(d, 0), axis=0) + np.minimum(np.maximum(d[0], d[1]), 0) - r


def p_triangle2(co, p0, p1, p2):                # :60-82
    p = [np.asarray(x, dtype=float)[:2] for x in (p0, p1, p2)]
    e = [p[1] - p[0], p[2] - p[1], p[0] - p[2]]
    s = np.sign(e[0][0] * e[2][1] - e[0][1] * e[2][0])
    dist, cross = [], []
    for pi, ei in zip(p, e):
        v = (co[:2].T - pi).T
        pq = v - np.outer(ei, np.clip(np.dot(v.T, ei) / np.dot(ei, ei), 0, 1))
        dist.append(np.sum(pq * pq, axis=0))
        cross.append(s * (v[0] * ei[1] - v[1] * ei[0]))
    return -np.sqrt(np.amin(dist, axis=0)) * np.sign(np.amin(cross, axis=0))


def p_arc2(co, radius, start_angle, end_angle):  # :85-103
    xy, psi = _arc_fold(co[:2], start_angle, end_angle)
    return norm(xy - np.asarray((radius * np.cos(psi), radius * np.sin(psi))), axis=0)


def p_sector(co, radius, angle_1, angle_2):     # :105-129
    xy = _rot2((angle_2 + angle_1) / 2).dot(co[:2])
    xy[1] = np.abs(xy[1])
    return _sector_tail(xy, radius, np.abs(angle_2 - angle_1) / 2)


def p_infsector(co, angle_1, angle_2):          # :132-150
    half = np.abs(angle_2 - angle_1) / 2
    xy = _rot2((angle_2 + angle_1) / 2).dot(co[:2])
    xy[1] = np.abs(xy[1])
    phi = np.arctan2(xy[1], xy[0])
    c = np.asarray([np.cos(half), np.sin(half)])
    m = norm(xy - np.outer(c, np.clip(np.dot(xy.T, c), 0, np.inf)), axis=0)
    return np.sign(phi - half) * m


def p_ngon(co, radius, n):                      # :153-177
    beta, alpha = np.pi * (0.5 - 1 / n), 2 * np.pi / n
    phi = np.arctan2(co[1], co[0])
    phi[phi < 0] = 2 * np.pi + phi[phi < 0]
    phi = np.mod(phi, alpha)
    xy = np.asarray([np.cos(phi), np.sin(phi)]) * norm(co[:2], axis=0)
    q = (xy.T - [radius, 0]).T
    t = np.asarray([-np.cos(beta), np.sin(beta)])
    no = np.asarray([np.sin(beta), np.cos(beta)])
    h = np.clip(np.dot(q.T, t), 0, 2 * radius * np.sin(alpha / 2))
    return norm(q - np.outer(t, h), axis=0) * np.sign(np.dot(q.T, no))


def p_segcurve2(co, points, t):                 # :180-188
    return _nearest(co[:2], _resample(np.asarray(points, dtype=float)[:2], t))


def p_segline2(co, points):                     # :191-198
    points = np.asarray(points, dtype=float)
    out = np.ones(co.shape[1]) * 1e16
    for i in range(points.shape[1] - 1):
        out = np.minimum(out, p_segment2(co, points[:, i], points[:, i + 1]))
    return out


def p_paramcurve2(co, f, fp, t):                # :214-218
    return _nearest(co[:2], np.asarray(f(t, *fp), dtype=float))


def p_cloud2(co, points):                       # :221-224
    return _nearest(co[:2], np.asarray(points, dtype=float)[:2])


# ---- polygon interior  C/triangulation_functions.py ----
def _cross2(a, b):
    return a[0] * b[1] - a[1] * b[0]


def interior_convex(co, points):                # :355-387
    # Two passes like the reference — all normals first, as COLUMNS of a copy of `points` (:370-381), then one dot
    # product per edge with the strided column (:383-385): a point exactly on an edge (grid points on the line y = 2x of
    # the hourglass example) gets its sign from the last bit of that product, and BLAS sums a strided operand in a
    # different order than a contiguous one. With a contiguous normal 120 of 241,001 points of hourglass_2D came out
    # on the other side.
    m = points.shape[1]
    zero = np.average(points, axis=1)
    normals = points.copy()
    for i in range(m):
        k = (i + 1) % m
        ci = points[:, i] / 2 + points[:, k] / 2 - zero
        vi = points[:, k] - points[:, i]
        vin = vi / norm(vi)
        ni = vin.copy()
        ni[0] = -vin[1]
        ni[1] = vin[0]
        ni = ni * (1 - 2 * (ni.dot(ci) < 0))
        normals[:, i] = ni
    sp = -np.ones(co.shape[1])
    for k in range(m):
        sp[:] = np.maximum(sp, np.sign(np.dot(co.T - points[:, k], normals[:, k])))
    return sp


def _tri_contains(vs, t):                       # :40-60
    d = (t[1, 1] - t[1, 2]) * (t[0, 0] - t[0, 2]) + (t[0, 2] - t[0, 1]) * (t[1, 0] - t[1, 2])
    l1 = ((t[1, 1] - t[1, 2]) * (vs[0] - t[0, 2]) + (t[0, 2] - t[0, 1]) * (vs[1] - t[1, 2])) / d
    l2 = ((t[1, 2] - t[1, 0]) * (vs[0] - t[0, 2]) + (t[0, 0] - t[0, 2]) * (vs[1] - t[1, 2])) / d
    l3 = 1 - l1 - l2
    return (l1 * l2 * l3 >= 0) * (l1 < 1) * (l2 < 1) * (l3 < 1)


def triangulate(vs):                            # :81-105
    points = vs.copy()
    tris = []
    i = 0
    while points.shape[1] > 3:
        ix = [i - 1, i, (i + 1) % points.shape[1]]
        t = points[:, ix]
        ear = (_cross2(t[:2, 1] - t[:2, 0], t[:2, 2] - t[:2, 1]) > 0) and not np.any(_tri_contains(points, t))
        if ear:
            tris.append(t.copy())
            points = np.delete(points, i, axis=1)
            i = 0
        else:
            i += 1
    tris.append(points)
    return tris


def check_intersection_all(vs):                 # :128-183 (debug prints omitted)
    ixs = np.linspace(0, vs.shape[1] + 1, vs.shape[1] + 1, endpoint=False, dtype=int)
    ixs[-1] = 0
    mask = np.zeros(vs.shape[1] + 1, dtype=bool)
    vs = vs[:, ixs]
    six, eix, cs = [], [], []
    for c in range(vs.shape[1] - 2):
        m1, m2 = mask.copy(), mask.copy()
        m1[c + 1:-1] = True
        m2[c + 2:] = True
        x, y = vs[:2, m1], vs[:2, m2]
        v, u = y - x, vs[:2, c + 1] - vs[:2, c]
        v, u = v / norm(v, axis=0), u / norm(u)
        dot_mask = np.abs(v[0] * u[0] + v[1] * u[1]) == 1
        x, y = x[:, ~dot_mask], y[:, ~dot_mask]
        m1[m1] = ~dot_mask
        if x.size == 0 or y.size == 0:
            continue
        v1, v2 = vs[:2, c], vs[:2, c + 1]           # check_intersection :108-125
        b = (v2[0] - v1[0]) * (y[1] - x[1]) - (v2[1] - v1[1]) * (y[0] - x[0])
        t1 = ((x[0] - v1[0]) * (y[1] - x[1]) - (x[1] - v1[1]) * (y[0] - x[0])) / b
        t2 = ((x[0] - v1[0]) * (v2[1] - v1[1]) - (x[1] - v1[1]) * (v2[0] - v1[0])) / b
        hit = (t1 > 0) * (t1 < 1) * (t2 > 0) * (t2 < 1)
        if np.any(hit):
            six.append(c)
            eix.append(ixs[m1][hit])
            cs.append(np.asarray((v1[0] + (v2[0] - v1[0]) * t1, v1[1] + (v2[1] - v1[1]) * t1))[:, hit])
    return six, eix, cs


def create_points_sets(vs, idata):              # :186-302 (debug prints omitted)
    nv = vs.shape[1]
    leading_ixs = idata[0]
    cross_ixs = np.asarray(idata[1])            # ragged crossing counts: ValueError, as in the reference
    coors = np.asarray(idata[2])
    n_intersections = cross_ixs.size
    intersection_ixs = np.linspace(0, n_intersections, n_intersections, endpoint=False, dtype=int) + nv
    c_cross_ixs = np.concatenate(cross_ixs, axis=0)
    groups = []
    c = 0
    for i in range(len(leading_ixs)):
        for j in range(cross_ixs[i].size):
            groups.append([leading_ixs[i], int(intersection_ixs[c]), int(np.mod(cross_ixs[i][j] + 1, nv))])
            groups.append([int(cross_ixs[i][j]), int(intersection_ixs[c]), int(np.mod(leading_ixs[i] + 1, nv))])
            c += 1
    i = 0
    for _v in range(len(groups)):
        group_reformat = 0
        group = np.squeeze(groups[i]).tolist()
        for g in range(len(groups)):
            if g == i:
                continue
            if group[0] == groups[g][-1] and group[-1] == groups[g][0]:
                group = np.concatenate((group, groups[g][1:-1])).tolist()
                group_reformat = 1
                break
            c4 = group[1] - groups[g][1] == -1
            c5 = int(np.mod(group[0] + 1, nv)) == groups[g][2]
            c10 = (group[0] < group[2]) + (groups[g][0] < groups[g][2])
            c3 = (group[1] == intersection_ixs[0]) * (groups[g][1] == intersection_ixs[-1])
            c6 = int(np.mod(groups[g][0] + 1, nv)) == group[2]
            c7 = len(group) == 3 and len(groups[g]) == 3
            c8 = group[0] in leading_ixs
            c9 = (group[2] not in intersection_ixs) and (group[0] not in intersection_ixs)
            c11 = group[1] != groups[g][1]          # (`is not` on small ints in the reference)
            if c4 * c5 * c7 * c10:
                group = [groups[g][0], groups[g][1], group[1], group[-1]]
                if group[0] == group[-1]:
                    group = group[:-1]
                group_reformat = 1
                break
            if c3 * c6 * c7 * c8 * c9 * c11:
                group = [group[0], group[1], groups[g][1], groups[g][2]]
                if group[0] == group[-1]:
                    group = group[:-1]
                group_reformat = 1
                break
        if group_reformat:
            groups[i] = group
            groups = [groups[k] for k in range(len(groups)) if not k == g]
            continue
        for _j in range(nv):
            if (group[-1] not in c_cross_ixs) and (group[-1] not in leading_ixs):
                if np.mod(group[-1] + 1, nv) == group[0]:
                    break
                elif group[-1] >= nv - 1:
                    pass
                else:
                    group.append(group[-1] + 1)
            if (group[0] not in c_cross_ixs) and (group[0] - 1 not in leading_ixs):
                if np.mod(group[-1] + 1, nv) == group[0]:
                    break
                elif group[0] == 0 or group[0] >= nv:
                    pass
                else:
                    group.append(group[0] - 1)
        groups[i] = group
        i += 1
        if i == len(groups):
            break
    coors_c = np.reshape(np.moveaxis(coors, 1, 0), (2, n_intersections))
    evs = np.concatenate((vs[:2, :], coors_c), axis=1)
    evs = np.concatenate((evs, np.zeros((1, evs.shape[1]))), axis=0)
    return [evs[:, [int(q) for q in groups[i]]] for i in range(n_intersections + 1)]


def interior_polygon(co, points):               # :390-430
    points = np.array(points, dtype=float)
    if points.shape[0] == 2:
        points = np.concatenate([points, np.zeros((1, points.shape[1]))])
    m = points.shape[1]
    conv = [_cross2(points[:2, i] - points[:2, i - 1], points[:2, i + 1] - points[:2, i]) for i in range(1, m - 1)]
    conv.append(_cross2(points[:2, 0] - points[:2, -1], points[:2, 1] - points[:2, 0]))
    conv = np.asarray(conv)
    interior = np.ones(co.shape[1])
    if np.all(conv >= 0):
        interior[interior_convex(co, points) <= 0] = -1
    elif np.all(conv <= 0):
        interior[interior_convex(co, points[:, ::-1]) <= 0] = -1
    else:
        with np.errstate(all="ignore"):
            intersection_data = check_intersection_all(points)
        if intersection_data[0]:                # self-intersecting outline: union of its loops (:403-412)
            for new_points in create_points_sets(points, intersection_data):
                interior[interior_polygon(co, new_points) <= 0] = -1
            return interior
        if np.count_nonzero(conv >= 0) < points.shape[0] // 2:
            points = points[:, ::-1]
        for t in triangulate(points):
            interior[interior_convex(co, t) <= 0] = -1
    return interior


def p_polygon2(co, points):                     # sdf_2D.py:201-211
    points = np.asarray(points, dtype=float)
    m = points.shape[1]
    out = np.ones(co.shape[1]) * 1e16
    for i in range(m):
        out = np.minimum(out, p_segment2(co, points[:, i], points[:, (i + 1) % m]))
    return out * interior_polygon(co, points)


def _closed(open_fn, seg_fn, ends):             # C/geom_3d.py:610-618 etc.
    def fn(co, *args):
        p0, p1 = ends(*args)
        return np.minimum(open_fn(co, *args), seg_fn(co, p0, p1))
    return fn


_param_ends = lambda f, fp, t: (np.asarray(f(t[0], *fp)).ravel(), np.asarray(f(t[-1], *fp)).ravel())  # noqa: E731
_point_ends = lambda points, *rest: (np.asarray(points)[:, 0], np.asarray(points)[:, -1])              # noqa: E731

PRIMS = {
    "sdf_x": lambda co, o: co[0] - o, "sdf_y": lambda co, o: co[1] - o, "sdf_z": lambda co, o: co[2] - o,
    "sdf_sphere": p_sphere, "sdf_cylinder": p_cylinder, "sdf_box": p_box, "sdf_torus": p_torus,
    "sdf_chainlink": p_chainlink, "sdf_braid": p_braid, "sdf_arc_3d": p_arc3d, "sdf_plane": p_plane,
    "sudf_plane": p_uplane, "sdf_segment_3d": p_segment3, "sdf_cone": p_cone, "sdf_infinite_cone": p_infcone,
    "sdf_oriented_infinite_cone": p_oriented_infcone, "sdf_solid_angle": p_solidangle,
    "sdf_triangle_3d": p_triangle3, "sdf_quad_3d": p_quad3, "sdf_segmented_curve_3d": p_segcurve3,
    "sdf_segmented_line_3d": p_segline3, "sdf_parametric_curve_3d": p_paramcurve3, "sdf_point_cloud_3d": p_cloud3,
    "sdf_circle": p_circle, "sdf_neu_circle": p_neucircle, "sdf_box_2d": p_box2, "sdf_segment_2d": p_segment2,
    "sdf_rounded_box_2d": p_rbox2, "sdf_triangle_2d": p_triangle2, "sdf_arc": p_arc2, "sdf_sector": p_sector,
    "sdf_inf_sector": p_infsector, "sdf_ngon": p_ngon, "sdf_segmented_curve_2d": p_segcurve2,
    "sdf_segmented_line_2d": p_segline2, "sdf_polygon_2d": p_polygon2, "sdf_parametric_curve_2d": p_paramcurve2,
    "sdf_point_cloud_2d": p_cloud2,
    "closed_parametric_curve_3d": _closed(p_paramcurve3, p_segment3, _param_ends),
    "closed_segmented_curve_3d": _closed(p_segcurve3, p_segment3, _point_ends),
    "closed_line_curve_3d": _closed(p_segline3, p_segment3, _point_ends),
    "closed_parametric_curve_2d": _closed(p_paramcurve2, p_segment2, _param_ends),
    "closed_segmented_curve_2d": _closed(p_segcurve2, p_segment2, _point_ends),
    "closed_line_curve_2d": _closed(p_segline2, p_segment2, _point_ends),
}


# =================================================================================================
# modifications  C/modifications.py
# =================================================================================================
MODS = {}


def _m(name):
    def deco(fn):
        MODS[name] = fn
        return fn
    return deco


def _inner(e, co, params):
    return eval_expr(e.inner, co, params)


@_m("elongation")            # :88-93
def _(e, co, params):
    ev = e.args["ev"]
    return _inner(e, np.asarray([co[k] - np.clip(co[k], -ev[k] / 2, ev[k] / 2) for k in range(3)]), params)


@_m("rounding")              # :113-114
def _(e, co, params):
    return _inner(e, co, params) - e.args["rounding_radius"]


@_m("rounding_cs")           # :138-141
def _(e, co, params):
    scale = np.maximum(1 - 2 * e.args["rounding_radius"] / e.args["bb_size"] + 1e-8, 1e-8)
    return scale * _inner(e, co / scale, params) - e.args["rounding_radius"]


@_m("boundary")
def _(e, co, params):
    return np.abs(_inner(e, co, params))


@_m("invert")
def _(e, co, params):
    return -_inner(e, co, params)


@_m("sign")
def _(e, co, params):
    return np.sign(_inner(e, co, params))


@_m("recover_volume")        # :343
def _(e, co, params):
    return _inner(e, co, params) * _call(e.second, co, params)


@_m("define_volume")         # :366
def _(e, co, params):
    return _inner(e, co, params) * _call(e.second, co, e.second_params)


@_m("displacement")          # :798
def _(e, co, params):
    return _inner(e, co, params) + _call(e.second, co, e.second_params)


@_m("onion")
def _(e, co, params):
    return np.abs(_inner(e, co, params)) - e.args["thickness"]


@_m("concentric")
def _(e, co, params):
    return np.abs(_inner(e, co, params) - e.args["width"] / 2)


@_m("revolution")            # :426-431
def _(e, co, params):
    q = np.zeros(co.shape)
    q[0] = norm([co[0], co[2]], axis=0) - e.args["radius"]
    q[1] = co[1]
    return _inner(e, q, params)


@_m("axis_revolution")       # :456-466
def _(e, co, params):
    rot = _rot2(e.args["angle"])
    co[:2, :] = rot.dot(co[:2])                     # in place, visible to the caller
    q = np.zeros(co.shape)
    q[0] = norm([co[0], co[2]], axis=0)
    q[1] = co[1]
    q[:2] = rot.T.dot(q[:2])
    q[0] -= e.args["radius"]
    return _inner(e, q, params)


@_m("extrusion")             # :489-496
def _(e, co, params):
    q = co.copy()
    q[2, :] = 0
    w = np.asarray((_inner(e, q, params), np.abs(co[2]) - e.args["distance"] / 2))
    return np.minimum(np.maximum(w[0], w[1]), 0) + norm(np.maximum(w, 0), axis=0)


@_m("twist")                 # :517-522
def _(e, co, params):
    c, s = np.cos(e.args["pitch"] * co[2]), np.sin(e.args["pitch"] * co[2])
    q = co.copy()
    q[0] = c * co[0] - s * co[1]
    q[1] = s * co[0] + c * co[1]
    return _inner(e, q, params)


@_m("bend")                  # :546-571
def _(e, co, params):
    radius, angle = e.args["radius"], e.args["angle"]
    c, s = np.cos(angle / 2), np.sin(angle / 2)
    rot = np.asarray([[c, s], [-s, c]])
    q = co.copy()
    q[1] -= radius
    phi = np.arctan2(q[0], -q[1])
    q[1] = -radius + norm(q[:2], axis=0)
    q[0] = radius * phi
    far = radius * angle / 2 <= np.abs(q[0])
    right = co[0, far] >= 0
    w = co[:2][:, far].copy()
    sg = np.sign(co[0, far])
    w[0] -= radius * s * sg
    w[1] -= radius * (1 - c)
    w[:, right] = rot.dot(w[:, right])
    w[:, ~right] = rot.T.dot(w[:, ~right])
    w[0] += radius * (angle / 2) * sg
    q[:2, far] = w
    return _inner(e, q, params)


def _shear(e, co, params):
    from aegolius_amd._mods import _NAMED_SHEARS, _shear_matrix
    sa, fa = (e.args["sheared_axis"], e.args["fixed_axis"]) if e.name == "shear" else _NAMED_SHEARS[e.name]
    return _inner(e, _shear_matrix(sa, fa, np.tan(e.args["angle"])).dot(co), params)


for _n in ("shear", "shear_xz", "shear_yz", "shear_xy", "shear_zy", "shear_yx", "shear_zx"):
    MODS[_n] = _shear


@_m("infinite_repetition")   # :819-821
def _(e, co, params):
    d = np.asarray(e.args["distances"])
    return _inner(e, (np.mod(co.T + d / 2, d) - d / 2).T, params)


def _finite_cells(co, size, rep):               # :846-868
    with np.errstate(divide="ignore", invalid="ignore"):
        c = size * (1 - 1 / rep) / 2
        d = size * (1 / 2 - 1 / rep)
        s = size / rep
        inside = [(co[k] >= -d[k]) * (co[k] <= d[k]) for k in range(3)]
        v = np.abs(co).T - c
        for k in range(3):
            v[:, k] -= 2 * v[:, k] * (co[k] < 0)
        u = np.mod(co.T - d, s) - s / 2
    for k in range(3):
        v[inside[k], k] = u[inside[k], k]
    return v, s


@_m("finite_repetition")
def _(e, co, params):
    v, _s = _finite_cells(co, np.asarray(e.args["size"]), np.asarray(e.args["repetitions"]))
    return _inner(e, v.T, params)


@_m("finite_repetition_rescaled")  # :925-927
def _(e, co, params):
    v, s = _finite_cells(co, np.asarray(e.args["size"]), np.asarray(e.args["repetitions"]))
    k = np.min(s / (np.asarray(e.args["instance_size"]) + np.asarray(e.args["padding"])))
    return k * _inner(e, (v / k).T, params)


@_m("symmetry")              # :948-952
def _(e, co, params):
    axis = e.args["axis"]
    if axis > co.shape[0]:
        return _inner(e, co, params)
    co[axis, :] = np.abs(co[axis, :])               # in place
    return _inner(e, co, params)


def _frame(a, b):                               # :978-988
    w = b - a
    length = norm(w)
    x = w / length
    y = np.asarray([-x[1], x[0], 0])
    y = y / norm(y)
    return np.asarray([x, y, np.cross(x, y)]), (b + a) / 2, length


@_m("mirror")                # :976-994
def _(e, co, params):
    rot, c, length = _frame(np.asarray(e.args["a"]), np.asarray(e.args["b"]))
    p = rot.dot((co.T - c).T)
    v = np.zeros(p.shape)
    v[1:] = p[1:]
    v[0] = np.abs(p[0]) - length / 2
    return _inner(e, v, params)


@_m("rotational_symmetry")   # :1017-1029
def _(e, co, params):
    angle = 2 * np.pi / e.args["n"]
    co[:2] = _rot2(angle / 2 - e.args["phase"]).dot(co[:2])          # in place
    phi = np.arctan2(co[1], co[0])
    phi[phi < 0] = 2 * np.pi + phi[phi < 0]
    phi = np.mod(phi, angle) - angle / 2
    radii = norm(co[:2], axis=0)
    co[0] = radii * np.cos(phi) - e.args["radius"]
    co[1] = radii * np.sin(phi)
    return _inner(e, co, params)


@_m("linear_instancing")     # :1056-1084
def _(e, co, params):
    n = e.args["n"]
    rot, c, length = _frame(np.asarray(e.args["a"]), np.asarray(e.args["b"]))
    p = rot.dot((co.T - c).T)
    s = length / (n - 1)
    d = s / 2
    band = (p[0] >= -length / 2 + d) * (p[0] <= length / 2 - d)
    v = np.zeros(p.shape)
    v[1:] = p[1:]
    v[0] = np.abs(p[0]) - length / 2
    v[0] -= 2 * v[0] * (p[0] < 0)
    if n > 2:
        u0 = np.mod(p[0] - (length / 2 - d), s) - d
        v[0, band] = u0[band]
    return _inner(e, v, params)


def _curve_instancing(e, co, params):           # :1108-1263
    from aegolius_amd._mods import _curve_samples
    n, rows, frames = _curve_samples(e)         # host-side sampling of the user's curve (not per-point work)
    centres = rows[:, :3]
    d = co[:, :, None] - centres.T[:, None, :]
    idx = np.argmin(np.sum(d * d, axis=0), axis=1)
    w = co - centres[idx].T
    if frames:
        f = rows[:, 3:].reshape(n, 3, 3)[idx]   # (N, 3, 3) rows dx, dy, dz
        w = np.einsum("nij,jn->in", f, w)
    return _inner(e, w, params)


for _n in ("curve_instancing", "aligned_curve_instancing", "fully_aligned_curve_instancing"):
    MODS[_n] = _curve_instancing


@_m("move_sdf")              # :1283
def _(e, co, params):
    return _inner(e, (co.T - e.args["move_vector"]).T, params)


@_m("scale_sdf")             # :1302
def _(e, co, params):
    k = e.args["scale_factor"]
    return k * _inner(e, co / k, params)


@_m("rotate_sdf")            # :1323-1324
def _(e, co, params):
    return _inner(e, np.asarray(e.args["rotation_matrix"]).T.dot(co), params)


# post-processing  C/post_processing.py:380-558
def _post(fn):
    def wrapped(e, co, params):
        return fn(_inner(e, co, params), e.args)
    return wrapped


# u = the field, a = the keyword arguments of the function (the array-level functions of C/post_processing.py and the
# modification methods of the same names share these formulas)
def _slowstart(u, a):
    b = (2 * a["smooth_width"] + a["threshold"]) * a["threshold"]
    return np.sqrt(np.maximum(u / a["width"], 0) ** 2 + b / a["width"]) - np.sqrt(b / a["width"]) * a["ground"]


POST_FUNCTIONS = {
    "sigmoid_falloff": lambda u, a: a["amplitude"] * (1 / (1 + np.exp(4 * u / a["width"]))),
    "positive_sigmoid_falloff": lambda u, a: a["amplitude"] * (1 / (1 + np.exp(4 * (u - a["width"]) / a["width"]))),
    "capped_exponential": lambda u, a: a["amplitude"] * np.minimum(np.exp(-4 * u / a["width"]), 1),
    "hard_binarization": lambda u, a: (u <= a["threshold"]).astype(float),
    "linear_falloff": lambda u, a: np.clip(1 - u / a["width"], 0, 1) * a["amplitude"],
    "relu": lambda u, a: np.maximum(u / a["width"], 0),
    "smooth_relu": lambda u, a: (u / a["width"] + np.sqrt((u / a["width"]) ** 2
                                                          + (a["smooth_width"] + a["threshold"]) * 4 * a["threshold"])) / 2,
    "slowstart": _slowstart,
    "gaussian_boundary": lambda u, a: a["amplitude"] * np.exp(-4 * (u / a["width"]) ** 2),
    "gaussian_falloff": lambda u, a: a["amplitude"] * np.exp(-4 * (np.maximum(u, 0) / a["width"]) ** 2),
}
for _name, _fn in POST_FUNCTIONS.items():
    MODS[_name] = _post(_fn)


@_m("custom_modification")   # C/modifications.py:1353-1356
def _(e, co, params):
    return e.args["modification"](lambda co_, *p: eval_expr(e.inner, co_, p), co, params, e.args["modification_parameters"])


@_m("custom_post_process")   # C/modifications.py:1658-1660
def _(e, co, params):
    return e.args["function"](_inner(e, co, params), *e.args["parameters"])


# grid-neighbourhood operators  C/modifications.py:163-275, 1589-1637; C/post_processing.py:561-623;
# reshapes: C/helper_functions.py:96-199 (smarter_reshape, vector_smarter_reshape)
def _resolution_conversion(r):                    # C/helper_functions.py:10-20
    return int(r) if int(r) % 2 == 1 else int(r) + 1


def _smarter_reshape(pattern, resolution):        # C/helper_functions.py:96-148
    n = pattern.shape[0]
    res = [_resolution_conversion(r) for r in np.atleast_1d(np.asarray(resolution)).ravel()]
    if len(res) == 1:
        r = res[0]
        if n // r == 1:
            return pattern
        if n // r ** 2 == 1:
            return pattern.reshape(r, r)
        if n // r ** 3 == 1:
            return pattern.reshape(r, r, r)
        raise ValueError("Cannot reshape the pattern with shape %r" % (pattern.shape,))
    if len(res) == 2:
        div = n // (res[0] * res[1])
        return pattern.reshape(res[0], res[1]) if div == 1 else pattern.reshape(res[0], res[1], int(div))
    div = n // (res[0] * res[1] * res[2])
    if div != 1:
        raise ValueError("Cannot reshape the pattern with shape %r" % (pattern.shape,))
    return pattern.reshape(res[0], res[1], res[2])


def _conv_averaging(u, kernel_size, iterations):  # C/post_processing.py:561-600
    from scipy.ndimage import convolve
    if iterations == 0:
        return u
    if isinstance(kernel_size, (int, np.integer)):
        kernel_size = (int(kernel_size),) * u.ndim
    kernel_size = tuple(int(k) for k in np.asarray(kernel_size).ravel())
    if len(kernel_size) != u.ndim:
        raise ValueError("Dimension of the kernel and the field must match!")
    filt = np.ones(kernel_size) / float(np.prod(kernel_size))
    new = convolve(u, filt)
    for _ in range(iterations - 1):
        new = convolve(new, filt)
    return new


@_m("conv_averaging")        # C/modifications.py:1607-1612
def _(e, co, params):
    u = _smarter_reshape(_inner(e, co, params), e.args["co_resolution"])
    return _conv_averaging(u, e.args["kernel_size"], e.args["iterations"]).flatten()


@_m("conv_edge_detection")   # C/modifications.py:1631-1634 (the grid-shaped result is returned as is)
def _(e, co, params):
    from scipy.ndimage import convolve
    u = _smarter_reshape(_inner(e, co, params), e.args["co_resolution"])
    f = np.asarray([[-1, -1, -1], [-1, 8, -1], [-1, -1, -1]], dtype=np.float64)
    f = f if u.ndim == 2 else f[:, :, None]
    _note(convolve(np.abs(u), np.abs(f)).ravel())   # sum |w||u|: what an input rounding error is multiplied by
    return convolve(u, f)


def _signed(e, co, params, crop):                 # C/modifications.py:163-218 (old), 220-275
    sp = _inner(e, co, params)
    if np.amin(sp) < 0:
        return sp
    res = e.args["co_resolution"]
    s = _smarter_reshape(sp, res)
    c = np.stack([_smarter_reshape(co[i], res) for i in range(co.shape[0])])
    seps = tuple(np.abs(c[i, 1 * (i == 0), 1 * (i == 1), 1 * (i == 2)] - c[i, 0, 0, 0]) for i in range(c.shape[0]))
    boundary = s < np.min(seps)
    interior = None
    for axis in (0, 1):
        b = np.moveaxis(boundary, axis, 0)
        chu, chuu = np.zeros(b.shape), np.zeros(b.shape)
        chu[1:] = b[1:] * ~b[:-1]
        chuu[:-1] = b[:-1] * ~b[1:]
        mark = np.cumsum(chu, axis=0)
        if crop:
            fmark = np.flip(np.cumsum(chuu, axis=0), axis=0)   # :250-254 the cumulative sum, flipped (not a reverse sum)
        else:
            # signed_old :185-203: fmark[i] accumulates the falling edges met walking back from the far end, then is
            # flipped: the number of falling edges at or after the point
            fmark = np.flip(np.cumsum(np.flip(chuu, axis=0), axis=0), axis=0)
        part = np.moveaxis(np.clip(mark % 2 + fmark % 2, 0, 1), 0, axis)
        interior = part if interior is None else interior * part
    interior = _conv_averaging(interior, (2, 2, 1), 1)
    if crop:
        interior = np.pad(interior[1:-1, 1:-1, 1:-1], pad_width=1, mode="edge")
    sign = 1 - 2 * (interior > 0.5)
    return sp * sign.flatten()


MODS["signed"] = lambda e, co, params: _signed(e, co, params, True)
MODS["signed_old"] = lambda e, co, params: _signed(e, co, params, False)


@_m("polygon")               # C/geom_2d.py:537-550
def _(e, co, params):
    d = _inner(e, co, params)
    if np.any(d < 0):
        return _inner(e, co, params)
    return d * interior_polygon(co, e.args["points"].copy())


@_m("shape")                 # C/geom_2d.py:425-452
def _(e, co, params):
    d = _inner(e, co, params)
    if np.any(d < 0):
        return _inner(e, co, params)
    points = np.asarray(e.args["points"], dtype=float)
    interior = np.ones(co.shape[1])
    for i in range(points.shape[1] - 1):
        t = points[:, i + 1] - points[:, i]
        t = t / norm(t)
        n = np.asarray([-t[1], abs(t[0])])
        lx, ux = min(points[0, i], points[0, i + 1]), max(points[0, i], points[0, i + 1])
        m = (co[0] >= lx) * (co[0] < ux)
        interior[m] *= np.sign(np.dot(co[:2, m].T - points[:2, i], n))
    return _inner(e, co, params) * interior


# =================================================================================================
# consumers of the field
# =================================================================================================
def point_cloud(field, co):                          # C/geom.py:62-74
    """Interior points of an already evaluated field: (3, M) float64, z = 0."""
    mask = np.asarray(field) <= 0
    pts = np.zeros((3, np.count_nonzero(mask)))
    pts[:2, :] = np.asarray(co)[:2, mask]
    return pts


def from_sdf(field, co_resolution):                  # C/vector_functions.py:130-140
    """Direction of the field's gradient on the grid: (D, N) float64."""
    dimensions = np.asarray(co_resolution).shape[0]
    g = _smarter_reshape(np.asarray(field, dtype=np.float64), co_resolution)
    vec = np.asarray(np.gradient(g)).reshape(dimensions, -1)
    m = norm(vec, axis=0)                            # batch_normalize, C/vector_modification_functions.py:14-20
    keep = ~(m == 0)
    vec[:, keep] = vec[:, keep] / m[keep]
    return vec
