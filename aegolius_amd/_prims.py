"""The `sdf_*` primitives (reference cores/sdf_3D.py, cores/sdf_2D.py) as symbolic objects.

Each PrimSDF keeps the reference function's name and positional argument meaning
(`sdf_box(co, size)`, `sdf_torus(co, R, r)` ...). Its `lower` routine turns the arguments into the
fp32 parameter block documented next to the matching device function in csrc/sdfk_device.h; all
constant arithmetic happens here in float64.
"""
import numpy as np

from ._ir import PrimSDF

_REG = {}

# parameter names after `co`, as the reference spells them (cores/sdf_3D.py:13-286, cores/sdf_2D.py:12-224): the
# primitives report these signatures and take keyword arguments like the reference's functions
ARG_NAMES = {
    "sdf_x": ('offset',),
    "sdf_y": ('offset',),
    "sdf_z": ('offset',),
    "sdf_sphere": ('radius',),
    "sdf_cylinder": ('radius', 'height'),
    "sdf_box": ('size',),
    "sdf_torus": ('R', 'r'),
    "sdf_chainlink": ('R', 'r', 'length'),
    "sdf_braid": ('length', 'R', 'r', 'pitch'),
    "sdf_arc_3d": ('R', 'r', 'start_angle', 'end_angle'),
    "sdf_plane": ('normal', 'offset'),
    "sudf_plane": ('normal', 'thickness'),
    "sdf_segment_3d": ('a', 'b'),
    "sdf_cone": ('height', 'angle'),
    "sdf_oriented_infinite_cone": ('angle',),
    "sdf_infinite_cone": ('angle',),
    "sdf_solid_angle": ('radius', 'angle_1', 'angle_2'),
    "sdf_triangle_3d": ('a', 'b', 'c'),
    "sdf_quad_3d": ('a', 'b', 'c', 'd'),
    "sdf_segmented_curve_3d": ('points', 't'),
    "sdf_segmented_line_3d": ('points',),
    "sdf_parametric_curve_3d": ('f', 'f_parameters', 't'),
    "sdf_point_cloud_3d": ('points',),
    "sdf_circle": ('radius',),
    "sdf_neu_circle": ('radius', 'norm'),
    "sdf_box_2d": ('size',),
    "sdf_segment_2d": ('a', 'b'),
    "sdf_rounded_box_2d": ('size', 'rounding'),
    "sdf_triangle_2d": ('p0', 'p1', 'p2'),
    "sdf_arc": ('radius', 'start_angle', 'end_angle'),
    "sdf_sector": ('radius', 'angle_1', 'angle_2'),
    "sdf_inf_sector": ('angle_1', 'angle_2'),
    "sdf_ngon": ('radius', 'n'),
    "sdf_segmented_curve_2d": ('points', 't'),
    "sdf_segmented_line_2d": ('points',),
    "sdf_polygon_2d": ('points',),
    "sdf_parametric_curve_2d": ('f', 'f_parameters', 't'),
    "sdf_point_cloud_2d": ('points',),
}


def _prim(name, doc=""):
    def deco(fn):
        _REG[name] = PrimSDF(name, fn, doc, ARG_NAMES.get(name))
        return fn
    return deco


def _f(x):
    return float(x)


def _vec(x, n, what):
    v = np.asarray(x, dtype=np.float64).ravel()
    if v.size < n:
        raise ValueError("%s needs %d components; got %r" % (what, n, x))
    return v[:n]


def _inv(x):
    with np.errstate(divide="ignore"):
        return 1.0 / np.float64(x)


def _nargs(name, args, n):
    if len(args) != n:
        raise TypeError("%s() takes %d positional arguments after co but %d were given" % (name, n, len(args)))


# ---- 3-D --------------------------------------------------------------------------------------
def _axis(idx, name):
    def fn(L, v, c, args):
        _nargs(name, args, 1)
        L.emit("P_AXIS", v, c, params=[args[0], idx])
    return fn


_prim("sdf_x")(_axis(0, "sdf_x"))   # sdf_3D.py:13-14
_prim("sdf_y")(_axis(1, "sdf_y"))   # :17-18
_prim("sdf_z")(_axis(2, "sdf_z"))   # :21-22


@_prim("sdf_sphere")                # :25-27
def _(L, v, c, args):
    _nargs("sdf_sphere", args, 1)
    L.emit("P_SPHERE", v, c, params=[args[0]])


@_prim("sdf_cylinder")              # :30-37
def _(L, v, c, args):
    _nargs("sdf_cylinder", args, 2)
    L.emit("P_CYLINDER", v, c, params=[args[0], args[1] / 2])


@_prim("sdf_box")                   # :40-47
def _(L, v, c, args):
    _nargs("sdf_box", args, 1)
    L.emit("P_BOX", v, c, params=_vec(args[0], 3, "size") / 2)


@_prim("sdf_torus")                 # :50-53
def _(L, v, c, args):
    _nargs("sdf_torus", args, 2)
    L.emit("P_TORUS", v, c, params=[args[0], args[1]])


@_prim("sdf_chainlink")             # :56-61
def _(L, v, c, args):
    _nargs("sdf_chainlink", args, 3)
    L.emit("P_CHAINLINK", v, c, params=[args[0], args[1], args[2] / 2])


@_prim("sdf_braid")                 # :64-75
def _(L, v, c, args):
    _nargs("sdf_braid", args, 4)
    length, R, r, pitch = args
    L.emit("P_BRAID", v, c, params=[length / 2, R, r, pitch])


@_prim("sdf_arc_3d")                # :78-96
def _(L, v, c, args):
    _nargs("sdf_arc_3d", args, 4)
    R, r, a0, a1 = args
    mid = (a0 + a1) / 2
    L.emit("P_ARC3D", v, c, params=[R, r, np.cos(mid), np.sin(mid), np.abs(a1 - mid)])


def _plane(op, name):
    def fn(L, v, c, args):
        _nargs(name, args, 2)
        normal = _vec(args[0], 3, "normal")
        with np.errstate(divide="ignore", invalid="ignore"):
            n = normal / np.linalg.norm(normal)
        L.emit(op, v, c, params=[n[0], n[1], n[2], args[1] if op == "P_PLANE" else args[1] / 2])
    return fn


_prim("sdf_plane")(_plane("P_PLANE", "sdf_plane"))      # :99-102
_prim("sudf_plane")(_plane("P_UPLANE", "sudf_plane"))   # :105-108


def _seg3(a, b):
    a, b = _vec(a, 3, "a"), _vec(b, 3, "b")
    ba = b - a
    return np.concatenate([a, ba, [_inv(np.dot(ba, ba))]])


def _seg2(a, b):
    a, b = _vec(a, 2, "a"), _vec(b, 2, "b")
    ba = b - a
    return np.concatenate([a, ba, [_inv(np.dot(ba, ba))]])


@_prim("sdf_segment_3d")            # :111-118
def _(L, v, c, args):
    _nargs("sdf_segment_3d", args, 2)
    L.emit("P_SEGMENT3", v, c, params=_seg3(args[0], args[1]))


@_prim("sdf_cone")                  # :121-136
def _(L, v, c, args):
    _nargs("sdf_cone", args, 2)
    height, angle = args
    q = np.asarray((np.tan(angle), -1.0)) * height
    L.emit("P_CONE", v, c, params=[q[0], q[1], height * (0.5 ** (1 / 3)), _inv(np.dot(q, q)), _inv(q[0])])


def _infcone(oriented, name):
    def fn(L, v, c, args):
        _nargs(name, args, 1)
        L.emit("P_INFCONE", v, c, params=[np.sin(args[0]), np.cos(args[0]), oriented])
    return fn


_prim("sdf_oriented_infinite_cone")(_infcone(1.0, "sdf_oriented_infinite_cone"))  # :139-148
_prim("sdf_infinite_cone")(_infcone(0.0, "sdf_infinite_cone"))                    # :151-157


def _sector_params(radius, a1, a2):
    half = np.abs(a2 - a1) / 2
    mid = (a2 + a1) / 2
    return [radius, np.cos(mid), np.sin(mid), half, np.cos(half), np.sin(half)]


@_prim("sdf_solid_angle")           # :160-183
def _(L, v, c, args):
    _nargs("sdf_solid_angle", args, 3)
    L.emit("P_SOLIDANGLE", v, c, params=_sector_params(*args))


def _polygon3(name, op, k):
    def fn(L, v, c, args):
        _nargs(name, args, k)
        pts = [_vec(p, 3, "vertex") for p in args]
        edges = [pts[(i + 1) % k] - pts[i] for i in range(k)]          # s1 = b-a, s2 = c-b, ... , last = a - last
        normal = np.cross(edges[0], edges[-1])
        crosses = [np.cross(s, normal) for s in edges]
        invs = [_inv(np.dot(s, s)) for s in edges]
        L.emit(op, v, c, params=np.concatenate(pts + edges + crosses + [normal, invs, [_inv(np.dot(normal, normal))]]))
    return fn


_prim("sdf_triangle_3d")(_polygon3("sdf_triangle_3d", "P_TRIANGLE3", 3))  # :186-214
_prim("sdf_quad_3d")(_polygon3("sdf_quad_3d", "P_QUAD3", 4))              # :217-250


def _points3(points):
    p = np.asarray(points, dtype=np.float64)
    if p.ndim != 2 or p.shape[0] < 3:
        raise ValueError("points must have shape (3, M); got %r" % (p.shape,))
    return p[:3]


def _points2(points):
    p = np.asarray(points, dtype=np.float64)
    if p.ndim != 2 or p.shape[0] < 2:
        raise ValueError("points must have shape (2+, M); got %r" % (p.shape,))
    return p[:2]


def emit_segline3(L, v, c, points, extra=()):
    """min over consecutive segments (:264-271); `extra` = additional (a, b) pairs (closing segment)."""
    p = _points3(points)
    rows = [_seg3(p[:, i], p[:, i + 1]) for i in range(p.shape[1] - 1)] + [_seg3(a, b) for a, b in extra]
    off = L.add_table(np.concatenate(rows) if rows else [])
    L.emit("P_SEGLINE3", v, c, params=[len(rows), off])


def emit_segline2(L, v, c, points, extra=(), closed_loop=False):
    p = _points2(points)
    m = p.shape[1]
    idx = [(i, (i + 1) % m) for i in range(m)] if closed_loop else [(i, i + 1) for i in range(m - 1)]
    rows = [_seg2(p[:, i], p[:, j]) for i, j in idx] + [_seg2(a, b) for a, b in extra]
    off = L.add_table(np.concatenate(rows) if rows else [])
    L.emit("P_SEGLINE2", v, c, params=[len(rows), off])


TREE_THRESHOLD = 256     # tables up to this many points are scanned (P_NEAREST*), larger ones go through a box tree
TREE_LEAF = 32           # points per leaf box = children per box on the two levels above


def build_point_tree(points32, with_order=False):
    """(M, 3) float32 points -> flat float32 table for P_NEARTREE and the number of root boxes
    (with_order: also the index of the tree's first point and the original indices of the points in leaf order).
    k-d median splits along the longest axis down to leaves of <= TREE_LEAF points; consecutive leaves (spatially
    coherent in k-d order) are grouped TREE_LEAF at a time under one middle box, middle boxes TREE_LEAF at a time under
    one root box. Boxes are the exact float32 bounds of their points. Built by the library on the host
    (sdfk_point_tree_build: compiled code, as the reference's scipy KDTree is; 1.7 M points in a fraction of a second)
    — no GPU involved."""
    from . import _engine
    table, n_root, point_base, order = _engine.point_tree(points32, TREE_LEAF, with_order)
    if with_order:
        return table, n_root, point_base, order
    return table, n_root


def emit_nearest(L, v, c, samples, dim):
    s = np.asarray(samples, dtype=np.float64)
    if s.ndim != 2 or s.shape[0] != dim:
        raise ValueError("nearest-point table must have shape (%d, M); got %r" % (dim, s.shape))
    if s.shape[1] < 1:
        raise ValueError("nearest-point table is empty")
    if s.shape[1] > TREE_THRESHOLD:
        with np.errstate(over="ignore"):
            p32 = np.zeros((s.shape[1], 3), dtype=np.float32)
            p32[:, :dim] = s.T.astype(np.float32)
        if np.all(np.isfinite(p32)):
            table, n_root = build_point_tree(p32)
            off = L.add_table(table)
            L.emit("P_NEARTREE", v, c, params=[n_root, off, dim])
            return
    off = L.add_table(s.T)
    L.emit("P_NEAREST3" if dim == 3 else "P_NEAREST2", v, c, params=[s.shape[1], off])


def resample_curve(points, t):
    """fval of sdf_segmented_curve_* (sdf_3D.py:255-257, sdf_2D.py:182-184)."""
    t = np.asarray(t, dtype=np.float64)
    vi = np.floor(t).astype(int)
    u = t - vi
    return points[:, vi + 1] * u + points[:, vi] * (1 - u)


@_prim("sdf_segmented_line_3d")     # :264-271
def _(L, v, c, args):
    _nargs("sdf_segmented_line_3d", args, 1)
    emit_segline3(L, v, c, args[0])


@_prim("sdf_segmented_curve_3d")    # :253-261
def _(L, v, c, args):
    _nargs("sdf_segmented_curve_3d", args, 2)
    emit_nearest(L, v, c, resample_curve(_points3(args[0]), args[1]), 3)


@_prim("sdf_parametric_curve_3d")   # :274-280
def _(L, v, c, args):
    _nargs("sdf_parametric_curve_3d", args, 3)
    f, fp, t = args
    emit_nearest(L, v, c, np.asarray(f(t, *fp), dtype=np.float64), 3)


@_prim("sdf_point_cloud_3d")        # :283-286
def _(L, v, c, args):
    _nargs("sdf_point_cloud_3d", args, 1)
    emit_nearest(L, v, c, np.asarray(args[0], dtype=np.float64), 3)


# ---- 2-D --------------------------------------------------------------------------------------
@_prim("sdf_circle")                # sdf_2D.py:12-14
def _(L, v, c, args):
    _nargs("sdf_circle", args, 1)
    L.emit("P_CIRCLE", v, c, params=[args[0]])


@_prim("sdf_neu_circle")            # :17-19
def _(L, v, c, args):
    _nargs("sdf_neu_circle", args, 2)
    radius, order = args
    kind, o = 0.0, order
    if order == np.inf:
        kind, o = 1.0, 0.0
    elif order == -np.inf:
        kind, o = 2.0, 0.0
    elif order == 0:
        kind, o = 3.0, 0.0
    L.emit("P_NEUCIRCLE", v, c, params=[radius, o, kind])


@_prim("sdf_box_2d")                # :22-28
def _(L, v, c, args):
    _nargs("sdf_box_2d", args, 1)
    L.emit("P_BOX2", v, c, params=_vec(args[0], 2, "size") / 2)


@_prim("sdf_segment_2d")            # :31-38
def _(L, v, c, args):
    _nargs("sdf_segment_2d", args, 2)
    L.emit("P_SEGMENT2", v, c, params=_seg2(args[0], args[1]))


@_prim("sdf_rounded_box_2d")        # :41-57
def _(L, v, c, args):
    _nargs("sdf_rounded_box_2d", args, 2)
    L.emit("P_RBOX2", v, c, params=np.concatenate([_vec(args[0], 2, "size") / 2, _vec(args[1], 4, "rounding")]))


@_prim("sdf_triangle_2d")           # :60-82
def _(L, v, c, args):
    _nargs("sdf_triangle_2d", args, 3)
    p = [_vec(q, 2, "vertex") for q in args]
    e = [p[1] - p[0], p[2] - p[1], p[0] - p[2]]
    s = np.sign(e[0][0] * e[2][1] - e[0][1] * e[2][0])
    L.emit("P_TRIANGLE2", v, c, params=np.concatenate(p + e + [[_inv(np.dot(x, x)) for x in e], [s]]))


@_prim("sdf_arc")                   # :85-103
def _(L, v, c, args):
    _nargs("sdf_arc", args, 3)
    radius, a0, a1 = args
    mid = (a0 + a1) / 2
    L.emit("P_ARC2", v, c, params=[radius, np.cos(mid), np.sin(mid), np.abs(a1 - mid)])


@_prim("sdf_sector")                # :105-129
def _(L, v, c, args):
    _nargs("sdf_sector", args, 3)
    L.emit("P_SECTOR", v, c, params=_sector_params(*args))


@_prim("sdf_inf_sector")            # :132-150
def _(L, v, c, args):
    _nargs("sdf_inf_sector", args, 2)
    L.emit("P_INFSECTOR", v, c, params=_sector_params(0.0, *args)[1:])


@_prim("sdf_ngon")                  # :153-177
def _(L, v, c, args):
    _nargs("sdf_ngon", args, 2)
    radius, n = args
    beta = np.pi * (0.5 - 1 / n)
    alpha = 2 * np.pi / n
    s, co = np.sin(beta), np.cos(beta)
    # integer n up to 16: fold by successive rotations instead of atan2 / mod / sincos (prim_ngon)
    whole = float(n) == int(n) and 3 <= int(n) <= 16
    L.emit("P_NGON", v, c, params=[radius, alpha, 1 / alpha, -co, s, s, co, 2 * radius * np.sin(alpha / 2),
                                   np.cos(alpha), np.sin(alpha), int(n) // 2 if whole else 0])


@_prim("sdf_segmented_line_2d")     # :191-198
def _(L, v, c, args):
    _nargs("sdf_segmented_line_2d", args, 1)
    emit_segline2(L, v, c, args[0])


@_prim("sdf_segmented_curve_2d")    # :180-188
def _(L, v, c, args):
    _nargs("sdf_segmented_curve_2d", args, 2)
    emit_nearest(L, v, c, resample_curve(_points2(args[0]), args[1]), 2)


@_prim("sdf_parametric_curve_2d")   # :214-218
def _(L, v, c, args):
    _nargs("sdf_parametric_curve_2d", args, 3)
    f, fp, t = args
    emit_nearest(L, v, c, np.asarray(f(t, *fp), dtype=np.float64), 2)


@_prim("sdf_point_cloud_2d")        # :221-224
def _(L, v, c, args):
    _nargs("sdf_point_cloud_2d", args, 1)
    emit_nearest(L, v, c, _points2(args[0]), 2)


@_prim("sdf_polygon_2d")            # :201-211
def _(L, v, c, args):
    _nargs("sdf_polygon_2d", args, 1)
    from ._polygon import emit_polygon_sign
    pts = np.asarray(args[0], dtype=np.float64)
    emit_segline2(L, v, c, pts, closed_loop=True)
    s = L.new_v()
    emit_polygon_sign(L, s, c, pts)
    L.emit("VMUL", v, v, s)
    L.free_v(s)


def get(name):
    return _REG[name]


ALL = dict(_REG)


# ---- closed-curve composites (the `sdf_closed_curve` closures of the curve classes:
#      reference cores/geom_3d.py:610-618,677-682,741-746 ; cores/geom_2d.py:375-383,491-496,584-589)
#      min(open curve, closing segment) ----
def _closed(name, dim, kind):
    seg_op, seg = ("P_SEGMENT3", _seg3) if dim == 3 else ("P_SEGMENT2", _seg2)
    pts = _points3 if dim == 3 else _points2

    def fn(L, v, c, args):
        if kind == "parametric":
            _nargs(name, args, 3)
            f, fp, t = args
            p0 = np.asarray(f(t[0], *fp), dtype=np.float64).ravel()
            p1 = np.asarray(f(t[-1], *fp), dtype=np.float64).ravel()
            emit_nearest(L, v, c, np.asarray(f(t, *fp), dtype=np.float64), dim)
        elif kind == "segmented":
            _nargs(name, args, 2)
            p = pts(args[0])
            p0, p1 = p[:, 0], p[:, -1]
            emit_nearest(L, v, c, resample_curve(p, args[1]), dim)
        else:
            _nargs(name, args, 1)
            p = pts(args[0])
            p0, p1 = p[:, 0], p[:, -1]
            (emit_segline3 if dim == 3 else emit_segline2)(L, v, c, p)
        w = L.new_v()
        L.emit(seg_op, w, c, params=seg(p0, p1))
        L.emit("VMIN", v, v, w)
        L.free_v(w)
    _REG[name] = PrimSDF(name, fn)


for _dim in (2, 3):
    for _kind in ("parametric", "segmented", "line"):
        _closed("closed_%s_curve_%dd" % (_kind, _dim), _dim, _kind)

ALL = dict(_REG)
