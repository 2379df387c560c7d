"""Lowering of the ModifyObject closures (reference cores/modifications.py:75-1663).

Each function receives the Lowerer, the ModSDF node, the coordinate register holding the closure's
`co` argument, that register's aliasing mode (see _lower.py) and the geometry parameters that the
reference threads through every closure as `*params`; it returns the value register with the result.
Constants are computed in float64 exactly as the reference computes them at evaluation time.
"""
import numpy as np

from ._lower import ALIASED, FROZEN, OWNED, LoweringError

MOD_LOWER = {}


def _mod(name):
    def deco(fn):
        MOD_LOWER[name] = fn
        return fn
    return deco


def _vec3(x, what):
    v = np.asarray(x, dtype=np.float64).ravel()
    if v.size != 3:
        raise ValueError("%s must have 3 components; got %r" % (what, x))
    return v


def _inv(x):
    x = np.asarray(x, dtype=np.float64)
    with np.errstate(divide="ignore"):
        return 1.0 / x


# ------------------------------------------------------------------------------------------------
# generic shapes
# ------------------------------------------------------------------------------------------------
def _coord_mod(L, expr, creg, mode, params, ops):
    """Array-creating domain warp: ops = [(opname, imm, params), ...] applied in sequence."""
    dst = L.writable(creg, mode)
    src = creg
    for opname, imm, prm in ops:
        L.emit(opname, dst, src, imm, params=prm)
        src = dst
    v = L.lower_expr(expr.inner, dst, OWNED, params)
    L.release(dst, creg)
    return v


def _inplace_mod(L, expr, creg, mode, params, ops):
    """Modification that overwrites the array it was given (mutation visible to ALIASED readers)."""
    if mode == FROZEN:
        dst, inner_mode = L.new_c(), OWNED
    else:
        dst, inner_mode = creg, mode
    src = creg
    for opname, imm, prm in ops:
        L.emit(opname, dst, src, imm, params=prm)
        src = dst
    v = L.lower_expr(expr.inner, dst, inner_mode, params)
    L.release(dst, creg)
    return v


def _value_mod(L, expr, creg, mode, params, opname, prm=()):
    v = L.lower_expr(expr.inner, creg, mode, params)
    L.emit(opname, v, v, params=prm)
    return v


def _two_field(L, expr, creg, mode, params, opname):
    """geo_object(co, *params) <op> second(co, *second_params)   (:343, :366, :798)"""
    if mode == FROZEN:
        c = L.new_c()
        L.emit("MOVC", c, creg)
        after = OWNED
    else:
        c, after = creg, mode
    v1 = L.lower_expr(expr.inner, c, ALIASED, params)
    sp = params if expr.second_params is None else tuple(expr.second_params)
    v2 = L.lower_callable(expr.second, c, after, sp)
    L.release(c, creg)
    L.emit(opname, v1, v1, v2)
    L.free_v(v2)
    return v1


# ------------------------------------------------------------------------------------------------
# domain warps
# ------------------------------------------------------------------------------------------------
@_mod("elongation")          # :75-98
def _(L, e, creg, mode, params):
    return _coord_mod(L, e, creg, mode, params, [("ELONGATE", 0, _vec3(e.args["ev"], "elongate_vector") / 2)])


@_mod("revolution")          # :411-436
def _(L, e, creg, mode, params):
    return _coord_mod(L, e, creg, mode, params, [("REVOLVE", 0, [e.args["radius"]])])


@_mod("axis_revolution")     # :438-472  (rotation is in place, qo is a new array)
def _(L, e, creg, mode, params):
    a, r = float(e.args["angle"]), e.args["radius"]
    cs = [np.cos(a), np.sin(a)]
    if mode == ALIASED:
        L.emit("ROT2D", creg, creg, params=cs)
        t = L.new_c()
        L.emit("AXREV", t, creg, params=cs + [r])
    else:
        t = creg if mode == OWNED else L.new_c()
        L.emit("ROT2D", t, creg, params=cs)
        L.emit("AXREV", t, t, params=cs + [r])
    v = L.lower_expr(e.inner, t, OWNED, params)
    L.release(t, creg)
    return v


@_mod("twist")               # :502-527
def _(L, e, creg, mode, params):
    return _coord_mod(L, e, creg, mode, params, [("TWIST", 0, [e.args["pitch"]])])


@_mod("bend")                # :529-577
def _(L, e, creg, mode, params):
    R, a = float(e.args["radius"]), float(e.args["angle"])
    c, s = np.cos(a / 2), np.sin(a / 2)
    return _coord_mod(L, e, creg, mode, params,
                      [("BEND", 0, [R, c, s, R * a / 2, R * s, R * (1 - c)])])


def _shear_matrix(sheared_axis, fixed_axis, t):
    # the exact matrices of reference cores/modifications.py:745-769
    if sheared_axis == 0:
        if fixed_axis == 1:
            return np.asarray([[1, 0, 0], [0, 1, 0], [-t, 0, 1]], dtype=np.float64).T
        if fixed_axis == 2:
            return np.asarray([[1, 0, 0], [-t, 1, 0], [0, 0, 1]], dtype=np.float64).T
    elif sheared_axis == 1:
        if fixed_axis == 0:
            return np.asarray([[1, 0, 0], [0, 1, 0], [0, -t, 1]], dtype=np.float64).T
        if fixed_axis == 2:
            return np.asarray([[1, 0, 0], [-t, 1, 0], [0, 0, 1]], dtype=np.float64)
    elif sheared_axis == 2:
        if fixed_axis == 0:
            return np.asarray([[1, 0, 0], [0, 1, 0], [0, -t, 1]], dtype=np.float64)
        if fixed_axis == 1:
            return np.asarray([[1, 0, 0], [0, 1, 0], [-t, 0, 1]], dtype=np.float64)
    raise ValueError("Specify a valid axis index")


# the six named variants (:579-726) are the (sheared, fixed) pairs below
_NAMED_SHEARS = {"shear_xz": (0, 2), "shear_yz": (1, 2), "shear_xy": (0, 1), "shear_zy": (2, 1), "shear_yx": (1, 0),
                 "shear_zx": (2, 0)}


def _shear(L, e, creg, mode, params):
    if e.name == "shear":
        sa, fa = e.args["sheared_axis"], e.args["fixed_axis"]
    else:
        sa, fa = _NAMED_SHEARS[e.name]
    o = _shear_matrix(sa, fa, np.tan(e.args["angle"]))
    return _coord_mod(L, e, creg, mode, params, [("LIN3", 0, o.ravel())])


for _n in list(_NAMED_SHEARS) + ["shear"]:
    MOD_LOWER[_n] = _shear


@_mod("infinite_repetition")  # :803-825
def _(L, e, creg, mode, params):
    d = _vec3(e.args["distances"], "distances")
    return _coord_mod(L, e, creg, mode, params, [("INFREP", 0, np.concatenate([d / 2, d, _inv(d)]))])


def _finrep_params(size, rep):
    size, rep = _vec3(size, "size"), _vec3(rep, "repetitions")
    with np.errstate(divide="ignore", invalid="ignore"):
        c = size * (1 - 1 / rep) / 2
        d = size * (1 / 2 - 1 / rep)
        s = size / rep
    return np.concatenate([c, d, s, s / 2, _inv(s)]), s


@_mod("finite_repetition")   # :827-873
def _(L, e, creg, mode, params):
    prm, _s = _finrep_params(e.args["size"], e.args["repetitions"])
    return _coord_mod(L, e, creg, mode, params, [("FINREP", 0, prm)])


@_mod("finite_repetition_rescaled")  # :875-930
def _(L, e, creg, mode, params):
    prm, s = _finrep_params(e.args["size"], e.args["repetitions"])
    f, pad = _vec3(e.args["instance_size"], "instance_size"), _vec3(e.args["padding"], "padding")
    sss = float(np.min(s / (f + pad)))
    v = _coord_mod(L, e, creg, mode, params, [("FINREP", 0, prm), ("CSCALE", 0, [1.0 / sss])])
    L.emit("VSCALE", v, v, params=[sss])
    return v


@_mod("symmetry")            # :932-955
def _(L, e, creg, mode, params):
    axis = e.args["axis"]
    if axis > 3:             # the reference's guard `axis > co.shape[0]` (off by one): untouched
        return L.lower_expr(e.inner, creg, mode, params)
    if axis == 3 or axis < -3:
        raise IndexError("index %d is out of bounds for axis 0 with size 3" % axis)
    return _inplace_mod(L, e, creg, mode, params, [("SYMMETRY", int(axis) % 3, [])])


def _segment_frame(a, b):
    """Frame of mirror / linear_instancing (:978-988, :1058-1068): rows x̂, ŷ, ẑ ; centre ; length."""
    a, b = _vec3(a, "a"), _vec3(b, "b")
    w = b - a
    c = (b + a) / 2
    with np.errstate(divide="ignore", invalid="ignore"):
        length = np.linalg.norm(w)
        x = w / length
        y = np.asarray([-x[1], x[0], 0])
        y = y / np.linalg.norm(y)
        z = np.cross(x, y)
    rot = np.asarray([x, y, z])
    return rot, c, float(length)


def _frame_xform(rot, c):
    # co' = rot·(co - c) = rot·co - rot·c   -> XFORM with M = rot
    return ("XFORM", 0, np.concatenate([rot.ravel(), rot.dot(c)]))


@_mod("mirror")              # :957-997
def _(L, e, creg, mode, params):
    rot, c, length = _segment_frame(e.args["a"], e.args["b"])
    return _coord_mod(L, e, creg, mode, params, [_frame_xform(rot, c), ("FOLDX", 0, [length / 2])])


@_mod("rotational_symmetry")  # :999-1033
def _(L, e, creg, mode, params):
    n, radius, phase = e.args["n"], e.args["radius"], e.args["phase"]
    angle = 2 * np.pi / n
    h = angle / 2 - phase
    return _inplace_mod(L, e, creg, mode, params,
                        [("ROT2D", 0, [np.cos(h), np.sin(h)]), ("ROTSYM", 0, [angle, angle / 2, 1 / angle, radius])])


@_mod("linear_instancing")   # :1035-1088
def _(L, e, creg, mode, params):
    n = e.args["n"]
    rot, c, length = _segment_frame(e.args["a"], e.args["b"])
    with np.errstate(divide="ignore", invalid="ignore"):
        s = np.float64(length) / (n - 1)
    d = s / 2
    prm = [length / 2, -length / 2 + d, length / 2 - d, length / 2 - d, s, d, _inv(s), 1.0 if n > 2 else 0.0]
    return _coord_mod(L, e, creg, mode, params, [_frame_xform(rot, c), ("LININST", 0, prm)])


def _curve_samples(e):
    """Instance centres (and frames) of the curve_instancing family (:1108-1114, 1150-1181, 1215-1251)."""
    f, fp, t_range = e.args["f"], tuple(e.args["f_parameters"]), tuple(e.args["t_range"])
    n = int(t_range[-1])
    ts = np.linspace(*t_range)
    fval = np.asarray(f(ts, *fp), dtype=np.float64)
    va = np.zeros((3, n))
    va[:fval.shape[0]] = fval
    if e.name == "curve_instancing":
        return n, va.T.copy(), False
    tol = 0.001
    fv_min = np.asarray(f(ts - tol, *fp), dtype=np.float64)
    fv_max = np.asarray(f(ts + tol, *fp), dtype=np.float64)
    der = (fv_max - fv_min) / (2 * tol)
    dx, dy = np.zeros((3, n)), np.zeros((3, n))
    if e.name == "aligned_curve_instancing":
        der = der / np.linalg.norm(der, axis=0)
        dx[:fval.shape[0]] = der
        dy[0, :] = -dx[1]
        dy[1, :] = dx[0]
    else:
        dermag = np.linalg.norm(der, axis=0)
        der = der / dermag
        der2 = (fv_max - 2 * fval + fv_min) / (tol ** 2)
        der2 = der2 / dermag
        der2 = der2 / np.linalg.norm(der2, axis=0)
        dx[:fval.shape[0]] = der
        dy[:fval.shape[0]] = der2
    dz = np.cross(dx.T, dy.T).T
    # trot[:, :, i] has rows dx_i, dy_i, dz_i ; table row = centre, dx, dy, dz
    rows = np.concatenate([va.T, dx.T, dy.T, dz.T], axis=1)
    return n, rows, True


def _curve_instancing(L, e, creg, mode, params):
    n, rows, frames = _curve_samples(e)
    if n < 1:
        raise ValueError("curve instancing needs at least one instance")
    from ._prims import TREE_THRESHOLD, build_point_tree
    centres = np.ascontiguousarray(rows[:, :3], dtype=np.float32)
    if n > TREE_THRESHOLD and np.all(np.isfinite(centres)):
        # many instances: nearest centre through the box tree of the point clouds
        tree, n_root, point_base, order = build_point_tree(centres, with_order=True)
        tree_off = L.add_table(tree)
        rows_off = L.add_table(rows)
        order_off = L.add_table(order)                         # original index of the tree's points, leaf order
        return _coord_mod(L, e, creg, mode, params,
                          [("CURVEINSTT", 0, [n_root, tree_off, 1.0 if frames else 0.0, rows_off, point_base, order_off])])
    off = L.add_table(rows)
    return _coord_mod(L, e, creg, mode, params, [("CURVEINST", 0, [n, off, 1.0 if frames else 0.0])])


for _n in ("curve_instancing", "aligned_curve_instancing", "fully_aligned_curve_instancing"):
    MOD_LOWER[_n] = _curve_instancing


@_mod("move_sdf")            # :1268-1287
def _(L, e, creg, mode, params):
    return _coord_mod(L, e, creg, mode, params, [("XLATE", 0, _vec3(e.args["move_vector"], "move_vector"))])


@_mod("rotate_sdf")          # :1308-1328
def _(L, e, creg, mode, params):
    rm = np.asarray(e.args["rotation_matrix"], dtype=np.float64)
    if rm.shape != (3, 3):
        raise ValueError("rotation_matrix must have shape (3, 3)")
    return _coord_mod(L, e, creg, mode, params, [("LIN3", 0, rm.T.ravel())])


# ------------------------------------------------------------------------------------------------
# domain + value
# ------------------------------------------------------------------------------------------------
@_mod("rounding_cs")         # :120-144
def _(L, e, creg, mode, params):
    r, bb = e.args["rounding_radius"], e.args["bb_size"]
    scale = 1 - 2 * r / bb + 1e-8
    scale = float(np.maximum(scale, 1e-8))
    v = _coord_mod(L, e, creg, mode, params, [("CSCALE", 0, [1.0 / scale])])
    L.emit("VAFFINE", v, v, params=[scale, r])
    return v


@_mod("scale_sdf")           # :1289-1306
def _(L, e, creg, mode, params):
    k = e.args["scale_factor"]
    v = _coord_mod(L, e, creg, mode, params, [("CSCALE", 0, [_inv(k)])])
    L.emit("VSCALE", v, v, params=[k])
    return v


@_mod("extrusion")           # :474-500
def _(L, e, creg, mode, params):
    w1 = L.new_v()
    L.emit("P_ZSLAB", w1, creg, params=[e.args["distance"] / 2])
    d = _coord_mod(L, e, creg, mode, params, [("ZEROZ", 0, [])])
    L.emit("EXTRUDE", d, d, w1)
    L.free_v(w1)
    return d


# ------------------------------------------------------------------------------------------------
# value-only
# ------------------------------------------------------------------------------------------------
VALUE_OPS = {}      # modification name -> (opcode, arguments -> parameter block): also the array-level functions of
                    # cores/post_processing.py run through these (aegolius_amd._eval.apply_fields)


def _simple_value(opname, prm_fn, name=None):
    if name is not None:
        VALUE_OPS[name] = (opname, prm_fn)

    def fn(L, e, creg, mode, params):
        return _value_mod(L, e, creg, mode, params, opname, prm_fn(e.args))
    return fn


MOD_LOWER["rounding"] = _simple_value("VSUBC", lambda a: [a["rounding_radius"]], name="rounding")            # :100-118
MOD_LOWER["boundary"] = _simple_value("VABS", lambda a: [], name="boundary")                                  # :146-161
MOD_LOWER["invert"] = _simple_value("VNEG", lambda a: [], name="invert")                                    # :277-299
MOD_LOWER["sign"] = _simple_value("VSIGN", lambda a: [], name="sign")                                     # :301-323
MOD_LOWER["onion"] = _simple_value("VONION", lambda a: [a["thickness"]], name="onion")                     # :371-388
MOD_LOWER["concentric"] = _simple_value("VCONCENTRIC", lambda a: [a["width"] / 2], name="concentric")           # :390-409
# post-processing wrappers :1361-1587 -> reference cores/post_processing.py:380-558
MOD_LOWER["sigmoid_falloff"] = _simple_value("VSIGMOID", lambda a: [a["amplitude"], 4 * _inv(a["width"]), 0.0], name="sigmoid_falloff")
MOD_LOWER["positive_sigmoid_falloff"] = _simple_value("VSIGMOID", lambda a: [a["amplitude"], 4 * _inv(a["width"]), a["width"]], name="positive_sigmoid_falloff")
MOD_LOWER["capped_exponential"] = _simple_value("VCAPEXP", lambda a: [a["amplitude"], -4 * _inv(a["width"])], name="capped_exponential")
MOD_LOWER["hard_binarization"] = _simple_value("VHARDBIN", lambda a: [a["threshold"]], name="hard_binarization")
MOD_LOWER["linear_falloff"] = _simple_value("VLINFALL", lambda a: [a["amplitude"], _inv(a["width"])], name="linear_falloff")
MOD_LOWER["relu"] = _simple_value("VRELU", lambda a: [_inv(a["width"])], name="relu")
MOD_LOWER["smooth_relu"] = _simple_value("VSMOOTHRELU", lambda a: [_inv(a["width"]), (a["smooth_width"] + a["threshold"]) * 4 * a["threshold"]], name="smooth_relu")


def _slowstart_params(a):
    b = (2 * a["smooth_width"] + a["threshold"]) * a["threshold"]
    bw = b / a["width"]
    return [_inv(a["width"]), bw, np.sqrt(bw) * a["ground"]]


MOD_LOWER["slowstart"] = _simple_value("VSLOWSTART", _slowstart_params, name="slowstart")
MOD_LOWER["gaussian_boundary"] = _simple_value("VGAUSS", lambda a: [a["amplitude"], _inv(a["width"]), 0.0], name="gaussian_boundary")
MOD_LOWER["gaussian_falloff"] = _simple_value("VGAUSS", lambda a: [a["amplitude"], _inv(a["width"]), 1.0], name="gaussian_falloff")


# ------------------------------------------------------------------------------------------------
# second field
# ------------------------------------------------------------------------------------------------
@_mod("recover_volume")      # :325-346
def _(L, e, creg, mode, params):
    return _two_field(L, e, creg, mode, params, "VMUL")


@_mod("define_volume")       # :348-369
def _(L, e, creg, mode, params):
    return _two_field(L, e, creg, mode, params, "VMUL")


@_mod("displacement")        # :776-801
def _(L, e, creg, mode, params):
    return _two_field(L, e, creg, mode, params, "VADD")


# ------------------------------------------------------------------------------------------------
# not pointwise / opaque: rejected loudly (next rows)
# ------------------------------------------------------------------------------------------------
def _unsupported(reason):
    def fn(L, e, creg, mode, params):
        raise NotImplementedError("modification %r: %s" % (e.name, reason))
    return fn


# ------------------------------------------------------------------------------------------------
# grid-neighbourhood operators  C/modifications.py:163-275 (signed_old, signed), :1589-1637 (conv_*)
# They need the whole field of their inner expression: the evaluation is cut into stages at these nodes
# (_eval._run_staged). In a program they are either the point where a stage stops or a V_FIELD read of the
# field an earlier stage produced.
# ------------------------------------------------------------------------------------------------
GRID_OPS = ("signed", "signed_old", "conv_averaging", "conv_edge_detection")
# opaque user code (C/modifications.py:1330-1359 custom_modification, :1639-1663 custom_post_process, and plain
# Python callables used as SDF functions): same staging, but the operator runs on the HOST between two GPU stages
HOST_OPS = ("custom_modification", "custom_post_process", "python_callable")
# operators whose stage program produces the inner field (the others are handed coordinates only)
FIELD_STAGE_OPS = GRID_OPS + ("custom_post_process",)


def _staged_op(L, e, creg, mode, params):
    from ._lower import NeedsStage, StageStop
    key = tuple(L._stack)                      # WHERE this operator sits: one node object may be used at several places
    idx = L.fields.get(key)
    if idx is not None:
        v = L.new_v()
        L.emit("V_FIELD", v, creg, idx)
        return v
    if L.stop_at == key:
        if L.probe_axis is not None:            # the coordinates this operator is handed (signed: grid spacings)
            v = L.new_v()
            L.emit("P_AXIS", v, creg, params=[0.0, float(L.probe_axis)])
        else:
            v = L.lower_expr(e.inner, creg, mode, params)
        raise StageStop(v, params)
    if L.stop_at is not None and e.name in FIELD_STAGE_OPS:
        # on the way to another operator: if that one lies inside this one, StageStop passes through here;
        # if not, this operator is met first and must get its own stage first
        L.lower_expr(e.inner, creg, mode, params)
    raise NeedsStage(e, key)


for _n in GRID_OPS + ("custom_modification", "custom_post_process"):
    MOD_LOWER[_n] = _staged_op
del _n


# ------------------------------------------------------------------------------------------------
# outline -> signed region (methods of the 2-D curve classes, reference cores/geom_2d.py)
# ------------------------------------------------------------------------------------------------
_UNSIGNED_PRIMS = {"sdf_segmented_line_2d", "sdf_segmented_curve_2d", "sdf_parametric_curve_2d", "sdf_point_cloud_2d",
                   "closed_parametric_curve_2d", "closed_segmented_curve_2d", "closed_line_curve_2d",
                   "sdf_segment_2d", "sdf_arc"}
_COORD_ONLY = {"elongation", "revolution", "axis_revolution", "twist", "bend", "shear", "infinite_repetition",
               "finite_repetition", "symmetry", "mirror", "rotational_symmetry", "linear_instancing",
               "curve_instancing", "aligned_curve_instancing", "fully_aligned_curve_instancing", "move_sdf",
               "rotate_sdf", "boundary"} | set(_NAMED_SHEARS)


def _provably_unsigned(expr):
    from ._ir import ModSDF, PrimSDF
    while isinstance(expr, ModSDF) and expr.name in _COORD_ONLY:
        expr = expr.inner
    return isinstance(expr, PrimSDF) and expr.name in _UNSIGNED_PRIMS


def _signed_region(L, e, creg, mode, params, emit_sign):
    # The reference evaluates the outline first and leaves it untouched `if np.any(d < 0)` (a whole-field
    # test, geom_2d.py:428-430, 541-543). That test is decidable here only for chains that cannot go negative.
    if not _provably_unsigned(e.inner):
        raise NotImplementedError("%s(): the reference decides at run time (np.any(d < 0)) whether the outline is "
                                  "already signed; only provably unsigned outlines are supported" % e.name)
    if mode == FROZEN:
        c = L.new_c()
        L.emit("MOVC", c, creg)
    else:
        c = creg
    v = L.lower_expr(e.inner, c, ALIASED, params)
    s = L.new_v()
    emit_sign(L, s, c)
    L.release(c, creg)
    L.emit("VMUL", v, v, s)
    L.free_v(s)
    return v


@_mod("polygon")             # geom_2d.py:530-555, 601-626
def _(L, e, creg, mode, params):
    from ._polygon import emit_polygon_sign
    return _signed_region(L, e, creg, mode, params, lambda L_, s, c: emit_polygon_sign(L_, s, c, e.args["points"]))


def _shape_rows(points):
    rows = []
    for i in range(points.shape[1] - 1):
        t = points[:, i + 1] - points[:, i]
        with np.errstate(divide="ignore", invalid="ignore"):
            t = t / np.linalg.norm(t)
        ny = abs(t[0])                       # n = (-t1, t0, 0) with n[1] made non-negative
        lx, ux = min(points[0, i], points[0, i + 1]), max(points[0, i], points[0, i + 1])
        rows.append((points[0, i], points[1, i], -t[1], ny, lx, ux))
    return rows


@_mod("shape")               # geom_2d.py:415-457
def _(L, e, creg, mode, params):
    def emit_sign(L_, s, c):
        rows = _shape_rows(np.asarray(e.args["points"], dtype=np.float64))
        off = L_.add_table(np.asarray(rows).ravel() if rows else [])
        L_.emit("P_SHAPESIGN", s, c, params=[len(rows), off])
    return _signed_region(L, e, creg, mode, params, emit_sign)
