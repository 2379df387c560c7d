"""create()/propagate() back end: lower the tree, fetch (or build) the native program, run it.

There is exactly one evaluation path — libsdfk.so on an MI355X. No NumPy evaluation exists in this
package; without the extension or without a GPU these functions raise.
"""
import collections
import threading

import numpy as np

from . import _engine
from ._lower import lower_expression, lower_geometry


class Config:
    """Process-wide evaluation settings."""
    device = 0                    # HIP device ordinal used by create()
    mode = _engine.MODE_AUTO      # MODE_AUTO (specialised, hiprtc) | MODE_INTERPRET | MODE_SPECIALIZED
    output_dtype = np.float32     # the reference returns float64; set to np.float64 to get an upcast copy
    cache_size = 128
    grid_fast_path = True         # generate_grid tags its (read-only) result; create() then evaluates the grid
                                  # from the per-axis tables on the GPU instead of uploading 12 B/point


config = Config()

_cache = collections.OrderedDict()
_lock = threading.Lock()


def program_for(lowered):
    key = lowered.key()
    with _lock:
        prog = _cache.get(key)
        if prog is not None:
            _cache.move_to_end(key)
            return prog
    prog = _engine.Program.from_lowered(lowered)
    with _lock:
        _cache[key] = prog
        while len(_cache) > config.cache_size:
            _cache.popitem(last=False)
    return prog


def _run(lowered, co, resident=False):
    axes = getattr(co, "grid_axes", None) if config.grid_fast_path else None
    if resident:
        return _run_resident(lowered, co, axes)
    if axes is not None:
        out = program_for(lowered).eval_grid_host(axes, device=config.device, mode=config.mode)
    else:
        out = program_for(lowered).eval_host(co, device=config.device, mode=config.mode)
    if config.output_dtype is not np.float32:
        out = out.astype(config.output_dtype)
    return out


def _run_resident(lowered, co, axes):
    """Same evaluation, field left in HBM as an _engine.DeviceField."""
    prog = program_for(lowered)
    if axes is not None:
        n = int(np.prod([np.asarray(a).size for a in axes]))
        field = _engine.DeviceField(n, config.device)
        prog.eval_grid(axes, 0, n, field.ptr, mode=config.mode)
        _engine.check(_engine.lib().sdfk_sync(None), "sdfk_sync")
        return field
    co = np.asarray(co)
    if co.ndim != 2 or co.shape[0] != 3:
        raise ValueError("coordinates must have shape (3, N); got %r" % (co.shape,))
    if co.dtype not in (np.float32, np.float64):
        co = co.astype(np.float64)
    co = np.ascontiguousarray(co)
    n = co.shape[1]
    field = _engine.DeviceField(n, config.device)
    _engine.check(_engine.lib().sdfk_eval_host_resident(prog.handle, _engine._ptr(co), 0 if co.dtype == np.float32 else 1, n,
                                                        n, _engine._vp(field.ptr), config.device, config.mode),
                  "sdfk_eval_host_resident")
    return field


def select_geometry(node, co, threshold=0.0):
    """numpy.flatnonzero(node.create(co) <= threshold) — for a tree that lowers to ONE program without the field ever
    existing: the evaluation kernels write flag bits, the compaction reads them (csrc/sdfk_fieldops.inc, "fused selection").
    Trees with grid-neighbourhood operators or user callables are evaluated to a resident field first."""
    from ._lower import NeedsStage
    try:
        lowered = lower_geometry(node)
    except NeedsStage:
        field = evaluate_geometry(node, co, resident=True)
        try:
            return field.select(threshold)
        finally:
            field.free()
    prog = program_for(lowered)
    axes = getattr(co, "grid_axes", None) if config.grid_fast_path else None
    if axes is not None:
        return prog.select_grid(axes, threshold, device=config.device, mode=config.mode)
    return prog.select_host(co, threshold, device=config.device, mode=config.mode)


def evaluate_geometry(node, co, resident=False):
    return _evaluate(lambda **kw: lower_geometry(node, **kw), co, node.modified_object, resident)


def evaluate_expr(expr, co, params):
    return _evaluate(lambda **kw: lower_expression(expr, params, **kw), co, expr)


def _evaluate(lower, co, root=None, resident=False):
    from ._lower import NeedsStage
    try:
        lowered = lower()
    except NeedsStage:
        return _run_staged(lower, co, root, resident)
    return _run(lowered, co, resident)


# ---------------------------------------------------------------------------------------------------
# pointwise operations on given fields (no tree): the array-level functions of cores/post_processing.py, cores/combine.py
# ---------------------------------------------------------------------------------------------------
def apply_fields(fields, emit):
    """`fields`: 1..32 scalar fields of the same size (ndarrays of any shape, or _engine.DeviceField); `emit(L, regs)`
    emits value instructions on a fresh Lowerer (regs[k] holds field k) and returns the result register. The fields
    enter as auxiliary rows read by V_FIELD instructions, so the arithmetic is that of the same device functions
    (csrc/sdfk_device.h val_* / cmb_*) a tree would run. Returns an ndarray shaped like the first ndarray field, or a
    DeviceField when every field is one."""
    from ._lower import Lowerer
    _engine.require_gpu()
    lib = _engine.lib()
    resident = all(isinstance(f, _engine.DeviceField) for f in fields)
    shape, host, n = None, [], None
    for f in fields:
        if isinstance(f, _engine.DeviceField):
            host.append(None)
            n_f = f.n
        else:
            arr = np.asarray(f)
            if shape is None:
                shape = arr.shape
            host.append(np.ascontiguousarray(arr, dtype=np.float32).ravel())
            n_f = host[-1].size
        if n is not None and n_f != n:
            raise ValueError("operands could not be broadcast together: %d and %d values" % (n, n_f))
        n = n_f
    L = Lowerer()
    regs = []
    for k in range(len(fields)):
        v = L.new_v()
        L.emit("V_FIELD", v, 0, k)
        regs.append(v)
    lowered = L.finish(emit(L, regs))
    if n == 0:
        return np.empty(shape or (0,), dtype=config.output_dtype)
    device = fields[0].device if resident else config.device
    _engine.check(lib.sdfk_set_device(device), "sdfk_set_device")
    vp = _engine._vp
    stride = (n + 63) // 64 * 64
    rows = max(len(fields), 3)                       # the rows double as the (never read) coordinate array
    d_aux = lib.sdfk_malloc(rows * stride * 4)
    out = _engine.DeviceField(n, device)
    if not d_aux:
        raise _engine.SdfkError("apply_fields: out of device memory")
    try:
        for k, f in enumerate(fields):
            row = d_aux + 4 * k * stride
            if host[k] is None:
                f._live()
                _engine.check(lib.sdfk_memcpy_d2d(vp(row), vp(f.ptr), n * 4), "d2d")
            else:
                _engine.check(lib.sdfk_memcpy_h2d(vp(row), _engine._ptr(host[k]), n * 4), "h2d")
        _engine.check(lib.sdfk_eval_device_aux(program_for(lowered).handle, vp(d_aux), n, stride, vp(d_aux), len(fields), stride,
                                               vp(out.ptr), None, config.mode), "sdfk_eval_device_aux")
        _engine.check(lib.sdfk_sync(None), "sdfk_sync")
    finally:
        lib.sdfk_free(vp(d_aux))
    if resident:
        return out
    res = out.numpy()
    out.free()
    if config.output_dtype is not np.float32:
        res = res.astype(config.output_dtype)
    return res.reshape(shape)


def apply_value_op(name, u, args):
    """One of the value operations of _mods.VALUE_OPS (the post-processing wrappers of the reference) on a field."""
    from ._mods import VALUE_OPS
    opname, prm = VALUE_OPS[name]

    def emit(L, regs):
        L.emit(opname, regs[0], regs[0], params=prm(args))
        return regs[0]
    return apply_fields([u], emit)


def apply_grid_op(name, u, args):
    """conv_averaging / conv_edge_detection on a GRID-shaped field (2-D or 3-D array), on the device."""
    from ._ir import ModSDF
    _engine.require_gpu()
    lib = _engine.lib()
    arr = np.asarray(u)
    if arr.ndim not in (2, 3):
        raise ValueError("the field must be a 2-D or 3-D grid; got shape %r" % (arr.shape,))
    n = arr.size
    field = _engine.DeviceField.from_host(arr, config.device)
    d_scratch = lib.sdfk_malloc(max(n, 1) * 4)
    try:
        if not d_scratch:
            raise _engine.SdfkError("out of device memory")
        node = ModSDF(name, dict(args, co_resolution=arr.shape), None)
        _apply_grid_op(lib, node, None, field.ptr, n, None, None, None, d_scratch, shape=arr.shape)
        _engine.check(lib.sdfk_sync(None), "sdfk_sync")
        out = field.numpy()
    finally:
        if d_scratch:
            lib.sdfk_free(_engine._vp(d_scratch))
        field.free()
    if config.output_dtype is not np.float32:
        out = out.astype(config.output_dtype)
    return out.reshape(arr.shape)


# ---------------------------------------------------------------------------------------------------
# staged evaluation: trees with grid-neighbourhood operators (signed, conv_averaging, conv_edge_detection)
# ---------------------------------------------------------------------------------------------------
def _grid_shape(n, resolution):
    """Shape smarter_reshape gives an (n,) field (reference cores/helper_functions.py:96-148)."""
    from .cores.helper_functions import resolution_conversion
    res = [resolution_conversion(int(r)) for r in np.atleast_1d(np.asarray(resolution)).ravel()]
    if len(res) == 1:
        r = res[0]
        for shape in ((r,), (r, r), (r, r, r)):
            if n // int(np.prod(shape)) == 1:
                return shape
        raise ValueError("Cannot reshape the pattern with shape (%d,)" % n)
    if len(res) == 2:
        div = n // (res[0] * res[1])
        shape = (res[0], res[1]) if div == 1 else (res[0], res[1], int(div))
    elif len(res) == 3:
        div = n // (res[0] * res[1] * res[2])
        if div != 1:
            raise ValueError("Cannot reshape the pattern with shape (%d,)" % n)
        shape = tuple(res)
    else:
        raise ValueError("resolution must have 1, 2 or 3 entries")
    if int(np.prod(shape)) != n:
        raise ValueError("cannot reshape array of size %d into shape %r" % (n, shape))
    return shape


def _plan_stages(lower):
    """[(stage program or None, operator node, position key, geometry parameters)], final program, {key: field index}:
    every staged operator gets its input from the stage before it (innermost / first-met operators first). Operators
    that are handed coordinates only (custom_modification, opaque callables) have no stage program of their own.
    Operators are identified by their POSITION in the tree (the same node object can be used at several places)."""
    from ._lower import NeedsStage
    from ._mods import FIELD_STAGE_OPS
    fields, stages = {}, []
    while True:
        try:
            return stages, lower(fields=fields), fields
        except NeedsStage as need:
            node, key = need.expr, need.key
        while True:
            try:
                if node.name in FIELD_STAGE_OPS:
                    prog = lower(fields=fields, stop_at=key)
                    params = prog.stage_params
                else:
                    params = lower(fields=fields, stop_at=key, probe_axis=0).stage_params   # reachable so far?
                    prog = None
                break
            except NeedsStage as inner:            # an operator nested in (or met before) the target: that one first
                node, key = inner.expr, inner.key
        stages.append((prog, node, key, params))
        fields[key] = len(fields)
        if len(fields) > 32:
            raise NotImplementedError("more than 32 staged operators in one tree")


def _apply_grid_op(lib, node, key, d_field, n, lower, fields, points4, d_scratch, shape=None):
    """Run one operator in place on a device field of n points (`shape`: the field is a slab of planes of the grid
    the operator's co_resolution describes)."""
    name, args = node.name, node.args
    shape = _grid_shape(n, args["co_resolution"]) if shape is None else shape
    dims = tuple(shape) + (1,) * (3 - len(shape))
    vp = _engine._vp
    if name == "conv_averaging":
        ks = args["kernel_size"]
        if isinstance(ks, (int, np.integer)):
            ks = (int(ks),) * len(shape)
        ks = tuple(int(k) for k in np.asarray(ks).ravel())
        if len(ks) != len(shape):
            raise ValueError("Dimension of the kernel and the field must match!")
        ks = ks + (1,) * (3 - len(ks))
        _engine.check(lib.sdfk_grid_box_average(vp(d_field), dims[0], dims[1], dims[2], ks[0], ks[1], ks[2],
                                                int(args["iterations"]), vp(d_scratch), None), "sdfk_grid_box_average")
    elif name == "conv_edge_detection":
        _engine.check(lib.sdfk_grid_edge_detect(vp(d_field), dims[0], dims[1], dims[2], vp(d_scratch), None),
                      "sdfk_grid_edge_detect")
    else:                                           # signed, signed_old
        if len(shape) != 3:
            raise ValueError("Dimension of the kernel and the field must match!")      # conv_averaging((2, 2, 1)) on 2-D
        # grid spacings as the operator sees them: coordinate i of the neighbour along axis i minus that of point 0
        seps = []
        for axis in range(3):
            vals = _eval_few(lib, lower(fields=fields, stop_at=key, probe_axis=axis), points4, len(fields))
            seps.append(abs(float(vals[1 + axis]) - float(vals[0])))
        _engine.check(lib.sdfk_grid_signed(vp(d_field), dims[0], dims[1], dims[2], float(np.float32(min(seps))),
                                           0 if name == "signed_old" else 1, vp(d_scratch), None), "sdfk_grid_signed")


def _eval_few(lib, lowered, points, n_aux):
    """A program on a handful of points (fields of earlier stages, if it reads any, are dead code there: zeros)."""
    m = int(points.shape[1])
    vp = _engine._vp
    host = np.zeros((3 + max(n_aux, 1) + 1, 64), dtype=np.float32)
    host[:3, :m] = points
    d = lib.sdfk_malloc(host.nbytes)
    if not d:
        raise _engine.SdfkError("staged evaluation: out of device memory")
    try:
        _engine.check(lib.sdfk_memcpy_h2d(vp(d), _engine._ptr(host), host.nbytes), "h2d")
        d_aux, d_out = d + 4 * 3 * 64, d + 4 * (3 + max(n_aux, 1)) * 64
        _engine.check(lib.sdfk_eval_device_aux(program_for(lowered).handle, vp(d), m, 64, vp(d_aux), max(n_aux, 1), 64,
                                               vp(d_out), None, config.mode), "sdfk_eval_device_aux")
        _engine.check(lib.sdfk_sync(None), "sdfk_sync")
        out = np.empty(m, dtype=np.float32)
        _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(out), vp(d_out), m * 4), "d2h")
        return out
    finally:
        lib.sdfk_free(vp(d))


def _edge_detection_is_outermost(expr):
    """conv_edge_detection returns the GRID-shaped array (reference cores/modifications.py:1631-1634: no flatten);
    when nothing but pointwise value operations follows, that is the shape create() hands back."""
    from ._ir import ModSDF, NodeSDF
    while True:
        if isinstance(expr, NodeSDF):
            expr = expr.obj.modified_object
        elif isinstance(expr, ModSDF):
            if expr.name == "conv_edge_detection":
                return expr
            expr = expr.inner
        else:
            return None


def _run_host_op(lib, node, key, params, lowered, k, n, stride, d_aux, d_out, run_program, lower, known):
    """custom_post_process / custom_modification / an opaque SDF callable: user code on the host between two GPU
    stages. The field (or the coordinates the closure is handed) comes back over PCIe, the result goes up as
    auxiliary field k. Functional completeness, not a fast path."""
    vp = _engine._vp
    row = d_aux + 4 * k * stride
    buf = np.empty(n, dtype=np.float32)

    def fetch(d_src):
        _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(buf), vp(d_src), n * 4), "d2h")
        return buf.astype(np.float64)

    if node.name == "custom_post_process":
        run_program(lowered, row, k)
        result = node.args["function"](fetch(row), *node.args["parameters"])
    else:
        co_here = np.empty((3, n), dtype=np.float64)
        for axis in range(3):
            run_program(lower(fields=known, stop_at=key, probe_axis=axis), d_out, k)
            co_here[axis] = fetch(d_out)
        if node.name == "custom_modification":
            inner = node.inner

            def geo_object(co_, *p):               # the closure the reference hands to the user's modification
                return evaluate_expr(inner, np.asarray(co_), p).astype(np.float64)
            result = node.args["modification"](geo_object, co_here, params, node.args["modification_parameters"])
        else:
            result = node.fn(co_here, *params)
    result = np.asarray(result, dtype=np.float32)
    if result.shape != (n,):
        raise ValueError("%s returned an array of shape %r, expected (%d,)" % (node.name, result.shape, n))
    result = np.ascontiguousarray(result)
    _engine.check(lib.sdfk_memcpy_h2d(vp(row), _engine._ptr(result), n * 4), "h2d")


def _run_staged(lower, co, root=None, resident=False):
    _engine.require_gpu()
    lib = _engine.lib()
    _engine.check(lib.sdfk_set_device(config.device), "sdfk_set_device")
    stages, final, fields = _plan_stages(lower)
    n = int(co.shape[1])
    axes = getattr(co, "grid_axes", None) if config.grid_fast_path else None
    vp = _engine._vp
    stride = (n + 63) // 64 * 64
    d_aux = lib.sdfk_malloc(len(stages) * stride * 4)
    d_out = lib.sdfk_malloc(stride * 4)
    d_co = None
    if not d_aux or not d_out:
        raise _engine.SdfkError("staged evaluation: out of device memory")
    try:
        if axes is None:
            host = np.ascontiguousarray(co, dtype=np.float32)
            if host.shape[0] != 3:
                raise ValueError("coordinates must have shape (3, N)")
            d_co = lib.sdfk_malloc(3 * stride * 4)
            if not d_co:
                raise _engine.SdfkError("staged evaluation: out of device memory")
            for r in range(3):
                _engine.check(lib.sdfk_memcpy_h2d(vp(d_co + 4 * r * stride), _engine._ptr(host[r]), n * 4), "h2d")
        ax = None if axes is None else [np.ascontiguousarray(a, dtype=np.float32) for a in axes]

        def run_program(lowered, d_dst, n_aux):
            prog = program_for(lowered)
            if ax is not None:
                _engine.check(lib.sdfk_eval_grid_aux(prog.handle, _engine._ptr(ax[0]), ax[0].size, _engine._ptr(ax[1]),
                                                     ax[1].size, _engine._ptr(ax[2]), ax[2].size, 0, n, vp(d_aux), n_aux,
                                                     stride, vp(d_dst), None, config.mode), "sdfk_eval_grid_aux")
            else:
                _engine.check(lib.sdfk_eval_device_aux(prog.handle, vp(d_co), n, stride, vp(d_aux), n_aux, stride,
                                                       vp(d_dst), None, config.mode), "sdfk_eval_device_aux")
                _engine.check(lib.sdfk_sync(None), "sdfk_sync")

        # the 4 points `signed` reads its grid spacings from: point 0 and its neighbour along each axis
        points4 = None
        known = {}
        from ._mods import HOST_OPS
        for k, (lowered, node, key, params) in enumerate(stages):
            if node.name in HOST_OPS:
                _run_host_op(lib, node, key, params, lowered, k, n, stride, d_aux, d_out, run_program, lower, known)
                known[key] = k
                continue
            run_program(lowered, d_aux + 4 * k * stride, k)
            if node.name in ("signed", "signed_old") and points4 is None:
                shape = _grid_shape(n, node.args["co_resolution"])
                if len(shape) == 3:
                    idx = [0, shape[1] * shape[2], shape[2], 1]
                    points4 = np.stack([np.asarray(co[r])[idx] for r in range(3)]).astype(np.float64)
            _apply_grid_op(lib, node, key, d_aux + 4 * k * stride, n, lower, known, points4, d_out)   # d_out doubles as scratch
            known[key] = k
        run_program(final, d_out, len(stages))
        if resident:
            field = _engine.DeviceField(n, config.device)
            _engine.check(lib.sdfk_memcpy_d2d(vp(field.ptr), vp(d_out), n * 4), "d2d")
            return field
        out = np.empty(n, dtype=np.float32)
        _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(out), vp(d_out), n * 4), "d2h")
    finally:
        for d in (d_aux, d_out, d_co):
            if d:
                lib.sdfk_free(vp(d))
    if config.output_dtype is not np.float32:
        out = out.astype(config.output_dtype)
    edge = _edge_detection_is_outermost(root) if root is not None else None
    if edge is not None:
        out = out.reshape(_grid_shape(n, edge.args["co_resolution"]))
    return out


# ---------------------------------------------------------------------------------------------------
# staged evaluation of ONE SLAB of a grid (multi-GPU: aegolius_amd.distributed)
# ---------------------------------------------------------------------------------------------------
def _halo_planes(stages, grid_shape, n_total, exchange=False):
    """Planes of halo a slab needs so that its interior is exact after every operator of `stages`: the reach
    of each box / edge kernel along the first axis, times its iterations, summed over the chain. `signed` needs no
    halo — its scan lines need the whole grid's boundary bits, which the ranks exchange (`exchange`)."""
    halo = 0
    for _low, node, _key, _params in stages:
        if exchange and node.name in ("signed", "signed_old"):
            continue
        if node.name not in ("conv_averaging", "conv_edge_detection"):
            raise NotImplementedError(
                "%r cannot run on a slab of the grid (it scans whole grid lines or runs user code on the whole "
                "field): evaluate the tree with geometry.create(co) on a single GPU" % (node.name,))
        shape = _grid_shape(n_total, node.args["co_resolution"])
        if tuple(shape) != tuple(grid_shape):
            raise ValueError("co_resolution %r of %s does not describe the evaluated grid %r"
                             % (node.args["co_resolution"], node.name, tuple(grid_shape)))
        if node.name == "conv_edge_detection":
            halo += 1
        else:
            ks = node.args["kernel_size"]
            k0 = int(ks) if isinstance(ks, (int, np.integer)) else int(np.asarray(ks).ravel()[0])
            halo += (k0 // 2) * int(node.args["iterations"])
    return halo


def evaluate_slab_staged(lower, axes, plane0, planes, out_ptr, comm=None):
    """Planes [plane0, plane0 + planes) of the grid spanned by `axes` (three per-axis tables) for a tree with
    conv_averaging / conv_edge_detection / signed nodes, written to the device buffer `out_ptr` (planes * n1 * n2 fp32).
    The slab is evaluated together with a halo of neighbouring planes (recomputed locally: the per-point stages are
    cheap, nothing is exchanged); at the true ends of the grid the operators' reflect boundary applies, at the cut
    ends the contaminated halo planes are dropped. Bit-identical to the same planes of a whole-grid evaluation.

    `signed` scans whole grid lines: it needs `comm`, the exchange between the slabs — an object with
    `allreduce_min(float) -> float` and `allgather_bytes(ptr, start, count, total) -> ptr` (device pointers; the
    bytes [start, start + count) of a `total`-byte array, returns the whole array) — see distributed._TorchComm."""
    _engine.require_gpu()
    lib = _engine.lib()
    vp = _engine._vp
    ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
    n0, n1, n2 = (int(a.size) for a in ax)
    flat2d = n2 == 1
    grid_shape = (n0, n1) if flat2d else (n0, n1, n2)
    plane = n1 * n2
    stages, final, _fields = _plan_stages(lower)
    halo = _halo_planes(stages, grid_shape, n0 * plane, exchange=comm is not None)
    e0, e1 = max(0, plane0 - halo), min(n0, plane0 + planes + halo)
    n = (e1 - e0) * plane
    ext_shape = (e1 - e0, n1) if flat2d else (e1 - e0, n1, n2)
    stride = (n + 63) // 64 * 64
    d_aux = lib.sdfk_malloc(max(1, len(stages)) * stride * 4)
    d_tmp = lib.sdfk_malloc(stride * 4)
    d_mask = None
    if not d_aux or not d_tmp:
        raise _engine.SdfkError("staged slab evaluation: out of device memory")
    try:
        def run_program(lowered, d_dst, n_aux):
            _engine.check(lib.sdfk_eval_grid_aux(program_for(lowered).handle, _engine._ptr(ax[0]), n0, _engine._ptr(ax[1]), n1,
                                                 _engine._ptr(ax[2]), n2, e0 * plane, n, vp(d_aux), n_aux, stride, vp(d_dst),
                                                 None, config.mode), "sdfk_eval_grid_aux")
        known = {}
        for k, (lowered, node, key, _params) in enumerate(stages):
            row = d_aux + 4 * k * stride
            run_program(lowered, row, k)
            if node.name in ("signed", "signed_old"):
                if flat2d:
                    raise ValueError("Dimension of the kernel and the field must match!")   # conv_averaging((2, 2, 1)) on 2-D
                shape = _grid_shape(n0 * plane, node.args["co_resolution"])
                if tuple(shape) != tuple(grid_shape):
                    raise ValueError("co_resolution %r of %s does not describe the evaluated grid %r"
                                     % (node.args["co_resolution"], node.name, tuple(grid_shape)))
                # "already signed" (C/modifications.py:236-237): the minimum over the whole grid = over every rank's
                # planes (the halo planes are grid planes too, with their owners' values: the union is the grid)
                own = row + 4 * (plane0 - e0) * plane
                fmin = _engine._c.c_float(0.0)
                if n > 0:
                    _engine.check(lib.sdfk_field_min(vp(row), n, _engine.ctypes.byref(fmin), None), "sdfk_field_min")
                gmin = comm.allreduce_min(float(fmin.value) if n > 0 else float("inf"))
                if not gmin < 0.0:
                    # grid spacings as the operator sees them: coordinate i of the neighbour along axis i minus that of point 0
                    points4 = np.asarray([[ax[0][0], ax[0][1], ax[0][0], ax[0][0]], [ax[1][0], ax[1][0], ax[1][1], ax[1][0]],
                                          [ax[2][0], ax[2][0], ax[2][0], ax[2][1]]], dtype=np.float64)
                    seps = []
                    for axis in range(3):
                        vals = _eval_few(lib, lower(fields=known, stop_at=key, probe_axis=axis), points4, len(known))
                        seps.append(abs(float(vals[1 + axis]) - float(vals[0])))
                    sep = float(np.float32(min(seps)))
                    d_mask = lib.sdfk_malloc(max(1, planes * plane))
                    if not d_mask:
                        raise _engine.SdfkError("staged slab evaluation: out of device memory")
                    _engine.check(lib.sdfk_grid_boundary_mask(vp(own), planes * plane, sep, vp(d_mask), None), "sdfk_grid_boundary_mask")
                    _engine.check(lib.sdfk_sync(None), "sdfk_sync")
                    d_full = comm.allgather_bytes(d_mask, plane0 * plane, planes * plane, n0 * plane)
                    _engine.check(lib.sdfk_grid_signed_slab(vp(row), e0, e1 - e0, vp(d_full), n0, n1, n2,
                                                            0 if node.name == "signed_old" else 1, None, None),
                                  "sdfk_grid_signed_slab")
                    lib.sdfk_free(vp(d_mask))
                    d_mask = None
            else:
                _apply_grid_op(lib, node, key, row, n, lower, known, None, d_tmp, shape=ext_shape)
            known[key] = k
        run_program(final, d_tmp, len(stages))
        # hipMemcpy device-to-device through the plumbing entry point (any direction works for device pointers)
        _engine.check(lib.sdfk_memcpy_d2d(vp(out_ptr), vp(d_tmp + 4 * (plane0 - e0) * plane), planes * plane * 4), "d2d")
    finally:
        lib.sdfk_free(vp(d_aux))
        lib.sdfk_free(vp(d_tmp))
        if d_mask:
            lib.sdfk_free(vp(d_mask))
