"""Vector-field programs: the reference's closures over (3, N) arrays as one GPU kernel.

The reference builds a vector field as `vf(p, *params)` wrapped by one closure per modification
(cores/modifications.py:1666-1975). Here the same calls record a `VecClosure` — a leaf (a field definition of
cores/vector_functions.py, or any user callable) plus the list of modifications — and calling it lowers the chain
to a program for `sdfk_vec_eval_host` (include/sdfk.h): numbers and 3-vectors become immediates, NumPy arrays the
reference would broadcast become rows of a `streams` array. There is no NumPy evaluation of a chain here; without
the extension or a GPU the call raises.
"""
import ctypes

import numpy as np

from . import _engine

OP = {name: k for k, name in enumerate(
    ["INIT_P", "INIT_SPHERICAL", "INIT_CYLINDRICAL", "INIT_RADIAL_SPH", "INIT_RADIAL_CYL", "INIT_VORTEX", "INIT_AAR",
     "INIT_AAV", "INIT_CONST", "INIT_STREAM", "ADD", "SUB", "MUL", "ROT_Z", "ROT_X", "ROT_Y", "ROT_THETA", "ROT_AXIS",
     "REVOLVE_X", "REVOLVE_Y", "REVOLVE_Z", "NORMALIZE"])}
K_NONE, K_IMM1, K_IMM3, K_ROW1, K_ROW3, K_P = range(6)
OUT_KINDS = {"vector": 0, "x": 1, "y": 2, "z": 3, "phi": 4, "theta": 5, "length": 6}
MAX_INSTR = 64


class VecInstr(ctypes.Structure):
    _fields_ = [("op", ctypes.c_int32), ("src", ctypes.c_int32 * 2), ("imm", ctypes.c_float * 4)]


class _Builder:
    """Instruction list + stream rows of one evaluation on N points."""

    def __init__(self, n, p=None):
        self.n = n
        self.p = p
        self.instr = []
        self.rows = []
        self._seen = {}

    def _row(self, a):
        if isinstance(a, tuple):                                 # (device pointer, owner): a row that is already in HBM
            self.rows.append(a)
            return len(self.rows) - 1
        a = np.ascontiguousarray(a, dtype=np.float32).reshape(-1)
        if a.size != self.n:
            raise ValueError("operands could not be broadcast together with shapes (3,%d) (%d,)" % (self.n, a.size))
        self.rows.append(a)
        return len(self.rows) - 1

    def rows_of(self, arr, count):
        """First stream row of a (count, N) operand; identical arrays are uploaded once."""
        key = (id(arr), count)
        if key not in self._seen and isinstance(arr, (_engine.DeviceField, _engine.DeviceVectorField)):
            if arr.n != self.n:
                raise ValueError("operands could not be broadcast together with shapes (3,%d) (%d,)" % (self.n, arr.n))
            if isinstance(arr, _engine.DeviceField):
                first = self._row((arr.ptr, arr))
            else:
                first = self._row((arr.row_ptr(0), arr))
                self._row((arr.row_ptr(1), arr))
                self._row((arr.row_ptr(2), arr))
            self._seen[key] = (first, arr)
        if key not in self._seen:
            a = np.asarray(arr)
            first = self._row(a if count == 1 else a[0])
            for r in range(1, count):
                self._row(a[r])
            self._seen[key] = (first, arr)                      # keep `arr` alive: ids are only unique while it lives
        return self._seen[key][0]

    def emit(self, name, a=None, b=None):
        ins = VecInstr()
        kinds = [K_NONE, K_NONE]
        for w, operand in enumerate((a, b)):
            if operand is None:
                continue
            kind, value = operand
            kinds[w] = kind
            if kind == K_IMM1:
                ins.imm[3 if w else 0] = float(value)
            elif kind == K_IMM3:
                for k in range(3):
                    ins.imm[k] = float(value[k])
            elif kind in (K_ROW1, K_ROW3):
                ins.src[w] = int(value)
        ins.op = OP[name] | kinds[0] << 8 | kinds[1] << 12
        self.instr.append(ins)
        if len(self.instr) > MAX_INSTR:
            raise ValueError("vector-field chains are limited to %d modifications" % (MAX_INSTR - 1))

    # ---- operand classification: what NumPy's broadcasting does in the reference --------------------------------------
    def number_or_row(self, value, what):
        """A spatially independent or dependent number (angles): scalar or (N,)."""
        dev = self.device_operand(value, 1)
        if dev is not None:
            return dev
        a = np.asarray(value, dtype=np.float64)
        if a.size == 1:
            return K_IMM1, float(a.reshape(-1)[0])
        a = np.squeeze(a)
        if a.ndim == 1 and a.size == self.n:
            return K_ROW1, self.rows_of(value, 1)
        raise ValueError("%s must be a number or an array of %d values; got shape %r" % (what, self.n, np.shape(value)))

    def addend(self, value):
        """add_vectors / subtract_vectors (cores/vector_modification_functions.py:23-36): size 3 -> one vector for every
        point, anything else is broadcast against (3, N)."""
        dev = self.device_operand(value, None)
        if dev is not None:
            return dev
        a = np.asarray(value, dtype=np.float64)
        if a.size == 3:
            if a.shape not in ((3,), (1, 3)) and self.n != 3:    # np.add(vec.T, add_vec): (N, 3) against a column fails
                raise ValueError("operands could not be broadcast together with shapes (%d,3) %s " % (self.n, a.shape))
            return K_IMM3, a.reshape(-1)
        return self.broadcast(a, value)

    def device_operand(self, value, rows):
        """A DeviceField (one number per point) or DeviceVectorField (one vector per point) used as an operand."""
        if isinstance(value, tuple) and len(value) == 1:
            value = value[0]                                     # the ready-made classes wrap their angle in a 1-tuple
        if isinstance(value, _engine.DeviceField) and rows in (1, None):
            return K_ROW1, self.rows_of(value, 1)
        if isinstance(value, _engine.DeviceVectorField) and rows in (3, None):
            return K_ROW3, self.rows_of(value, 3)
        if isinstance(value, (_engine.DeviceField, _engine.DeviceVectorField)):
            raise ValueError("a %s cannot be used here" % type(value).__name__)
        return None

    def broadcast(self, a, value):
        """numpy broadcasting of an operand against (3, N)."""
        if a.size == 1:
            return K_IMM1, float(a.reshape(-1)[0])
        if a.shape in ((self.n,), (1, self.n)):
            return K_ROW1, self.rows_of(value, 1)
        if a.shape == (3, self.n):
            return K_ROW3, self.rows_of(value, 3)
        if a.shape == (3, 1):
            return K_IMM3, a.reshape(-1)
        raise ValueError("operands could not be broadcast together with shapes (3,%d) %r" % (self.n, a.shape))

    def vectors(self, value, what):
        """One 3-vector or one per point (rotation axes)."""
        dev = self.device_operand(value, 3)
        if dev is not None:
            return dev
        a = np.asarray(value, dtype=np.float64)
        if a.size == 3:
            return K_IMM3, a.reshape(-1)
        if a.shape == (3, self.n):
            return K_ROW3, self.rows_of(value, 3)
        raise ValueError("%s must be a 3-vector or a (3, %d) array; got shape %r" % (what, self.n, a.shape))

    def coordinates(self, co):
        """The coordinate cloud of a revolution: the evaluation's own input when it is the same array."""
        if co is self.p:
            return K_P, None
        dev = self.device_operand(co, 3)
        if dev is not None:
            return dev
        a = np.asarray(co)
        if a.shape != (3, self.n):
            raise ValueError("the coordinates of a revolution must have shape (3, %d); got %r" % (self.n, a.shape))
        return K_ROW3, self.rows_of(co, 3)


# ---- leaves: the field definitions of cores/vector_functions.py ------------------------------------------------------
def _no_params(name):
    def init(b, params):
        b.emit(name)
    return init


def _angle_param(name, what):
    def init(b, params):
        if len(params) != 1:
            raise TypeError("%s() missing 1 required positional argument: %r" % (what, "gamma" if "awn" in what else "alpha"))
        b.emit(name, b.number_or_row(params[0], "the angle"))
    return init


def _const(vec):
    def init(b, params):
        b.emit("INIT_CONST", (K_IMM3, vec))
    return init


LEAVES = {
    "cartesian_define": _no_params("INIT_P"),
    "spherical_define": _no_params("INIT_SPHERICAL"),
    "cylindrical_define": _no_params("INIT_CYLINDRICAL"),
    "radial_vector_field_spherical": _no_params("INIT_RADIAL_SPH"),
    "radial_vector_field_cylindrical": _no_params("INIT_RADIAL_CYL"),
    "vortex_vector_field_cylindrical": _no_params("INIT_VORTEX"),
    "aar_vector_field_cylindrical": _angle_param("INIT_AAR", "aar_vector_field_cylindrical"),
    "aav_vector_field_cylindrical": _angle_param("INIT_AAV", "aav_vector_field_cylindrical"),
    "x_vector_field": _const((1.0, 0.0, 0.0)),
    "y_vector_field": _const((0.0, 1.0, 0.0)),
    "z_vector_field": _const((0.0, 0.0, 1.0)),
}
# exact arity of the definitions that take none of the extra parameters (the reference's signatures are f(p))
STRICT_ARITY = {"cartesian_define", "spherical_define", "cylindrical_define"}


def _apply_mod(b, name, args):
    if name in ("add", "subtract"):
        b.emit("ADD" if name == "add" else "SUB", b.addend(args[0]))
    elif name == "rescale":
        dev = b.device_operand(args[0], None)
        b.emit("MUL", dev if dev is not None else b.broadcast(np.asarray(args[0], dtype=np.float64), args[0]))
    elif name in ("rotate_phi", "rotate_z"):
        b.emit("ROT_Z", b.number_or_row(args[0], "the angle"))
    elif name == "rotate_x":
        b.emit("ROT_X", b.number_or_row(args[0], "the angle"))
    elif name == "rotate_y":
        b.emit("ROT_Y", b.number_or_row(args[0], "the angle"))
    elif name == "rotate_theta":
        b.emit("ROT_THETA", b.number_or_row(args[0], "the angle"))
    elif name == "rotate_axis":
        b.emit("ROT_AXIS", b.vectors(args[0], "the axis"), b.number_or_row(args[1], "the angle"))
    elif name in ("revolution_x", "revolution_y", "revolution_z"):
        b.emit("REVOLVE_" + name[-1].upper(), b.coordinates(args[0]))
    elif name == "normalize":
        b.emit("NORMALIZE")
    else:
        raise ValueError("unknown vector modification %r" % (name,))


class VecClosure:
    """What the reference's `vf` / `new_vf` closures are here: callable as `closure(p, *params)`, evaluated on the GPU."""

    def __init__(self, leaf, mods=()):
        self.leaf = leaf
        self.mods = tuple(mods)

    def then(self, name, *args):
        return VecClosure(self.leaf, self.mods + ((name, args),))

    def __call__(self, p, *params):
        return evaluate(self, p, params, "vector")


def as_closure(fn):
    return fn if isinstance(fn, VecClosure) else VecClosure(fn)


def _leaf_name(fn):
    return getattr(fn, "_vec_leaf", None)


def evaluate(closure, p, params, out="vector", resident=False):
    """closure(p, *params) of the reference -> (3, N) array (or (N,) for a component / angle / length read-out).
    `resident`: leave the result in HBM (DeviceVectorField / DeviceField). `p` and any per-point operand may already
    live there (DeviceVectorField / DeviceField): nothing of theirs crosses PCIe."""
    from ._eval import config
    closure = as_closure(closure)
    leaf = closure.leaf
    inner = leaf
    prefix = ()
    while isinstance(inner, VecClosure):                         # a closure used as the leaf of another field
        prefix = inner.mods + prefix
        inner = inner.leaf
    mods = prefix + closure.mods
    name = _leaf_name(inner)
    if name == "from_sdf" and isinstance(p, _engine.DeviceField) and (mods or resident):
        from ._eval import _grid_shape                           # a resident SDF: its gradient never leaves the device
        shape = _grid_shape(p.n, params[0])
        if len(shape) != 3:
            raise NotImplementedError("modifications of a from_sdf field need a 3-D grid")
        p_arr, grid_axes = p.gradient_resident(shape), None
        b = _Builder(p.n, None)
        b.emit("INIT_P")
    elif name == "from_sdf":                                     # VectorFieldFromSDF: p is the (N,) field
        from .cores.vector_functions import from_sdf
        start = np.asarray(from_sdf(p, *params))
        if not mods and out == "vector":
            return start
        if start.shape[0] != 3:
            raise NotImplementedError("modifications of a from_sdf field need a 3-D grid")
        p_arr, grid_axes = start, None
        b = _Builder(start.shape[1], None)
        b.emit("INIT_P")
    elif name in LEAVES:
        if name in STRICT_ARITY and params:
            raise TypeError("%s() takes 1 positional argument but %d were given" % (name, 1 + len(params)))
        p_arr = p if isinstance(p, (np.ndarray, _engine.DeviceVectorField)) else np.asarray(p, dtype=np.float64)
        if len(p_arr.shape) != 2 or p_arr.shape[0] != 3:
            raise ValueError("vector fields take a (3, N) array; got shape %r" % (p_arr.shape,))
        grid_axes = getattr(p, "grid_axes", None) if config.grid_fast_path else None
        b = _Builder(p_arr.shape[1], p)
        LEAVES[name](b, params)
    elif name in ("hyperbolic_vector_field_cylindrical", "awn_vector_field_cylindrical"):
        # the reference calls cylindrical_define(1, alpha, zeros) — three arguments to a one-argument function
        # (cores/vector_functions.py:58-68) — so these two definitions raise for every input; kept as they are
        raise TypeError("cylindrical_define() takes 1 positional argument but 3 were given")
    else:                                                        # user code: evaluated on the host, then the chain
        start = np.asarray(inner(p, *params))
        if not mods and out == "vector":
            return start
        if start.ndim != 2 or start.shape[0] != 3:
            raise ValueError("a vector field function must return a (3, N) array; got shape %r" % (start.shape,))
        p_arr, grid_axes = start, None
        b = _Builder(start.shape[1], None)
        b.emit("INIT_P")
    for mod_name, args in mods:
        _apply_mod(b, mod_name, args)
    on_device = resident or isinstance(p_arr, _engine.DeviceVectorField) or any(isinstance(r, tuple) for r in b.rows)
    if on_device:
        return _run_device(b, p_arr, grid_axes, OUT_KINDS[out], config, resident)
    return _run(b, p_arr, grid_axes, OUT_KINDS[out], config)


def evaluate_slab(field, axes, start, count, out="vector", resident=False):
    """Points [start, start + count) of field.create(generate_grid cloud of `axes`) — the multi-GPU path
    (aegolius_amd.distributed.vector_field_sharded). The chain is lowered for the WHOLE cloud (operand shapes are
    checked against N) and run on the slab: coordinates expanded on the device from the tables, per-point operands
    sliced on their way up. `field` is a VectorField whose leaf is one of the built-in definitions."""
    from ._eval import config
    closure = as_closure(field.vf)
    mods, inner = closure.mods, closure.leaf
    while isinstance(inner, VecClosure):
        mods, inner = inner.mods + mods, inner.leaf
    name = _leaf_name(inner)
    if name not in LEAVES:
        raise NotImplementedError("sharded evaluation takes the built-in field definitions (got %r)" % (name or inner,))
    n_total = int(np.prod([np.asarray(a).size for a in axes]))
    if not (0 <= start and start + count <= n_total):
        raise ValueError("slab [%d, %d) outside the grid of %d points" % (start, start + count, n_total))
    whole = _GridCloud(n_total)
    b = _Builder(n_total, whole)
    LEAVES[name](b, field._vf_parameters)
    for mod_name, args in mods:
        _apply_mod(b, mod_name, [whole if getattr(a, "grid_axes", None) is not None else a for a in args])
    return _run_device(b, None, axes, OUT_KINDS[out], config, resident, (int(start), int(count)))


class _GridCloud:
    """Stands for the generate_grid cloud itself while a chain is lowered for a slab (revolutions about its axes)."""

    def __init__(self, n):
        self.shape = (3, n)


def _run_device(b, p_arr, grid_axes, out_kind, config, resident, slab=None):
    """The chain on whole device arrays through sdfk_vec_eval_device: p is expanded from the grid tables, taken from a
    DeviceVectorField or uploaded; operand rows that already live in HBM are copied device-to-device into the streams
    array, host rows are uploaded; the result stays there when `resident`."""
    _engine.require_gpu()
    lib, vp = _engine.lib(), _engine._vp
    first, n = slab if slab is not None else (0, b.n)
    _engine.check(lib.sdfk_set_device(config.device), "sdfk_set_device")
    prog = (VecInstr * len(b.instr))(*b.instr)
    stride = (n + 63) // 64 * 64
    own_p = None
    d_streams = None
    result = _engine.DeviceVectorField(n, config.device) if out_kind == 0 else _engine.DeviceField(stride, config.device)
    try:
        if isinstance(p_arr, _engine.DeviceVectorField):
            d_p, p_stride = p_arr.ptr, p_arr.stride
        else:
            own_p = _engine.DeviceVectorField(n, config.device)
            d_p, p_stride = own_p.ptr, own_p.stride
            if grid_axes is not None:
                _engine.grid_fill(d_p, p_stride, grid_axes, first, n)
            else:
                host = np.ascontiguousarray(p_arr, dtype=np.float32)
                for r in range(3):
                    if n:
                        _engine.check(lib.sdfk_memcpy_h2d(vp(own_p.row_ptr(r)), _engine._ptr(host[r]), n * 4), "h2d")
        if b.rows:
            d_streams = lib.sdfk_malloc(len(b.rows) * stride * 4)
            if not d_streams:
                raise _engine.SdfkError("vector chain: out of device memory")
            for k, row in enumerate(b.rows):
                dst = vp(d_streams + 4 * k * stride)
                if isinstance(row, tuple):
                    if n:
                        _engine.check(lib.sdfk_memcpy_d2d(dst, vp(row[0] + 4 * first), n * 4), "d2d")
                elif n:
                    _engine.check(lib.sdfk_memcpy_h2d(dst, _engine._ptr(row[first:first + n]), n * 4), "h2d")
        out_stride = result.stride if out_kind == 0 else stride
        _engine.check(lib.sdfk_vec_eval_device(prog, len(b.instr), vp(d_p), n, p_stride, vp(d_streams) if d_streams else None,
                                               len(b.rows), stride, out_kind, vp(result.ptr), out_stride, None),
                      "sdfk_vec_eval_device")
        _engine.check(lib.sdfk_sync(None), "sdfk_sync")
    finally:
        if own_p is not None:
            own_p.free()
        if d_streams:
            lib.sdfk_free(vp(d_streams))
    if out_kind != 0:
        result.n = n                                             # allocated with the padded length: 16-byte row ends
    if resident:
        return result
    out = result.numpy()
    result.free()
    if config.output_dtype is not np.float32:
        out = out.astype(config.output_dtype)
    return out


def _run(b, p_arr, grid_axes, out_kind, config):
    _engine.require_gpu()
    lib = _engine.lib()
    n = b.n
    prog = (VecInstr * len(b.instr))(*b.instr)
    streams = np.stack(b.rows) if b.rows else None
    out = np.empty((3, n) if out_kind == 0 else (n,), dtype=np.float32)
    if grid_axes is not None:
        ax = [np.ascontiguousarray(a, dtype=np.float32) for a in grid_axes]
        host, dtype = None, 0
    else:
        ax = [None, None, None]
        host = p_arr
        if host.dtype not in (np.float32, np.float64):
            host = host.astype(np.float64)
        host = np.ascontiguousarray(host)
        dtype = 0 if host.dtype == np.float32 else 1
    _engine.check(lib.sdfk_vec_eval_host(
        prog, len(b.instr), _engine._ptr(host) if host is not None else None, dtype, n,
        _engine._ptr(ax[0]) if ax[0] is not None else None, ax[0].size if ax[0] is not None else 0,
        _engine._ptr(ax[1]) if ax[1] is not None else None, ax[1].size if ax[1] is not None else 0,
        _engine._ptr(ax[2]) if ax[2] is not None else None, ax[2].size if ax[2] is not None else 0,
        _engine._ptr(streams) if streams is not None else None, len(b.rows), out_kind, _engine._ptr(out), config.device),
        "sdfk_vec_eval_host")
    if config.output_dtype is not np.float32:
        out = out.astype(config.output_dtype)
    return out


def program_array(instr):
    """ctypes array of a lower_only() instruction list."""
    prog = (VecInstr * len(instr))()
    for k, (op, ka, kb, src, imm) in enumerate(instr):
        prog[k].op = op | ka << 8 | kb << 12
        prog[k].src[0], prog[k].src[1] = src
        for j in range(4):
            prog[k].imm[j] = imm[j]
    return prog


def lower_only(closure, p, params=()):
    """(instructions, stream rows) of closure(p, *params) without running it — CPU tests of the lowering."""
    closure = as_closure(closure)
    name = _leaf_name(closure.leaf)
    p_arr = np.asarray(p)
    b = _Builder(p_arr.shape[1], p)
    LEAVES[name](b, params)
    for mod_name, args in closure.mods:
        _apply_mod(b, mod_name, args)
    return [(i.op & 255, (i.op >> 8) & 15, (i.op >> 12) & 15, tuple(i.src), tuple(i.imm)) for i in b.instr], b.rows
