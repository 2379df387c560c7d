"""Lowering: expression tree -> sdfk register-machine program.

Everything that does not depend on the point is computed HERE, in float64, and rounded once to
fp32 (R^T/s, R^T t, reciprocals, sin/cos of fixed angles, instancing frames ...). The kernels only
see fp32 constants in SGPRs.

Coordinate registers and the reference's array aliasing
-------------------------------------------------------
The reference passes NumPy arrays between closures; three modifications overwrite the array they
were given (`symmetry` cores/modifications.py:951, `rotational_symmetry` :1022-1028,
`axis_revolution` :459) and that mutation is visible to a sibling evaluated later by an enclosing
`displacement` / `recover_volume` / `define_volume` (:798, :343, :366). A coordinate register is
therefore lowered under one of three modes:

  OWNED    nobody reads this register afterwards: any op may overwrite it in place
  ALIASED  an enclosing two-field modification re-reads it afterwards AND, in the reference, it is
           the very same array: in-place modifications write through (visible), array-creating
           modifications take a fresh register
  FROZEN   somebody re-reads it afterwards but the reference would have made a private copy (the
           skipped identity transform of a node): any writer must first move to a fresh register
"""
import numpy as np

from . import _lipschitz as _lip
from . import _ops
from ._ir import CombineSDF, ModSDF, NodeSDF, PrimSDF, SDFExpr, UnsupportedSDF

OWNED, ALIASED, FROZEN = 0, 1, 2


class LoweringError(Exception):
    pass


class LoweredProgram:
    __slots__ = ("code", "params", "tables", "result_reg", "n_creg", "n_vreg", "cull_sites", "cull_k", "stage_params")

    def __init__(self, code, params, tables, result_reg, n_creg, n_vreg, cull_sites=None, cull_k=None):
        self.code, self.params, self.tables = code, params, tables
        self.result_reg, self.n_creg, self.n_vreg = result_reg, n_creg, n_vreg
        # brick-culling sites: rows (combiner index, a_start, a_end, b_start, b_end) + K = L_a + L_b
        self.cull_sites = np.zeros((0, 5), dtype=np.uint32) if cull_sites is None else cull_sites
        self.cull_k = np.zeros(0, dtype=np.float32) if cull_k is None else cull_k
        self.stage_params = ()     # stage programs: the geometry parameters the staged operator's closure is called with

    def key(self):
        return (self.code.tobytes(), self.params.tobytes(), self.tables.tobytes(), self.result_reg,
                self.cull_sites.tobytes(), self.cull_k.tobytes())

    @property
    def fits_interpreter(self):
        return self.n_creg <= _ops.INTERP_NC and self.n_vreg <= _ops.INTERP_NV


class NeedsStage(Exception):
    """Lowering met a grid-neighbourhood operator (signed / conv_*) whose input field has not been computed
    by an earlier evaluation stage yet (see _eval._run_staged)."""

    def __init__(self, expr, key=None):
        Exception.__init__(self, "operator %r needs a staged evaluation" % (expr.name,))
        self.expr = expr
        self.key = key          # where in the tree: the same node can occur at several places (Lowerer._stack)


class StageStop(Exception):
    """Raised at the operator a stage program stops at: `vreg` holds the value that stage has to produce."""

    def __init__(self, vreg, params=()):
        Exception.__init__(self)
        self.vreg = vreg
        self.params = tuple(params)   # the geometry parameters the operator's closure is called with


class Lowerer:
    def __init__(self):
        self.code = []
        self.params = []
        self.tables = []            # chunks (float64 arrays), concatenated by finish(); offsets count floats
        self._table_len = 0
        self._c_used = {0}
        self._v_used = set()
        self.n_creg = 1
        self.n_vreg = 0
        self.lip_c = {0: 1.0}      # Lipschitz bound of each register w.r.t. the root point (_lipschitz.py)
        self.lip_v = {}
        self.cull = []             # (combiner index, a_start, a_end, b_start, b_end, K)
        self.fields = {}           # position key of a staged operator -> auxiliary field index (earlier stages)
        self.stop_at = None        # position key of the operator this (stage) program ends at
        self._stack = []           # path from the root to the expression being lowered: the position key of a
                                   # staged operator (the same node object may occur at several places of a tree)
        self.probe_axis = None     # with stop_at: produce coordinate component `probe_axis` there instead
        self._fold_floor = 0       # instructions below this index are never merged into (range boundaries)
        self._affine = {}          # instruction index -> (A (3,3), c (3,)) of an affine coordinate op, float64

    # ---- registers ----
    def new_c(self):
        i = 1
        while i in self._c_used:
            i += 1
        if i > 255:
            raise LoweringError("expression needs more than 256 coordinate registers")
        self._c_used.add(i)
        self.n_creg = max(self.n_creg, i + 1)
        return i

    def free_c(self, i):
        self._c_used.discard(i)

    def new_v(self):
        i = 0
        while i in self._v_used:
            i += 1
        if i > 255:
            raise LoweringError("expression needs more than 256 value registers")
        self._v_used.add(i)
        self.n_vreg = max(self.n_vreg, i + 1)
        return i

    def free_v(self, i):
        self._v_used.discard(i)

    # ---- emission ----
    _AFFINE = ("XFORM", "XLATE", "LIN3", "CSCALE")

    @staticmethod
    def _as_affine(opname, p):
        """q = A p - c of the affine coordinate ops (float64)."""
        p = np.asarray(p, dtype=np.float64)
        if opname == "XFORM":
            return p[:9].reshape(3, 3), p[9:12].copy()
        if opname == "XLATE":
            return np.eye(3), p[:3].copy()
        if opname == "LIN3":
            return p[:9].reshape(3, 3), np.zeros(3)
        return np.eye(3) * p[0], np.zeros(3)      # CSCALE

    def _emit_affine(self, a, b, A, c):
        if np.array_equal(A, np.eye(3)):
            name, prm = "XLATE", c
        else:
            name, prm = "XFORM", np.concatenate([A.ravel(), c])
        self._affine[len(self.code)] = (A, c, self.lip_c.get(b, _lip.INF))
        self.emit(name, a, b, params=prm, _fold=False)

    # float64 limits of the exponent of the strictly positive value maps (C/post_processing.py:380-429, 526-558):
    # exp(t) is non-zero down to t = -745.13; 1 + exp(x) is finite up to x = 709.78
    _EXP_ZERO, _EXP_INF = -745.13, -709.78

    def _fold_positive_map(self, opname, a, b, params):
        """sign(v) / hard_binarization(v, 0) right after a strictly positive value map on the same register: in the
        reference (float64) the map only reaches 0 where exp underflows at an exponent of -745; in fp32 it is 0 from -103
        on, and the sign of the fp32 value would be 0 where the reference has 1 (fuzz seeds 70109 / 70139 / 70197 of
        round 2: off by 1.0 at > 1000 points). The pair becomes ONE operator that decides on the exponent
        (csrc/sdfk_device.h val_expflag). Returns True when it has emitted it."""
        if a != b or not self.code:
            return False
        if opname == "VHARDBIN" and params[0] != 0.0:
            return False
        last = len(self.code) - 1
        if last < self._fold_floor:
            return False

        def positive_map(i, reg):
            """parameter block of the exponent test if instruction i is a positive map applied in place to V[reg]"""
            w, poff = self.code[i]
            name = _ops.OPS[w & 255].name
            if name not in ("VGAUSS", "VCAPEXP", "VSIGMOID") or (w >> 8) & 255 != reg or (w >> 16) & 255 != reg:
                return None
            lp = self.params[poff:poff + 3]
            if not (lp[0] > 0.0 and np.isfinite(lp[0]) and np.isfinite(lp[1])):
                return None
            if name == "VGAUSS":
                return [0.0, lp[1], lp[2], self._EXP_ZERO]
            if name == "VCAPEXP":
                return [1.0, lp[1], 0.0, self._EXP_ZERO]
            return [2.0, lp[1], lp[2], self._EXP_INF]

        block = positive_map(last, a)
        if block is not None:
            # sign: 1 while the map is non-zero, 0 once it has underflowed; hard_binarization(v <= 0): the other way round
            block += [1.0, 0.0] if opname == "VSIGN" else [0.0, 1.0]
            lpoff = self.code[last][1]
            self.code.pop()
            del self.params[lpoff:]
            self.emit("VEXPFLAG", a, a, params=block, _fold=False)
            return True
        # sign(map * g) (recover_volume / define_volume of a mapped field, C/modifications.py:325-369): the sign of the
        # product is the sign of g while the map is non-zero in float64 and 0 once it has underflowed there — the map is
        # replaced IN PLACE by its 1 / 0 flag (new parameters at the end of the table) and product and sign stay
        lw = self.code[last][0]
        if opname != "VSIGN" or _ops.OPS[lw & 255].name != "VMUL" or (lw >> 8) & 255 != a:
            return False
        for reg in {(lw >> 16) & 255, lw >> 24}:
            for j in range(last - 1, -1, -1):
                w = self.code[j][0]
                kind = _ops.OPS[w & 255].kind
                if kind == "C_C":
                    continue
                if (w >> 8) & 255 == reg:                         # the instruction that defines this factor
                    block = positive_map(j, reg)
                    if block is not None:
                        self.code[j] = (_ops.BY_NAME["VEXPFLAG"].code | (reg << 8) | (reg << 16), len(self.params))
                        self.params.extend(block + [1.0, 0.0])
                        self.lip_v[reg] = _lip.INF
                        return False                              # (the sign itself is still emitted)
                    break
                if (kind == "V_V" and (w >> 16) & 255 == reg) or \
                        (kind == "V_VV" and reg in ((w >> 16) & 255, w >> 24)):
                    break                                          # somebody else reads the factor in between
        return False

    def emit(self, opname, a, b=0, c=0, params=(), _fold=True):
        info = _ops.BY_NAME[opname]
        params = [float(x) for x in np.asarray(params, dtype=np.float64).ravel()]
        if len(params) != info.nparams:
            raise LoweringError("%s expects %d parameters, got %d" % (opname, info.nparams, len(params)))
        if _fold and opname in ("VSIGN", "VHARDBIN") and self._fold_positive_map(opname, a, b, params):
            return
        if _fold and opname in self._AFFINE:
            # Consecutive affine maps on the same register (nested Euclidean transforms, move_sdf / scale_sdf /
            # shear chains) are composed here in float64 and rounded ONCE: g(f(p)) = A2 A1 p - (A2 c1 + c2).
            # Only the in-place continuation `C[a] = g(C[a])` right after `C[a] = f(C[b])` is merged: the
            # intermediate value is overwritten anyway, so nothing else can have read it.
            A2, c2 = self._as_affine(opname, params)
            last = len(self.code) - 1
            if a == b and last >= self._fold_floor and last in self._affine:
                lw, lpoff = self.code[last]
                if (lw >> 8) & 255 == a and all(np.isfinite(A2.ravel())) and all(np.isfinite(c2)):
                    A1, c1, lip_in = self._affine.pop(last)
                    src = (lw >> 16) & 255
                    self.code.pop()
                    del self.params[lpoff:]
                    # the merged map starts from what `src` held BEFORE the first map: undo that map's factor on the
                    # Lipschitz bound when it was applied in place (else the bound is multiplied twice — too small for
                    # contractions, which would let the culling skip operands it must not)
                    if src == a:
                        self.lip_c[a] = lip_in
                    self._emit_affine(a, src, A2.dot(A1), A2.dot(c1) + c2)
                    return
            self._affine[len(self.code)] = (A2, c2, self.lip_c.get(b, _lip.INF))
        poff = len(self.params)
        self.params.extend(params)
        self.code.append((info.code | (a << 8) | (b << 16) | (c << 24), poff))
        if info.kind == "C_C":
            self.lip_c[a] = self.lip_c.get(b, _lip.INF) * _lip.factor(_lip.C_C, opname, params)
        elif info.kind == "V_C":
            self.lip_v[a] = self.lip_c.get(b, _lip.INF) * _lip.factor(_lip.V_C, opname, params)
        elif info.kind == "V_V":
            self.lip_v[a] = self.lip_v.get(b, _lip.INF) * _lip.factor(_lip.V_V, opname, params)
        else:
            fn = _lip.V_VV.get(opname)
            la, lb = self.lip_v.get(b, _lip.INF), self.lip_v.get(c, _lip.INF)
            self.lip_v[a] = fn(la, lb) if fn and np.isfinite(la) and np.isfinite(lb) else _lip.INF

    def add_table(self, values):
        off = self._table_len
        chunk = np.array(values, dtype=np.float64).ravel()       # (a copy: the caller's array may change afterwards)
        self.tables.append(chunk)
        self._table_len += chunk.size
        if self._table_len >= (1 << 24):
            raise LoweringError("tables exceed 2^24 floats (offsets are carried as fp32)")
        return off

    def finish(self, vreg):
        with np.errstate(over="ignore"):
            params = np.asarray(self.params, dtype=np.float64).astype(np.float32)
            tables = (np.concatenate(self.tables) if self.tables else np.zeros(0)).astype(np.float32)
        # the largest subtrees first if there are more sites than mask bits
        sites = sorted(self.cull, key=lambda r: -((r[2] - r[1]) + (r[4] - r[3])))[:_lip.MAX_SITES]
        sites.sort(key=lambda r: r[0])
        return LoweredProgram(np.asarray(self.code, dtype=np.uint32).reshape(-1, 2), params, tables, vreg,
                              self.n_creg, self.n_vreg,
                              np.asarray([r[:5] for r in sites], dtype=np.uint32).reshape(-1, 5),
                              np.asarray([r[5] for r in sites], dtype=np.float32))

    # ---- coordinate helpers ----
    def writable(self, creg, mode):
        """Destination register for an op that produces a NEW array from `creg`."""
        return creg if mode == OWNED else self.new_c()

    def release(self, dst, creg):
        if dst != creg:
            self.free_c(dst)

    # ---- nodes: Euclidean transform around an expression (reference transformations.py:232-242) ----
    def lower_node(self, node, creg, mode):
        self._stack.append(("n", getattr(node, "path_key", None) or id(node)))
        try:
            return self._lower_node(node, creg, mode)
        finally:
            self._stack.pop()

    def _lower_node(self, node, creg, mode):
        if not _is_geometry(node):
            raise LoweringError("object %r is not an aegolius_amd geometry (needs the symbolic node protocol)"
                                % (node,))
        R = np.asarray(node.rotation_matrix, dtype=np.float64)
        if R.shape != (3, 3):
            raise ValueError("rotation matrix must have shape (3, 3); got %r" % (R.shape,))
        t = np.asarray(node.center, dtype=np.float64).reshape(3)
        s = node.scale
        pushed = _push_transform_into_members(node)
        if pushed is not None:
            # a large hard union that was moved / rotated / rescaled as a whole (value modifications on top included):
            # every member behind that transform (see _flatten_hard), so that the members still start from the input
            # point — what the chain kernels need; the factor on the value stays where it was, after everything
            v = self.lower_expr(pushed, creg, OWNED if mode == OWNED else FROZEN, node._geo_parameters)
            if s != 1:
                self.emit("VSCALE", v, v, params=[s])
            return v
        ident_r = np.array_equal(R, np.eye(3))
        unit_s = (s == 1)
        if ident_r and unit_s and not np.any(t):
            # (I·co)/1.0 - 0 is a bit-exact copy in the reference: alias the register instead
            c, inner_mode = creg, (OWNED if mode == OWNED else FROZEN)
        else:
            c = self.writable(creg, mode)
            rt = R.T
            if ident_r and unit_s:
                self.emit("XLATE", c, creg, params=rt.dot(t))
            else:
                self.emit("XFORM", c, creg, params=np.concatenate([(rt / s).ravel(), rt.dot(t)]))
            inner_mode = OWNED
        v = self.lower_expr(node.modified_object, c, inner_mode, node._geo_parameters)
        self.release(c, creg)
        if not unit_s and not getattr(node, "coord_only", False):
            self.emit("VSCALE", v, v, params=[s])
        return v

    # ---- expressions ----
    def lower_expr(self, expr, creg, mode, params):
        if isinstance(expr, NodeSDF):              # (a fresh wrapper per lowering: the object it wraps is the identity)
            return self.lower_node(expr.obj, creg, OWNED if mode == OWNED else FROZEN)
        self._stack.append(getattr(expr, "path_key", None) or id(expr))
        try:
            return self._lower_expr(expr, creg, mode, params)
        finally:
            self._stack.pop()

    def _lower_expr(self, expr, creg, mode, params):
        if isinstance(expr, PrimSDF):
            v = self.new_v()
            expr.lower(self, v, creg, params)
            return v
        if isinstance(expr, ModSDF):
            from ._mods import MOD_LOWER
            return MOD_LOWER[expr.name](self, expr, creg, mode, params)
        if isinstance(expr, CombineSDF):
            return self._lower_combine(expr, creg, mode)
        if isinstance(expr, NodeSDF):
            return self.lower_node(expr.obj, creg, OWNED if mode == OWNED else FROZEN)
        if isinstance(expr, UnsupportedSDF):
            from ._mods import _staged_op
            return _staged_op(self, expr, creg, mode, params)
        raise NotImplementedError(
            "aegolius_amd cannot fuse the opaque Python callable %r into the GPU evaluation; build the field "
            "from aegolius_amd primitives / geometry objects (obj.propagate, obj.sign(direct=True), ...)" % (expr,))

    def lower_callable(self, fn, creg, mode, params):
        """Second-field argument of displacement / define_volume / recover_volume."""
        self._stack.append("second")
        try:
            return self.lower_expr(as_expr(fn), creg, mode, params)
        finally:
            self._stack.pop()

    # ---- combiners (reference combine.py:51-78, 129-135, 154-160) ----
    def _lower_combine(self, expr, creg, mode):
        from .cores.combine import BINARY_OPS, NARY_OPS, PARAMETRIC_OPS
        op = expr.owner.operation_type
        kids = expr.children
        if expr.parametric:
            if op not in PARAMETRIC_OPS:
                raise KeyError(op)
            opcode = PARAMETRIC_OPS[op]
            w = expr.parameters
            if isinstance(w, (tuple, list, np.ndarray)):
                raise TypeError("unsupported operand type(s) for the parametric operation %s: parameters must be "
                                "a scalar" % op)
            w = float(w)
            if opcode in ("BOLTZ", "BOLTZSUB"):
                prm = [1.0 / w] if w != 0 else [np.inf]
            elif w == 0:
                # reference combine.py:14-18, 22-26: zero width degenerates to the hard operator
                opcode, prm = {"SMIN2": "VMIN", "SMIN3": "VMIN", "SMAX3": "VMAX", "SSUB3": "VSUBTRACT"}[opcode], []
            else:
                prm = [w, 1.0 / (4.0 * w)] if opcode == "SMIN2" else [w, 1.0 / (6.0 * w * w)]
            if len(kids) != 2:
                raise TypeError("%s takes exactly 2 objects (%d given)" % (op, len(kids)))
        else:
            if op in NARY_OPS:
                opcode, prm = NARY_OPS[op], []
                if len(kids) < 1:
                    raise ValueError("zero-size array to reduction operation which has no identity")
            elif op in BINARY_OPS:
                opcode, prm = BINARY_OPS[op], []
                if len(kids) != 2:
                    raise TypeError("%s takes exactly 2 objects (%d given)" % (op, len(kids)))
            else:
                raise KeyError(op)
        if not expr.parametric and opcode == "VSUBTRACT" and not _holds_combiner(kids[0]):
            # (a body that is a combination of its own stays an operand: the chain kernels then run the union as a chain
            #  and body and subtraction as the REST of the program, sdfk_codegen.cpp chain_analyse)
            holes = _subtracted_union(kids[1])
            if holes is not None:
                # body minus a UNION of many (a perforated plate, a porous block): max(a, -min_j b_j) = max(a, max_j -b_j),
                # an n-ary INTERSECT of the body and the negated members (negation is exact) — one chain instead of a
                # chain inside an operand, which the chain kernels do not take
                kids, opcode = (kids[0],) + holes, "VMAX"
        if not expr.parametric and opcode in ("VMIN", "VMAX"):
            kids = _flatten_hard(kids, opcode)
        acc = None
        first = len(self.code)
        for i, kid in enumerate(kids):
            last = (i == len(kids) - 1)
            b_start = len(self.code)
            self._fold_floor = b_start                     # never merge across an operand-range boundary
            self._stack.append(("child", i))
            try:
                v = self.lower_node(kid, creg, OWNED if (last and mode == OWNED) else FROZEN)
            finally:
                self._stack.pop()
            if acc is None:
                acc = v
            else:
                k = self.lip_v.get(acc, _lip.INF) + self.lip_v.get(v, _lip.INF)
                idx = len(self.code)
                self.emit(opcode, acc, acc, v, params=prm)
                if opcode in _lip.CULLABLE and np.isfinite(k) and b_start > first and idx > b_start:
                    self.cull.append((idx, first, b_start - 1, b_start, idx - 1, float(k)))
                self.free_v(v)
        return acc


class _Reframed:
    """A member of a flattened group: geometry `inner` seen through the transform of the group it belonged to (the
    node protocol of lower_node: the group's XFORM / XLATE, then the member, then the group's VSCALE)."""
    _geo_parameters = ()

    def __init__(self, group, inner, coord_only=False):
        self.rotation_matrix = group.rotation_matrix
        self.center = group.center
        self.scale = group.scale
        self.modified_object = NodeSDF(inner)
        self.coord_only = coord_only               # the group's factor on the VALUE is applied by the caller, once
        # position keys of staged operators (Lowerer._stack) must be the same in every lowering of one tree: wrappers
        # are made anew each time, so they are named after the objects they stand for
        self.path_key = ("reframed", id(group), getattr(inner, "path_key", None) or id(inner))


def _flatten_min():
    """Fewest members at which nested hard unions are flattened: where the table-driven chain kernels take over
    (sdfk_codegen.cpp chain_min_leaves, same environment variable); smaller trees keep their hierarchy of cull sites."""
    import os
    try:
        return max(2, int(os.environ.get("SDFK_CHAIN_MIN", "22")))
    except ValueError:
        return 22


def _flatten_hard(kids, opcode, always=False):
    """Nested hard unions / intersections as ONE n-ary operation. min and max are exact and associative, a rigid
    transform of a group can be applied to every member instead (the same arithmetic per member, recomputed), and a
    positive scale commutes with them bit for bit (x -> fl(s x) is monotone): UNION(move(UNION(a, b)), c) is
    UNION(move∘a, move∘b, c). The group's transform then meets the member's own inside one operand range, where emit()
    composes consecutive affine maps in float64 into one (as it does along any chain of transforms): one rounding
    instead of two — the flattened field differs from the nested program's by fp32 rounding (1e-7) and is no further
    from the float64 reference. A group is flattened when nothing but its transform lies between
    the two combiners (no modification of the group's value or coordinates). Done only when the flattened operation
    reaches the size of the chain kernels — 20 instances of a 50-sphere cluster become one 1000-member chain with
    per-brick survivor lists instead of a program beyond the specialisation limit."""
    from .cores.combine import BINARY_OPS, NARY_OPS
    if _env_flag("SDFK_NO_FLATTEN"):
        return kids
    same = {k for k, v in list(NARY_OPS.items()) + list(BINARY_OPS.items()) if v == opcode}

    def identity(g):
        R = np.asarray(g.rotation_matrix, dtype=np.float64)
        t = np.asarray(g.center, dtype=np.float64).reshape(-1)
        return R.shape == (3, 3) and np.array_equal(R, np.eye(3)) and g.scale == 1 and t.size == 3 and not np.any(t)

    def expand(frames, kid, out):
        """frames: the groups `kid` lies in, outermost first (only those with a transform)"""
        inner = getattr(kid, "modified_object", None) if _is_geometry(kid) else None
        if (len(frames) < 16 and isinstance(inner, CombineSDF) and not inner.parametric
                and getattr(inner.owner, "operation_type", None) in same and len(inner.children) >= 1):
            s = kid.scale
            try:
                ok = bool(np.isfinite(s)) and s > 0
            except TypeError:
                ok = False
            if ok:
                below = frames if identity(kid) else frames + [kid]
                for g in inner.children:
                    expand(below, g, out)
                return True
        for f in reversed(frames):                                 # innermost group first
            kid = _Reframed(f, kid)
        out.append(kid)
        return False

    flat, changed = [], False
    for kid in kids:
        changed |= expand([], kid, flat)
    if not changed or (len(flat) < _flatten_min() and not always):
        return kids
    return tuple(flat)


class _Negated:
    """-inner as a geometry of its own (identity frame): a member of the INTERSECT that a subtracted UNION turns into."""
    rotation_matrix = np.eye(3)
    center = np.zeros(3)
    scale = 1.0
    _geo_parameters = ()

    def __init__(self, inner):
        self.modified_object = ModSDF("invert", {}, NodeSDF(inner))
        key = getattr(inner, "path_key", None) or id(inner)
        self.modified_object.path_key = ("negated-mod", key)
        self.path_key = ("negated", key)


def _subtracted_union(node):
    """node: the second operand of a SUBTRACT2. -> its members, re-framed and negated, when it is a hard UNION (bare, with
    at most a transform of its own) of at least as many members as the chain kernels take; else None."""
    from .cores.combine import BINARY_OPS, NARY_OPS
    if _env_flag("SDFK_NO_FLATTEN") or not _is_geometry(node):
        return None
    inner = node.modified_object
    if not isinstance(inner, CombineSDF) or inner.parametric:
        return None
    op = getattr(inner.owner, "operation_type", None)
    if (NARY_OPS.get(op) or BINARY_OPS.get(op)) != "VMIN":
        return None
    members = _flatten_hard((node,), "VMIN", always=True)
    if (len(members) == 1 and members[0] is node) or len(members) + 1 < _flatten_min():
        return None
    return tuple(_Negated(m) for m in members)


def _large_hard_union(node):
    """node: an operand. -> (opcode, members framed by the operand's own transform) when it is a bare hard n-ary combination
    of at least as many members as the chain kernels take, else None."""
    from .cores.combine import BINARY_OPS, NARY_OPS
    if not _is_geometry(node):
        return None
    inner = node.modified_object
    if not isinstance(inner, CombineSDF) or inner.parametric:
        return None
    op = getattr(inner.owner, "operation_type", None)
    opcode = NARY_OPS.get(op) or BINARY_OPS.get(op)
    if opcode == "VSUBTRACT" and len(inner.children) == 2 and not _holds_combiner(inner.children[0]):
        # a simple body minus a large union: the INTERSECT of the body and the negated members (see _lower_combine), here
        # framed by the operand's own transform (a positive scale commutes with max and with the negation)
        holes = _subtracted_union(inner.children[1])
        s = node.scale
        try:
            ok = holes is not None and bool(np.isfinite(s)) and s > 0
        except TypeError:
            ok = False
        if not ok:
            return None
        members = (inner.children[0],) + holes
        R = np.asarray(node.rotation_matrix, dtype=np.float64)
        t = np.asarray(node.center, dtype=np.float64).reshape(-1)
        if not (R.shape == (3, 3) and np.array_equal(R, np.eye(3)) and s == 1 and t.size == 3 and not np.any(t)):
            members = tuple(_Reframed(node, m) for m in members)
        return "VMAX", members
    if opcode not in ("VMIN", "VMAX"):
        return None
    members = _flatten_hard((node,), opcode, always=True)
    if (len(members) == 1 and members[0] is node) or len(members) < _flatten_min():
        return None
    return opcode, members


def _holds_combiner(node, depth=0):
    """Is there a CombineGeometry anywhere in the expression of this geometry (a cull site inside it once lowered)?"""
    if depth > 64:
        return True
    stack = [node.modified_object if _is_geometry(node) else node]
    seen = 0
    while stack:
        e = stack.pop()
        seen += 1
        if seen > 4096 or isinstance(e, CombineSDF):
            return True
        if isinstance(e, NodeSDF):
            if _is_geometry(e.obj):
                stack.append(e.obj.modified_object)
        elif isinstance(e, ModSDF):
            stack.append(e.inner)
            if isinstance(e.second, SDFExpr):
                stack.append(e.second)
            elif e.second is not None and _is_geometry(getattr(e.second, "__self__", None)):
                stack.append(e.second.__self__.modified_object)
    return False


class _Operation:
    def __init__(self, operation_type):
        self.operation_type = operation_type


def _push_transform_into_members(node):
    """node: a geometry with a transform of its own whose SDF is a hard n-ary combination, bare or under pointwise
    VALUE modifications (rounding, onion, ... — they never see the coordinates). -> the equivalent expression with the
    transform's coordinate part applied to every member instead (the caller applies the factor on the value), or None
    when the rules of _flatten_hard do not apply or the combination is smaller than the chain kernels' minimum."""
    from ._mods import VALUE_OPS
    from .cores.combine import BINARY_OPS, NARY_OPS
    if isinstance(node, _Reframed) or _env_flag("SDFK_NO_FLATTEN"):
        return None
    R = np.asarray(node.rotation_matrix, dtype=np.float64)
    t = np.asarray(node.center, dtype=np.float64).reshape(-1)
    s = node.scale
    try:
        if not (np.isfinite(s) and s > 0) or R.shape != (3, 3) or t.size != 3:
            return None
    except TypeError:
        return None
    if np.array_equal(R, np.eye(3)) and s == 1 and not np.any(t):
        return None
    mods, inner = [], node.modified_object
    while isinstance(inner, ModSDF) and inner.name in VALUE_OPS and inner.second is None and len(mods) < 64:
        mods.append(inner)
        inner = inner.inner
    if not isinstance(inner, CombineSDF):
        return None
    members = None
    if not inner.parametric:
        op = getattr(inner.owner, "operation_type", None)
        opcode = NARY_OPS.get(op) or BINARY_OPS.get(op)
        kids = inner.children
        if opcode == "VSUBTRACT" and len(kids) == 2 and not _holds_combiner(kids[0]):
            holes = _subtracted_union(kids[1])                  # body minus a large UNION: an INTERSECT (see _lower_combine)
            if holes is not None:
                kids, opcode = (kids[0],) + holes, "VMAX"
        if opcode in ("VMIN", "VMAX"):
            members = _flatten_hard(kids, opcode, always=True)
            if len(members) < _flatten_min():
                members = None
    if members is not None:
        expr = CombineSDF(_Operation("UNION" if opcode == "VMIN" else "INTERSECT"),
                          [_Reframed(node, m, coord_only=True) for m in members], parametric=False)
    else:
        # any other combination with a large hard union among its operands (a union clipped by a box, blended with a
        # ground plane, subtracted from a body that is a combination itself — and then placed): the coordinate part of
        # the transform goes into every operand, and into the members of those unions; the combination is kept
        unions = [_large_hard_union(k) for k in inner.children]
        if not any(u is not None for u in unions):
            return None
        new_kids = []
        for k, u in zip(inner.children, unions):
            if u is None:
                new_kids.append(_Reframed(node, k, coord_only=True))
                continue
            sub = CombineSDF(_Operation("UNION" if u[0] == "VMIN" else "INTERSECT"),
                             [_Reframed(node, m, coord_only=True) for m in u[1]], parametric=False)
            sub.path_key = ("pushed-operand", id(k))
            holder = _ExprNode(sub, ())
            holder.path_key = ("pushed-holder", id(k))
            new_kids.append(holder)
        expr = CombineSDF(inner.owner, new_kids, inner.parametric, inner.parameters)
    expr.path_key = ("pushed", id(inner))
    for m in reversed(mods):
        expr = ModSDF(m.name, m.args, expr)
        expr.path_key = ("pushed", id(m))
    return expr


def _env_flag(name):
    """An experiment switch of the lowering (SDFK_NO_FLATTEN): set and not "0"."""
    import os
    return os.environ.get(name, "") not in ("", "0")


def _is_geometry(obj):
    return hasattr(obj, "_geo_parameters") and hasattr(obj, "modified_object") and hasattr(obj, "rotation_matrix")


def as_expr(fn):
    """Map a user-supplied callable onto the symbolic world."""
    if isinstance(fn, SDFExpr):
        return fn
    owner = getattr(fn, "__self__", None)
    if owner is not None and _is_geometry(owner) and getattr(fn, "__name__", "") in ("propagate", "create"):
        return NodeSDF(owner)
    key = id(fn)                                   # one node per callable: stage fields are keyed by node identity
    known = _OPAQUE.get(key)
    if known is not None and known.fn is fn:
        return known
    if len(_OPAQUE) > 4096:
        _OPAQUE.clear()
    _OPAQUE[key] = node = UnsupportedSDF(fn, "aegolius_amd cannot fuse the opaque Python callable %r into the GPU evaluation; "
                              "pass an aegolius_amd sdf_* function, a modification closure returned by an "
                              "aegolius_amd geometry, or obj.propagate" % (fn,))
    return node


_OPAQUE = {}


def _staged(L, fields, stop_at, probe_axis):
    L.fields = dict(fields or {})
    L.stop_at = stop_at
    L.probe_axis = probe_axis
    return L


def _deep(fn):
    """Run fn(); a tree nested deeper than the interpreter's recursion limit allows (a left-deep chain of hundreds of
    pairwise combinations: the lowering recurses a few frames per level, as the reference's nested closures do when they are
    called) is lowered again on a thread with a large stack and no practical limit."""
    try:
        return fn()
    except RecursionError:
        pass
    import sys
    import threading
    box = {}

    def work():
        old = sys.getrecursionlimit()
        sys.setrecursionlimit(1000000)
        try:
            box["value"] = fn()
        except BaseException as exc:  # noqa: BLE001
            box["error"] = exc
        finally:
            sys.setrecursionlimit(old)
    old_size = threading.stack_size(512 * 1024 * 1024)
    try:
        t = threading.Thread(target=work)
        t.start()
        t.join()
    finally:
        threading.stack_size(old_size)
    if "error" in box:
        raise box["error"]
    return box["value"]


def lower_geometry(node, fields=None, stop_at=None, probe_axis=None):
    """Lower `node.create(co)` to a program. `fields` / `stop_at` / `probe_axis`: stage programs of a tree with
    grid-neighbourhood operators (see _eval._run_staged)."""
    return _deep(lambda: _lower_geometry(node, fields, stop_at, probe_axis))


def _lower_geometry(node, fields, stop_at, probe_axis):
    L = _staged(Lowerer(), fields, stop_at, probe_axis)
    try:
        v = L.lower_node(node, 0, OWNED)
    except StageStop as stop:
        low = L.finish(stop.vreg)
        low.stage_params = stop.params
        return low
    if stop_at is not None:
        raise LoweringError("stage operator not reached")
    return L.finish(v)


class _ExprNode:
    """Adapter: evaluate a bare expression `expr(co, *params)` (identity transform)."""
    rotation_matrix = np.eye(3)
    center = np.zeros(3)
    scale = 1.0

    def __init__(self, expr, params):
        self.modified_object = expr
        self._geo_parameters = tuple(params)


def lower_expression(expr, params, fields=None, stop_at=None, probe_axis=None):
    return _deep(lambda: _lower_expression(expr, params, fields, stop_at, probe_axis))


def _lower_expression(expr, params, fields, stop_at, probe_axis):
    # a bare closure call gets the caller's array itself (no private copy): OWNED is safe because the
    # register is loaded from memory and the caller's array is never written
    L = _staged(Lowerer(), fields, stop_at, probe_axis)
    try:
        v = L.lower_expr(expr, 0, OWNED, tuple(params))
    except StageStop as stop:
        low = L.finish(stop.vreg)
        low.stage_params = stop.params
        return low
    if stop_at is not None:
        raise LoweringError("stage operator not reached")
    return L.finish(v)
