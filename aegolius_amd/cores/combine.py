"""`CombineGeometry` — API mirror of reference cores/combine.py:37-163 (7 non-parametric and 6
parametric operators). `combine()` validates the operator name exactly like the reference
(SyntaxError) and returns a fresh `GenericGeometry` whose SDF is a CombineSDF node; the operator is
looked up through `operation_type` when the tree is evaluated, as in the reference (:135, :160).
The arithmetic is in csrc/sdfk_device.h (cmb_*).
"""
from .._ir import CombineSDF
from .geom import GenericGeometry


# ---- the smooth kernels on plain arrays (reference cores/combine.py:12-34) -----------------------------------------
def _pair(opname, x, y, prm):
    from .. import _eval

    def emit(L, regs):
        L.emit(opname, regs[0], regs[0], regs[1], params=prm)
        return regs[0]
    return _eval.apply_fields([x, y], emit)


def smoothmin_poly2(x, y, a):
    """min(x, y) - h^2 a / 4 with h = max(a - |x - y|, 0) / a; a == 0: min(x, y) (:12-18). On the GPU (cmb_smin2)."""
    a = float(a)
    return _pair("VMIN", x, y, []) if a == 0 else _pair("SMIN2", x, y, [a, 1.0 / (4.0 * a)])


def smoothmin_poly3(x, y, a):
    """min(x, y) - h^3 a / 6; a == 0: min(x, y) (:20-26). On the GPU (cmb_smin3)."""
    a = float(a)
    return _pair("VMIN", x, y, []) if a == 0 else _pair("SMIN3", x, y, [a, 1.0 / (6.0 * a * a)])


def smoothmax_boltz(x, y, a):
    """Boltzmann-weighted mean (x e^{x/a} + y e^{y/a}) / (e^{x/a} + e^{y/a}) (:29-34), evaluated in the overflow-free
    form. On the GPU (cmb_boltz)."""
    a = float(a)
    return _pair("BOLTZ", x, y, [1.0 / a if a != 0 else float("inf")])

# operation name -> opcode of the folding instruction
NARY_OPS = {"UNION": "VMIN", "INTERSECT": "VMAX"}                                  # np.amin / np.amax over all
BINARY_OPS = {"UNION2": "VMIN", "SUBTRACT2": "VSUBTRACT", "INTERSECT2": "VMAX", "SUM": "VADD",
              "DIFFERENCE": "VDIFF"}
PARAMETRIC_OPS = {"SMOOTH_UNION2_2": "SMIN2", "SMOOTH_UNION2": "SMIN3", "SMOOTH_INTERSECT2": "SMAX3",
                  "SMOOTH_INTERSECT2_BOLTZMANN": "BOLTZ", "SMOOTH_SUBTRACT2": "SSUB3",
                  "SMOOTH_SUBTRACT2_BOLTZMANN": "BOLTZSUB"}

_NONPARAMETRIC_ORDER = ["UNION2", "UNION", "SUBTRACT2", "INTERSECT2", "INTERSECT", "SUM", "DIFFERENCE"]


class UnknownOperation(SyntaxError, TypeError):
    """The reference means to raise SyntaxError(msg, hint) for an unknown operation name (reference
    cores/combine.py:125-127, 150-152); because the hint is a plain string, CPython rejects that
    constructor call and what callers actually see is a TypeError. This class is both."""


class CombineGeometry:
    """Combination operations on scalar fields.

    Args:
        operation_type: name of the operation used by combine() / combine_parametric().
    """

    def __init__(self, operation_type):
        self.operation_type = operation_type
        self._combined_geometry = None
        self.operations = {k: (NARY_OPS.get(k) or BINARY_OPS[k]) for k in _NONPARAMETRIC_ORDER}
        self.parametric_operations = dict(PARAMETRIC_OPS)

    @property
    def available_operations(self):
        names = list(self.operations.keys())
        print(f"Available non-parametric operations are: {names}")
        return names

    @property
    def available_parametric_operations(self):
        names = list(self.parametric_operations.keys())
        print(f"Available parametric operations are: {names}")
        return names

    @property
    def combined_geometry(self):
        """SDF of the combined geometries; usable as `GenericGeometry(combined_geometry, ())`."""
        return self._combined_geometry

    def combine(self, *combined_objects):
        if self.operation_type not in self.operations:
            raise UnknownOperation(f"{self.operation_type} is not an implemented non-parametric operation. "
                                   f"Possible operations are {list(self.operations)}")
        self._combined_geometry = CombineSDF(self, combined_objects, parametric=False)
        return GenericGeometry(self._combined_geometry, ())

    def combine_parametric(self, *combined_objects, parameters):
        if self.operation_type not in self.parametric_operations:
            raise UnknownOperation(f"{self.operation_type} is not an implemented parametric operation. "
                                   f"Possible parametric operations are {list(self.parametric_operations)}")
        self._combined_geometry = CombineSDF(self, combined_objects, parametric=True, parameters=parameters)
        return GenericGeometry(self._combined_geometry, ())
