"""Lipschitz bookkeeping for exact brick culling.

For every register the lowering tracks an upper bound L of the Lipschitz constant of its content as
a function of the ROOT input point (Euclidean norm): |v(p) - v(q)| <= L |p - q|. Distance functions
are 1-Lipschitz, rigid transforms keep the constant, a uniform scale s gives 1/s on coordinates and
s on the value, min / max / polynomial smooth-min keep the larger constant of their operands (their
partial derivatives are a convex combination). Anything that jumps (sign, modulo cells, nearest
instance) or whose stretch is unbounded (twist, bend) is `inf`, which disables culling around it.

The culling kernel evaluates the tree once at the centre c of a brick of radius rho; for a combiner
`smin_w(a, b)` it may skip the whole b subtree on that brick when
    b(c) - a(c) >= w + (L_a + L_b) rho        (=> b(p) - a(p) >= w for every p in the brick)
in which case the combiner returns a(p) bit-exactly (t = max(w - |a-b|, 0) = 0).
"""
import numpy as np

INF = float("inf")


def _norm2(m9):
    m = np.asarray(m9, dtype=np.float64).reshape(3, 3)
    if not np.all(np.isfinite(m)):
        return INF
    return float(np.linalg.norm(m, 2))


def _neucircle(p):
    order, kind = p[1], p[2]
    if kind in (1.0, 2.0):
        return 1.0
    if kind == 3.0 or order <= 0:
        return INF
    return 1.0 if order >= 2 else float(2.0 ** (1.0 / order - 0.5))


# coordinate -> coordinate : factor on the coordinate constant
C_C = {
    "MOVC": lambda p: 1.0, "XFORM": lambda p: _norm2(p[:9]), "XLATE": lambda p: 1.0, "LIN3": lambda p: _norm2(p[:9]),
    "CSCALE": lambda p: abs(float(p[0])), "ELONGATE": lambda p: 1.0, "REVOLVE": lambda p: 1.0, "ROT2D": lambda p: 1.0,
    "AXREV": lambda p: 1.0, "ZEROZ": lambda p: 1.0, "SYMMETRY": lambda p: 1.0, "FOLDX": lambda p: 1.0,
}
# coordinate -> value : Lipschitz constant of the primitive itself
V_C = {name: (lambda p: 1.0) for name in (
    "P_AXIS", "P_SPHERE", "P_CYLINDER", "P_BOX", "P_TORUS", "P_CHAINLINK", "P_PLANE", "P_UPLANE", "P_SEGMENT3",
    "P_CONE", "P_INFCONE", "P_SOLIDANGLE", "P_TRIANGLE3", "P_QUAD3", "P_SEGLINE3", "P_NEAREST3", "P_CIRCLE", "P_BOX2",
    "P_SEGMENT2", "P_RBOX2", "P_TRIANGLE2", "P_ARC2", "P_ARC3D", "P_SECTOR", "P_INFSECTOR", "P_NGON", "P_SEGLINE2",
    "P_NEAREST2", "P_ZSLAB", "P_NEARTREE")}
V_C["P_NEUCIRCLE"] = _neucircle
# value -> value : factor
V_V = {
    "VSCALE": lambda p: abs(float(p[0])), "VSUBC": lambda p: 1.0, "VAFFINE": lambda p: abs(float(p[0])),
    "VABS": lambda p: 1.0, "VNEG": lambda p: 1.0, "VONION": lambda p: 1.0, "VCONCENTRIC": lambda p: 1.0,
    "VRELU": lambda p: abs(float(p[0])), "VLINFALL": lambda p: abs(float(p[0] * p[1])),
}
_MAX = lambda a, b: max(a, b)   # noqa: E731
V_VV = {
    "VADD": lambda a, b: a + b, "VDIFF": lambda a, b: a + b, "VMIN": _MAX, "VMAX": _MAX, "VSUBTRACT": _MAX,
    "SMIN2": _MAX, "SMIN3": _MAX, "SMAX3": _MAX, "SSUB3": _MAX, "EXTRUDE": lambda a, b: float(np.sqrt(2.0)) * max(a, b),
}
# combiners at which a whole operand subtree can be skipped
CULLABLE = ("VMIN", "VMAX", "VSUBTRACT", "SMIN2", "SMIN3", "SMAX3", "SSUB3")
MAX_SITES = 32767  # what sdfk_program_set_cull takes; the mask kernels use the 64 widest (the flat tile kernel 31), long
                   # n-ary chains run table-driven with every site (csrc/sdfk_codegen.cpp, "chain mode")


def factor(table, name, params):
    fn = table.get(name)
    if fn is None:
        return INF
    with np.errstate(all="ignore"):
        v = fn(np.asarray(params, dtype=np.float64))
    return v if np.isfinite(v) else INF
