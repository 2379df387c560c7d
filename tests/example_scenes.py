"""Scenes of the reference's own example scripts (Code/examples/scalar/{2D,3D}/*.py), restated as builders over the
operator API so that they can be evaluated by the reference (fixture generator, build container), by the oracle and by
aegolius_amd alike. SURVEY.md §8(c) asks for these tree-level fixtures.

Every entry: builder(ns), grid size, the script's own resolution, the reduced resolution of the committed fixture, and
where the script lives + the name of the array it computes. `tests/golden/generate_example_golden.py` EXECUTES the
script itself against the real reference (plots stubbed) and checks that the builder below reproduces the script's
array exactly at the script's resolution, so each recipe is pinned to the script it restates without the script's text
living in this repository. The reduced grids keep the scripts' spacing ratios, so the tie classes of the full-size
scenes (grid points exactly on polygon edges, points equidistant from two curve samples) occur in the fixtures too.
"""
import numpy as np

EXAMPLES = {}


def example(name, size, full_res, res, script, variable, overrides=None):
    """overrides: {script variable: value} the generator sets in the script's text before running it (the scripts
    select their variants with module-level constants)."""
    def deco(fn):
        assert name not in EXAMPLES, name
        EXAMPLES[name] = dict(build=fn, size=size, full_res=full_res, res=res, script=script, variable=variable,
                              overrides=overrides or {})
        return fn
    return deco


# ---- 3-D -------------------------------------------------------------------------------------------------------------
@example("pawn_3D", (1.4, 1.4, 2.2), (100, 100, 200), (34, 34, 66), "3D/pawn_3D.py", "pawn_pattern")
def pawn(ns):
    torso = ns.Cone(1.8, np.pi / 10)
    torso.move((0, 0, -0.55))
    torso.rounding_cs(0.1, 0.4)
    base = ns.Cylinder(0.4, 0.07)
    base.rounding(0.1)
    base.move((0, 0, -0.95))
    head = ns.Sphere(0.25)
    head.move((0, 0, 0.6))
    collar = ns.Cylinder(0.3, 0.05)
    collar.move((0, 0, 0.25))
    union = ns.CombineGeometry("UNION2")
    statue = union.combine(union.combine(base, torso), head)
    out = ns.CombineGeometry("SMOOTH_UNION2_2").combine_parametric(statue, collar, parameters=0.2)
    out.move((0, 0, 0.2))
    return out


def _chip_displace(co_, a_, p_):
    return a_ * np.sin(co_[1] * p_ * np.pi * 2)


@example("chip_3D", (3, 3, 2), (150, 150, 100), (38, 38, 26), "3D/chip_3D.py", "chip_pattern")
def chip(ns):
    cy = ns.Cylinder(1, 0.05)
    cy.rotate(np.pi / 2, (1, 0, 0))
    s1 = ns.GenericGeometry(cy.propagate, ())
    s1.bend(1.75, np.pi)
    s1.rotate(np.pi / 2, (0, 1, 0))
    s2 = ns.GenericGeometry(s1.propagate, ())
    s2.bend(1.75, np.pi)
    s2.rotate(np.pi / 2, (1, 0, 0))
    out = ns.GenericGeometry(s2.propagate, ())
    out.displacement(_chip_displace, (0.02, 10))
    return out


@example("braid_3D", (1.5, 1.5, 4), (150, 150, 200), (30, 30, 40), "3D/braid_3D.py", "braid_pattern")
def braid(ns):
    torus = ns.Torus(0.25, 0.2)
    torus.elongation((2., 0., 0.0))
    torus.rotate(np.pi / 2, (0, 1, 0))
    out = ns.GenericGeometry(torus.propagate, ())
    out.twist(np.pi)
    return out


@example("sphere_in_sphere_3D", (2.3, 2.3, 2.3), (150, 150, 150), (38, 38, 38), "3D/sphere_in_sphere_3D.py", "combined_pattern")
def sphere_in_sphere(ns):
    inner = ns.Sphere(0.25)
    inner.rescale(2)
    inner.set_scale(1.5)
    outer = ns.Arc(0.75, -np.pi / 4, np.pi / 3)
    outer.rounding(0.1)
    outer.revolution(0)
    out = ns.CombineGeometry("UNION2").combine(inner, outer)
    out.rotate(np.pi / 4, (0, 0, 1))
    out.rescale(1.25)
    return out


def _repetition(kind):
    def build(ns):
        box = ns.Box(1.0, 0.5, 0.25)
        if kind == "INFINITE":
            box.infinite_repetition((1.2, 1, 0.5))
        elif kind == "FINITE":
            box.finite_repetition((2., 3., 2.), (2, 3, 4))
        else:
            box.finite_repetition_rescaled((2., 3., 2.), (2, 3, 5), (1, 0.5, 0.25), (0.2, 0.3, 0.1))
        return box
    return build


for _kind in ("INFINITE", "FINITE", "FINITE_RESCALED"):
    example("repetitions_3D_" + _kind.lower(), (3, 3, 3), (100, 100, 100), (34, 34, 34),
            "3D/finite_infinite_repetitions_3D.py", "box_pattern", {"repetition_type": _kind})(_repetition(_kind))


def _spiral(t, radius, height, freq):
    x = radius * np.cos(2 * np.pi * freq * t)
    y = radius * np.sin(2 * np.pi * freq * t)
    z = height * t - height / 2
    return np.asarray((x, y, z))


def _instancing(kind):
    def build(ns):
        box = ns.Box(0.5, 0.2, 0.3)
        if kind == "SIMPLE":
            box.curve_instancing(_spiral, (1, 2, 2), (0, 1, 21))
        elif kind == "ALIGNED":
            box.aligned_curve_instancing(_spiral, (1, 2, 2), (0, 1, 21))
        else:
            box.fully_aligned_curve_instancing(_spiral, (1, 2, 2), (0, 1, 21))
        return box
    return build


for _kind in ("SIMPLE", "ALIGNED", "FULLY_ALIGNED"):
    example("spiral_instancing_3D_" + _kind.lower(), (3, 3, 3), (100, 100, 100), (34, 34, 34),
            "3D/spiral_instancing_3D.py", "spiral_pattern", {"instancing_type": _kind})(_instancing(_kind))


# ---- 2-D -------------------------------------------------------------------------------------------------------------
@example("olympic_rings_2D", (5, 3), (500, 300), (126, 76), "2D/olympic_rings_2D.py", "olympic_rings_pattern")
def olympic_rings(ns):
    radius, thickness, x_sep, y_sep = 0.5, 0.05, 1.2, 0.5
    rings = []
    for cx, cy in ((-x_sep, y_sep / 2), (0, y_sep / 2), (x_sep, y_sep / 2), (-x_sep / 2, -y_sep / 2), (x_sep / 2, -y_sep / 2)):
        c = ns.Circle(radius)
        c.onion(thickness)
        c.move((cx, cy, 0))
        rings.append(c)
    return ns.CombineGeometry("UNION").combine(*rings)


_HOURGLASS = [[-1, -2, 0], [1, 2, 0], [-1, 2, 0], [1, -2, 0]]


@example("hourglass_2D_parametric_polygon", (4, 6), (400, 600), (100, 150), "2D/hourglass_2D.py", "segmented_line_pattern",
         {"parametric_evaluate": True, "polygon": True})
def hourglass_parametric(ns):
    spc = ns.SegmentedParametricCurve(_HOURGLASS, (0, 4, 200), closed=True)
    spc.polygon()
    return spc


@example("hourglass_2D_line_polygon", (4, 6), (400, 600), (100, 150), "2D/hourglass_2D.py", "segmented_line_pattern",
         {"parametric_evaluate": False, "polygon": True})
def hourglass_line(ns):
    spc = ns.SegmentedLine(_HOURGLASS, closed=True)
    spc.polygon()
    return spc


@example("hourglass_2D_line_rounded", (4, 6), (400, 600), (100, 150), "2D/hourglass_2D.py", "segmented_line_pattern",
         {"parametric_evaluate": False, "polygon": False})
def hourglass_rounded(ns):
    spc = ns.SegmentedLine(_HOURGLASS, closed=True)
    spc.rounding(0.1)
    return spc
