"""Scene builders shared by the golden-vector generator (run against the real reference in the build
container) and by the parity tests (run against aegolius_amd + the oracle).

Every builder takes `ns`, a namespace with the `spomso.cores` layout — the reference package itself
(`import spomso.cores as ns`) or the drop-in (`import aegolius_amd.cores as ns`) — and returns a
geometry object. The same Python text therefore drives both implementations: that is the drop-in
claim under test.
"""
import numpy as np

SCENES = {}
# scenes whose value jumps (sign / binarisation / cell boundaries): fp32 may legitimately take the other
# branch for a point sitting within rounding of the jump — the GPU tests count such points instead of
# bounding them
DISCONTINUOUS = set()


def scene(name, discontinuous=False):
    def deco(fn):
        assert name not in SCENES, name
        SCENES[name] = fn
        if discontinuous:
            DISCONTINUOUS.add(name)
        return fn
    return deco


def placed(obj, angle=0.7, axis=(0.3, -0.5, 0.8), move=(0.15, -0.1, 0.2), scale=None):
    obj.rotate(angle, axis)
    obj.move(move)
    if scale is not None:
        obj.set_scale(scale)
    return obj


# -------------------------------------------------------------------------------------------------
# inputs
# -------------------------------------------------------------------------------------------------
def input_points():
    """(3, 2042) float64 holding fp32-representable values: 1024 random points in [-2,2]^3, a 9^3 grid
    over [-1.5,1.5]^3 (exact zeros and symmetry planes) and a 17^2 planar grid with z = 0."""
    rng = np.random.default_rng(20240607)
    rnd = rng.uniform(-2.0, 2.0, size=(3, 1024))
    ax = np.linspace(-1.5, 1.5, 9)
    g3 = np.asarray(np.meshgrid(ax, ax, ax, indexing="ij")).reshape(3, -1)
    ax2 = np.linspace(-1.6, 1.6, 17)
    g2 = np.zeros((3, 17 * 17))
    g2[:2] = np.asarray(np.meshgrid(ax2, ax2, indexing="ij")).reshape(2, -1)
    co = np.concatenate([rnd, g3, g2], axis=1)
    return co.astype(np.float32).astype(np.float64)


# -------------------------------------------------------------------------------------------------
# curves used by parametric shapes / instancing
# -------------------------------------------------------------------------------------------------
def helix(t, radius, pitch):
    return np.asarray((radius * np.cos(t), radius * np.sin(t), pitch * t))


def ellipse(t, a, b):
    return np.asarray((a * np.cos(t), b * np.sin(t)))


def ellipse3(t, a, b):
    return np.asarray((a * np.cos(t), b * np.sin(t), 0 * t))


ZIGZAG3 = np.asarray([[-0.9, -0.3, 0.2, 0.8, 0.4], [-0.5, 0.6, -0.4, 0.5, -0.7], [-0.2, 0.3, 0.1, -0.3, 0.4]])
ZIGZAG2 = np.asarray([[-0.9, -0.3, 0.2, 0.8, 0.4], [-0.5, 0.6, -0.4, 0.5, -0.7], [0.0, 0.0, 0.0, 0.0, 0.0]])
CONVEX_POLY = np.asarray([[-0.8, 0.7, 0.9, 0.1, -0.7], [-0.6, -0.7, 0.3, 0.9, 0.4], [0, 0, 0, 0, 0]], dtype=float)
CONCAVE_POLY = np.asarray([[-0.9, 0.0, 0.9, 0.6, 0.0, -0.6], [-0.7, -0.2, -0.7, 0.8, 0.1, 0.8], [0, 0, 0, 0, 0, 0]],
                          dtype=float)
BOWTIE_POLY = np.asarray([[-0.8, 0.7, 0.8, -0.7], [-0.6, 0.5, -0.7, 0.6], [0, 0, 0, 0]], dtype=float)
FIGURE8_POLY = np.asarray([[-0.9, -0.3, 0.3, 0.9, 0.9, 0.3, -0.3, -0.9], [-0.3, -0.6, 0.6, 0.3, -0.3, -0.6, 0.6, 0.3],
                           [0, 0, 0, 0, 0, 0, 0, 0]], dtype=float)
FISH_POLY = np.asarray([[-0.9, -0.9, 0.2, 0.8, 0.2], [0.4, -0.4, 0.35, 0.0, -0.35], [0, 0, 0, 0, 0]], dtype=float)
# every turn of a pentagram goes the same way: the reference takes it for a convex outline (its inner pentagon)
PENTAGRAM_POLY = np.asarray([[np.cos(2 * np.pi * k * 2 / 5 + np.pi / 2) for k in range(5)],
                             [np.sin(2 * np.pi * k * 2 / 5 + np.pi / 2) for k in range(5)], [0.0] * 5])
# edges that cross different numbers of other edges: the reference's decomposition raises ValueError (ragged index array)
RAGGED_CROSSINGS_POLY = np.asarray([[-0.9, 0.8, 0.2, -0.2, -0.8, 0.9], [0.0, 0.1, -0.8, 0.8, -0.1, 0.05], [0.0] * 6])
CLOUD3 = np.random.default_rng(5).uniform(-1.0, 1.0, size=(3, 37))


# -------------------------------------------------------------------------------------------------
# primitives, each under a general rotation + translation (and some under a scale)
# -------------------------------------------------------------------------------------------------
_PRIMS = {
    "x": lambda ns: ns.X(0.2), "y": lambda ns: ns.Y(-0.3), "z": lambda ns: ns.Z(0.1),
    "sphere": lambda ns: ns.Sphere(0.5),
    "cylinder": lambda ns: ns.Cylinder(0.4, 1.1),
    "infinite_cylinder": lambda ns: ns.InfiniteCylinder(0.45),
    "box": lambda ns: ns.Box(0.9, 0.6, 0.4),
    "torus": lambda ns: ns.Torus(0.6, 0.17),
    "chainlink": lambda ns: ns.ChainLink(0.4, 0.1, 0.9),
    "braid": lambda ns: ns.Braid(1.6, 0.3, 0.08, 2.5),
    "arc3d": lambda ns: ns.Arc3D(0.7, 0.12, 0.3, 2.4),
    "plane": lambda ns: ns.Plane((0.2, -0.4, 0.9), 0.3),
    "oriented_plane": lambda ns: ns.OrientedPlane((0.5, 0.1, -0.7), 0.25),
    "line": lambda ns: ns.Line((-0.4, 0.3, -0.5), (0.6, -0.2, 0.7)),
    "cone": lambda ns: ns.Cone(0.9, np.pi / 7),
    "infinite_cone": lambda ns: ns.InfiniteCone(np.pi / 5),
    "oriented_infinite_cone": lambda ns: ns.OrientedInfiniteCone(np.pi / 6),
    "solid_angle": lambda ns: ns.geom_3d.SolidAngle(0.8, 0.2, 1.5),
    "triangle3d": lambda ns: ns.Triangle3D((-0.5, -0.4, 0.1), (0.7, -0.2, -0.3), (0.1, 0.8, 0.4)),
    "quad": lambda ns: ns.Quad((-0.6, -0.5, 0.0), (0.6, -0.6, 0.2), (0.7, 0.5, 0.0), (-0.5, 0.6, -0.2)),
    "segmented_line3d_closed": lambda ns: ns.SegmentedLine3D(ZIGZAG3, closed=True),
    "segmented_curve3d": lambda ns: ns.SegmentedParametricCurve3D(ZIGZAG3, (0, 5, 23)),
    "segmented_curve3d_closed": lambda ns: ns.SegmentedParametricCurve3D(ZIGZAG3, (0, 5, 23), closed=True),
    "parametric_curve3d": lambda ns: ns.ParametricCurve3D(helix, (0.6, 0.1), (0, 2 * np.pi, 41)),
    "parametric_curve3d_closed": lambda ns: ns.ParametricCurve3D(helix, (0.6, 0.1), (0, 2 * np.pi, 41), closed=True),
    "point_cloud3d": lambda ns: ns.geom_3d.PointCloud3D(CLOUD3),
    # 2-D
    "circle": lambda ns: ns.Circle(0.55),
    "neu_circle_3": lambda ns: ns.NEUCircle(0.6, 3),
    "neu_circle_1": lambda ns: ns.NEUCircle(0.6, 1),
    "neu_circle_inf": lambda ns: ns.NEUCircle(0.6, np.inf),
    "ngon": lambda ns: ns.NGon(0.6, 5),
    "ngon_3": lambda ns: ns.NGon(0.5, 3),
    "ngon_8": lambda ns: ns.NGon(0.7, 8),
    "ngon_16": lambda ns: ns.NGon(0.65, 16),
    "ngon_17": lambda ns: ns.NGon(0.65, 17),          # beyond the rotation fold of the GPU kernel: by angle
    "rectangle": lambda ns: ns.Rectangle(0.9, 0.5),
    "rounded_rectangle": lambda ns: ns.RoundedRectangle(1.0, 0.7, (0.1, 0.05, 0.2, 0.0)),
    "segment": lambda ns: ns.Segment((-0.5, -0.2, 0.0), (0.6, 0.4, 0.0)),
    "triangle": lambda ns: ns.Triangle(np.asarray((-0.6, -0.4)), np.asarray((0.7, -0.3)), np.asarray((0.0, 0.8))),
    "sector": lambda ns: ns.Sector(0.8, 0.3, 1.9),
    "infinite_sector": lambda ns: ns.InfiniteSector(0.4, 1.6),
    "arc": lambda ns: ns.Arc(0.7, 0.2, 2.6),
    "polygon_convex": lambda ns: ns.Polygon(CONVEX_POLY.copy()),
    "polygon_concave": lambda ns: ns.Polygon(CONCAVE_POLY.copy()),
    # self-intersecting outlines: cut into loops at the crossing points (C/triangulation_functions.py:128-302, 403-412)
    "polygon_bowtie": lambda ns: ns.Polygon(BOWTIE_POLY.copy()),
    "polygon_figure8": lambda ns: ns.Polygon(FIGURE8_POLY.copy()),
    "polygon_fish": lambda ns: ns.Polygon(FISH_POLY.copy()),
    "polygon_pentagram": lambda ns: ns.Polygon(PENTAGRAM_POLY.copy()),
    "segmented_line2d": lambda ns: ns.SegmentedLine(ZIGZAG2),
    "segmented_line2d_closed": lambda ns: ns.SegmentedLine(ZIGZAG2, closed=True),
    "segmented_curve2d": lambda ns: ns.SegmentedParametricCurve(ZIGZAG2, (0, 5, 19)),
    "segmented_curve2d_closed": lambda ns: ns.SegmentedParametricCurve(ZIGZAG2, (0, 5, 19), closed=True),
    "parametric_curve2d": lambda ns: ns.ParametricCurve(ellipse, (0.8, 0.5), (0, 2 * np.pi, 33)),
    "parametric_curve2d_closed": lambda ns: ns.ParametricCurve(ellipse, (0.8, 0.5), (0, 5.5, 33), closed=True),
    "point_cloud2d": lambda ns: ns.PointCloud2D(CLOUD3),
}
_DISCONT_PRIMS = {"polygon_convex", "polygon_concave", "polygon_bowtie", "polygon_figure8", "polygon_fish", "polygon_pentagram", "oriented_infinite_cone", "infinite_sector", "solid_angle",
                  "sector", "cone", "triangle", "ngon", "ngon_3", "ngon_8", "ngon_16", "ngon_17"}

for _name, _make in _PRIMS.items():
    scene("prim_" + _name, _name in _DISCONT_PRIMS)(lambda ns, m=_make: placed(m(ns)))
    scene("prim_" + _name + "_raw", _name in _DISCONT_PRIMS)(lambda ns, m=_make: m(ns))

scene("prim_sphere_scaled")(lambda ns: placed(ns.Sphere(0.5), scale=1.7))
scene("prim_box_scaled_int")(lambda ns: placed(ns.Box(0.9, 0.6, 0.4), scale=2))
scene("prim_torus_translated")(lambda ns: placed(ns.Torus(0.6, 0.17), angle=0.0, axis=(0, 0, 1)))


@scene("prim_sphere_set_ops")
def _(ns):
    s = ns.Sphere(0.4)
    s.set_location((0.3, 0.1))
    s.set_rotation(0.9, (0.0, 1.0, 0.0))
    s.rescale(1.5)
    s.rotate(0.4, (1, 0, 0))
    return s


@scene("prim_box_rotate_matrix")
def _(ns):
    b = ns.Box(0.8, 0.5, 0.3)
    b.set_rotation(0.6, (0.0, 0.0, 1.0))
    b.move((0.1, 0.2, -0.1))
    return b


# -------------------------------------------------------------------------------------------------
# modifications
# -------------------------------------------------------------------------------------------------
def _mod_scene(name, base, apply, discontinuous=False, place=True):
    def build(ns):
        obj = base(ns)
        apply(obj, ns)
        return placed(obj) if place else obj
    scene("mod_" + name, discontinuous)(build)


_box = lambda ns: ns.Box(0.6, 0.3, 0.2)          # noqa: E731
_sph = lambda ns: ns.Sphere(0.4)                 # noqa: E731
_rect = lambda ns: ns.Rectangle(0.5, 0.3)        # noqa: E731
_circ = lambda ns: ns.Circle(0.3)                # noqa: E731
_tor = lambda ns: ns.Torus(0.4, 0.12)            # noqa: E731

_mod_scene("elongation", _box, lambda o, ns: o.elongation((0.4, 0.0, 0.1)))
_mod_scene("rounding", _box, lambda o, ns: o.rounding(0.07))
_mod_scene("rounding_cs", _box, lambda o, ns: o.rounding_cs(0.05, 0.6))
_mod_scene("boundary", _sph, lambda o, ns: o.boundary())
_mod_scene("invert", _box, lambda o, ns: o.invert())
_mod_scene("sign", _box, lambda o, ns: o.sign(), True)
_mod_scene("onion", _sph, lambda o, ns: o.onion(0.05))
_mod_scene("concentric", _sph, lambda o, ns: o.concentric(0.2))
_mod_scene("revolution", _rect, lambda o, ns: o.revolution(0.6))
_mod_scene("axis_revolution", _rect, lambda o, ns: o.axis_revolution(0.6, 0.4))
_mod_scene("extrusion", _circ, lambda o, ns: o.extrusion(0.7))
_mod_scene("twist", _box, lambda o, ns: o.twist(np.pi / 2))
_mod_scene("bend", _box, lambda o, ns: o.bend(1.5, np.pi / 3))
_mod_scene("bend_tight", lambda ns: ns.Box(1.6, 0.2, 0.2), lambda o, ns: o.bend(0.5, 2.0))
for _sh in ("shear_xz", "shear_yz", "shear_xy", "shear_zy", "shear_yx", "shear_zx"):
    _mod_scene(_sh, _box, lambda o, ns, s=_sh: getattr(o, s)(0.35))
for _sa, _fa in ((0, 1), (0, 2), (1, 0), (1, 2), (2, 0), (2, 1)):
    _mod_scene("shear_%d%d" % (_sa, _fa), _box, lambda o, ns, a=_sa, f=_fa: o.shear(0.3, a, f))
_mod_scene("infinite_repetition", _sph, lambda o, ns: o.infinite_repetition((1.1, 1.3, 0.9)), True)
_mod_scene("finite_repetition", lambda ns: ns.Sphere(0.12),
           lambda o, ns: o.finite_repetition((2.0, 1.5, 1.2), (4, 3, 2)), True)
_mod_scene("finite_repetition_rescaled", lambda ns: ns.Sphere(0.4),
           lambda o, ns: o.finite_repetition_rescaled((2.0, 1.5, 1.2), (4, 3, 2), (0.8, 0.8, 0.8), (0.1, 0.1, 0.1)),
           True)
_mod_scene("symmetry_x", lambda ns: placed(ns.Box(0.5, 0.3, 0.2), move=(0.4, 0.2, 0.1)),
           lambda o, ns: o.symmetry(0), place=False)
_mod_scene("symmetry_z", lambda ns: placed(ns.Box(0.5, 0.3, 0.2), move=(0.4, 0.2, 0.3)),
           lambda o, ns: o.symmetry(2), place=False)
_mod_scene("symmetry_noop", _box, lambda o, ns: o.symmetry(4))
_mod_scene("mirror", _sph, lambda o, ns: o.mirror((-0.6, 0.1, 0.0), (0.5, 0.4, 0.3)))
_mod_scene("rotational_symmetry", lambda ns: ns.Sphere(0.15), lambda o, ns: o.rotational_symmetry(5, 0.7, 0.3), True)
_mod_scene("linear_instancing", lambda ns: ns.Sphere(0.12),
           lambda o, ns: o.linear_instancing(5, (-0.8, -0.2, 0.1), (0.9, 0.3, -0.2)), True)
_mod_scene("linear_instancing_2", lambda ns: ns.Sphere(0.12),
           lambda o, ns: o.linear_instancing(2, (-0.8, -0.2, 0.1), (0.9, 0.3, -0.2)))
_mod_scene("curve_instancing", lambda ns: ns.Sphere(0.1),
           lambda o, ns: o.curve_instancing(helix, (0.6, 0.15), (0, 2 * np.pi, 7)), True)
_mod_scene("aligned_curve_instancing", lambda ns: ns.Box(0.3, 0.1, 0.05),
           lambda o, ns: o.aligned_curve_instancing(ellipse3, (0.8, 0.5), (0.1, 6.0, 7)), True)
_mod_scene("fully_aligned_curve_instancing", lambda ns: ns.Box(0.3, 0.1, 0.05),
           lambda o, ns: o.fully_aligned_curve_instancing(helix, (0.6, 0.15), (0.1, 6.0, 7)), True)
# more instances than the scan threshold (256): nearest centre through the box tree
_mod_scene("curve_instancing_many", lambda ns: ns.Sphere(0.02),
           lambda o, ns: o.curve_instancing(helix, (0.7, 0.05), (0, 6 * np.pi, 333)), True)
_mod_scene("fully_aligned_curve_instancing_many", lambda ns: ns.Box(0.05, 0.02, 0.01),
           lambda o, ns: o.fully_aligned_curve_instancing(helix, (0.6, 0.08), (0.1, 18.0, 400)), True)
_mod_scene("move_sdf", _box, lambda o, ns: o.move_sdf((0.2, -0.3, 0.1)))
_mod_scene("scale_sdf", _box, lambda o, ns: o.scale_sdf(1.6))
_mod_scene("rotate_sdf", _box, lambda o, ns: o.rotate_sdf(
    np.asarray([[np.cos(0.5), -np.sin(0.5), 0.0], [np.sin(0.5), np.cos(0.5), 0.0], [0.0, 0.0, 1.0]])))
_mod_scene("sigmoid_falloff", _sph, lambda o, ns: o.sigmoid_falloff(2.0, 0.3))
_mod_scene("positive_sigmoid_falloff", _sph, lambda o, ns: o.positive_sigmoid_falloff(2.0, 0.3))
_mod_scene("capped_exponential", _sph, lambda o, ns: o.capped_exponential(1.5, 0.4))
_mod_scene("hard_binarization", _sph, lambda o, ns: o.hard_binarization(0.0), True)
_mod_scene("linear_falloff", _sph, lambda o, ns: o.linear_falloff(1.5, 0.5))
_mod_scene("relu", _sph, lambda o, ns: o.relu(0.5))
_mod_scene("smooth_relu", _sph, lambda o, ns: o.smooth_relu(0.1, 0.5, 0.02))
_mod_scene("slowstart", _sph, lambda o, ns: o.slowstart(0.1, 0.5, 0.02))
_mod_scene("slowstart_noground", _sph, lambda o, ns: o.slowstart(0.1, 0.5, 0.02, ground=False))
_mod_scene("gaussian_boundary", _sph, lambda o, ns: o.gaussian_boundary(1.2, 0.4))
_mod_scene("gaussian_falloff", _sph, lambda o, ns: o.gaussian_falloff(1.2, 0.4))
_mod_scene("polygon_from_line", lambda ns: ns.SegmentedLine(CONCAVE_POLY.copy(), closed=True),
           lambda o, ns: o.polygon(), True)
_mod_scene("shape_from_curve", lambda ns: ns.ParametricCurve(ellipse, (0.8, 0.5), (0, 2 * np.pi, 33), closed=True),
           lambda o, ns: o.shape(), True)


@scene("mod_recover_volume", True)
def _(ns):
    b = ns.Box(0.6, 0.4, 0.3)
    inside = b.sign(direct=True)
    b.boundary()
    b.recover_volume(inside)
    return placed(b)


@scene("mod_define_volume", True)
def _(ns):
    s = ns.Sphere(0.5)
    other = ns.Box(0.5, 0.5, 0.5)
    s.boundary()
    s.define_volume(other.sign(direct=True), ((0.5, 0.5, 0.5),))
    return placed(s)


@scene("mod_displacement_node")
def _(ns):
    s = ns.Sphere(0.5)
    bump = ns.Sphere(0.2)
    bump.move((0.4, 0.0, 0.0))
    bump.gaussian_boundary(0.1, 0.3)
    s.displacement(bump.propagate, ())
    return placed(s)


@scene("mod_displacement_sdf_function")
def _(ns):
    s = ns.Box(0.6, 0.4, 0.3)
    s.displacement(ns.sdf_sphere, (0.3,))
    return placed(s)


# ---- in-place aliasing: the second field sees the coordinates the inner chain mutated ----
@scene("alias_symmetry_displacement")
def _(ns):
    s = ns.Sphere(0.5)
    s.move_sdf((0.2, 0.0, 0.0))
    s.symmetry(0)
    s.displacement(ns.sdf_x, (0.0,))       # evaluates to |x| (mutated array), not x
    return placed(s)


@scene("alias_rotsym_recover", True)
def _(ns):
    s = ns.Sphere(0.2)
    s.rotational_symmetry(4, 0.5, 0.1)
    s.rounding(0.02)
    s.recover_volume(ns.sdf_y)             # sees the rotated / folded coordinates
    return placed(s)


@scene("alias_axis_revolution_displacement")
def _(ns):
    r = ns.Rectangle(0.4, 0.2)
    r.axis_revolution(0.5, 0.3)
    r.displacement(ns.sdf_x, (0.1,))       # sees xy rotated in place, but not the revolved array
    return placed(r)


@scene("alias_symmetry_in_child_not_visible")
def _(ns):
    a = ns.Sphere(0.3)
    a.symmetry(0)
    b = ns.Box(0.4, 0.3, 0.2)
    b.move((0.3, 0.1, 0.0))
    u = ns.CombineGeometry("UNION2").combine(a, b)   # b must see the untouched coordinates
    u.displacement(ns.sdf_x, (0.0,))
    return placed(u)


@scene("alias_nested_two_field", True)
def _(ns):
    s = ns.Sphere(0.45)
    s.symmetry(1)
    s.displacement(ns.sdf_y, (0.0,))
    s.recover_volume(ns.sdf_y)
    return placed(s)


@scene("frozen_node_chain")
def _(ns):
    base = placed(ns.Box(0.5, 0.3, 0.2))
    frozen = ns.GenericGeometry(base.propagate)
    frozen.twist(1.1)
    frozen.symmetry(0)
    return placed(frozen, angle=0.4, axis=(1, 0, 0), move=(0.0, 0.1, 0.0))


@scene("modified_object_as_sdf")
def _(ns):
    b = ns.Box(0.5, 0.4, 0.3)
    b.rounding(0.05)
    g = ns.GenericGeometry(b.modified_object, (0.7, 0.2, 0.3))   # same chain, other parameters
    return placed(g)


# -------------------------------------------------------------------------------------------------
# combiners
# -------------------------------------------------------------------------------------------------
def _pair(ns):
    a = placed(ns.Sphere(0.45), move=(-0.15, 0.05, 0.0))
    b = placed(ns.Box(0.7, 0.4, 0.5), angle=0.5, axis=(1, 1, 0), move=(0.25, -0.05, 0.1))
    return a, b


for _op in ("UNION2", "UNION", "SUBTRACT2", "INTERSECT2", "INTERSECT", "SUM", "DIFFERENCE"):
    scene("combine_" + _op)(lambda ns, op=_op: ns.CombineGeometry(op).combine(*_pair(ns)))
for _op in ("SMOOTH_UNION2_2", "SMOOTH_UNION2", "SMOOTH_INTERSECT2", "SMOOTH_INTERSECT2_BOLTZMANN",
            "SMOOTH_SUBTRACT2", "SMOOTH_SUBTRACT2_BOLTZMANN"):
    scene("combine_" + _op)(lambda ns, op=_op: ns.CombineGeometry(op).combine_parametric(*_pair(ns), parameters=0.2))
scene("combine_SMOOTH_UNION2_zero_width")(
    lambda ns: ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(*_pair(ns), parameters=0.0))


@scene("combine_UNION_nary")
def _(ns):
    objs = [placed(ns.Sphere(0.2), move=(0.5 * np.cos(k), 0.5 * np.sin(k), 0.1 * k)) for k in range(6)]
    return placed(ns.CombineGeometry("UNION").combine(*objs), scale=1.3)


@scene("combine_INTERSECT_nary")
def _(ns):
    objs = [placed(ns.Sphere(0.7), move=(0.2 * np.cos(k), 0.2 * np.sin(k), 0.0)) for k in range(4)]
    return ns.CombineGeometry("INTERSECT").combine(*objs)


@scene("combine_modified_result")
def _(ns):
    u = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(*_pair(ns), parameters=0.15)
    u.onion(0.03)
    u.elongation((0.2, 0.1, 0.0))
    return placed(u, angle=1.1, axis=(0, 1, 0), move=(0.1, 0.0, -0.1), scale=0.8)


# -------------------------------------------------------------------------------------------------
# tree-level scenes: the five BASELINE configs at reduced resolution, plus example-like ones
# -------------------------------------------------------------------------------------------------
# the recipes live in the package (aegolius_amd/workloads.py: bench.py and smoke() build from the same ones)
from aegolius_amd.workloads import _cfg2_prims, cfg2_tree, cfg3_chain, cfg4_scene2d, cfg5_tree, sphere_union  # noqa: E402,F401


scene("tree_cfg1_sphere")(lambda ns: ns.Sphere(0.5))
scene("tree_cfg2_smooth_union10")(cfg2_tree)
scene("tree_cfg3_mod_chain", True)(cfg3_chain)
scene("tree_cfg4_union50_2d", True)(cfg4_scene2d)
scene("tree_cfg5_three_level")(cfg5_tree)


# the reference's own example scenes (Code/examples/scalar/3D/pawn_3D.py, chip_3D.py, 2D/olympic_rings_2D.py): the builders
# are proven equal to the scripts by tests/golden/generate_example_golden.py; here on the shared input cloud
import example_scenes  # noqa: E402

scene("tree_pawn_3D")(example_scenes.pawn)
scene("host_chip_3D")(example_scenes.chip)          # (a Python callable as displacement: host stage)
scene("tree_olympic_rings_2D")(example_scenes.olympic_rings)


# large hard unions in the forms the lowering rewrites (nested groups flattened, a transform of the whole pushed into the
# members, body minus union as one INTERSECT) or runs as a chain inside a larger program — pinned against the REAL reference
def _scattered(ns, count, seed, radius=0.12, extent=1.0, kinds=2):
    rng = np.random.default_rng(seed)
    objs = []
    for k in range(count):
        o = ns.Sphere(float(radius * rng.uniform(0.5, 1.5))) if k % kinds == 0 else ns.Box(*(float(x) for x in rng.uniform(0.1, 0.3, 3)))
        if k % 5 == 0:
            o.rounding(0.01)
        if k % 3 == 0:
            o.rotate(float(rng.uniform(0, 3)), tuple(rng.normal(size=3)))
        o.move(rng.uniform(-extent, extent, 3))
        objs.append(o)
    return objs


@scene("chain_union_60_flat")
def _(ns):
    return ns.CombineGeometry("UNION").combine(*_scattered(ns, 60, 1))


@scene("chain_clusters_nested_placed")
def _(ns):
    rng = np.random.default_rng(2)
    groups = []
    for g in range(6):
        u = ns.CombineGeometry("UNION").combine(*_scattered(ns, 12, 10 + g, radius=0.08, extent=0.3))
        u.rotate(float(rng.uniform(0, 3)), tuple(rng.normal(size=3)))
        u.move(rng.uniform(-0.8, 0.8, 3))
        if g % 2:
            u.rescale(1.3)
        groups.append(u)
    pair = ns.CombineGeometry("UNION2").combine(groups[0], groups[1])
    pair.move((0.1, 0.0, -0.1))
    scene_ = ns.CombineGeometry("UNION").combine(pair, *groups[2:])
    scene_.rotate(0.4, (0, 1, 0))
    scene_.rescale(0.9)
    scene_.onion(0.02)
    return scene_


@scene("chain_perforated_plate")
def _(ns):
    plate = ns.Box(2.2, 2.0, 0.5)
    holes = ns.CombineGeometry("UNION").combine(*_scattered(ns, 40, 3, radius=0.1, kinds=1))
    holes.move((0.05, 0, 0))
    out = ns.CombineGeometry("SUBTRACT2").combine(plate, holes)
    out.rotate(0.3, (1, 0, 0))
    return out


@scene("chain_intersection_40")
def _(ns):
    objs = []
    rng = np.random.default_rng(4)
    for _k in range(40):
        b = ns.Box(*(float(x) for x in rng.uniform(2.0, 3.0, 3)))
        b.rotate(float(rng.uniform(0, 3)), tuple(rng.normal(size=3)))
        b.move(rng.uniform(-0.4, 0.4, 3))
        objs.append(b)
    return ns.CombineGeometry("INTERSECT").combine(*objs)


@scene("chain_clipped_and_placed")
def _(ns):
    u = ns.CombineGeometry("UNION").combine(*_scattered(ns, 50, 5))
    t = ns.CombineGeometry("INTERSECT2").combine(u, ns.Sphere(1.1))
    t.rotate(0.3, (0, 1, 0))
    t.move((0.1, 0, 0))
    t.rescale(1.1)
    return t


@scene("chain_blended_with_ground")
def _(ns):
    ground = ns.Box(3.0, 3.0, 0.3)
    ground.move((0, 0, -0.9))
    u = ns.CombineGeometry("UNION").combine(*_scattered(ns, 50, 6))
    t = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(ground, u, parameters=0.15)
    t.rounding(0.01)
    return t


@scene("chain_composite_body_minus_union")
def _(ns):
    body = ns.CombineGeometry("SMOOTH_UNION2_2").combine_parametric(ns.Box(2.0, 2.0, 2.0), ns.Sphere(1.3), parameters=0.2)
    u = ns.CombineGeometry("UNION").combine(*_scattered(ns, 40, 7, kinds=1))
    return ns.CombineGeometry("SUBTRACT2").combine(body, u)


@scene("tree_deep_right")
def _(ns):
    """Right-deep nesting: exercises the value-register stack."""
    rng = np.random.default_rng(99)
    prims = _cfg2_prims(ns, rng, 6)
    acc = prims[-1]
    for p in reversed(prims[:-1]):
        acc = ns.CombineGeometry("SMOOTH_UNION2_2").combine_parametric(p, acc, parameters=0.12)
    return acc


# -------------------------------------------------------------------------------------------------
# seeded random trees: compositions nobody thought of (aliasing modes, nested transforms, register reuse)
# -------------------------------------------------------------------------------------------------
def _random_leaf(ns, rng):
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    kind = int(rng.integers(0, 12))
    if kind == 0:
        return ns.Sphere(u(0.2, 0.6))
    if kind == 1:
        return ns.Box(u(0.3, 0.9), u(0.2, 0.7), u(0.2, 0.6))
    if kind == 2:
        return ns.Cylinder(u(0.15, 0.4), u(0.3, 1.0))
    if kind == 3:
        return ns.Torus(u(0.3, 0.6), u(0.05, 0.15))
    if kind == 4:
        return ns.Cone(u(0.4, 0.9), u(0.2, 0.6))
    if kind == 5:
        return ns.ChainLink(u(0.2, 0.4), u(0.05, 0.1), u(0.4, 0.9))
    if kind == 6:
        return ns.Line((u(-0.6, 0), u(-0.5, 0.5), u(-0.5, 0.5)), (u(0, 0.6), u(-0.5, 0.5), u(-0.5, 0.5)))
    if kind == 7:
        c = ns.Circle(u(0.2, 0.5))
        c.extrusion(u(0.2, 0.8))
        return c
    if kind == 8:
        r = ns.Rectangle(u(0.2, 0.4), u(0.1, 0.3))
        r.revolution(u(0.3, 0.6))
        return r
    if kind == 9:
        return ns.Plane((u(-1, 1), u(-1, 1), u(0.2, 1)), u(0.05, 0.3))
    if kind == 10:
        return ns.InfiniteCylinder(u(0.1, 0.4))
    return ns.Arc3D(u(0.4, 0.7), u(0.05, 0.15), u(0, 1), u(1.5, 3))


def _random_mods(obj, ns, rng):
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    for _ in range(int(rng.integers(0, 4))):
        m = int(rng.integers(0, 14))
        if m == 0:
            obj.rounding(u(0.01, 0.08))
        elif m == 1:
            obj.onion(u(0.01, 0.05))
        elif m == 2:
            obj.elongation((u(0, 0.4), u(0, 0.3), u(0, 0.2)))
        elif m == 3:
            obj.symmetry(int(rng.integers(0, 3)))
        elif m == 4:
            obj.twist(u(-1.5, 1.5))
        elif m == 5:
            obj.mirror((u(-0.6, -0.1), u(-0.3, 0.3), u(-0.2, 0.2)), (u(0.1, 0.6), u(-0.3, 0.3), u(-0.2, 0.2)))
        elif m == 6:
            obj.displacement(ns.sdf_x, (u(-0.2, 0.2),))
        elif m == 7:
            obj.scale_sdf(u(0.6, 1.5))
        elif m == 8:
            obj.move_sdf((u(-0.3, 0.3), u(-0.3, 0.3), u(-0.3, 0.3)))
        elif m == 9:
            obj.concentric(u(0.05, 0.2))
        elif m == 10:
            obj.shear_xz(u(-0.4, 0.4))
        elif m == 11:
            obj.rounding_cs(u(0.01, 0.05), u(0.8, 1.5))
        elif m == 12:
            other = ns.Sphere(u(0.3, 0.7))
            obj.displacement(other.propagate, ())
        else:
            obj.invert()
            obj.invert()
    return obj


def _random_place(obj, rng):
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    t = int(rng.integers(0, 5))
    if t >= 1:
        obj.move((u(-0.6, 0.6), u(-0.6, 0.6), u(-0.6, 0.6)))
    if t >= 2:
        obj.rotate(u(0, np.pi), rng.normal(0, 1, 3))
    if t == 4:
        obj.set_scale(u(0.6, 1.6))
    return obj


_RANDOM_BINARY = ["UNION2", "SUBTRACT2", "INTERSECT2", "SUM", "DIFFERENCE"]
_RANDOM_PARAMETRIC = ["SMOOTH_UNION2_2", "SMOOTH_UNION2", "SMOOTH_INTERSECT2", "SMOOTH_SUBTRACT2",
                      "SMOOTH_INTERSECT2_BOLTZMANN"]


def _random_tree(ns, rng, depth):
    if depth == 0 or rng.uniform() < 0.25:
        return _random_place(_random_mods(_random_leaf(ns, rng), ns, rng), rng)
    c = int(rng.integers(0, 3))
    if c == 0:
        kids = [_random_tree(ns, rng, depth - 1) for _ in range(int(rng.integers(2, 5)))]
        node = ns.CombineGeometry(["UNION", "INTERSECT"][int(rng.integers(0, 2))]).combine(*kids)
    elif c == 1:
        node = ns.CombineGeometry(_RANDOM_BINARY[int(rng.integers(0, len(_RANDOM_BINARY)))]).combine(
            _random_tree(ns, rng, depth - 1), _random_tree(ns, rng, depth - 1))
    else:
        node = ns.CombineGeometry(_RANDOM_PARAMETRIC[int(rng.integers(0, len(_RANDOM_PARAMETRIC)))]).combine_parametric(
            _random_tree(ns, rng, depth - 1), _random_tree(ns, rng, depth - 1), parameters=float(rng.uniform(0.05, 0.3)))
    if rng.uniform() < 0.4:
        node = ns.GenericGeometry(node.propagate, ())          # freeze, then modify the frozen node
    return _random_place(_random_mods(node, ns, rng), rng)


def random_tree(ns, seed, depth=3):
    return _random_tree(ns, np.random.default_rng(seed), depth)


for _seed in range(48):
    scene("random_tree_%02d" % _seed, True)(lambda ns, s=_seed: random_tree(ns, 1000 + s))


# -------------------------------------------------------------------------------------------------
# opaque user code: custom_post_process, custom_modification, plain Python callables as SDF functions
# (evaluated on the host between two GPU stages)
# -------------------------------------------------------------------------------------------------
def _user_tanh(u, amplitude, width):
    return amplitude * np.tanh(u / width)


def _user_zoom(geo_object, co, params, mod_params):
    k, drift = mod_params
    return geo_object(co * k, *params) / k + drift * co[2]


def _user_blob(co, radius, bump):
    return np.linalg.norm(co, axis=0) - radius + bump * np.sin(4.0 * co[0]) * np.cos(3.0 * co[1])


def _user_ripple(co, amplitude):
    return amplitude * np.sin(5.0 * co[0] + 2.0 * co[2])


@scene("host_custom_post_process")
def _(ns):
    b = ns.Box(0.9, 0.7, 0.5)
    b.rotate(0.4, (0, 1, 1))
    b.custom_post_process(_user_tanh, (0.5, 0.3), post_process_name="tanh")
    b.rounding(0.01)
    return b


@scene("host_custom_modification")
def _(ns):
    t = ns.Torus(0.5, 0.15)
    t.twist(0.4)
    t.custom_modification(_user_zoom, (1.3, 0.02), modification_name="zoom")
    t.onion(0.01)
    t.move((0.1, 0.0, -0.2))
    return t


@scene("host_python_callable_leaf")
def _(ns):
    g = ns.GenericGeometry(_user_blob, 0.6, 0.05)
    g.elongation((0.2, 0.0, 0.1))
    g.rotate(0.7, (1, 0, 0))
    g.set_scale(1.2)
    return g


@scene("host_python_callable_as_displacement_in_union")
def _(ns):
    s = ns.Sphere(0.55)
    s.displacement(_user_ripple, (0.03,))
    c = ns.Cylinder(0.25, 1.2)
    c.custom_post_process(_user_tanh, (1.0, 2.0))
    u = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(s, c, parameters=0.1)
    u.move((0.05, 0.05, 0.0))
    return u


@scene("host_same_callable_used_twice_with_other_parameters")
def _(ns):
    # one Python function, two use sites, different parameters: each use is its own stage (stages are keyed by the
    # position in the tree, not by the callable)
    s = ns.Sphere(0.55)
    s.displacement(_user_ripple, (0.03,))
    s.displacement(_user_ripple, (-0.015,))
    b = ns.Box(0.5, 0.4, 0.3)
    b.displacement(_user_ripple, (0.05,))
    b.move((0.3, 0.0, 0.1))
    return ns.CombineGeometry("UNION2").combine(s, b)


@scene("host_same_object_used_at_two_places")
def _(ns):
    # the SAME geometry object (with a user post-process inside) as a child of two different parents that hand it
    # different coordinates
    shared = ns.Torus(0.4, 0.12)
    shared.custom_post_process(_user_tanh, (0.8, 0.5))
    a = ns.CombineGeometry("UNION2").combine(shared, ns.Sphere(0.2))
    a.move((0.4, 0.0, 0.0))
    b = ns.CombineGeometry("INTERSECT2").combine(shared, ns.Box(1.2, 1.2, 0.4))
    b.rotate(0.8, (1, 0, 0))
    return ns.CombineGeometry("UNION2").combine(a, b)


# -------------------------------------------------------------------------------------------------
# grid-neighbourhood modifications (signed, conv_averaging, conv_edge_detection): these reshape the field to the
# grid, so they are evaluated on whole generate_grid clouds, not on the shared point cloud.
#   GRID_SCENES[name] = (builder(ns, co_resolution), grid key);  GRIDS[key] = (size, resolution as the user passes it)
# -------------------------------------------------------------------------------------------------
GRIDS = {"g3": ((2.0, 2.0, 2.0), (14, 12, 10)), "g2": ((3.0, 3.0), (30, 24))}
GRID_SCENES = {}

# -------------------------------------------------------------------------------------------------
# the fp32 RANGE class: sign() / hard_binarization(0) of a value map that is strictly positive in exact arithmetic. The
# reference (float64) keeps a tiny positive number where fp32 underflows to 0, so the sign of the fp32 value would be
# off by 1.0; the lowering decides on the exponent instead (Lowerer._fold_positive_map). The three random chains are
# the cases the round-2 fuzzer flagged (tests/fuzz_mods.py seeds 70109 / 70139 / 70197).
# -------------------------------------------------------------------------------------------------
def _fuzz_mods_scene(seed):
    def build(ns):
        import sys
        import fuzz_mods
        return fuzz_mods.build(ns, sys.modules[__name__], seed)[0]
    return build


for _seed in (70109, 70139, 70197):
    scene("range_fuzz_mods_%d" % _seed, True)(_fuzz_mods_scene(_seed))


@scene("range_gaussian_boundary_sign", True)
def _(ns):
    s = ns.Sphere(0.5)                      # |v / w| up to 30: fp32 underflows at 5.1, float64 at 13.6
    s.gaussian_boundary(2.0, 0.1)
    s.sign()
    return placed(s)


@scene("range_gaussian_falloff_hardbin", True)
def _(ns):
    b = ns.Box(0.6, 0.5, 0.4)
    b.gaussian_falloff(0.7, 0.08)
    b.hard_binarization(0.0)
    return placed(b)


@scene("range_capped_exponential_sign", True)
def _(ns):
    t = ns.Torus(0.5, 0.1)
    t.capped_exponential(1.5, 0.004)        # exponent -4 v / w down to -3000: beyond float64's -745 too
    t.sign()
    return placed(t)


@scene("range_sigmoid_sign", True)
def _(ns):
    c = ns.Cylinder(0.3, 0.8)
    c.sigmoid_falloff(1.0, 0.005)           # 1 + exp(4 v / w) overflows in fp32 at v / w = 22, in float64 at 177
    c.sign()
    return placed(c)



def grid_scene(name, grid, discontinuous=False):
    def deco(fn):
        assert name not in GRID_SCENES and name not in SCENES, name
        GRID_SCENES[name] = (fn, grid)
        if discontinuous:
            DISCONTINUOUS.add(name)
        return fn
    return deco


def grid_inputs(ns, key):
    """(co, co_resolution): generate_grid of the namespace under test, coordinates rounded to fp32 values."""
    size, resolution = GRIDS[key]
    co, _ = ns.generate_grid(size, resolution)
    return np.asarray(co).astype(np.float32).astype(np.float64), resolution


@grid_scene("grid_conv_sphere_3x3x3", "g3")
def _(ns, res):
    s = ns.Sphere(0.6)
    s.conv_averaging((3, 3, 3), 1, res)
    return s


@grid_scene("grid_conv_even_kernel_iterated_then_rounding", "g3")
def _(ns, res):
    b = ns.Box(0.9, 0.7, 0.5)
    b.rotate(0.6, (1, 1, 0))
    b.conv_averaging((2, 4, 1), 3, res)
    b.rounding(0.03)
    b.move((0.1, -0.05, 0.0))
    return b


@grid_scene("grid_conv_integer_kernel", "g3")
def _(ns, res):
    t = ns.Torus(0.5, 0.15)
    t.conv_averaging(3, 2, res)
    return t


@grid_scene("grid_conv_point_cloud_2d", "g2")
def _(ns, res):
    phi = np.linspace(0, 2 * np.pi, 26)
    pts = np.asarray([np.cos(phi), np.sin(phi), 0 * phi])
    pc = ns.PointCloud2D(pts)
    pc.conv_averaging((5, 5), 4, res)
    return pc


@grid_scene("grid_conv_under_a_twist", "g3")
def _(ns, res):
    b = ns.Box(0.8, 0.4, 0.9)
    b.conv_averaging((3, 3, 1), 1, res)
    b.twist(0.8)                              # declared later: warps the coordinates the averaged closure is given
    b.set_scale(1.2)
    return b


@grid_scene("grid_edge_detection_3d", "g3")
def _(ns, res):
    s = ns.Sphere(0.55)
    s.move((0.1, 0.0, -0.1))
    s.conv_edge_detection(res)
    return s


@grid_scene("grid_edge_detection_2d", "g2")
def _(ns, res):
    c = ns.Circle(0.8)
    c.conv_edge_detection(res)
    return c


@grid_scene("grid_signed_sphere_shell", "g3", True)
def _(ns, res):
    s = ns.Sphere(0.62)
    s.boundary()
    s.signed(res)
    return s


@grid_scene("grid_signed_box_transformed_node", "g3", True)
def _(ns, res):
    b = ns.Box(0.9, 0.8, 0.7)
    b.boundary()
    b.signed(res)
    b.move((0.07, -0.04, 0.05))
    b.set_scale(1.1)
    return b


@grid_scene("grid_signed_old_rotated_torus", "g3", True)
def _(ns, res):
    t = ns.Torus(0.55, 0.22)
    t.boundary()
    t.signed_old(res)
    t.rotate(0.5, (1, 0, 0))
    return t


@grid_scene("grid_signed_already_signed_is_untouched", "g3")
def _(ns, res):
    s = ns.Sphere(0.5)
    s.signed(res)
    return s


@grid_scene("grid_signed_then_conv_then_onion", "g3", True)
def _(ns, res):
    c = ns.Cylinder(0.5, 0.9)
    c.boundary()
    c.signed(res)
    c.conv_averaging((3, 3, 3), 1, res)
    c.onion(0.02)
    return c


@grid_scene("grid_union_of_signed_and_plain", "g3", True)
def _(ns, res):
    a = ns.Box(0.7, 0.7, 0.7)
    a.boundary()
    a.signed(res)
    a.rounding(0.02)
    b = ns.Sphere(0.35)
    b.move((0.45, 0.3, 0.2))
    c = ns.Torus(0.6, 0.1)
    c.conv_averaging((3, 3, 1), 2, res)
    u = ns.CombineGeometry("UNION").combine(a, b, c)
    u.rotate(0.0, (0, 0, 1))
    return u


# -------------------------------------------------------------------------------------------------
# consumers of the field: point_cloud (interior extraction) and from_sdf (gradient direction)
#   CONSUMER_SCENES[name] = (builder(ns, co_resolution), grid key)
# -------------------------------------------------------------------------------------------------
CONSUMER_SCENES = {}


def consumer_scene(name, grid):
    def deco(fn):
        assert name not in CONSUMER_SCENES, name
        CONSUMER_SCENES[name] = (fn, grid)
        return fn
    return deco


@consumer_scene("consume_sphere", "g3")
def _(ns, res):
    s = ns.Sphere(0.6)
    s.move((0.1, -0.05, 0.2))
    return s


@consumer_scene("consume_smooth_union_3d", "g3")
def _(ns, res):
    a = ns.Box(0.7, 0.5, 0.9)
    a.rotate(0.5, (1, 0.2, 0.4))
    b = ns.Torus(0.5, 0.12)
    b.move((0.2, 0.1, -0.3))
    return ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(a, b, parameters=0.15)


@consumer_scene("consume_flat_plateaus", "g3")
def _(ns, res):
    b = ns.Box(1.25, 1.05, 0.85)               # faces between lattice planes
    b.hard_binarization(0.0)                   # 0 / 1 field: most gradients vanish, the rest sit on the faces
    return b


@consumer_scene("consume_empty_interior", "g3")
def _(ns, res):
    s = ns.Sphere(0.2)
    s.move((5.0, 0.0, 0.0))                    # nothing inside the grid
    return s


@consumer_scene("consume_circle_2d", "g2")
def _(ns, res):
    c = ns.Circle(0.9)
    c.move((0.3, -0.2, 0.0))
    return c


@consumer_scene("consume_rounded_rectangle_2d", "g2")
def _(ns, res):
    r = ns.RoundedRectangle(1.8, 1.1, (0.3, 0.1, 0.2, 0.0))
    r.rotate(0.4, (0, 0, 1))
    r.onion(0.1)
    return r


def input_sensitivity(evaluate, co, trials=5, seed=12345):
    """How far the REFERENCE arithmetic (float64 oracle, `evaluate(co) -> field`) moves when every coordinate changes
    by one fp32 ulp: per-point max |f(co') - f(co)| over a few seeded perturbations. A GPU deviation within a small
    multiple of this is the conditioning of the scene (steep value maps, fractional powers near an axis), not an
    error of the kernel; the fuzzers report such points separately."""
    rng = np.random.default_rng(seed)
    base = np.asarray(evaluate(co.copy()), dtype=np.float64)
    worst = np.zeros_like(base)
    for _ in range(trials):
        moved = co * (1.0 + 6e-8 * rng.choice([-1.0, 1.0], size=co.shape))
        with np.errstate(all="ignore"):
            worst = np.fmax(worst, np.abs(np.asarray(evaluate(moved), dtype=np.float64) - base))
    return worst
