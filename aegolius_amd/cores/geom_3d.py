"""3-D shapes — API mirror of reference cores/geom_3d.py (constructor signatures, parameter
massaging and read-only properties), generated from the table below; the curve classes that carry
extra logic are written out. All of them are GenericGeometry objects evaluated on the GPU."""
import numpy as np

from .. import _prims as P
from ._shapes import shape
from .geom import GenericGeometry

_a = np.asarray

X = shape("X", P.get("sdf_x"), ["offset"], doc="Value of the x coordinate zeroed at `offset`.")
Y = shape("Y", P.get("sdf_y"), ["offset"], doc="Value of the y coordinate zeroed at `offset`.")
Z = shape("Z", P.get("sdf_z"), ["offset"], doc="Value of the z coordinate zeroed at `offset`.")
InfiniteCylinder = shape("InfiniteCylinder", P.get("sdf_circle"), ["radius"],
                         doc="Cylinder of a given radius and infinite height (axis z).")
Cylinder = shape("Cylinder", P.get("sdf_cylinder"), ["radius", "height"], doc="Cylinder of radius and height (axis z).")
Sphere = shape("Sphere", P.get("sdf_sphere"), ["radius"], doc="Sphere of a given radius.")
Box = shape("Box", P.get("sdf_box"), ["a", "b", "c"], pack=lambda a, b, c: ((a, b, c),),
            doc="Box with side lengths a, b, c along x, y, z.")
Plane = shape("Plane", P.get("sudf_plane"), ["normal", "thickness"],
              pack=lambda normal, thickness: (_a(normal), thickness),
              props={"normal": lambda v: _a(v["normal"]), "thickness": lambda v: v["thickness"]},
              doc="Slab of a given thickness around the plane through the origin with the given normal.")
OrientedPlane = shape("OrientedPlane", P.get("sdf_plane"), ["normal", "offset"],
                      pack=lambda normal, offset: (_a(normal), offset),
                      props={"normal": lambda v: _a(v["normal"]), "offset": lambda v: v["offset"]},
                      doc="Signed distance to a plane with the given normal, offset along it.")
Line = shape("Line", P.get("sdf_segment_3d"), ["a", "b"],
             props={"point_a": lambda v: _a(v["a"]), "point_b": lambda v: _a(v["b"])},
             doc="Segment between points a and b (unsigned distance).")
Triangle3D = shape("Triangle3D", P.get("sdf_triangle_3d"), ["a", "b", "c"],
                   pack=lambda a, b, c: (_a(a), _a(b), _a(c)),
                   props={k: (lambda v, k=k: _a(v[k])) for k in "abc"}, doc="Triangle with vertices a, b, c.")
Quad = shape("Quad", P.get("sdf_quad_3d"), ["a", "b", "c", "d"],
             pack=lambda a, b, c, d: (_a(a), _a(b), _a(c), _a(d)),
             props={k: (lambda v, k=k: _a(v[k])) for k in "abcd"}, doc="Quadrilateral with vertices a, b, c, d.")
Torus = shape("Torus", P.get("sdf_torus"), ["primary_radius", "secondary_radius"],
              doc="Torus in the xy-plane: ring radius and tube radius.")
ChainLink = shape("ChainLink", P.get("sdf_chainlink"), ["primary_radius", "secondary_radius", "length"],
                  pack=lambda primary_radius, secondary_radius, length: (primary_radius, secondary_radius, length / 2),
                  props={"primary_radius": lambda v: v["primary_radius"],
                         "secondary_radius": lambda v: v["secondary_radius"], "length": lambda v: v["length"] / 2},
                  doc="Chain link: width, wire thickness, length.")
Braid = shape("Braid", P.get("sdf_braid"), ["length", "primary_radius", "secondary_radius", "pitch"],
              pack=lambda length, primary_radius, secondary_radius, pitch:
              (length / 2, primary_radius, secondary_radius, pitch),
              props={"primary_radius": lambda v: v["primary_radius"],
                     "secondary_radius": lambda v: v["secondary_radius"], "length": lambda v: v["length"] / 2,
                     "pitch": lambda v: v["pitch"]},
              doc="Two-strand braid along z.")
Arc3D = shape("Arc3D", P.get("sdf_arc_3d"), ["radius", "thickness", "start_angle", "end_angle"],
              props={"start_angle": lambda v: v["start_angle"], "end_angle": lambda v: v["end_angle"],
                     "primary_radius": lambda v: v["radius"], "secondary_radius": lambda v: v["thickness"]},
              doc="Arc of a torus between two angles.")
Cone = shape("Cone", P.get("sdf_cone"), ["height", "angle"],
             props={"height": lambda v: v["height"], "height_offset": lambda v: v["height"] * 0.5 ** (1 / 3),
                    "angle": lambda v: v["angle"], "base_radius": lambda v: v["height"] * np.tan(v["angle"])},
             doc="Cone of a given height and slope angle.")
InfiniteCone = shape("InfiniteCone", P.get("sdf_infinite_cone"), ["angle"], doc="Unsigned infinite cone, tip at origin.")
OrientedInfiniteCone = shape("OrientedInfiniteCone", P.get("sdf_oriented_infinite_cone"), ["angle"],
                             doc="Signed infinite cone, tip at the origin, negative below the surface.")
SolidAngle = shape("SolidAngle", P.get("sdf_solid_angle"), ["radius", "angle_1", "angle_2"],
                   doc="Spherical sector between two angles.")


def _columns(points):
    p = np.asarray(points)
    return p.T if p.shape[1] < p.shape[0] else p


class _CurveBase(GenericGeometry):
    closed = property(lambda self: self._closed, doc="Whether the curve is closed on itself.")


class ParametricCurve3D(_CurveBase):
    """Curve through the samples of a user-provided parametric curve f(t, *parameters) -> (3, M)."""

    def __init__(self, parametric_curve, parametric_curve_parameters, t_range, closed=False):
        self._curve, self._c_params, self._t_range, self._closed = \
            parametric_curve, parametric_curve_parameters, t_range, closed
        GenericGeometry.__init__(self, self.sdf_closed_curve(), parametric_curve, parametric_curve_parameters, self.ts)

    def sdf_closed_curve(self):
        return P.get("closed_parametric_curve_3d" if self.closed else "sdf_parametric_curve_3d")

    steps = property(lambda self: self._t_range[2])
    t_start = property(lambda self: self._t_range[0])
    t_end = property(lambda self: self._t_range[1])
    ts = property(lambda self: np.linspace(*self._t_range))


class SegmentedParametricCurve3D(_CurveBase):
    """Curve through points resampled along a poly-line at fractional indices."""

    def __init__(self, points, t_range, closed=False):
        self._points, self._t_range, self._closed = _columns(points), t_range, closed
        GenericGeometry.__init__(self, self.sdf_closed_curve(), self._points, self.ts)

    def sdf_closed_curve(self):
        return P.get("closed_segmented_curve_3d" if self.closed else "sdf_segmented_curve_3d")

    steps = property(lambda self: self._t_range[2])
    t_start = property(lambda self: self._t_range[0])
    t_end = property(lambda self: self._t_range[1])

    @property
    def ts(self):
        tt = np.linspace(self._t_range[0], self._t_range[1] - 1, self._t_range[2]) - self._t_range[0]
        return np.clip(tt, 0, self._points.shape[1] - 1.0001)


class SegmentedLine3D(_CurveBase):
    """Poly-line through the given points. As in the reference (cores/geom_3d.py:748-751) the open
    variant is wired to `sdf_segmented_curve_3d` with one argument too few and raises TypeError on
    create(); only closed=True evaluates."""

    def __init__(self, points, closed=False):
        self._points, self._closed = _columns(points), closed
        GenericGeometry.__init__(self, self.sdf_closed_curve(), self._points)

    def sdf_closed_curve(self):
        return P.get("closed_line_curve_3d" if self.closed else "sdf_segmented_curve_3d")


class PointCloud3D(GenericGeometry):
    """Unsigned distance to the nearest point of a (3, M) cloud."""

    def __init__(self, points):
        self._points = _columns(points)
        GenericGeometry.__init__(self, P.get("sdf_point_cloud_3d"), self._points)

    points = property(lambda self: self._points)


class GenericGeometry3D(GenericGeometry):
    """Reference cores/geom_3d.py:22-36: the same constructor and create / propagate as GenericGeometry."""
