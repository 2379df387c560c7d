"""2-D shapes — API mirror of reference cores/geom_2d.py, generated from the table below (see
_shapes.shape); the curve classes with extra methods are written out. 2-D SDFs read x and y only;
coordinates are still (3, N) with the z row present, as produced by generate_grid."""
import numpy as np

from .. import _prims as P
from .._ir import ModSDF
from ._shapes import shape
from .geom import GenericGeometry
from .geom_3d import _columns, _CurveBase

_a = np.asarray

Circle = shape("Circle", P.get("sdf_circle"), ["radius"], doc="Circle of a given radius.")
NEUCircle = shape("NEUCircle", P.get("sdf_neu_circle"), ["radius", "order"],
                  doc="'Circle' in the p-norm of the given order (non-Euclidean).")
NGon = shape("NGon", P.get("sdf_ngon"), ["radius", "n_sides"], doc="Regular polygon with n sides.")
Rectangle = shape("Rectangle", P.get("sdf_box_2d"), ["a", "b"], pack=lambda a, b: ((a, b),),
                  props={"a": lambda v: v["a"], "b": lambda v: v["b"], "size": lambda v: _a((v["a"], v["b"]))},
                  doc="Rectangle with side lengths a (x) and b (y).")
RoundedRectangle = shape("RoundedRectangle", P.get("sdf_rounded_box_2d"), ["a", "b", "rounding"],
                         pack=lambda a, b, rounding: ((a, b), rounding[:4]),
                         props={"a": lambda v: v["a"], "b": lambda v: v["b"], "size": lambda v: _a((v["a"], v["b"])),
                                "round_corners": lambda v: _a(v["rounding"][:4])},
                         doc="Rectangle with per-corner rounding radii.")
Segment = shape("Segment", P.get("sdf_segment_2d"), ["a", "b"],
                props={"point_a": lambda v: _a(v["a"]), "point_b": lambda v: _a(v["b"])},
                doc="Segment between a and b (unsigned distance).")
Triangle = shape("Triangle", P.get("sdf_triangle_2d"), ["a", "b", "c"],
                 pack=lambda a, b, c: (_a(a), _a(b), _a(c)),
                 props={k: (lambda v, k=k: _a(v[k])) for k in "abc"}, doc="Triangle with vertices a, b, c.")
Sector = shape("Sector", P.get("sdf_sector"), ["radius", "angle_1", "angle_2"], doc="Circular sector.")
InfiniteSector = shape("InfiniteSector", P.get("sdf_inf_sector"), ["angle_1", "angle_2"], doc="Infinite wedge.")
Arc = shape("Arc", P.get("sdf_arc"), ["radius", "start_angle", "end_angle"], doc="Circular arc (unsigned).")


class Polygon(GenericGeometry):
    """Simple polygon through `vertices` ((3, M), M >= 3)."""

    def __init__(self, vertices):
        GenericGeometry.__init__(self, P.get("sdf_polygon_2d"), vertices)
        vertices = np.array(vertices)
        if not (vertices.shape[1] >= 3 and vertices.shape[0] >= 3):
            raise ValueError("There must be at least 3 vertices defined by their coordinates in 3D space.")
        if 3 not in vertices.shape:
            raise ValueError("The coordinates of vertices should be defined in 3D space.")
        if not vertices.shape[0] == 3:
            vertices = vertices.T
        self._vertices = vertices
        self._n_sides = vertices.shape[1]

    n_sides = property(lambda self: self._n_sides)


class _PolygonMixin:
    def polygon(self):
        """Turn the closed outline into a signed polygon (reference cores/geom_2d.py:530-555)."""
        if not self.closed:
            return self.geo_object
        self._mod.append("polygon")
        node = ModSDF("polygon", {"points": self._points.copy()}, self.geo_object)
        self.geo_object = node
        return node


class ParametricCurve(_CurveBase):
    """Curve through the samples of a user-provided parametric curve f(t, *parameters) -> (2, M)."""

    def __init__(self, parametric_curve, parametric_curve_parameters, t_range, closed=False):
        self._curve, self._c_params, self._t_range, self._closed = \
            parametric_curve, parametric_curve_parameters, t_range, closed
        GenericGeometry.__init__(self, self.sdf_closed_curve(), parametric_curve, parametric_curve_parameters, self.ts)

    def sdf_closed_curve(self):
        return P.get("closed_parametric_curve_2d" if self.closed else "sdf_parametric_curve_2d")

    steps = property(lambda self: self._t_range[2])
    t_start = property(lambda self: self._t_range[0])
    t_end = property(lambda self: self._t_range[1])
    ts = property(lambda self: np.linspace(*self._t_range))

    def shape(self):
        """Signed interior of a closed curve (reference cores/geom_2d.py:415-457)."""
        if not self.closed:
            return self.geo_object
        self._mod.append("shape")
        ts_ = np.zeros(self.steps + 1)
        ts_[:self.steps] = self.ts
        node = ModSDF("shape", {"points": np.asarray(self._curve(ts_, *self._c_params), dtype=float)},
                      self.geo_object)
        self.geo_object = node
        return node


class SegmentedParametricCurve(_CurveBase, _PolygonMixin):
    """Curve through points resampled along a poly-line at fractional indices."""

    def __init__(self, points, t_range, closed=False):
        self._points, self._t_range, self._closed = _columns(points), t_range, closed
        GenericGeometry.__init__(self, self.sdf_closed_curve(), self._points, self.ts)

    def sdf_closed_curve(self):
        return P.get("closed_segmented_curve_2d" if self.closed else "sdf_segmented_curve_2d")

    steps = property(lambda self: self._t_range[2])
    t_start = property(lambda self: self._t_range[0])
    t_end = property(lambda self: self._t_range[1])

    @property
    def ts(self):
        tt = np.linspace(self._t_range[0], self._t_range[1] - 1, self._t_range[2])
        return np.clip(tt, 0, self._points.shape[1] - 1.0001)


class SegmentedLine(_CurveBase, _PolygonMixin):
    """Poly-line through the given points."""

    def __init__(self, points, closed=False):
        self._points, self._closed = _columns(points), closed
        GenericGeometry.__init__(self, self.sdf_closed_curve(), self._points)

    def sdf_closed_curve(self):
        return P.get("closed_line_curve_2d" if self.closed else "sdf_segmented_line_2d")


class PointCloud2D(GenericGeometry):
    """Unsigned distance to the nearest point of a (2+, M) cloud (x, y used)."""

    def __init__(self, points):
        self._points = _columns(points)
        GenericGeometry.__init__(self, P.get("sdf_point_cloud_2d"), self._points)

    points = property(lambda self: self._points)


class GenericGeometry2D(GenericGeometry):
    """Reference cores/geom_2d.py:22-36: the same constructor and create / propagate as GenericGeometry."""
