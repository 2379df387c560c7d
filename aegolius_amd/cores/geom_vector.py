"""Ready-made vector fields (reference cores/geom_vector.py:18-198): a `VectorField` bound to one of the definitions
of `vector_functions`, with the reference's constructor signatures and read-only properties."""
from .geom import VectorField
from . import vector_functions as _vf


def _fixed(name, definition, doc):
    def __init__(self):
        VectorField.__init__(self, definition)
    return type(name, (VectorField,), {"__init__": __init__, "__doc__": doc})


CartesianVectorField = _fixed("CartesianVectorField", _vf.cartesian_define,
                              "Field given by its cartesian components (x, y, z); output cartesian (:18-24).")
CylindricalVectorField = _fixed("CylindricalVectorField", _vf.cylindrical_define,
                                "Field given by its cylindrical components (r, phi, z); output cartesian (:27-33).")
SphericalVectorField = _fixed("SphericalVectorField", _vf.spherical_define,
                              "Field given by its spherical components (r, phi, theta); output cartesian (:36-42).")
RadialSphericalVectorField = _fixed("RadialSphericalVectorField", _vf.radial_vector_field_spherical,
                                    "Unit vectors pointing away from the origin; input = positions (:45-52).")
RadialCylindricalVectorField = _fixed("RadialCylindricalVectorField", _vf.radial_vector_field_cylindrical,
                                      "Unit vectors pointing away from the line x = 0, y = 0 (:55-62).")
HyperbolicCylindricalVectorField = _fixed("HyperbolicCylindricalVectorField", _vf.hyperbolic_vector_field_cylindrical,
                                          "Constructs like the reference's; create() raises TypeError as there (:65-72).")
VortexCylindricalVectorField = _fixed("VortexCylindricalVectorField", _vf.vortex_vector_field_cylindrical,
                                      "Unit vectors circling the line x = 0, y = 0 (:124-131).")
XVectorField = _fixed("XVectorField", _vf.x_vector_field, "All vectors point along x (:158-165).")
YVectorField = _fixed("YVectorField", _vf.y_vector_field, "All vectors point along y (:168-175).")
ZVectorField = _fixed("ZVectorField", _vf.z_vector_field, "All vectors point along z (:178-185).")


class WindingCylindricalVectorField(VectorField):
    """Field with winding number gamma about the z-axis (:75-95); create() raises TypeError as in the reference."""

    def __init__(self, gamma):
        self._gamma = gamma
        VectorField.__init__(self, _vf.awn_vector_field_cylindrical, (gamma,))

    @property
    def gamma(self):
        return self._gamma


class AngledRadialCylindricalVectorField(VectorField):
    """Radial cylindrical field turned about z by alpha, a number or one angle per point (:98-121)."""

    def __init__(self, alpha):
        self._alpha = alpha
        VectorField.__init__(self, _vf.aar_vector_field_cylindrical, (alpha,))

    @property
    def alpha(self):
        return self._alpha


class AngledVortexCylindricalVectorField(VectorField):
    """Vortex field turned about z by alpha, a number or one angle per point (:134-155)."""

    def __init__(self, alpha):
        self._alpha = alpha
        VectorField.__init__(self, _vf.aav_vector_field_cylindrical, (alpha,))

    @property
    def alpha(self):
        return self._alpha


class VectorFieldFromSDF(VectorField):
    """Direction of the gradient of an SDF sampled on a grid; create() takes the (N,) field (:188-198).

    Args:
        grid_resolution: Number of points along each axis of the grid on which the SDF is evaluated.
    """

    def __init__(self, grid_resolution):
        VectorField.__init__(self, _vf.from_sdf, grid_resolution)
