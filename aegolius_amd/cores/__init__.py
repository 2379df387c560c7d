"""Same flat namespace as the reference's `spomso.cores` (reference cores/__init__.py:7-48) for the
scalar-field path: `from aegolius_amd.cores import Sphere, CombineGeometry, generate_grid ...`."""
from . import post_processing
from . import triangulation_functions
from .combine import CombineGeometry
from .combine import smoothmin_poly2, smoothmin_poly3, smoothmax_boltz
from .transformations import EuclideanTransform
from .modifications import ModifyObject
from .geom import GenericGeometry, VectorField
from .modifications import ModifyVectorObject

from .post_processing import sigmoid_falloff, positive_sigmoid_falloff, capped_exponential
from .post_processing import hard_binarization, linear_falloff
from .post_processing import relu, smooth_relu, slowstart
from .post_processing import gaussian_boundary, gaussian_falloff
from .post_processing import conv_averaging, conv_edge_detection
from .post_processing import custom_post_process

from .triangulation_functions import check_convex, check_convex_all, is_inside_triangle, is_ear, triangulate
from .triangulation_functions import interior_triangle, interior_convex, interior_polygon

from .helper_functions import resolution_conversion, generate_grid, smarter_reshape
from .helper_functions import vector_smarter_reshape, nd_vector_smarter_reshape
from .vector_functions import from_sdf
from .vector_functions import cartesian_define, cylindrical_define, spherical_define
from .vector_functions import radial_vector_field_cylindrical, radial_vector_field_spherical
from .vector_functions import hyperbolic_vector_field_cylindrical, awn_vector_field_cylindrical
from .vector_functions import aar_vector_field_cylindrical, vortex_vector_field_cylindrical, aav_vector_field_cylindrical
from .vector_functions import x_vector_field, y_vector_field, z_vector_field

from .vector_modification_functions import batch_normalize
from .vector_modification_functions import add_vectors, subtract_vectors, rescale_vectors
from .vector_modification_functions import rotate_vectors_phi, rotate_vectors_theta
from .vector_modification_functions import rotate_vectors_x_axis, rotate_vectors_y_axis, rotate_vectors_z_axis
from .vector_modification_functions import rotate_vectors_axis
from .vector_modification_functions import revolve_field_x, revolve_field_y, revolve_field_z

from .geom_vector import CartesianVectorField, CylindricalVectorField, SphericalVectorField
from .geom_vector import RadialSphericalVectorField
from .geom_vector import RadialCylindricalVectorField, HyperbolicCylindricalVectorField
from .geom_vector import AngledRadialCylindricalVectorField, WindingCylindricalVectorField
from .geom_vector import VortexCylindricalVectorField, AngledVortexCylindricalVectorField
from .geom_vector import XVectorField, YVectorField, ZVectorField
from .geom_vector import VectorFieldFromSDF

from .sdf_2D import *  # noqa: F401,F403
from .sdf_3D import *  # noqa: F401,F403

from .geom_2d import Circle, NEUCircle, NGon, Rectangle, RoundedRectangle
from .geom_2d import Segment, Triangle, Sector, InfiniteSector, Arc, Polygon
from .geom_2d import ParametricCurve, SegmentedParametricCurve, SegmentedLine
from .geom_2d import PointCloud2D

from .geom_3d import InfiniteCylinder, Cylinder, Sphere, Box, Plane, OrientedPlane, Line, Triangle3D, Quad
from .geom_3d import Torus, ChainLink, Braid, Arc3D, Cone, InfiniteCone, OrientedInfiniteCone
from .geom_3d import ParametricCurve3D, SegmentedParametricCurve3D, SegmentedLine3D
from .geom_3d import X, Y, Z
