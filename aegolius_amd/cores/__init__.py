"""Same flat namespace as the reference's `spomso.cores` (reference cores/__init__.py:7-48) for the
scalar-field path: `from aegolius_amd.cores import Sphere, CombineGeometry, generate_grid ...`."""
from .combine import CombineGeometry
from .transformations import EuclideanTransform
from .modifications import ModifyObject
from .geom import GenericGeometry

from .helper_functions import resolution_conversion, generate_grid, smarter_reshape
from .helper_functions import vector_smarter_reshape, nd_vector_smarter_reshape
from .vector_functions import from_sdf

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
