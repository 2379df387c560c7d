"""`GenericGeometry` — the drop-in boundary (reference cores/geom.py:15-74).

`GenericGeometry(geo_sdf, *geo_parameters)`, `.create(co)`, `.propagate(co, *ignored)` and
`.point_cloud(co)` keep the reference's signatures. `create` lowers the whole expression tree that
hangs off this object (its modifications, its Euclidean transform, and — through CombineGeometry —
every child object) to ONE fused GPU program and runs it through libsdfk.so.
"""
import numpy as np

from .._eval import evaluate_geometry
from .modifications import ModifyObject, ModifyVectorObject
from .transformations import EuclideanTransform


class GenericGeometry(EuclideanTransform, ModifyObject):
    """Constructs geometry based on an SDF.

    Args:
        geo_sdf: SDF of a geometry - geo_sdf(co, *geo_parameters): an aegolius_amd `sdf_*` function, a
            modification/combination closure returned by this package, or `other.propagate`.
        geo_parameters: Parameters of the SDF.
    """

    def __init__(self, geo_sdf, *geo_parameters):
        EuclideanTransform.__init__(self)
        ModifyObject.__init__(self, geo_sdf)
        self._geo_parameters = geo_parameters
        self._sdf = self.geo_object

    def create(self, co):
        """Signed distance field of shape (N,) on the (3, N) point cloud `co` (modifications first,
        then the Euclidean transform — reference cores/geom.py:29-43)."""
        self._sdf = self.modified_object
        return evaluate_geometry(self, co)

    def propagate(self, co, *parameters_):
        """Same as create(); extra arguments are ignored (reference cores/geom.py:45-60). Pass
        `obj.propagate` wherever an SDF function is expected to use this object as a sub-tree."""
        self._sdf = self.modified_object
        return evaluate_geometry(self, co)

    def create_resident(self, co):
        """create(), but the field stays in HBM: returns an `aegolius_amd.DeviceField` (`.numpy()`, `.select()`,
        `.gradient()`) so that thresholding and gradients run on the device without a PCIe round trip."""
        self._sdf = self.modified_object
        return evaluate_geometry(self, co, resident=True)

    def point_cloud(self, co):
        """Interior points (field <= 0) as a (3, M) cloud with z = 0 (reference cores/geom.py:62-74). Evaluation and
        mask are fused on the device — the kernels write one flag bit per point instead of the field — and only the
        indices of the interior points come back."""
        from .._eval import select_geometry
        inside = select_geometry(self, co, 0.0)
        pts = np.zeros((3, inside.size))
        pts[:2, :] = np.asarray(co)[:2, inside]
        return pts


class VectorField(ModifyVectorObject):
    """Constructs a VectorField object from a vector field function (reference cores/geom.py:213-362).

    Args:
        vf: The vector field function - vf(p, *vf_parameters): a definition of `aegolius_amd.cores.vector_functions`
            (lowered to the GPU kernel), a field returned by a modification method, or any callable returning a
            (3, N) array (run on the host; the modifications still run on the GPU).
        vf_parameters: The parameters of the vector field function.
    """

    def __init__(self, vf, *vf_parameters):
        ModifyVectorObject.__init__(self, vf)
        self._vf_parameters = vf_parameters
        self._vf = self.vf

    def _evaluate(self, p, out):
        from .._vector import evaluate
        self._vf = self.vf
        return evaluate(self._vf, p, self._vf_parameters, out)

    def create(self, p):
        """Applies the modifications and returns the map of the vector field, shape (3, N)."""
        return self._evaluate(p, "vector")

    def create_resident(self, p):
        """create(), but the field stays in HBM: returns an `aegolius_amd.DeviceVectorField` (`.numpy()`), usable as
        the input, a second field or the revolution coordinates of another chain. `p` may itself be a
        DeviceVectorField (for `VectorFieldFromSDF`: a DeviceField from `GenericGeometry.create_resident`), and so may
        every per-point operand handed to a modification or a field constructor."""
        from .._vector import evaluate
        self._vf = self.vf
        return evaluate(self._vf, p, self._vf_parameters, "vector", resident=True)

    def propagate(self, p, *parameters_):
        """Same as create(); extra arguments are ignored."""
        return self._evaluate(p, "vector")

    def x(self, p, *parameters_):
        """x component of the modified field, shape (N,)."""
        return self._evaluate(p, "x")

    def y(self, p, *parameters_):
        return self._evaluate(p, "y")

    def z(self, p, *parameters_):
        return self._evaluate(p, "z")

    def phi(self, p, *parameters_):
        """Azimuthal angle arctan2(v_y, v_x) of every vector."""
        return self._evaluate(p, "phi")

    def theta(self, p, *parameters_):
        """Polar angle arccos(v_z) of every vector."""
        return self._evaluate(p, "theta")

    def length(self, p, *parameters_):
        """Length of every vector."""
        return self._evaluate(p, "length")
