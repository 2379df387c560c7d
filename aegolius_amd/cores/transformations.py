"""Euclidean transform state of a geometry — API mirror of the reference's `EuclideanTransform`
(reference cores/transformations.py:12-264): same method names, argument meaning, recorded history
and exceptions. Only the state lives here; applying it to coordinates is instruction XFORM / XLATE
of the GPU program (aegolius_amd/_lower.py: Lowerer.lower_node).
"""
import numpy as np
from scipy.spatial.transform import Rotation

_NUMBER = (float, int)


def _padded3(values, what):
    arr = np.asarray(values)
    if arr.size > 3:
        raise SyntaxError(f"Array {arr} is of incorrect size!")
    out = np.zeros(3)
    out[:arr.size] = arr.ravel()
    return out, arr.size


class EuclideanTransform:
    """Translation `center`, uniform `scale` and rotation of one geometry.

    Evaluation applies  co' = (R^T co)/s - R^T t  and returns  s * f(co')  (reference :232-242)."""

    def __init__(self):
        self._et = []
        self._center = np.zeros(3)
        self._scale = 1.0
        self._rot_matrix = np.eye(3)
        self._angle = 0.0
        self._axis = np.asarray((0.0, 0.0, 1.0))

    # ---- read-only views (reference :25-77) ----
    transformations = property(lambda self: self._et, doc="Chronological list of applied transformations.")
    center = property(lambda self: self._center, doc="Position of the geometry.")
    scale = property(lambda self: self._scale, doc="Scale factor.")
    rotation_matrix = property(lambda self: self._rot_matrix, doc="Rotation matrix (3, 3).")
    rotation_axis = property(lambda self: self._axis, doc="Axis of rotation.")
    rotation_angle = property(lambda self: self._angle, doc="Angle of rotation about the axis.")

    # ---- translation (reference :79-106) ----
    def set_location(self, center):
        self._et.append("set_location")
        vec, k = _padded3(center, "center")
        self._center[:k] = vec[:k]

    def move(self, move_vector):
        self._et.append("move")
        vector = np.asarray(move_vector)
        if vector.size > 3:
            raise SyntaxError(f"Array {vector} is of incorrect size!")
        self._center += vector

    # ---- scale (reference :108-134) ----
    def set_scale(self, scale):
        self._et.append("set_scale")
        if not isinstance(scale, _NUMBER):
            raise TypeError("Scale must be a float or an int")
        self._scale = scale

    def rescale(self, scale):
        self._et.append("rescale")
        if not isinstance(scale, _NUMBER):
            raise TypeError("Scale must be a float or an int")
        self._scale *= scale

    # ---- rotation (reference :136-230) ----
    @staticmethod
    def get_rotation_matrix(angle, axis):
        axis_, _ = _padded3(axis, "axis")
        if not isinstance(angle, _NUMBER):
            raise TypeError("Rotation angle must be a float or an int")
        return Rotation.from_rotvec(angle * axis_).as_matrix(), angle, axis_

    def set_rotation(self, angle, axis):
        self._et.append("set_rotation")
        self._rot_matrix, self._angle, self._axis = self.get_rotation_matrix(angle, axis)

    def rotate_rotvec(self, angle, axis):
        axis = np.asarray(axis)
        if np.array_equal(axis, np.zeros(3)[:axis.size]):
            raise ValueError("Axis cannot be zero!")
        unit = axis / np.linalg.norm(axis)
        self.rotate_matrix(self.get_rotation_matrix(angle, unit)[0])

    def rotate_matrix(self, rotation_matrix):
        m = np.asarray(rotation_matrix, dtype=float)
        if m.shape == (1, 3, 3):          # the reference's rotate(matrix) wraps its 1-tuple this way (:221-223)
            m = m[0]
        if m.shape != (3, 3):
            raise ValueError("rotation matrix must have shape (3, 3); got %r" % (m.shape,))
        self._rot_matrix = np.matmul(m, self._rot_matrix)
        rotvec = Rotation.from_matrix(self._rot_matrix).as_rotvec()
        self._angle = np.linalg.norm(rotvec)
        self._axis = rotvec / self._angle if self._angle != 0 else np.asarray((0.0, 0.0, 1.0))

    def rotate(self, *inputs):
        """rotate(matrix)  or  rotate(angle, axis)."""
        self._et.append("rotate")
        if len(inputs) == 1:
            self.rotate_matrix(inputs[0])
        elif len(inputs) == 2:
            self.rotate_rotvec(inputs[0], np.asarray(inputs[1]))
        elif len(inputs) > 2:
            raise SyntaxError("Wrong number of inputs!")

    # ---- evaluation through the transform (reference cores/transformations.py:232-264) -----------------
    @staticmethod
    def apply_ec_transforms(function_, co_, params_, rm, tm, sm):
        """`sm * function_(co', *params_)` with `co' = (rm^T co_) / sm - rm^T tm`, evaluated on the GPU.
        `function_`: anything this package accepts as an SDF function (sdf_* functions, modification / combine
        closures, `other.propagate`, or a plain Python callable — the latter with its code on the host)."""
        from .._eval import evaluate_geometry
        from .._lower import as_expr

        class _Transformed:                      # the node protocol the lowering walks
            rotation_matrix = np.asarray(rm, dtype=float)
            center = np.asarray(tm, dtype=float)
            scale = sm
            modified_object = as_expr(function_)
            _geo_parameters = tuple(params_)
        return evaluate_geometry(_Transformed, co_)

    def apply(self, function_, co_, params_):
        """Apply this object's transformations to an SDF function: field of shape (N,)."""
        return self.apply_ec_transforms(function_, co_, params_, self.rotation_matrix, self.center, self.scale)
