"""`from_sdf` — direction field of an SDF (reference cores/vector_functions.py:130-140).

The gradient stencil and the normalisation run on the GPU (libsdfk.so, sdfk_field_gradient); there is no NumPy
evaluation here. The analytic vector fields of the reference's vector-field path are out of scope.
"""
import numpy as np

from .. import _engine
from .._eval import _grid_shape, config


def from_sdf(sdf_, co_resolution):
    """Unit vectors along the gradient of the field `sdf_` sampled on the grid `co_resolution`.

    Args:
        sdf_: (N,) field — an ndarray, or an `aegolius_amd.DeviceField` from `create_resident` (no upload then).
        co_resolution: resolution of the grid the field was created on (what `generate_grid` returned / was given).
    Returns:
        (D, N) array, D = number of entries of `co_resolution`: numpy.gradient of the reshaped field with unit
        spacing, every vector divided by its norm (zero vectors stay zero).
    """
    dimensions = np.asarray(co_resolution).shape[0]
    own = not isinstance(sdf_, _engine.DeviceField)
    field = _engine.DeviceField.from_host(np.asarray(sdf_).ravel(), config.device) if own else sdf_
    try:
        shape = _grid_shape(field.n, co_resolution)
        if len(shape) != dimensions:
            raise NotImplementedError("from_sdf: a %d-entry resolution on a field of shape %r" % (dimensions, shape))
        vec = field.gradient(shape, normalize=True)
    finally:
        if own:
            field.free()
    if config.output_dtype is not np.float32:
        vec = vec.astype(config.output_dtype)
    return vec


# ---- field definitions (reference cores/vector_functions.py:15-127) -------------------------------------------------
# Each is a tag the lowering recognises (aegolius_amd._vector.LEAVES); called directly it evaluates on the GPU.
def _definition(name, doc, first, rest):
    """`first`: name of the point argument; `rest`: names of the further arguments, or "*p" for the reference's catch-all."""
    import inspect
    P = inspect.Parameter
    params = [P(first, P.POSITIONAL_OR_KEYWORD)]
    params += [P("p", P.VAR_POSITIONAL)] if rest == "*p" else [P(n, P.POSITIONAL_OR_KEYWORD) for n in rest]
    sig = inspect.Signature(params)

    def fn(*args, **kwargs):
        from .._vector import VecClosure, evaluate
        bound = sig.bind(*args, **kwargs)
        return evaluate(VecClosure(fn), bound.args[0], bound.args[1:])
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = doc
    fn.__signature__ = sig
    fn._vec_leaf = name
    return fn


cartesian_define = _definition("cartesian_define", "(ux, uy, uz) = p (:15-20).", "p", ())
spherical_define = _definition("spherical_define", "(r, phi, theta) = p -> r (cos phi sin theta, sin phi sin theta, cos theta) (:23-32).", "p", ())
cylindrical_define = _definition("cylindrical_define", "(r, phi, z) = p -> (r cos phi, r sin phi, z) (:35-43).", "p", ())
radial_vector_field_spherical = _definition("radial_vector_field_spherical", "r / |r|, zero at the origin (:46-48).", "r", "*p")
radial_vector_field_cylindrical = _definition("radial_vector_field_cylindrical", "(x, y, 0) / |(x, y)| (:51-55).", "r", "*p")
hyperbolic_vector_field_cylindrical = _definition(
    "hyperbolic_vector_field_cylindrical", "Raises TypeError for every input, as the reference does (:58-61).", "r", "*p")
awn_vector_field_cylindrical = _definition(
    "awn_vector_field_cylindrical", "Raises TypeError for every input, as the reference does (:64-68).", "r", ("gamma",))
vortex_vector_field_cylindrical = _definition("vortex_vector_field_cylindrical", "The radial cylindrical field turned by 90 degrees (:71-79).", "r", "*p")
aar_vector_field_cylindrical = _definition("aar_vector_field_cylindrical", "The radial cylindrical field turned by alpha (:82-94).", "r", ("alpha",))
aav_vector_field_cylindrical = _definition("aav_vector_field_cylindrical", "The vortex field turned by alpha (:97-109).", "r", ("alpha",))
x_vector_field = _definition("x_vector_field", "(1, 0, 0) everywhere (:112-115).", "r", "*p")
y_vector_field = _definition("y_vector_field", "(0, 1, 0) everywhere (:118-121).", "r", "*p")
z_vector_field = _definition("z_vector_field", "(0, 0, 1) everywhere (:124-127).", "r", "*p")
from_sdf._vec_leaf = "from_sdf"
