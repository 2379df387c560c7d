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
