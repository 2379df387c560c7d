"""The array-level vector operations of the reference (cores/vector_modification_functions.py:14-160), evaluated on the
GPU: each call is a two-instruction vector program (aegolius_amd._vector) on the given (3, N) array."""
import numpy as np

from .._vector import VecClosure, evaluate
from .vector_functions import cartesian_define


def _apply(vec, name, *args):
    return evaluate(VecClosure(cartesian_define).then(name, *args), vec, ())


def batch_normalize(vec):
    """Divide every vector by its norm unless the norm is 0 — in place, like the reference (:14-20)."""
    out = _apply(vec, "normalize")
    if isinstance(vec, np.ndarray) and vec.flags.writeable:
        vec[...] = out
        return vec
    return out


def add_vectors(vec, add_vec):                      # :23-28
    return _apply(vec, "add", add_vec)


def subtract_vectors(vec, subtract_vec):            # :31-36
    return _apply(vec, "subtract", subtract_vec)


def rescale_vectors(vec, scale):                    # :39-41
    return _apply(vec, "rescale", scale)


def rotate_vectors_phi(vec, phis):                  # :44-52
    return _apply(vec, "rotate_phi", phis)


def rotate_vectors_theta(vec, thetas):              # :55-68
    return _apply(vec, "rotate_theta", thetas)


def rotate_vectors_x_axis(vec, alpha):              # :71-80
    return _apply(vec, "rotate_x", alpha)


def rotate_vectors_y_axis(vec, alpha):              # :83-92
    return _apply(vec, "rotate_y", alpha)


def rotate_vectors_z_axis(vec, alpha):              # :95-103
    return _apply(vec, "rotate_z", alpha)


def rotate_vectors_axis(vec, axes, alpha):          # :106-119
    return _apply(vec, "rotate_axis", axes, alpha)


def revolve_field_x(r, vec):                        # :122-133
    return _apply(vec, "revolution_x", r)


def revolve_field_y(r, vec):                        # :136-146
    return _apply(vec, "revolution_y", r)


def revolve_field_z(r, vec):                        # :149-159
    return _apply(vec, "revolution_z", r)
