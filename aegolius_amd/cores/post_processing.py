"""The array-level scalar post-processing functions (reference cores/post_processing.py:380-642): same names,
arguments and defaults, applied to a field `u` of any shape. The arithmetic runs on the GPU through the value
instructions a tree would use (`ModifyObject.sigmoid_falloff` etc. lower to the same device functions); `u` may
also be an `aegolius_amd.DeviceField`, and then the result stays in HBM. `conv_averaging` / `conv_edge_detection`
take a GRID-shaped (2-D / 3-D) array like the reference and run the device stencil kernels.

The `PostProcess` closure-builder class of the reference (:13-375) duplicates the `ModifyObject` post-process
methods and is outside the scope of this package.
"""
from .. import _eval


def sigmoid_falloff(u, amplitude, width):
    """amplitude / (1 + exp(4 u / width)) (:380-394)."""
    return _eval.apply_value_op("sigmoid_falloff", u, {"amplitude": amplitude, "width": width})


def positive_sigmoid_falloff(u, amplitude, width):
    """The sigmoid shifted by `width` towards positive values (:397-412)."""
    return _eval.apply_value_op("positive_sigmoid_falloff", u, {"amplitude": amplitude, "width": width})


def capped_exponential(u, amplitude, width):
    """amplitude * min(exp(-4 u / width), 1) (:415-429)."""
    return _eval.apply_value_op("capped_exponential", u, {"amplitude": amplitude, "width": width})


def hard_binarization(u, threshold):
    """1.0 where u <= threshold, else 0.0 (:432-446)."""
    return _eval.apply_value_op("hard_binarization", u, {"threshold": threshold})


def linear_falloff(u, amplitude, width):
    """amplitude * clip(1 - u / width, 0, 1) (:449-463)."""
    return _eval.apply_value_op("linear_falloff", u, {"amplitude": amplitude, "width": width})


def relu(u, width=1):
    """max(u / width, 0) (:466-477)."""
    return _eval.apply_value_op("relu", u, {"width": width})


def smooth_relu(u, smooth_width, width=1, threshold=0.01):
    """Smooth approximation of the ReLU (:480-500)."""
    return _eval.apply_value_op("smooth_relu", u, {"smooth_width": smooth_width, "width": width, "threshold": threshold})


def slowstart(u, smooth_width, width=1, threshold=0.01, ground=True):
    """Smooth ReLU with a slow start, optionally grounded at zero (:503-523)."""
    return _eval.apply_value_op("slowstart", u, {"smooth_width": smooth_width, "width": width, "threshold": threshold,
                                                 "ground": ground})


def gaussian_boundary(u, amplitude, width):
    """amplitude * exp(-4 (u / width)^2) (:526-540)."""
    return _eval.apply_value_op("gaussian_boundary", u, {"amplitude": amplitude, "width": width})


def gaussian_falloff(u, amplitude, width):
    """The same on max(u, 0) (:543-558)."""
    return _eval.apply_value_op("gaussian_falloff", u, {"amplitude": amplitude, "width": width})


def conv_averaging(u, kernel_size, iterations):
    """`iterations` passes of a box filter of `kernel_size` over the grid-shaped field, reflect boundaries like
    scipy.ndimage.convolve (:561-597)."""
    if iterations == 0:
        return u
    return _eval.apply_grid_op("conv_averaging", u, {"kernel_size": kernel_size, "iterations": iterations})


def conv_edge_detection(u):
    """3 x 3 (x 1) edge-detection stencil over the grid-shaped field (:600-623)."""
    return _eval.apply_grid_op("conv_edge_detection", u, {})


def custom_post_process(u, function, parameters):
    """function(u, *parameters): user code, called on the host exactly like the reference does (:626-642)."""
    return function(u, *parameters)
