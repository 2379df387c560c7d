"""`ModifyObject` — API mirror of the reference's modification layer (reference
cores/modifications.py:30-1663): the same 52 method names, positional/keyword arguments, defaults,
recorded history (`modifications`) and `direct=` behaviour.

Where the reference wraps `self.geo_object` in a new Python closure, each method here records a
`ModSDF` node that captures the current chain (same snapshot semantics) and returns it; the node is
callable like the closure it replaces. The math of every modification lives in
csrc/sdfk_device.h; its host-side constant folding in aegolius_amd/_mods.py.
"""
import numpy as np

from .._ir import ModSDF, SDFExpr
from .._lower import as_expr


class ModifyObject:
    """All modifications that can be applied to a scalar field.

    Attributes:
        original_geo_object: SDF as constructed.
        geo_object: SDF with the modifications applied so far (outermost = most recent).
    """

    def __init__(self, geo_object):
        self._mod = []
        geo_object = as_expr(geo_object)
        self.original_geo_object = geo_object
        self.geo_object = geo_object

    modifications = property(lambda self: self._mod, doc="Chronological list of applied modifications.")
    modified_object = property(lambda self: self.geo_object, doc="SDF of the modified geometry.")
    original_object = property(lambda self: self.original_geo_object, doc="SDF of the unmodified geometry.")

    def _wrap(self, name, args, install=True, second=None, second_params=None, label=None):
        self._mod.append(label or name)
        node = ModSDF(name, args, self.geo_object, second, second_params)
        if install:
            self.geo_object = node
        return node

    # ---- shape-changing -------------------------------------------------------------------------
    def elongation(self, elongate_vector):
        return self._wrap("elongation", {"ev": np.array(elongate_vector, dtype=float)})

    def rounding(self, rounding_radius):
        return self._wrap("rounding", {"rounding_radius": rounding_radius})

    def rounding_cs(self, rounding_radius, bb_size):
        return self._wrap("rounding_cs", {"rounding_radius": rounding_radius, "bb_size": bb_size})

    def boundary(self):
        return self._wrap("boundary", {})

    def signed_old(self, co_resolution):
        return self._wrap("signed_old", {"co_resolution": co_resolution}, label="signed")

    def signed(self, co_resolution):
        return self._wrap("signed", {"co_resolution": co_resolution})

    def invert(self, direct=False):
        return self._wrap("invert", {}, install=not direct)

    def sign(self, direct=False):
        return self._wrap("sign", {}, install=not direct)

    def recover_volume(self, interior):
        return self._wrap("recover_volume", {}, second=interior)

    def define_volume(self, interior, interior_parameters):
        return self._wrap("define_volume", {}, second=interior, second_params=tuple(interior_parameters))

    def onion(self, thickness):
        return self._wrap("onion", {"thickness": thickness})

    def concentric(self, width):
        return self._wrap("concentric", {"width": width})

    def revolution(self, radius):
        return self._wrap("revolution", {"radius": radius})

    def axis_revolution(self, radius, angle):
        return self._wrap("axis_revolution", {"radius": radius, "angle": angle})

    def extrusion(self, distance):
        return self._wrap("extrusion", {"distance": distance})

    def twist(self, pitch):
        return self._wrap("twist", {"pitch": pitch})

    def bend(self, radius, angle):
        return self._wrap("bend", {"radius": radius, "angle": angle})

    def shear_xz(self, angle):
        return self._wrap("shear_xz", {"angle": angle})

    def shear_yz(self, angle):
        return self._wrap("shear_yz", {"angle": angle})

    def shear_xy(self, angle):
        return self._wrap("shear_xy", {"angle": angle})

    def shear_zy(self, angle):
        return self._wrap("shear_zy", {"angle": angle})

    def shear_yx(self, angle):
        return self._wrap("shear_yx", {"angle": angle})

    def shear_zx(self, angle):
        return self._wrap("shear_zx", {"angle": angle})

    def shear(self, angle, sheared_axis, fixed_axis):
        return self._wrap("shear", {"angle": angle, "sheared_axis": sheared_axis, "fixed_axis": fixed_axis})

    def displacement(self, displacement_function, displacement_function_parameters):
        return self._wrap("displacement", {}, second=displacement_function,
                          second_params=tuple(displacement_function_parameters))

    # ---- repetition / instancing ------------------------------------------------------------------
    def infinite_repetition(self, distances):
        return self._wrap("infinite_repetition", {"distances": np.array(distances, dtype=float)})

    def finite_repetition(self, size, repetitions):
        return self._wrap("finite_repetition", {"size": np.array(size, dtype=float),
                                                "repetitions": np.array(repetitions, dtype=float)})

    def finite_repetition_rescaled(self, size, repetitions, instance_size, padding):
        return self._wrap("finite_repetition_rescaled",
                          {"size": np.array(size, dtype=float), "repetitions": np.array(repetitions, dtype=float),
                           "instance_size": np.array(instance_size, dtype=float),
                           "padding": np.array(padding, dtype=float)})

    def symmetry(self, axis):
        return self._wrap("symmetry", {"axis": axis})

    def mirror(self, a, b):
        return self._wrap("mirror", {"a": np.array(a, dtype=float), "b": np.array(b, dtype=float)})

    def rotational_symmetry(self, n, radius, phase):
        return self._wrap("rotational_symmetry", {"n": n, "radius": radius, "phase": phase})

    def linear_instancing(self, n, a, b):
        return self._wrap("linear_instancing", {"n": n, "a": np.array(a, dtype=float), "b": np.array(b, dtype=float)})

    def curve_instancing(self, f, f_parameters, t_range):
        return self._wrap("curve_instancing", {"f": f, "f_parameters": f_parameters, "t_range": t_range})

    def aligned_curve_instancing(self, f, f_parameters, t_range):
        return self._wrap("aligned_curve_instancing", {"f": f, "f_parameters": f_parameters, "t_range": t_range})

    def fully_aligned_curve_instancing(self, f, f_parameters, t_range):
        return self._wrap("fully_aligned_curve_instancing",
                          {"f": f, "f_parameters": f_parameters, "t_range": t_range})

    # ---- re-positioning of the SDF itself -----------------------------------------------------------
    def move_sdf(self, move_vector):
        return self._wrap("move_sdf", {"move_vector": np.array(move_vector, dtype=float)})

    def scale_sdf(self, scale_factor):
        return self._wrap("scale_sdf", {"scale_factor": scale_factor})

    def rotate_sdf(self, rotation_matrix):
        return self._wrap("rotate_sdf", {"rotation_matrix": np.array(rotation_matrix, dtype=float)})

    def custom_modification(self, modification, modification_parameters, modification_name="custom"):
        return self._wrap("custom_modification", {"modification": modification,
                                                  "modification_parameters": modification_parameters},
                          label=modification_name)

    # ---- value post-processing (reference cores/post_processing.py:380-623) ---------------------------
    def sigmoid_falloff(self, amplitude, width):
        return self._wrap("sigmoid_falloff", {"amplitude": amplitude, "width": width})

    def positive_sigmoid_falloff(self, amplitude, width):
        return self._wrap("positive_sigmoid_falloff", {"amplitude": amplitude, "width": width})

    def capped_exponential(self, amplitude, width):
        return self._wrap("capped_exponential", {"amplitude": amplitude, "width": width})

    def hard_binarization(self, threshold):
        return self._wrap("hard_binarization", {"threshold": threshold})

    def linear_falloff(self, amplitude, width):
        return self._wrap("linear_falloff", {"amplitude": amplitude, "width": width})

    def relu(self, width):
        return self._wrap("relu", {"width": width})

    def smooth_relu(self, smooth_width, width=1, threshold=0.01):
        return self._wrap("smooth_relu", {"smooth_width": smooth_width, "width": width, "threshold": threshold})

    def slowstart(self, smooth_width, width=1, threshold=0.01, ground=True):
        return self._wrap("slowstart", {"smooth_width": smooth_width, "width": width, "threshold": threshold,
                                        "ground": ground})

    def gaussian_boundary(self, amplitude, width):
        return self._wrap("gaussian_boundary", {"amplitude": amplitude, "width": width})

    def gaussian_falloff(self, amplitude, width):
        return self._wrap("gaussian_falloff", {"amplitude": amplitude, "width": width})

    def conv_averaging(self, kernel_size, iterations, co_resolution):
        return self._wrap("conv_averaging", {"kernel_size": kernel_size, "iterations": iterations,
                                             "co_resolution": co_resolution})

    def conv_edge_detection(self, co_resolution):
        return self._wrap("conv_edge_detection", {"co_resolution": co_resolution})

    def custom_post_process(self, function, parameters, post_process_name="custom"):
        return self._wrap("custom_post_process", {"function": function, "parameters": parameters},
                          label=post_process_name)


class ModifyVectorObject:
    """All the modifications which can be applied to a vector field (reference cores/modifications.py:1666-1975): same
    methods, same bookkeeping (`modifications`, `modified_object`, `original_object`, the public `vf` /
    `original_vf`). Every method wraps the current field and returns the new one — an `aegolius_amd._vector.VecClosure`,
    callable as `vf(p, *params)` like the reference's closures, evaluated as one GPU kernel.

    Args:
        vf: Vector field function vf(p, *parameters).
    """

    def __init__(self, vf):
        from .._vector import as_closure
        self._mod = []
        self.original_vf = vf
        self.vf = as_closure(vf)

    @property
    def modifications(self):
        """All the modifications which were applied to the vector field in chronological order."""
        return self._mod

    @property
    def modified_object(self):
        return self.vf

    @property
    def original_object(self):
        return self.original_vf

    def _wrap(self, name, *args):
        from .._vector import as_closure
        self._mod.append(name)
        self.vf = as_closure(self.vf).then(name, *args)
        return self.vf

    def add(self, second_field):
        """Adds a number, a vector, or a vector field of the same shape (:1712-1730)."""
        return self._wrap("add", second_field)

    def subtract(self, second_field):
        return self._wrap("subtract", second_field)

    def rescale(self, second_field):
        """Multiplies by a number, a per-point number or a field of the same shape (:1752-1770)."""
        return self._wrap("rescale", second_field)

    def rotate_phi(self, phi):
        """Turns the vectors about z by a spatially dependent or independent angle (:1772-1790)."""
        return self._wrap("rotate_phi", phi)

    def rotate_theta(self, theta):
        """Turns the vectors in their own meridional plane (:1792-1810)."""
        return self._wrap("rotate_theta", theta)

    def rotate_x(self, alpha):
        return self._wrap("rotate_x", alpha)

    def rotate_y(self, alpha):
        return self._wrap("rotate_y", alpha)

    def rotate_z(self, alpha):
        return self._wrap("rotate_z", alpha)

    def rotate_axis(self, axis, alpha):
        """Turns the vectors about one axis, or one axis per point, by alpha (:1872-1893); axes are used as given."""
        return self._wrap("rotate_axis", axis, alpha)

    def revolution_x(self, co):
        """Revolves a 2D vector field about the x-axis of the coordinate system `co` (3, N) (:1895-1914)."""
        return self._wrap("revolution_x", co)

    def revolution_y(self, co):
        return self._wrap("revolution_y", co)

    def revolution_z(self, co):
        return self._wrap("revolution_z", co)

    def normalize(self):
        """Divides every vector by its length; zero vectors stay zero (:1956-1975)."""
        return self._wrap("normalize")


__all__ = ["ModifyObject", "ModifyVectorObject", "SDFExpr"]
