"""Workflows of the reference's own example scripts that are PIPELINES rather than one tree: the five in-scope vector
examples (Code/examples/vector/*.py) and two scalar ones (post-processing functions on a field; cloud -> field -> interior
points -> cloud -> field), restated over the operator API so that the real reference (fixture generator, build
container), the oracle and aegolius_amd can all walk them. They run over several rows of SURVEY.md §8: an SDF tree (a), a
grid-neighbourhood smoothing (f.1), nearest-point leaves at scale (f.2), interior selection and the gradient direction of
the field (f.3), a vector-field chain with per-point arguments and six read-outs (f.4).

A workflow is `fn(ns, ev, hook, size, res, **variant) -> {name: array}`:
  ns    the namespace under test (`spomso.cores` or `aegolius_amd.cores`), used to BUILD objects;
  ev    how objects are EVALUATED: `ProductEvaluator` calls the objects' own methods (reference and aegolius_amd alike),
        `OracleEvaluator` (tests/test_example_pipelines.py) hands the recorded trees / chains to `oracle/`;
  hook  `hook(stage, value) -> value` is called on every intermediate array: the generator records it and goes on from its
        fp32 rounding ("identical grids" for every stage: `continue_from`); the tests compare their value with the record
        and CONTINUE FROM THE RECORD the same way, so every stage is checked on exactly the reference's input — a gradient
        of an SDF does not amplify the previous stage's rounding into the next comparison.
`tests/golden/generate_example_pipeline_golden.py` EXECUTES each script against the real reference and checks that the
workflow below, walked by the real reference with a pass-through hook at the script's own resolution, reproduces the
script's arrays bit for bit; the scripts' text stays in /root/reference.
"""
import importlib

import numpy as np

EXAMPLES = {}
READ_OUTS = ("create", "x", "y", "z", "phi", "theta", "length")
# script variable of each read-out (the scripts reshape the scalar ones to the grid afterwards)
SCRIPT_VARIABLES = {"create": "final_field", "x": "x", "y": "y", "z": "z", "phi": "phi", "theta": "theta", "length": "length"}


def example(name, script, size, full_res, res, overrides=None, raises=None, outputs=None, cwd=None, **variant):
    """raises: the exception type the SCRIPT ITSELF ends with in the reference for this variant (two of the built-in
    fields cannot be evaluated by the reference: C/geom_vector.py hands `cylindrical_define` three arguments); the
    generator checks that the script and the workflow both raise it, the tests that this package raises the same.
    outputs: {name the workflow returns: (script variable, the script reshaped it to the grid afterwards)}, by default the
    seven read-outs of a vector field; script: relative to Code/examples/ ; cwd: the directory a script that looks for the
    reference's data files relative to os.getcwd() wants to be run from."""
    def deco(fn):
        assert name not in EXAMPLES, name
        EXAMPLES[name] = dict(run=fn, script=script, size=size, full_res=full_res, res=res, overrides=overrides or {},
                              variant=variant, raises=raises, cwd=cwd,
                              outputs=outputs or {r: (SCRIPT_VARIABLES[r], r != "create") for r in READ_OUTS})
        return fn
    return deco


def continue_from(stage, recorded):
    """What a walk goes on with after `stage`: the recorded array rounded to fp32 values (outputs are ends of the walk)."""
    recorded = np.asarray(recorded, dtype=np.float64)
    return recorded.copy() if stage.startswith("out/") else recorded.astype(np.float32).astype(np.float64)


def mod(ns, name):
    """Sub-module of the namespace under test (the reference does not re-export every class at the top)."""
    return importlib.import_module(ns.__name__ + "." + name)


class ProductEvaluator:
    """Evaluation through the objects' own methods: the reference's closures, or aegolius_amd's GPU path."""

    def __init__(self, ns):
        self.ns = ns

    def sdf(self, tree, co):
        return tree.create(co)

    def vector(self, field, arg, read_out):
        return getattr(field, read_out)(arg)

    def conv_averaging(self, grid, kernel_size, iterations):
        return mod(self.ns, "post_processing").conv_averaging(grid, kernel_size, iterations)

    def linear_falloff(self, u, amplitude, width):
        return mod(self.ns, "post_processing").linear_falloff(u, amplitude, width)

    def smarter_reshape(self, pattern, resolution):
        return mod(self.ns, "helper_functions").smarter_reshape(pattern, resolution)

    def post(self, name, u, **kwargs):
        return getattr(mod(self.ns, "post_processing"), name)(u, **kwargs)

    def point_cloud(self, tree, co):
        return tree.point_cloud(co)

    def batch_normalize(self, vec):
        return mod(self.ns, "vector_modification_functions").batch_normalize(vec)


def read_outs(ev, hook, field, arg):
    return {r: hook("out/" + r, ev.vector(field, arg, r)) for r in READ_OUTS}


# ---- buildin_vector_fields.py -----------------------------------------------------------------------------------------
_BUILDIN = {"RADIAL_SPHERICAL": ("RadialSphericalVectorField", ()), "RADIAL_CYLINDRICAL": ("RadialCylindricalVectorField", ()),
            "HYPERBOLIC_CYLINDRICAL": ("HyperbolicCylindricalVectorField", ()), "AWN": ("WindingCylindricalVectorField", (2,)),
            "AAR": ("AngledRadialCylindricalVectorField", (np.pi / 6,)), "VORTEX": ("VortexCylindricalVectorField", ()),
            "AAV": ("AngledVortexCylindricalVectorField", (np.pi / 6,)), "X": ("XVectorField", ()), "Y": ("YVectorField", ()),
            "Z": ("ZVectorField", ())}


def buildin(ns, ev, hook, size, res, field_type):
    coor, _ = mod(ns, "helper_functions").generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    cls, args = _BUILDIN[field_type]
    final = getattr(mod(ns, "geom_vector"), cls)(*args)
    return read_outs(ev, hook, final, coor)


for _t in _BUILDIN:
    example("buildin_" + _t.lower(), "vector/buildin_vector_fields.py", (4, 4, 4), (100, 100, 5), (24, 24, 5),
            overrides={"field_type": _t}, raises=TypeError if _t in ("HYPERBOLIC_CYLINDRICAL", "AWN") else None,
            field_type=_t)(buildin)


# ---- custom_vector_field.py -------------------------------------------------------------------------------------------
@example("custom_radial_order3", "vector/custom_vector_field.py", (25, 25, 25), (50, 50, 50), (14, 14, 14))
def custom(ns, ev, hook, size, res):
    coor, _ = mod(ns, "helper_functions").generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))

    def radial(co_cloud_, order_):                              # a user callable as the vector field: host code
        u = np.linalg.norm(co_cloud_.copy(), axis=0, ord=order_)
        vec = np.asarray(np.gradient(ev.smarter_reshape(u, res))).reshape(len(res), -1)
        return ev.batch_normalize(vec)

    final = mod(ns, "geom").VectorField(radial, 3)
    return read_outs(ev, hook, final, coor)


# ---- from_components.py -----------------------------------------------------------------------------------------------
def components(ns, ev, hook, size, res, define_type):
    gv = mod(ns, "geom_vector")
    coor, _ = mod(ns, "helper_functions").generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    r_ = np.ones(coor.shape[1])
    phi_ = np.pi * coor[0] / (size[0] / 2)
    theta_ = np.pi * coor[2] / size[2] + np.pi / 2
    if define_type == "XYZ":
        coordinates = np.asarray((r_ * np.cos(phi_) * np.sin(theta_), r_ * np.sin(phi_) * np.sin(theta_), r_ * np.cos(theta_)))
        final = gv.CartesianVectorField()
    elif define_type == "CYLINDRICAL":
        coordinates = np.asarray((r_ * np.sin(theta_), phi_, r_ * np.cos(theta_)))
        final = gv.CylindricalVectorField()
    else:
        coordinates = np.asarray((r_, phi_, theta_))
        final = gv.SphericalVectorField()
    coordinates = hook("coordinates", coordinates)
    final.rotate_phi(np.pi / 2)
    final.rotate_theta(-np.pi / 4)
    final.rescale(hook("scale", np.abs(coor[2] / size[2]) + 0.1))
    second = gv.SphericalVectorField()
    second_coordinates = hook("second_coordinates", np.asarray((r_, phi_ * 0, -theta_)))
    final.add(hook("second", ev.vector(second, second_coordinates, "create")))
    final.normalize()
    return read_outs(ev, hook, final, coordinates)


for _t in ("XYZ", "CYLINDRICAL", "SPHERICAL"):
    example("components_" + _t.lower(), "vector/from_components.py", (10, 10, 10), (100, 100, 50), (20, 20, 12),
            overrides={"define_type": _t}, define_type=_t)(components)


# ---- revolve_vector_field.py ------------------------------------------------------------------------------------------
def revolve(ns, ev, hook, size, res, revolve_axis):
    coor, _ = mod(ns, "helper_functions").generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    vfs = mod(ns, "geom_3d").X(0)
    vfs.rotate(np.pi / 4, (0, 0, 1))
    pattern = hook("sdf", ev.sdf(vfs, coor))
    final = mod(ns, "geom_vector").VectorFieldFromSDF(res)
    getattr(final, "revolution_" + revolve_axis.lower())(coor)
    return read_outs(ev, hook, final, pattern)


for _t in "XYZ":
    example("revolve_" + _t.lower(), "vector/revolve_vector_field.py", (100, 100, 100), (50, 50, 50), (14, 14, 14),
            overrides={"revolve_axis": _t}, revolve_axis=_t)(revolve)


# ---- sdf_vector_field.py ----------------------------------------------------------------------------------------------
def _lemniscate(t, amplitude):
    return np.asarray((amplitude * np.cos(t) / (1 + np.sin(t) ** 2), amplitude * np.cos(t) * np.sin(t) / (1 + np.sin(t) ** 2)))


def waveguide(ns, ev, hook, size, res, spline_type):
    hf, g2 = mod(ns, "helper_functions"), mod(ns, "geom_2d")
    coor, _ = hf.generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    if spline_type == "CIRCLE":
        wg = g2.Circle(30)
        wg.boundary()
    elif spline_type == "LINE":
        wg = g2.Segment((-20, 0, 0), (20, 0, 0))
        wg.rotate(np.pi / 12, (0, 0, 1))
    elif spline_type == "SEGMENTED_LINE":
        points = np.asarray(((-20, 30, 0), (20, 45, 0), (35, -35, 0), (-40, -40, 0))) * 0.95
        wg = g2.SegmentedLine(points, True)
    elif spline_type == "PARAMETRIC_CURVE":
        wg = g2.ParametricCurve(_lemniscate, (40,), (0, 2 * np.pi, 501), True)
        wg.rotate(-np.pi / 4, (0, 0, 1))
    else:
        p1 = _lemniscate(np.linspace(0, 2 * np.pi, 101), 40)
        wg1 = g2.SegmentedParametricCurve(p1, (0, p1.shape[1], 501), True)
        wg1.rotate(-np.pi / 4, (0, 0, 1))
        p2 = _lemniscate(np.linspace(0, 2 * np.pi, 101), 40)
        wg2 = g2.SegmentedParametricCurve(p2, (0, p2.shape[1], 501), True)
        wg2.rotate(np.pi / 4, (0, 0, 1))
        wg = mod(ns, "combine").CombineGeometry("UNION").combine(wg1, wg2)
    pattern = hook("sdf", ev.sdf(wg, coor))
    smooth = ev.conv_averaging(ev.smarter_reshape(pattern, res), (5, 5, 1), 1)
    pattern = hook("smooth", np.asarray(smooth).reshape(pattern.shape))
    final = mod(ns, "geom_vector").VectorFieldFromSDF(res)
    final.rotate_z(hook("phis", ev.linear_falloff(pattern, np.pi / 2, 20)))
    final.rotate_axis((1, 0, 0), hook("thetas", ev.linear_falloff(pattern, np.pi / 6, 20)))
    return read_outs(ev, hook, final, pattern)


for _t in ("CIRCLE", "LINE", "SEGMENTED_LINE", "PARAMETRIC_CURVE", "SEGMENTED_PARAMETRIC_CURVE"):
    example("waveguide_" + _t.lower(), "vector/sdf_vector_field.py", (100, 100, 5.5), (100, 100, 11), (30, 30, 5),
            overrides={"spline_type": _t}, spline_type=_t)(waveguide)


# ---- scalar/2D/post_processing_scalar_2D.py: a field and the seven predefined post-processing functions ------------------
_POST = {"ce": ("capped_exponential", dict(amplitude=1.0, width=0.5)), "rl": ("relu", dict(width=1.0)),
         "gb": ("gaussian_boundary", dict(amplitude=1.0, width=0.5)), "lf": ("linear_falloff", dict(amplitude=1.0, width=0.5)),
         "sf": ("sigmoid_falloff", dict(amplitude=1.0, width=0.5)), "gf": ("gaussian_falloff", dict(amplitude=1.0, width=0.5)),
         "hb": ("hard_binarization", dict(threshold=0))}


@example("post_processing_scalar_2D", "scalar/2D/post_processing_scalar_2D.py", (4, 4), (400, 400), (40, 40),
         outputs={k: (k, False) for k in _POST})
def post_processing(ns, ev, hook, size, res):
    coor, _ = mod(ns, "helper_functions").generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    pattern = hook("sdf", ev.sdf(mod(ns, "geom_2d").Circle(1), coor))
    field = ev.smarter_reshape(pattern, res)
    return {k: hook("out/" + k, ev.post(fn, field, **kw)) for k, (fn, kw) in _POST.items()}


# ---- scalar/2D/erosion_dilation_image_2D.py: cloud -> field -> interior points -> cloud -> field ---------------------------
def shapes(ns, ev, hook, size, res, morphology):
    from example_scenes import image_cloud
    g2 = mod(ns, "geom_2d")
    coor, _ = mod(ns, "helper_functions").generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    point_cloud = g2.PointCloud2D(image_cloud("shapes"))       # 201,874 points: tests/golden/image_clouds.npz
    if morphology == "DILATE":
        point_cloud.rounding(0.01)
    if morphology == "ERODE":
        point_cloud.rounding(-0.01)
    point_cloud_2 = mod(ns, "geom").GenericGeometry(point_cloud.propagate, ())
    point_cloud_2.rounding(0.2)
    point_cloud_2.onion(0.02)
    final = mod(ns, "combine").CombineGeometry("UNION2").combine(point_cloud, point_cloud_2)
    hook("sdf", ev.sdf(final, coor))                            # (the script never looks at this field: recorded for the replay)
    new_cloud = hook("new_cloud", ev.point_cloud(final, coor))
    return {"final_pattern": hook("out/final_pattern", ev.sdf(g2.PointCloud2D(new_cloud), coor))}


for _t in ("NOTHING", "DILATE", "ERODE"):
    example("erosion_dilation_" + _t.lower(), "scalar/2D/erosion_dilation_image_2D.py", (4, 2), (800, 400), (40, 20),
            overrides={"morphology": _t}, outputs={"final_pattern": ("final_pattern", False)}, cwd="scalar/2D",
            morphology=_t)(shapes)


# ---- scalar/2D/surface_reconstruction_2D.py: sparse cloud -> (smoothed) field -> back-propagated points -> field -----------
def reconstruction(ns, ev, hook, size, res, reconstruction_type):
    hf, g2 = mod(ns, "helper_functions"), mod(ns, "geom_2d")
    coor, res_new = hf.generate_grid(size, res)
    coor = hook("coor", np.asarray(coor))
    circle = np.asarray([[np.cos(phi_), np.sin(phi_), 0] for phi_ in np.linspace(0, 2 * np.pi, 25 + 1)]).T
    point_cloud = g2.PointCloud2D(circle)                        # (the script wraps the array in `Points`: an identity)
    original = hook("out/original", ev.sdf(point_cloud, coor))
    if reconstruction_type == "CONV":
        point_cloud.conv_averaging((5, 5), 50, res)
    elif reconstruction_type == "SLOWSTART":
        point_cloud.slowstart(1.5, threshold=0.005, ground=False)
    elif reconstruction_type == "SMOOTH_RELU":
        point_cloud.smooth_relu(1.5, threshold=0.005)
    modified = hook("out/modified", ev.sdf(point_cloud, coor))
    distance, threshold = (1.0, 0.02) if reconstruction_type == "CONV" else (1.5, 0.05)
    # the script's reconstruction_function: host code on the field (gradient, a band of the field, one Newton-like step)
    field_ = ev.smarter_reshape(modified, res)
    mask = (field_ >= distance - threshold) * (field_ <= distance + threshold)
    grad = np.asarray(np.gradient(field_, *tuple(size[i] / res_new[i] for i in range(len(size)))))
    masked = mod(ns, "helper_functions").vector_smarter_reshape(coor, res)[:, mask]
    new_coordinates = np.zeros((3, np.count_nonzero(mask)))
    new_coordinates[:len(size), :] = masked[:len(size), :] - field_[mask] * grad[:, mask]
    new_coordinates = hook("new_coordinates", new_coordinates)
    final = hook("out/final", ev.sdf(g2.PointCloud2D(new_coordinates), coor))
    return {"original": original, "modified": modified, "final": final}


for _t in ("CONV", "SLOWSTART", "SMOOTH_RELU", "NOTHING"):
    example("surface_reconstruction_" + _t.lower(), "scalar/2D/surface_reconstruction_2D.py", (5, 5), (400, 400), (60, 60),
            overrides={"reconstruction_type": _t}, outputs={k: (k, False) for k in ("original", "modified", "final")},
            reconstruction_type=_t)(reconstruction)
