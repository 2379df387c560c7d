"""Vector-field scenes shared by the golden generator (run against the REAL reference), the oracle tests and the GPU
parity tests. A scene is `builder(ns, aux) -> (field, input_key, read_out)`: `ns` is the namespace under test
(`spomso.cores` or `aegolius_amd.cores`), `aux` the seeded input arrays of `inputs()`, `input_key` names the array handed
to the field, `read_out` is create | x | y | z | phi | theta | length."""
import numpy as np

N = 1536


def inputs():
    """Seeded inputs, all fp32-representable: positions `p` (with the origin, points on the z-axis and in the planes),
    a second cloud `co2`, per-point angles, a second field, per-point axes, spherical / cylindrical components."""
    rng = np.random.default_rng(20251003)
    f32 = lambda a: np.asarray(a, dtype=np.float32).astype(np.float64)     # noqa: E731
    p = rng.uniform(-2, 2, size=(3, N))
    p[:, 0] = 0.0                                              # the origin: every normalisation meets a zero vector
    p[:2, 1:9] = 0.0                                           # on the z-axis: the planar part vanishes
    p[2, 9:20] = 0.0
    p[0, 20:30] = 0.0
    p[1, 30:36] = -0.0
    co2 = rng.uniform(-1.5, 1.5, size=(3, N))
    co2[1:, 0:4] = 0.0                                         # atan2(0, 0) inside the revolutions
    co2[:, 4] = (-1.0, 0.0, 0.0)                               # atan2(0, -1) = pi
    comps = np.stack([rng.uniform(0.2, 2.0, N), rng.uniform(-np.pi, np.pi, N), rng.uniform(0, np.pi, N)])
    return {"p": f32(p), "co2": f32(co2), "alpha": f32(rng.uniform(-np.pi, np.pi, N)), "beta": f32(rng.uniform(-7, 7, N)),
            "second": f32(rng.normal(size=(3, N))), "axes": f32(rng.normal(size=(3, N))), "scale": f32(rng.uniform(-2, 2, N)),
            "comps": f32(comps)}


SCENES = {}


def scene(name):
    def deco(fn):
        assert name not in SCENES, name
        SCENES[name] = fn
        return fn
    return deco


def _simple(name, cls, key="p"):
    SCENES[name] = lambda ns, aux: (getattr(ns, cls)(), key, "create")


_simple("cartesian", "CartesianVectorField")
_simple("cylindrical_components", "CylindricalVectorField", "comps")
_simple("spherical_components", "SphericalVectorField", "comps")
_simple("radial_spherical", "RadialSphericalVectorField")
_simple("radial_cylindrical", "RadialCylindricalVectorField")
_simple("vortex", "VortexCylindricalVectorField")
_simple("x_field", "XVectorField")
_simple("y_field", "YVectorField")
_simple("z_field", "ZVectorField")

SCENES["angled_radial_number"] = lambda ns, aux: (ns.AngledRadialCylindricalVectorField(0.7), "p", "create")
SCENES["angled_radial_per_point"] = lambda ns, aux: (ns.AngledRadialCylindricalVectorField(aux["alpha"]), "p", "create")
SCENES["angled_vortex_number"] = lambda ns, aux: (ns.AngledVortexCylindricalVectorField(-1.1), "p", "create")
SCENES["angled_vortex_per_point"] = lambda ns, aux: (ns.AngledVortexCylindricalVectorField(aux["beta"]), "p", "create")


def _modified(name, apply, cls="RadialSphericalVectorField", key="p", read="create"):
    def build(ns, aux):
        f = getattr(ns, cls)()
        apply(f, aux)
        return f, key, read
    assert name not in SCENES, name
    SCENES[name] = build


_modified("add_number", lambda f, a: f.add(0.25))
_modified("add_vector", lambda f, a: f.add((0.5, -1.0, 2.0)))
_modified("add_vector_row", lambda f, a: f.add(np.asarray([[0.5, -1.0, 2.0]])))
_modified("add_field", lambda f, a: f.add(a["second"]))
_modified("add_per_point_number", lambda f, a: f.add(a["scale"]))
_modified("subtract_number", lambda f, a: f.subtract(1.5))
_modified("subtract_vector", lambda f, a: f.subtract([1.0, 0.0, -0.5]))
_modified("subtract_field", lambda f, a: f.subtract(a["second"]))
_modified("rescale_number", lambda f, a: f.rescale(-2.5))
_modified("rescale_per_point", lambda f, a: f.rescale(a["scale"]))
_modified("rescale_field", lambda f, a: f.rescale(a["second"]))
_modified("rescale_column", lambda f, a: f.rescale(np.asarray([[2.0], [0.5], [-1.0]])))
_modified("rotate_phi_number", lambda f, a: f.rotate_phi(0.4))
_modified("rotate_phi_per_point", lambda f, a: f.rotate_phi(a["alpha"]))
_modified("rotate_theta_number", lambda f, a: f.rotate_theta(-0.9))
_modified("rotate_theta_per_point", lambda f, a: f.rotate_theta(a["alpha"]))
_modified("rotate_x_per_point", lambda f, a: f.rotate_x(a["beta"]))
_modified("rotate_y_per_point", lambda f, a: f.rotate_y(a["alpha"]))
_modified("rotate_z_number", lambda f, a: f.rotate_z(2.2))
_modified("rotate_axis_fixed", lambda f, a: f.rotate_axis((0.0, 0.6, 0.8), 1.3))
_modified("rotate_axis_fixed_not_unit", lambda f, a: f.rotate_axis((1.0, 2.0, -0.5), a["alpha"]))
_modified("rotate_axis_per_point", lambda f, a: f.rotate_axis(a["axes"], a["beta"]))
_modified("revolution_x_same_cloud", lambda f, a: f.revolution_x(a["p"]))
_modified("revolution_y_same_cloud", lambda f, a: f.revolution_y(a["p"]))
_modified("revolution_z_same_cloud", lambda f, a: f.revolution_z(a["p"]))
_modified("revolution_x_other_cloud", lambda f, a: f.revolution_x(a["co2"]), cls="XVectorField")
_modified("revolution_y_other_cloud", lambda f, a: f.revolution_y(a["co2"]), cls="CartesianVectorField", key="second")
_modified("revolution_z_other_cloud", lambda f, a: f.revolution_z(a["co2"]), cls="YVectorField")
_modified("normalize_after_add", lambda f, a: (f.add(a["second"]), f.normalize()))
_modified("normalize_zero_vectors", lambda f, a: (f.rescale((a["scale"] > 0).astype(float)), f.normalize()),
          cls="CartesianVectorField", key="second")


def _chain(f, a):
    f.rotate_phi(a["alpha"])
    f.add((0.1, 0.2, 0.3))
    f.rotate_axis(a["axes"], 0.5)
    f.rescale(a["scale"])
    f.revolution_z(a["p"])
    f.subtract(a["second"])
    f.rotate_theta(a["beta"])
    f.normalize()


_modified("long_chain", _chain, cls="VortexCylindricalVectorField")
for _read in ("x", "y", "z", "phi", "theta", "length"):
    _modified("read_out_" + _read, lambda f, a: (f.rotate_x(a["alpha"]), f.add(a["second"]), f.normalize()),
              cls="RadialCylindricalVectorField", read=_read)
_modified("read_out_length_not_unit", lambda f, a: f.rescale(a["second"]), cls="CartesianVectorField", key="second",
          read="length")


def _helix(p, pitch, amplitude):                               # a user-defined field: runs on the host in both namespaces
    return np.asarray([-amplitude * p[1], amplitude * p[0], pitch + 0 * p[2]])


@scene("user_function_with_parameters")
def _(ns, aux):
    f = ns.geom.VectorField(_helix, 0.3, 1.5)
    f.rotate_y(aux["alpha"])
    f.normalize()
    return f, "p", "create"


@scene("field_built_on_another_field")
def _(ns, aux):
    inner = ns.AngledVortexCylindricalVectorField(aux["alpha"])
    inner.add((0.0, 0.0, 0.5))
    outer = ns.geom.VectorField(inner.propagate)
    outer.rotate_x(0.3)
    outer.normalize()
    return outer, "p", "create"


@scene("closure_reused_as_a_field_function")
def _(ns, aux):
    inner = ns.RadialCylindricalVectorField()
    vf = inner.rotate_phi(aux["beta"])                         # the returned closure is itself a field function
    outer = ns.geom.VectorField(vf)
    outer.subtract(0.25)
    return outer, "p", "theta"


# eager array-level functions: name -> callable(ns, aux) -> array
FUNCTIONS = {
    "fn_batch_normalize": lambda ns, a: ns.batch_normalize(a["second"].copy()),
    "fn_add_vectors": lambda ns, a: ns.add_vectors(a["second"], (1.0, 2.0, 3.0)),
    "fn_subtract_vectors": lambda ns, a: ns.subtract_vectors(a["second"], a["axes"]),
    "fn_rescale_vectors": lambda ns, a: ns.rescale_vectors(a["second"], a["scale"]),
    "fn_rotate_vectors_phi": lambda ns, a: ns.rotate_vectors_phi(a["second"], a["alpha"]),
    "fn_rotate_vectors_theta": lambda ns, a: ns.rotate_vectors_theta(a["second"], a["alpha"]),
    "fn_rotate_vectors_x_axis": lambda ns, a: ns.rotate_vectors_x_axis(a["second"], 0.3),
    "fn_rotate_vectors_y_axis": lambda ns, a: ns.rotate_vectors_y_axis(a["second"], a["beta"]),
    "fn_rotate_vectors_z_axis": lambda ns, a: ns.rotate_vectors_z_axis(a["second"], a["beta"]),
    "fn_rotate_vectors_axis": lambda ns, a: ns.rotate_vectors_axis(a["second"], a["axes"], a["alpha"]),
    "fn_revolve_field_x": lambda ns, a: ns.revolve_field_x(a["co2"], a["second"]),
    "fn_revolve_field_y": lambda ns, a: ns.revolve_field_y(a["co2"], a["second"]),
    "fn_revolve_field_z": lambda ns, a: ns.revolve_field_z(a["co2"], a["second"]),
    "fn_radial_vector_field_spherical": lambda ns, a: ns.radial_vector_field_spherical(a["p"]),
    "fn_vortex_vector_field_cylindrical": lambda ns, a: ns.vortex_vector_field_cylindrical(a["p"]),
    "fn_spherical_define": lambda ns, a: ns.spherical_define(a["comps"]),
}

# scenes that raise in the reference (and must raise the same type here)
RAISING = {
    "hyperbolic": (lambda ns, a: ns.HyperbolicCylindricalVectorField().create(a["p"]), TypeError),
    "winding": (lambda ns, a: ns.WindingCylindricalVectorField(2.0).create(a["p"]), TypeError),
    "add_vector_column": (lambda ns, a: _raise_add_column(ns, a), ValueError),
    "rescale_three_numbers": (lambda ns, a: _raise_rescale3(ns, a), ValueError),
}


def _raise_add_column(ns, a):
    f = ns.CartesianVectorField()
    f.add(np.asarray([[0.5], [-1.0], [2.0]]))                  # np.add(vec.T, column) cannot broadcast
    return f.create(a["p"])


def _raise_rescale3(ns, a):
    f = ns.CartesianVectorField()
    f.rescale((1.0, 2.0, 3.0))                                 # (3, N) * (3,) cannot broadcast
    return f.create(a["p"])


def run(ns, name, aux):
    field, key, read = SCENES[name](ns, aux)
    return getattr(field, read)(aux[key])


# the same array-level calls through the oracle module `vo` (oracle/vector_oracle.py)
FUNCTION_ORACLE = {
    "fn_batch_normalize": lambda vo, a: vo.unit(a["second"]),
    "fn_add_vectors": lambda vo, a: vo.modify(a["second"], "add", ((1.0, 2.0, 3.0),)),
    "fn_subtract_vectors": lambda vo, a: vo.modify(a["second"], "subtract", (a["axes"],)),
    "fn_rescale_vectors": lambda vo, a: vo.modify(a["second"], "rescale", (a["scale"],)),
    "fn_rotate_vectors_phi": lambda vo, a: vo.modify(a["second"], "rotate_phi", (a["alpha"],)),
    "fn_rotate_vectors_theta": lambda vo, a: vo.modify(a["second"], "rotate_theta", (a["alpha"],)),
    "fn_rotate_vectors_x_axis": lambda vo, a: vo.modify(a["second"], "rotate_x", (0.3,)),
    "fn_rotate_vectors_y_axis": lambda vo, a: vo.modify(a["second"], "rotate_y", (a["beta"],)),
    "fn_rotate_vectors_z_axis": lambda vo, a: vo.modify(a["second"], "rotate_z", (a["beta"],)),
    "fn_rotate_vectors_axis": lambda vo, a: vo.modify(a["second"], "rotate_axis", (a["axes"], a["alpha"])),
    "fn_revolve_field_x": lambda vo, a: vo.modify(a["second"], "revolution_x", (a["co2"],)),
    "fn_revolve_field_y": lambda vo, a: vo.modify(a["second"], "revolution_y", (a["co2"],)),
    "fn_revolve_field_z": lambda vo, a: vo.modify(a["second"], "revolution_z", (a["co2"],)),
    "fn_radial_vector_field_spherical": lambda vo, a: vo.leaf("radial_vector_field_spherical", a["p"], ()),
    "fn_vortex_vector_field_cylindrical": lambda vo, a: vo.leaf("vortex_vector_field_cylindrical", a["p"], ()),
    "fn_spherical_define": lambda vo, a: vo.leaf("spherical_define", a["comps"], ()),
}
