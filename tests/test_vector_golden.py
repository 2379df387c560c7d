"""CPU: the vector-field oracle (oracle/vector_oracle.py) and the host-side API mirror, pinned against golden vectors
produced by the REAL reference (tests/golden/generate_vector_golden.py, run in the build container)."""
import json
import os

import numpy as np
import pytest

import vector_scenes as vs
import aegolius_amd.cores as ns
from aegolius_amd import _vector
from oracle import vector_oracle as vo

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def vgolden():
    data = np.load(os.path.join(HERE, "golden", "vector_golden.npz"))
    with open(os.path.join(HERE, "golden", "vector_golden_meta.json")) as f:
        return data, json.load(f)


@pytest.fixture(scope="module")
def aux(vgolden):
    aux = vs.inputs()
    for k, v in aux.items():                                   # reproducible from the seed, fp32-representable
        np.testing.assert_array_equal(v.astype(np.float32), vgolden[0]["input/" + k])
        np.testing.assert_array_equal(v, v.astype(np.float32).astype(np.float64))
    return aux


def oracle_run(name, aux):
    field, key, read = vs.SCENES[name](ns, aux)
    return vo.evaluate(field.vf, aux[key], field._vf_parameters, "vector" if read == "create" else read)


def test_fixture_covers_every_scene(vgolden):
    data, meta = vgolden
    assert set(meta["scenes"]) == set(vs.SCENES) and set(meta["functions"]) == set(vs.FUNCTIONS)
    assert set(meta["raising"]) == set(vs.RAISING)
    assert set(vs.FUNCTION_ORACLE) == set(vs.FUNCTIONS)


@pytest.mark.parametrize("name", sorted(vs.SCENES))
def test_oracle_matches_reference(name, vgolden, aux):
    ref = vgolden[0]["scene/" + name]
    with np.errstate(all="ignore"):
        out = oracle_run(name, {k: v.copy() for k, v in aux.items()})
    assert out.shape == ref.shape
    assert np.array_equal(np.isnan(out), np.isnan(ref))
    err = np.abs(out - ref) / np.maximum(1.0, np.abs(ref))
    assert np.nanmax(err, initial=0.0) <= 1e-12


@pytest.mark.parametrize("name", sorted(vs.FUNCTIONS))
def test_oracle_functions_match_reference(name, vgolden, aux):
    ref = vgolden[0]["function/" + name]
    out = vs.FUNCTION_ORACLE[name](vo, {k: v.copy() for k, v in aux.items()})
    assert out.shape == ref.shape and np.abs(out - ref).max() <= 1e-12


def test_zero_vectors_and_axis_points_are_in_the_inputs(vgolden, aux):
    data, _ = vgolden
    radial = data["scene/radial_spherical"]
    assert not radial[:, 0].any()                              # the origin stays a zero vector
    planar = data["scene/radial_cylindrical"]
    assert not planar[:, 1:9].any()                            # points on the z-axis
    assert (np.linalg.norm(data["scene/normalize_zero_vectors"], axis=0) == 0).sum() > 100


@pytest.mark.parametrize("name", sorted(vs.RAISING))
def test_errors_match_the_reference(name, vgolden, aux):
    _, meta = vgolden
    fn, exc = vs.RAISING[name]
    assert meta["raising"][name] == exc.__name__
    with pytest.raises(exc):
        fn(ns, aux)                                            # raised while lowering: before any GPU call


def test_api_mirror_of_the_vector_classes():
    f = ns.AngledRadialCylindricalVectorField(0.3)
    assert f.alpha == 0.3 and ns.WindingCylindricalVectorField(2).gamma == 2
    assert f.modifications == [] and f.original_object is ns.aar_vector_field_cylindrical
    closure = f.rotate_x(0.1)
    f.add((1, 2, 3))
    f.normalize()
    assert f.modifications == ["rotate_x", "add", "normalize"]
    assert callable(closure) and f.modified_object is f.vf and len(f.vf.mods) == 3 and len(closure.mods) == 1
    for method in ("add", "subtract", "rescale", "rotate_phi", "rotate_theta", "rotate_x", "rotate_y", "rotate_z",
                   "rotate_axis", "revolution_x", "revolution_y", "revolution_z", "normalize", "create", "propagate", "x",
                   "y", "z", "phi", "theta", "length"):
        assert callable(getattr(f, method))
    for name in ("CartesianVectorField", "CylindricalVectorField", "SphericalVectorField", "RadialSphericalVectorField",
                 "RadialCylindricalVectorField", "HyperbolicCylindricalVectorField", "AngledRadialCylindricalVectorField",
                 "WindingCylindricalVectorField", "VortexCylindricalVectorField", "AngledVortexCylindricalVectorField",
                 "XVectorField", "YVectorField", "ZVectorField", "VectorFieldFromSDF", "batch_normalize", "add_vectors",
                 "subtract_vectors", "rescale_vectors", "rotate_vectors_phi", "rotate_vectors_theta",
                 "rotate_vectors_x_axis", "rotate_vectors_y_axis", "rotate_vectors_z_axis", "rotate_vectors_axis",
                 "revolve_field_x", "revolve_field_y", "revolve_field_z", "from_sdf", "ModifyVectorObject"):
        assert hasattr(ns, name), name


def test_lowering_of_a_chain(aux):
    f = ns.VortexCylindricalVectorField()
    f.rotate_phi(aux["alpha"])
    f.add((0.1, 0.2, 0.3))
    f.rotate_axis(aux["axes"], 0.5)
    f.revolution_z(aux["p"])
    f.rescale(aux["alpha"])                                    # the same array again: uploaded once
    instr, rows = _vector.lower_only(f.vf, aux["p"], ())
    ops = [i[0] for i in instr]
    assert ops == [_vector.OP[k] for k in ("INIT_VORTEX", "ROT_Z", "ADD", "ROT_AXIS", "REVOLVE_Z", "MUL")]
    assert [i[1] for i in instr] == [_vector.K_NONE, _vector.K_ROW1, _vector.K_IMM3, _vector.K_ROW3, _vector.K_P,
                                     _vector.K_ROW1]
    assert len(rows) == 4 and instr[1][3][0] == instr[5][3][0] == 0 and instr[3][3][0] == 1
    assert instr[3][2] == _vector.K_IMM1 and abs(instr[3][4][3] - 0.5) < 1e-7
    with pytest.raises(ValueError):
        g = ns.CartesianVectorField()
        g.rotate_x(np.zeros(7))
        _vector.lower_only(g.vf, aux["p"], ())


@pytest.mark.parametrize("name", ["long_chain", "read_out_theta", "read_out_length", "rotate_axis_per_point", "cartesian",
                                  "revolution_y_other_cloud", "angled_vortex_per_point", "spherical_components"])
def test_chain_specialised_source_compiles_for_gfx950(name, built, aux):
    """hiprtc cross-compiles the straight-line kernel of a chain without a GPU (the build check of the vector path)."""
    import ctypes
    field, key, read = vs.SCENES[name](ns, aux)
    instr, rows = _vector.lower_only(field.vf, aux[key], field._vf_parameters)
    prog = _vector.program_array(instr)
    kind = _vector.OUT_KINDS["vector" if read == "create" else read]
    src = built.lib().sdfk_vec_source(prog, len(instr), len(rows), kind).decode()
    assert src.count("sdfk_vec_apply(") >= len(instr) and "sdfk_vspec" in src
    size = ctypes.c_size_t(0)
    built.check(built.lib().sdfk_vec_compile_check(prog, len(instr), len(rows), kind, ctypes.byref(size)), "compile")
    assert size.value > 1000


def test_operand_classification_follows_numpy_broadcasting(aux):
    """Which operands become immediates, rows or errors — decided at lowering, before any GPU call."""
    n = aux["p"].shape[1]
    K = _vector

    def kinds(apply):
        f = ns.CartesianVectorField()
        apply(f)
        instr, rows = K.lower_only(f.vf, aux["p"], ())
        return instr[-1][1], instr[-1][2], len(rows)
    assert kinds(lambda f: f.add(2)) == (K.K_IMM1, K.K_NONE, 0)
    assert kinds(lambda f: f.add(np.float32(2))) == (K.K_IMM1, K.K_NONE, 0)
    assert kinds(lambda f: f.add([1, 2, 3])) == (K.K_IMM3, K.K_NONE, 0)
    assert kinds(lambda f: f.add(np.ones((1, 3)))) == (K.K_IMM3, K.K_NONE, 0)
    assert kinds(lambda f: f.add(np.ones(n))) == (K.K_ROW1, K.K_NONE, 1)
    assert kinds(lambda f: f.add(np.ones((1, n)))) == (K.K_ROW1, K.K_NONE, 1)
    assert kinds(lambda f: f.subtract(np.ones((3, n)))) == (K.K_ROW3, K.K_NONE, 3)
    assert kinds(lambda f: f.rescale(np.ones((3, 1)))) == (K.K_IMM3, K.K_NONE, 0)
    assert kinds(lambda f: f.rotate_axis(np.ones((3, n)), np.ones(n))) == (K.K_ROW3, K.K_ROW1, 4)
    assert kinds(lambda f: f.rotate_axis((0, 0, 1), 0.5)) == (K.K_IMM3, K.K_IMM1, 0)
    assert kinds(lambda f: f.revolution_x(aux["p"])) == (K.K_P, K.K_NONE, 0)
    assert kinds(lambda f: f.revolution_x(aux["co2"])) == (K.K_ROW3, K.K_NONE, 3)
    for bad in (lambda f: f.add(np.ones((3, 1))), lambda f: f.add(np.ones(n + 1)), lambda f: f.rescale((1, 2, 3)),
                lambda f: f.rescale(np.ones((2, n))), lambda f: f.rotate_x(np.ones((3, n))),
                lambda f: f.rotate_axis(np.ones((2, n)), 0.1), lambda f: f.revolution_z(np.ones((3, n - 1)))):
        with pytest.raises(ValueError):
            kinds(bad)
    # NumPy raises for the same shapes
    vec = np.zeros((3, n))
    for operand in (np.ones(n + 1), np.ones((2, n))):
        with pytest.raises(ValueError):
            np.add(vec, operand)
    with pytest.raises(ValueError):
        np.add(vec.T, np.ones((3, 1)))
    with pytest.raises(ValueError):
        np.multiply(vec, (1, 2, 3))
