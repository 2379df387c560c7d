"""GPU parity of the vector-field path: every scene of tests/vector_scenes.py through the product API
(`aegolius_amd.cores`, libsdfk.so's sdfk_vec_eval_host) against the golden vectors of the real reference.

Tolerance: |gpu - ref| <= 1e-6 * max(1, |ref|) per component; zero vectors of the reference are exactly zero here.
Where a scene is ill-conditioned at a point (arccos at the poles, a difference of nearly equal vectors that is then
normalised) the bound is 8x what the float64 oracle itself moves under a one-ulp (fp32) change of its inputs."""
import ctypes
import json
import os

import numpy as np
import pytest

import vector_scenes as vs
import aegolius_amd.cores as ns
from aegolius_amd import _vector
from oracle import vector_oracle as vo

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
TOL = 1e-6


@pytest.fixture(scope="module")
def engine(built):
    built.require_gpu()
    return built


@pytest.fixture(scope="module")
def vgolden():
    data = np.load(os.path.join(HERE, "golden", "vector_golden.npz"))
    with open(os.path.join(HERE, "golden", "vector_golden_meta.json")) as f:
        return data, json.load(f)


@pytest.fixture(scope="module")
def aux():
    return vs.inputs()


def close(got, ref, slack=None):
    assert got.dtype == np.float32 and got.shape == ref.shape
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    bound = TOL * np.maximum(1.0, np.abs(ref))
    if slack is not None:
        bound = np.maximum(bound, slack)
    err = np.abs(got.astype(np.float64) - ref)
    bad = ~(err <= bound) & ~np.isnan(ref)
    assert not bad.any(), "%d values off, worst %.3g" % (bad.sum(), np.nanmax(err))
    assert not got[ref == 0].any() or np.abs(got[ref == 0]).max() <= TOL


def input_sensitivity(name, aux, trials=4):
    """How far the float64 oracle moves when every input (positions, angles, second fields) changes by one fp32 ulp:
    the conditioning of the scene (differences of nearly equal vectors before a normalisation, arccos at the poles)."""
    rng = np.random.default_rng(99)

    def run(arrays):
        field, key, read = vs.SCENES[name](ns, arrays)
        with np.errstate(all="ignore"):
            return vo.evaluate(field.vf, arrays[key], field._vf_parameters, "vector" if read == "create" else read)
    base = run(aux)
    worst = np.zeros_like(base)
    for _ in range(trials):
        moved = {k: v * (1.0 + 6e-8 * rng.choice([-1.0, 1.0], size=v.shape)) for k, v in aux.items()}
        worst = np.fmax(worst, np.abs(run(moved) - base))
    return worst


@pytest.mark.parametrize("name", sorted(vs.SCENES))
def test_scene_matches_reference_golden(name, engine, vgolden, aux):
    ref = vgolden[0]["scene/" + name]
    mine = {k: v.copy() for k, v in aux.items()}
    got = vs.run(ns, name, mine)
    for k in aux:                                              # inputs are never modified
        np.testing.assert_array_equal(mine[k], aux[k])
    slack = 8.0 * input_sensitivity(name, aux)                 # never more than the reference's own conditioning
    if name.endswith("theta"):                                 # arccos(v_z): an fp32 v_z (a few 1e-7) costs 1 / sin(theta)
        field, key, _ = vs.SCENES[name](ns, aux)
        vz = vo.evaluate(field.vf, aux[key], field._vf_parameters, "z")
        slack = np.maximum(slack, 3e-7 / np.sqrt(np.maximum(1.0 - vz * vz, 1e-12)))
    assert np.nanmedian(slack) <= 4 * TOL, "a scene that is ill-conditioned everywhere pins nothing"
    close(got, ref, slack)
    if ref.ndim == 2 and "normalize" in name:
        lengths = np.linalg.norm(got.astype(np.float64), axis=0)
        assert np.all((np.abs(lengths - 1) < 1e-6) | (lengths == 0))


@pytest.mark.parametrize("name", sorted(vs.FUNCTIONS))
def test_array_level_function_matches_reference_golden(name, engine, vgolden, aux):
    got = vs.FUNCTIONS[name](ns, {k: v.copy() for k, v in aux.items()})
    close(np.asarray(got, dtype=np.float32), vgolden[0]["function/" + name])


def test_batch_normalize_is_in_place_like_the_reference(engine, aux):
    vec = aux["second"].copy()
    out = ns.batch_normalize(vec)
    assert out is vec and np.abs(np.linalg.norm(vec, axis=0) - 1).max() < 1e-6


def test_grid_input_is_expanded_on_the_device(engine):
    """A generate_grid cloud handed to a vector field is never uploaded: same result as the plain array."""
    co, _ = ns.generate_grid((2, 2, 2), (20, 16, 12))
    alpha = np.linspace(-3, 3, co.shape[1])
    for build in (lambda: ns.RadialSphericalVectorField(), lambda: ns.AngledVortexCylindricalVectorField(alpha)):
        a, b = build(), build()
        for f, p in ((a, co), (b, np.array(co))):
            f.rotate_y(alpha)
            f.revolution_x(p)
            f.normalize()
        np.testing.assert_array_equal(a.create(co), b.create(np.array(co)))
        ref = vo.evaluate(a.vf, np.asarray(co, dtype=np.float32).astype(np.float64), a._vf_parameters)
        # the grid tables are fp32: compare against the oracle on the fp32-rounded coordinates
        close(a.create(co), ref, slack=4e-6)


def test_vector_field_from_sdf_feeds_a_chain(engine):
    co, res = ns.generate_grid((2, 2, 2), (24, 20, 16))
    sdf = ns.Sphere(0.6).create(co)
    f = ns.VectorFieldFromSDF(res)
    np.testing.assert_array_equal(f.create(sdf), ns.from_sdf(sdf, res))
    f.rotate_z(0.5)
    f.add((0.0, 0.0, 0.25))
    f.normalize()
    ref = vo.evaluate(f.vf, sdf.astype(np.float64), f._vf_parameters)
    close(f.create(sdf), ref, slack=2e-6)


def test_ragged_sizes_and_the_device_entry_point(engine):
    lib, vp = engine.lib(), ctypes.c_void_p
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 5, 1023, 1024, 1025, 4099):
        p = rng.normal(size=(3, n)).astype(np.float32)
        ang = rng.uniform(-3, 3, n).astype(np.float32)
        f = ns.RadialCylindricalVectorField()
        f.rotate_x(ang)
        f.normalize()
        got = f.create(p)
        close(got, vo.evaluate(f.vf, p.astype(np.float64), ()))
        # the same program through sdfk_vec_eval_device on caller-owned device memory
        instr, rows = _vector.lower_only(f.vf, p, ())
        prog = (_vector.VecInstr * len(instr))()
        for k, (op, ka, kb, src, imm) in enumerate(instr):
            prog[k].op = op | ka << 8 | kb << 12
            prog[k].src[0], prog[k].src[1] = src
            for j in range(4):
                prog[k].imm[j] = imm[j]
        stride = (n + 63) // 64 * 64
        d = lib.sdfk_malloc(stride * 4 * 7)
        for r in range(3):
            engine.check(lib.sdfk_memcpy_h2d(vp(d + 4 * r * stride), engine._ptr(p[r]), n * 4), "h2d")
        if rows:                                               # n = 1: a one-value angle is an immediate
            engine.check(lib.sdfk_memcpy_h2d(vp(d + 4 * 3 * stride), engine._ptr(rows[0]), n * 4), "h2d")
        engine.check(lib.sdfk_vec_eval_device(prog, len(instr), vp(d), n, stride, vp(d + 4 * 3 * stride), len(rows), stride, 0,
                                              vp(d + 4 * 4 * stride), stride, None), "sdfk_vec_eval_device")
        engine.check(lib.sdfk_sync(None), "sync")
        dev = np.empty((3, n), np.float32)
        for r in range(3):
            engine.check(lib.sdfk_memcpy_d2h(engine._ptr(dev[r]), vp(d + 4 * (4 + r) * stride), n * 4), "d2h")
        lib.sdfk_free(vp(d))
        np.testing.assert_array_equal(dev, got)
    assert ns.CartesianVectorField().create(np.zeros((3, 0))).shape == (3, 0)


def test_malformed_programs_are_rejected(engine):
    lib = engine.lib()
    prog = (_vector.VecInstr * 2)()
    out = np.zeros((3, 4), np.float32)
    p = np.zeros((3, 4), np.float32)

    def run(n_instr=2, n_streams=0):
        return lib.sdfk_vec_eval_host(prog, n_instr, engine._ptr(p), 0, 4, None, 0, None, 0, None, 0, None, n_streams, 0,
                                      engine._ptr(out), 0)
    prog[0].op, prog[1].op = _vector.OP["ADD"] | _vector.K_IMM1 << 8, _vector.OP["NORMALIZE"]
    assert run() != 0 and "initialiser" in engine.last_error()
    prog[0].op, prog[1].op = _vector.OP["INIT_P"], 99
    assert run() != 0
    prog[1].op = _vector.OP["ROT_Z"] | _vector.K_ROW1 << 8
    prog[1].src[0] = 0
    assert run() != 0 and "stream row" in engine.last_error()
    prog[1].op = _vector.OP["NORMALIZE"] | _vector.K_IMM1 << 8
    assert run() != 0 and "operand kind" in engine.last_error()
    prog[1].op = _vector.OP["NORMALIZE"]
    assert run() == 0


@pytest.mark.parametrize("name", sorted(vs.SCENES))
def test_specialised_and_interpreter_kernels_give_the_same_bits(name, engine, aux):
    """The hiprtc kernel of a chain's topology and the interpreter kernel call the same device functions."""
    lib = engine.lib()
    try:
        lib.sdfk_vec_set_interpret(1)
        interpreted = vs.run(ns, name, {k: v.copy() for k, v in aux.items()})
    finally:
        lib.sdfk_vec_set_interpret(0)
    specialised = vs.run(ns, name, {k: v.copy() for k, v in aux.items()})
    np.testing.assert_array_equal(specialised, interpreted)


def test_one_compiled_kernel_serves_every_chain_of_the_same_shape(engine, aux):
    import time
    def chain(angle, vec):
        f = ns.VortexCylindricalVectorField()
        f.rotate_y(angle)
        f.add(vec)
        f.rotate_x(aux["alpha"])
        f.normalize()
        return f
    first = chain(0.123, (1.0, 0.0, 0.5))
    first.create(aux["p"])                                      # builds (or finds) the kernel of this topology
    t0 = time.perf_counter()
    other = chain(-2.5, (0.0, 3.0, 0.25))
    got = other.create(aux["p"])
    assert time.perf_counter() - t0 < 0.2                       # new numbers, same shape: no compilation
    close(got, vo.evaluate(other.vf, aux["p"], ()), slack=8.0 * 1e-7)


def test_vector_chain_at_scale(engine):
    """513^3 (BASELINE cfg 2's single-box size) through the product API on a generate_grid cloud (expanded on the
    device): size-independent properties plus 100,000 sampled points against the oracle."""
    co, _ = ns.generate_grid((2, 2, 2), (512, 512, 512))
    n = co.shape[1]
    f = ns.RadialSphericalVectorField()
    f.rotate_phi(0.3)
    f.revolution_z(co)
    f.add((0.0, 0.0, 0.5))
    f.normalize()
    got = f.create(co)
    assert got.shape == (3, n) and got.dtype == np.float32
    pick = np.random.default_rng(11).integers(0, n, 100000)
    sample = np.asarray(co)[:, pick].astype(np.float32).astype(np.float64)     # the grid tables are fp32
    g = ns.RadialSphericalVectorField()
    g.rotate_phi(0.3)
    g.revolution_z(sample)
    g.add((0.0, 0.0, 0.5))
    g.normalize()
    want = vo.evaluate(g.vf, sample, ())
    close(got[:, pick], want, slack=4e-6)
    lengths = np.linalg.norm(got[:, pick].astype(np.float64), axis=0)
    assert np.abs(lengths - 1).max() < 1e-6
    # the read-outs agree with the vector they are taken from
    np.testing.assert_array_equal(f.z(co), got[2])
    assert np.abs(f.length(co)[pick] - 1).max() < 1e-6


def test_resident_pipeline_from_sdf_to_vector_field(engine):
    """SDF fields and vector fields that stay in HBM feed a chain as angle, second field, coordinates and input: the
    result equals, bit for bit, the same chain fed with the downloaded arrays."""
    from aegolius_amd import DeviceField, DeviceVectorField
    co, res = ns.generate_grid((2, 2, 2), (40, 36, 32))
    n = co.shape[1]
    ball = ns.Sphere(0.6)
    ball.move((0.1, 0.0, -0.2))
    slab = ns.Box(1.2, 0.8, 0.4)
    slab.rotate(0.4, (1, 1, 0))
    sdf_dev, ang_dev = ball.create_resident(co), slab.create_resident(co)
    sdf, ang = sdf_dev.numpy(), ang_dev.numpy()
    second = np.random.default_rng(2).normal(size=(3, n)).astype(np.float32)
    second_dev = DeviceVectorField.from_host(second)
    np.testing.assert_array_equal(second_dev.numpy(), second)

    def chain(angle, field2):
        f = ns.AngledRadialCylindricalVectorField(angle)
        f.rotate_theta(angle)
        f.add(field2)
        f.revolution_z(co)
        f.rescale(angle)
        f.normalize()
        return f
    on_device = chain(ang_dev, second_dev).create_resident(co)
    assert isinstance(on_device, DeviceVectorField) and on_device.shape == (3, n)
    from_host = chain(ang, second).create(co)
    np.testing.assert_array_equal(on_device.numpy(), from_host)
    np.testing.assert_array_equal(chain(ang_dev, second_dev).create(co), from_host)       # device operands, host result
    close(from_host, vo.evaluate(chain(ang.astype(np.float64), second.astype(np.float64)).vf,
                                 np.asarray(co, dtype=np.float32).astype(np.float64), (ang.astype(np.float64),)), slack=4e-6)
    # a resident vector field as the INPUT of the next chain, and a read-out of it
    nxt = ns.CartesianVectorField()
    nxt.rotate_axis(second_dev, 0.7)
    nxt.subtract(on_device)
    expect = ns.CartesianVectorField()
    expect.rotate_axis(second, 0.7)
    expect.subtract(from_host)
    np.testing.assert_array_equal(nxt.create(on_device), expect.create(from_host))
    np.testing.assert_array_equal(nxt.length(on_device), expect.length(from_host))
    # the gradient direction of a resident SDF enters a chain without leaving the device
    g = ns.VectorFieldFromSDF(res)
    g.rotate_z(ang_dev)
    g.normalize()
    h = ns.VectorFieldFromSDF(res)
    h.rotate_z(ang)
    h.normalize()
    np.testing.assert_array_equal(g.create_resident(sdf_dev).numpy(), h.create(sdf))
    bad = ns.CartesianVectorField()
    bad.add(DeviceField.from_host(np.zeros(5, np.float32)))     # a field of another size
    with pytest.raises(ValueError):
        bad.create(co)


def test_sharded_vector_field_equals_the_whole_cloud(engine):
    """§8(e) for vector fields: slabs of whole rows emulated on one device, per-point operands sliced per rank."""
    from aegolius_amd.distributed import vector_field_sharded
    size, resolution = (2, 2, 2), (20, 18, 26)
    co, _ = ns.generate_grid(size, resolution)
    n = co.shape[1]
    rng = np.random.default_rng(8)
    ang, second = rng.uniform(-3, 3, n), rng.normal(size=(3, n))
    f = ns.AngledVortexCylindricalVectorField(ang)
    f.rotate_x(ang)
    f.add(second)
    f.revolution_y(co)
    f.normalize()
    whole, whole_phi = f.create(co), f.phi(co)
    for world in (1, 2, 3, 7):
        parts = [vector_field_sharded(f, size, resolution, world_rank=(world, r))[0] for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(parts, axis=1), whole)
        phis = [vector_field_sharded(f, size, resolution, out="phi", world_rank=(world, r))[0] for r in range(world)]
        np.testing.assert_array_equal(np.concatenate(phis), whole_phi)
    dev, _ = vector_field_sharded(f, size, resolution, world_rank=(2, 1), resident=True)
    np.testing.assert_array_equal(dev.numpy(), whole[:, n - dev.n:])


def test_the_example_of_the_integration_notes(engine):
    from aegolius_amd.cores import AngledRadialCylindricalVectorField, VectorFieldFromSDF, Sphere, generate_grid
    co, res = generate_grid((2, 2, 2), (48, 48, 48))
    angle = Sphere(0.6).create_resident(co)
    field = AngledRadialCylindricalVectorField(angle)
    field.revolution_z(co)
    field.normalize()
    directions = field.create(co)
    assert directions.shape == (3, co.shape[1]) and directions.dtype == np.float32
    host = AngledRadialCylindricalVectorField(angle.numpy())
    host.revolution_z(co)
    host.normalize()
    np.testing.assert_array_equal(directions, host.create(co))
    np.testing.assert_array_equal(field.create_resident(co).numpy(), directions)
    normals = VectorFieldFromSDF(res).create(Sphere(0.6).create(co))
    lengths = np.linalg.norm(normals.astype(np.float64), axis=0)
    assert normals.shape == (3, co.shape[1]) and np.all((np.abs(lengths - 1) < 1e-6) | (lengths == 0))
