"""GPU parity of the field consumers: interior extraction (`GenericGeometry.point_cloud`,
sdfk_field_select) and gradient direction (`from_sdf`, sdfk_field_gradient) on a field that stays in HBM.

Selection is integer work: bit-exact against numpy.flatnonzero on the same fp32 field. The gradient direction is
floating point: |gpu - ref| <= 1e-6 per component (unit vectors), zero vectors exactly zero, the reference being
given the same fp32 field; the raw (un-normalised) gradient equals numpy's rounded to fp32, bit for bit."""
import os
import sys

import numpy as np
import pytest

import scenes
import aegolius_amd
import aegolius_amd.cores as ns
from aegolius_amd import DeviceField
from oracle import sdf_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-6


@pytest.fixture(scope="module")
def engine(built):
    built.require_gpu()
    return built


@pytest.mark.parametrize("n", [1, 3, 4, 63, 1024, 8191, 8192, 8193, 70001, 1 << 20, (1 << 22) + 5])
def test_select_equals_flatnonzero(n, engine):
    rng = np.random.default_rng(n)
    f = rng.normal(size=n).astype(np.float32)
    f[rng.random(n) < 0.01] = np.nan                       # NaN is never selected (NumPy: nan <= t is False)
    f[rng.random(n) < 0.01] = 0.0
    f[rng.random(n) < 0.01] = -0.0
    f[rng.random(n) < 0.005] = np.inf
    f[rng.random(n) < 0.005] = -np.inf
    f[rng.random(n) < 0.005] = 1e-42                       # subnormal
    dev = DeviceField.from_host(f)
    for thr in (0.0, -0.0, -0.5, 1.5, -10.0, 10.0, np.inf, -np.inf, 1e-45, np.nan):
        with np.errstate(invalid="ignore"):
            want = np.flatnonzero(f <= np.float32(thr))
        got = dev.select(thr)
        assert got.dtype == np.int64
        np.testing.assert_array_equal(got, want)
        assert dev.count(thr) == want.size
    np.testing.assert_array_equal(dev.numpy(), f)
    dev.free()
    with pytest.raises(aegolius_amd._engine.SdfkError):
        dev.select(0.0)


def test_select_patterns_that_stress_the_scan(engine):
    n = 3 * 8192 * 64 + 17
    for f in (np.full(n, -1.0, np.float32), np.full(n, 1.0, np.float32),
              np.where(np.arange(n) % 8192 == 8191, -1.0, 1.0).astype(np.float32),
              np.where((np.arange(n) // 8192) % 2 == 0, -1.0, 1.0).astype(np.float32)):
        dev = DeviceField.from_host(f)
        np.testing.assert_array_equal(dev.select(0.0), np.flatnonzero(f <= 0))
        dev.free()
    assert DeviceField.from_host(np.zeros(0, np.float32)).select().size == 0


@pytest.mark.parametrize("name", sorted(scenes.CONSUMER_SCENES))
def test_point_cloud_matches_reference_golden(name, engine, golden):
    data, _ = golden
    build, key = scenes.CONSUMER_SCENES[name]
    co, res = scenes.grid_inputs(ns, key)
    ref_field = data["consumer/%s/field" % name]
    ref_cloud = data["consumer/%s/cloud" % name]
    inside_ref = ref_field <= 0
    for points in (co, ns.generate_grid(*scenes.GRIDS[key])[0]):     # plain array and tagged grid (no upload)
        obj = build(ns, res)
        cloud = obj.point_cloud(points)
        assert cloud.dtype == np.float64 and cloud.shape[0] == 3 and not cloud[2].any()
        inside = obj.create(points) <= 0
        np.testing.assert_array_equal(cloud[:2], np.asarray(points)[:2, inside])      # the caller's own coordinates
        # the two masks may differ only where the reference's field is within tolerance of the threshold
        assert np.all(np.abs(ref_field[inside != inside_ref]) <= TOL)
        if name == "consume_flat_plateaus":                          # its field is exactly 0 / 1: nothing is close
            np.testing.assert_array_equal(inside, inside_ref)
    np.testing.assert_array_equal(build(ns, res).point_cloud(co), ref_cloud if np.array_equal(inside, inside_ref) else
                                  sdf_oracle.point_cloud(np.where(inside, -1.0, 1.0), co))
    # and the interior is exactly the mask of OUR field
    obj = build(ns, res)
    field = obj.create(co)
    np.testing.assert_array_equal(obj.point_cloud(co), sdf_oracle.point_cloud(field, co))


def numpy_direction(f, shape):
    """from_sdf's arithmetic on an arbitrary shape (resolution_conversion only produces odd extents)."""
    vec = np.asarray(np.gradient(f.astype(np.float64).reshape(shape))).reshape(len(shape), -1)
    m = np.linalg.norm(vec, axis=0)
    keep = ~(m == 0)
    raw = vec.copy()
    vec[:, keep] = vec[:, keep] / m[keep]
    return raw, vec


def direction_check(got, want):
    assert got.dtype == np.float32 and got.shape == want.shape
    zero = np.linalg.norm(want, axis=0) == 0
    assert not got[:, zero].any()
    assert np.abs(got.astype(np.float64) - want).max() <= TOL


@pytest.mark.parametrize("name", sorted(scenes.CONSUMER_SCENES))
def test_from_sdf_matches_reference_golden(name, engine, golden):
    data, _ = golden
    build, key = scenes.CONSUMER_SCENES[name]
    _, res = scenes.grid_inputs(ns, key)
    field32 = data["consumer/%s/field32" % name]
    want = data["consumer/%s/direction" % name]
    direction_check(ns.from_sdf(field32, res), want)
    direction_check(ns.from_sdf(field32.astype(np.float64), res), want)


@pytest.mark.parametrize("shape", [(2, 2, 2), (3, 70, 129), (33, 17, 64), (35, 5, 201), (2, 300), (129, 65), (7, 2), (77,),
                                   (2,), (5, 17, 1025), (4, 3, 5000), (3, 9000, 3), (40000, 3), (3, 40000), (65, 37, 130), (34, 64, 257),
                                   (97, 11, 1023), (66, 200, 7), (40, 3, 160), (9, 40, 161), (33, 20, 513), (70, 9, 322)])
def test_gradient_direction_on_random_fields(shape, engine):
    rng = np.random.default_rng(sum(shape))
    n = int(np.prod(shape))
    f = rng.normal(size=n).astype(np.float32)
    f[rng.random(n) < 0.3] = 0.25                              # plateaus: zero differences and zero vectors
    raw, want = numpy_direction(f, shape)
    dev = DeviceField.from_host(f)
    direction_check(dev.gradient(shape), want)
    # numpy.gradient itself: the fp32 difference is the float64 difference rounded once
    np.testing.assert_array_equal(dev.gradient(shape, normalize=False), raw.astype(np.float32))
    dev.free()
    if all(s % 2 for s in shape):                              # reachable through the reference's signature
        res = tuple(s - 1 for s in shape)
        np.testing.assert_array_equal(want, sdf_oracle.from_sdf(f.astype(np.float64), res))
        direction_check(ns.from_sdf(f, res), want)


@pytest.mark.parametrize("scale", [1e-42, 1e-30, 1e-12, 1e12, 1e30, 1.5e38])
def test_gradient_direction_does_not_overflow_or_vanish(scale, engine):
    """fp32 arithmetic with power-of-two scaling: squares of 1e30 or 1e-30 differences never reach the sum."""
    shape = (9, 20, 67)
    rng = np.random.default_rng(5)
    f = (rng.uniform(-1, 1, size=int(np.prod(shape))) * scale).astype(np.float32)
    f[rng.random(f.size) < 0.2] = np.float32(scale) * np.float32(0.5)
    with np.errstate(all="ignore"):
        _, want = numpy_direction(f, shape)
    dev = DeviceField.from_host(f)
    direction_check(dev.gradient(shape), want)
    dev.free()


def test_gradient_direction_with_nan_and_inf(engine):
    shape = (6, 8, 40)
    rng = np.random.default_rng(6)
    f = rng.normal(size=int(np.prod(shape))).astype(np.float32)
    f[[17, 400, 1333]] = np.nan
    f[[55, 900]] = np.inf
    with np.errstate(all="ignore"):
        _, want = numpy_direction(f, shape)
    dev = DeviceField.from_host(f)
    got = dev.gradient(shape)
    dev.free()
    np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.abs(got[ok].astype(np.float64) - want[ok]).max() <= TOL


def test_gradient_rejects_what_numpy_rejects(engine):
    dev = DeviceField.from_host(np.zeros(12, np.float32))
    with pytest.raises(ValueError):
        dev.gradient((12, 1))
    with pytest.raises(ValueError):
        dev.gradient((5, 2))
    with pytest.raises(ValueError):
        np.gradient(np.zeros((12, 1)))


def test_resident_field_feeds_both_consumers_without_a_round_trip(engine):
    co, res = ns.generate_grid((2, 2, 2), (64, 48, 40))
    a = ns.Sphere(0.5)
    b = ns.Box(0.6, 0.4, 0.9)
    b.rotate(0.7, (1, 1, 0))
    b.move((0.3, 0.1, 0.0))
    tree = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(a, b, parameters=0.1)
    host = tree.create(co)
    dev = tree.create_resident(co)
    assert isinstance(dev, DeviceField) and dev.n == host.size
    np.testing.assert_array_equal(dev.numpy(), host)
    np.testing.assert_array_equal(dev.select(), np.flatnonzero(host <= 0))
    np.testing.assert_array_equal(ns.from_sdf(dev, res), ns.from_sdf(host, res))
    direction_check(ns.from_sdf(dev, res), sdf_oracle.from_sdf(host.astype(np.float64), res))
    # a plain (untagged) array takes the staging path and leaves the same field on the device
    plain = tree.create_resident(np.array(co))
    np.testing.assert_array_equal(plain.numpy(), host)
    # staged trees (grid-neighbourhood operators) can stay resident too
    s = ns.Sphere(0.6)
    s.conv_averaging((3, 3, 3), 1, res)
    np.testing.assert_array_equal(s.create_resident(co).numpy(), s.create(co))


def test_consumers_at_scale(engine):
    """513^3 (BASELINE cfg 2's single-box size): properties that need no CPU evaluation of the tree."""
    co, res = ns.generate_grid((2, 2, 2), (512, 512, 512))
    s = ns.Sphere(0.5)
    s.move((0.125, -0.25, 0.0))
    dev = s.create_resident(co)
    idx = dev.select(0.0)
    field = dev.numpy()
    assert np.all(np.diff(idx) > 0)
    assert idx.size == np.count_nonzero(field <= 0) and np.all(field[idx] <= 0)
    # number of lattice points inside the sphere ~ its volume / cell volume
    cell = (2.0 / 512) ** 3
    assert abs(idx.size * cell / (4 / 3 * np.pi * 0.5 ** 3) - 1) < 1e-3
    vec = ns.from_sdf(dev, res)
    assert vec.shape == (3, field.size)
    pick = np.random.default_rng(0).integers(0, field.size, 200000)
    p = np.asarray(co)[:, pick].astype(np.float64) - np.array([[0.125], [-0.25], [0.0]])
    r = np.linalg.norm(p, axis=0)
    far = r > 0.05
    np.testing.assert_allclose(np.linalg.norm(vec[:, pick].astype(np.float64), axis=0)[far], 1.0, atol=1e-6)
    # the gradient of a distance field points radially; central differences on a 2/512 lattice are second order
    assert np.abs(vec[:, pick][:, far] - (p / r)[:, far]).max() < 2e-3
    cloud = s.point_cloud(co)
    assert cloud.shape == (3, idx.size) and np.array_equal(cloud[:2], np.asarray(co)[:2, idx])


@pytest.mark.parametrize("size,resolution", [((2, 2, 2), (40, 24, 66)), ((3, 3), (200, 130))])
def test_sharded_consumers_equal_the_single_device_result(size, resolution, engine):
    """§8(e) for the consumers: slabs (with their recomputed halo plane) emulated on one device, every rank's part
    concatenated — bit-identical to select / from_sdf on the whole field."""
    import torch
    from aegolius_amd.distributed import gradient_direction_sharded, interior_indices_sharded
    a = ns.Sphere(0.55)
    b = ns.Box(0.5, 0.9, 0.4)
    b.rotate(0.5, (0, 1, 1))
    b.move((0.2, -0.1, 0.1))
    tree = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(a, b, parameters=0.12)
    co, res = ns.generate_grid(size, resolution)
    dev = tree.create_resident(co)
    want_idx = dev.select(0.0)
    want_vec = ns.from_sdf(dev, resolution)
    for world in (1, 2, 3, 8):
        vec = [gradient_direction_sharded(tree, size, resolution, world_rank=(world, r))[0] for r in range(world)]
        assert all(v.is_cuda and v.dtype == torch.float32 for v in vec)
        np.testing.assert_array_equal(torch.cat(vec, dim=1).cpu().numpy(), want_vec)
        idx = [interior_indices_sharded(tree, size, resolution, world_rank=(world, r))[0] for r in range(world)]
        assert all(i.is_cuda and i.dtype == torch.int64 for i in idx)
        np.testing.assert_array_equal(torch.cat(idx).cpu().numpy(), want_idx)
    assert want_idx.size > 100


# ---- `signed` on a sharded grid: the slabs exchange the boundary bits ------------------------------
def _signed_worker(rank, world, port, resolution, q):
    import torch
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    sys.path.insert(0, os.path.join(root, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import aegolius_amd.cores as cores
        from aegolius_amd.distributed import evaluate_grid_sharded
        res = {}
        for name, build in _signed_trees(cores, resolution).items():
            # (slabs left distributed: gloo moves no device tensors, the parent puts them together)
            slab, _ = evaluate_grid_sharded(build(), (2.0, 2.0, 2.0), resolution, gather=False)
            res[name] = slab.cpu().numpy()
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


def _signed_trees(cores, resolution):
    def shell():
        s = cores.Sphere(0.6)
        s.boundary()
        s.signed(resolution)
        return s

    def shell_then_average():
        s = cores.Torus(0.5, 0.2)
        s.rotate(0.6, (1, 0.3, 0.2))
        s.boundary()
        s.signed(resolution)
        s.conv_averaging((3, 3, 3), 1, resolution)
        s.onion(0.05)
        return s

    def already_signed():
        s = cores.Sphere(0.6)
        s.signed(resolution)
        return s

    def old_variant():
        s = cores.Box(0.9, 0.7, 0.5)
        s.boundary()
        s.signed_old(resolution)
        return s
    return {"shell": shell, "shell_then_average": shell_then_average, "already_signed": already_signed, "old_variant": old_variant}


@pytest.mark.parametrize("world", [2, 3])
def test_signed_shards_across_ranks(world, engine):
    """Trees with `signed` / `signed_old` on a grid cut into slabs of whole planes, one rank per slab (here: ranks on
    one GPU over gloo; on a node: RCCL): the slabs put together are the field of the single-GPU evaluation, bit for bit.
    Slabs are uneven (21 planes over 2 / 3 ranks) and `signed` is followed by an averaging that needs a halo."""
    import socket
    import torch.multiprocessing as mp
    resolution = (20, 18, 16)
    with socket.socket() as sck:
        sck.bind(("127.0.0.1", 0))
        port = sck.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_signed_worker, args=(r, world, port, resolution, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    co, _ = ns.generate_grid((2.0, 2.0, 2.0), resolution)
    got = dict(got)
    for name, build in _signed_trees(ns, resolution).items():
        want = build().create(co)
        assert (want < 0).any() and (want > 0).any()
        np.testing.assert_array_equal(np.concatenate([got[r][name] for r in range(world)]), want, err_msg=name)


@pytest.mark.parametrize("name", ["tree_cfg2_smooth_union10", "tree_cfg5_three_level", "tree_cfg4_union50_2d", "tree_cfg3_mod_chain",
                                  "prim_sphere", "combine_SUBTRACT2"])
def test_fused_selection_equals_flatnonzero_of_the_field(name, engine):
    """Program.select_grid / select_host (evaluation kernels writing flag bits, compaction from the flags) return exactly
    numpy.flatnonzero(field <= threshold) of the field the same kernels write — row-block, chain-mode and plain kernels,
    tagged grids and plain arrays, odd row lengths (edge windows OR their bits in), thresholds that select nothing / all."""
    import scenes
    from aegolius_amd._lower import lower_geometry
    tree = scenes.SCENES[name](ns)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    flat = "2d" in name
    for shape in ((9, 37, 1025), (20, 33, 64), (7, 50, 40), (5, 5, 333)):
        if flat:
            co, _ = ns.generate_grid((10, 10), (shape[0] * shape[1] - 1, shape[2] - 1))
        else:
            co, _ = ns.generate_grid((2.6, 2.6, 2.6), tuple(r - 1 for r in shape))
        axes = [a.astype(np.float32) for a in co.grid_axes]
        field = prog.eval_grid_host(axes, mode=engine.MODE_SPECIALIZED)
        plain = np.asarray(co).astype(np.float32)
        for thr in (0.0, 0.13, -1e9, 1e9, float(np.median(field))):
            want = np.flatnonzero(field <= np.float32(thr))
            np.testing.assert_array_equal(prog.select_grid(axes, thr), want)
            np.testing.assert_array_equal(prog.select_host(plain, thr), want)
        # a slab of whole rows from the axis tables; scattered points (no row structure: plain kernel)
        row = shape[2] if not flat else shape[2]
        s0, cnt = 3 * row, (field.size // row - 5) * row
        np.testing.assert_array_equal(prog.select_grid(axes, 0.05, start=s0, count=cnt), np.flatnonzero(field[s0:s0 + cnt] <= np.float32(0.05)))
    rng = np.random.default_rng(3)
    pts = rng.uniform(-1.5, 1.5, (3, 10007)).astype(np.float32)
    if flat:
        pts[2] = 0
    got = prog.select_host(pts, 0.1)
    np.testing.assert_array_equal(got, np.flatnonzero(prog.eval_host(pts, mode=engine.MODE_SPECIALIZED) <= np.float32(0.1)))
    assert prog.select_host(pts, float("nan")).size == 0
