"""GPU parity tests (run with -m gpu on an MI355X). Everything goes through the C-ABI of
libsdfk.so; the checker is the golden vectors of the real reference (tests/golden) and the CPU
oracle (oracle/sdf_oracle.py).

Tolerance (BASELINE.json north_star): |gpu - ref| <= 1e-6 * max(1, |ref|), the
reference being fed the same fp32-rounded coordinates. Scenes with jumps (sign, binarisation, cell
boundaries — scenes.DISCONTINUOUS) may flip branch for points within rounding of the jump; the number of
such points is PINNED per scene (tests/golden/parity_budget.json: zero for all scenes but the ones listed
there), never bounded by a percentage. The seeded random trees nest SUM /
DIFFERENCE / displacement several levels deep, where the result is far smaller than the fields and
coordinates it is computed from: for those scenes (only) the denominator is
max(1, |ref|, largest intermediate magnitude at that point) as reported by the float64 oracle.
"""
import ctypes
import json
import os

import numpy as np
import pytest

import scenes
import aegolius_amd
import aegolius_amd.cores as ns
from aegolius_amd._lower import lower_geometry
from oracle import sdf_oracle

pytestmark = pytest.mark.gpu

TOL = 1e-6
ALL = sorted(scenes.SCENES)


def violations(out, ref, magnitude=None):
    out = out.astype(np.float64)
    both_nan = np.isnan(ref) & np.isnan(out)
    scale = np.maximum(1.0, np.abs(ref))
    if magnitude is not None:
        scale = np.maximum(scale, magnitude)
    err = np.abs(out - ref) / scale
    err[both_nan] = 0.0
    bad = ~(err <= TOL)
    return err, bad


# Off-point counts per scene, recorded on an MI355X by `python tests/report_gpu_parity.py --write-budget` (scenes that
# are not listed have NONE): a point within fp32 rounding of a jump of the scene (sign, binarisation, cell and sector
# boundaries) may land on the other side, and a handful of scenes have such points in the fixed input cloud. The count
# is pinned, not bounded by a percentage: a regression that corrupts a few points of any scene fails.
with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "parity_budget.json")) as _f:
    BUDGET = json.load(_f)


def check(name, out, ref, magnitude=None, budget=None):
    assert out.dtype == np.float32 and out.shape == ref.shape
    err, bad = violations(out, ref, magnitude)
    allowed = (BUDGET["scenes"].get(name, 0) if budget is None else budget)
    assert bad.sum() <= allowed, "%s: %d points off (recorded: %d), max rel err %.3e" % (name, bad.sum(), allowed,
                                                                                       np.nanmax(err))
    if allowed:
        # a point on the other side of a jump is off by the height of the jump, never by more than the scene's range
        span = np.nanmax(ref) - np.nanmin(ref)
        assert np.all(np.abs(out.astype(np.float64) - ref)[bad] <= span * (1 + 1e-6) + 1e-6), name


@pytest.fixture(scope="module")
def engine(built):
    built.require_gpu()
    return built


@pytest.fixture(autouse=True)
def _restore_mode():
    yield
    aegolius_amd.config.mode = 0


@pytest.mark.parametrize("name", ALL)
def test_scene_matches_reference_golden(name, engine, golden, golden_inputs):
    """Every primitive / modification / combiner / tree scene, both kernel flavours."""
    data, _ = golden
    ref = data["scene/" + name]
    magnitude = None
    if name.startswith("random_tree_"):
        again, magnitude = sdf_oracle.evaluate_with_magnitude(scenes.SCENES[name](ns), golden_inputs)
        np.testing.assert_allclose(again, ref, rtol=1e-12, atol=1e-12)
    outs = []
    for mode in (engine.MODE_SPECIALIZED, engine.MODE_INTERPRET):
        aegolius_amd.config.mode = mode
        co = golden_inputs.copy()
        out = scenes.SCENES[name](ns).create(co)
        np.testing.assert_array_equal(co, golden_inputs)
        check(name, out, ref, magnitude)
        outs.append(out)
    # same device functions, same contraction rules: the two flavours agree bit for bit
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("name", sorted(scenes.GRID_SCENES))
def test_grid_neighbourhood_scene_matches_reference_golden(name, engine, golden):
    """signed / conv_averaging / conv_edge_detection (staged evaluation: per-point programs + grid operators on the
    resident field + V_FIELD reads), both kernel flavours, tagged grid and plain array input."""
    data, meta = golden
    build, key = scenes.GRID_SCENES[name]
    co, res = scenes.grid_inputs(ns, key)
    tagged, _ = ns.generate_grid(*scenes.GRIDS[key])
    ref = data["gridscene/" + name]
    # edge detection subtracts neighbouring samples of an fp32 field: its error is judged against sum |w||u|
    # (reported by the oracle), like the intermediate magnitudes of the random trees
    magnitude = None
    if "edge_detection" in name:
        with np.errstate(all="ignore"):
            again, magnitude = sdf_oracle.evaluate_with_magnitude(build(ns, res), co)
        np.testing.assert_allclose(again, ref, rtol=1e-12, atol=1e-12)
    outs = []
    for mode in (engine.MODE_SPECIALIZED, engine.MODE_INTERPRET):
        aegolius_amd.config.mode = mode
        for arr in (co.copy(), tagged):
            out = build(ns, res).create(arr)
            assert list(out.shape) == meta["grid_scenes"][name]["shape"]
            check(name, out.ravel(), ref.ravel(), magnitude, budget=BUDGET["grid_scenes"].get(name, 0))
            outs.append(out)
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])


def test_grid_operators_at_scale(engine):
    """129^3 (cfg 1 size): `signed` of |sphere| followed by an averaging agrees with the oracle (sign flips only
    for the few points whose boundary test sits within rounding of the threshold);
    box averaging keeps a constant field constant (also with a kernel wider than the LDS chunk, several iterations,
    and on a flat 2-D field), and matches scipy-free arithmetic on a linear ramp away from the border."""
    co, res = ns.generate_grid((2, 2, 2), (128, 128, 128))
    def build():
        s = ns.Sphere(0.6)
        s.boundary()
        s.signed((128, 128, 128))
        s.conv_averaging((3, 3, 3), 1, (128, 128, 128))
        return s
    got = build().create(co)
    co64 = np.asarray(co).astype(np.float32).astype(np.float64)
    ref = sdf_oracle.evaluate(build(), co64)
    assert (ref < 0).sum() > 0.05 * ref.size
    err, bad = violations(got, ref)
    # sign flips of `signed` for the points whose boundary test sits within rounding of the threshold: recorded count
    assert bad.sum() <= BUDGET["grid_operators_at_scale"], (int(bad.sum()), float(np.nanmax(err)))
    # constant field / linear ramp through the operator alone
    lib = engine.lib()
    n0, n1, n2 = 33, 41, 130
    ramp = (np.arange(n0)[:, None, None] * 0.5 + np.arange(n1)[None, :, None] * 0.25 + np.arange(n2)[None, None, :] * 2.0)
    for field, shape, kern, iters in ((np.full((n0, n1, n2), 1.25), (n0, n1, n2), (9, 9, 9), 2),
                                      (ramp, (n0, n1, n2), (3, 5, 7), 1), (ramp, (n0, n1, n2), (2, 4, 6), 1),
                                      (np.full((n0 * n1, n2), -0.75), (n0 * n1, n2, 1), (5, 5, 1), 3)):
        host = np.ascontiguousarray(field, dtype=np.float32).ravel()
        d = lib.sdfk_malloc(host.nbytes)
        try:
            engine.check(lib.sdfk_memcpy_h2d(ctypes.c_void_p(d), host.ctypes.data_as(ctypes.c_void_p), host.nbytes), "h2d")
            engine.check(lib.sdfk_grid_box_average(ctypes.c_void_p(d), shape[0], shape[1], shape[2], kern[0], kern[1], kern[2],
                                                   iters, None, None), "box")
            out = np.empty_like(host)
            engine.check(lib.sdfk_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d), host.nbytes), "d2h")
        finally:
            lib.sdfk_free(ctypes.c_void_p(d))
        out = out.reshape(field.shape)
        if field is ramp:
            # a box average of a linear function is the function at the box centre: offsets -((k-1)//2) .. k//2
            shift = sum(w * (k // 2 - (k - 1) // 2) / 2.0 for w, k in zip((0.5, 0.25, 2.0), kern))
            inner = (slice(5, -5),) * 3
            np.testing.assert_allclose(out[inner], (ramp + shift)[inner], rtol=1e-6)
        else:
            np.testing.assert_array_equal(out, field.astype(np.float32))


def test_float32_and_float64_coordinates_agree(engine, golden_inputs):
    tree = scenes.cfg2_tree(ns)
    a = tree.create(golden_inputs)
    b = tree.create(golden_inputs.astype(np.float32))
    np.testing.assert_array_equal(a, b)


def test_empty_and_ragged_inputs(engine):
    tree = scenes.cfg2_tree(ns)
    assert tree.create(np.zeros((3, 0))).shape == (0,)
    rng = np.random.default_rng(3)
    for n in (1, 2, 3, 4, 5, 63, 64, 65, 255, 257, 1023, 1025, 4099):
        co = rng.uniform(-1, 1, (3, n)).astype(np.float32).astype(np.float64)
        out = tree.create(co)
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(scenes.cfg2_tree(ns), co)
        check("tree_cfg2_smooth_union10", out, ref)
    with pytest.raises(ValueError):
        tree.create(np.zeros((2, 10)))


os.environ.setdefault("SDFK_PLANE_BLOCKS", "1")      # the tests exercise the plane hint of sdfk_eval_device_rows3d


def _device_eval(engine, prog, co32, n, stride, misalign=0, mode=None, row_len=None, flat=False, plane_rows=None,
                 first_row_in_plane=0):
    """eval_device on raw HIP buffers (optionally shifted by `misalign` floats to defeat 16-B alignment)."""
    lib = engine.lib()
    d_co = lib.sdfk_malloc((3 * stride + misalign + 4) * 4)
    d_out = lib.sdfk_malloc((n + misalign + 4) * 4)
    try:
        host = np.zeros((3, stride), dtype=np.float32)
        host[:, :n] = co32
        engine.check(lib.sdfk_memcpy_h2d(ctypes.c_void_p(d_co + 4 * misalign), host.ctypes.data_as(ctypes.c_void_p),
                                         host.nbytes), "h2d")
        prog.eval_device(d_co + 4 * misalign, n, stride, d_out + 4 * misalign,
                         mode=engine.MODE_SPECIALIZED if mode is None else mode, row_len=row_len, flat=flat,
                         plane_rows=plane_rows, first_row_in_plane=first_row_in_plane)
        engine.check(lib.sdfk_sync(None), "sync")
        out = np.empty(n, dtype=np.float32)
        engine.check(lib.sdfk_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out + 4 * misalign),
                                         out.nbytes), "d2h")
        return out
    finally:
        lib.sdfk_free(ctypes.c_void_p(d_co))
        lib.sdfk_free(ctypes.c_void_p(d_out))


def test_device_pointers_aligned_and_unaligned(engine, golden_inputs):
    """16-byte vector path, scalar path (odd stride / shifted pointers) and the tail all agree."""
    low = lower_geometry(scenes.cfg5_tree(ns))
    prog = engine.Program(low.code, low.params, low.tables, low.result_reg)
    co32 = golden_inputs.astype(np.float32)
    n = co32.shape[1] - 1                                    # 2041: not a multiple of 4 -> tail launch
    want = prog.eval_host(co32[:, :n])
    for mode in (engine.MODE_SPECIALIZED, engine.MODE_INTERPRET):
        for stride, mis in ((2048, 0), (2041, 0), (2044, 1), (2045, 3)):
            got = _device_eval(engine, prog, co32[:, :n], n, stride, mis, mode)
            np.testing.assert_array_equal(got, want)


def test_program_rejects_malformed_code(engine):
    from aegolius_amd import _ops
    sphere = _ops.BY_NAME["P_SPHERE"].code
    with pytest.raises(engine.SdfkError):                     # unknown opcode
        engine.Program([[250, 0]], [0.5], [], 0)
    with pytest.raises(engine.SdfkError):                     # parameters out of range
        engine.Program([[sphere, 4]], [0.5], [], 0)
    with pytest.raises(engine.SdfkError):                     # reads coordinate register 5, never written
        engine.Program([[sphere | (5 << 16), 0]], [0.5], [], 0)
    with pytest.raises(engine.SdfkError):                     # result register never written
        engine.Program([[sphere, 0]], [0.5], [], 3)
    near = _ops.BY_NAME["P_NEAREST3"].code
    with pytest.raises(engine.SdfkError):                     # table rows beyond the table
        engine.Program([[near, 0]], [10.0, 0.0], np.zeros(9), 0)


def test_same_topology_new_parameters_reuses_kernel(engine, golden_inputs):
    """The specialised kernel is keyed by tree topology; shape parameters stay runtime data."""
    outs = []
    for r in (0.3, 0.45):
        s = scenes.placed(ns.Sphere(r))
        out = s.create(golden_inputs)
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(s, golden_inputs)
        check("prim_sphere", out, ref)
        outs.append(out)
    assert np.abs(outs[0] - outs[1]).max() > 0.1
    low = lower_geometry(scenes.placed(ns.Sphere(0.3)))
    prog = engine.Program(low.code, low.params, low.tables, low.result_reg)
    a = prog.eval_host(golden_inputs)
    p2 = low.params.copy()
    p2[-1] = 0.45
    prog.set_params(p2)
    b = prog.eval_host(golden_inputs)
    np.testing.assert_array_equal(a, outs[0])
    np.testing.assert_array_equal(b, outs[1])


def test_grid_fill_and_grid_eval_match_generate_grid(engine):
    from aegolius_amd.cores.helper_functions import grid_axes
    size, res = (2.0, 3.0, 1.0), (20, 12, 34)
    co, r = ns.generate_grid(size, res)
    axes64, r2 = grid_axes(size, res)
    assert r == r2
    axes = [a.astype(np.float32) for a in axes64]
    n = co.shape[1]
    lib = engine.lib()
    low = lower_geometry(scenes.cfg3_chain(ns))
    prog = engine.Program(low.code, low.params, low.tables, low.result_reg)
    want = prog.eval_host(co)
    for start, count in ((0, n), (7, n - 7), (1234, 3001), (n - 5, 5)):
        stride = (count + 3) // 4 * 4
        d_co = lib.sdfk_malloc(3 * stride * 4)
        d_out = lib.sdfk_malloc(stride * 4)
        try:
            engine.grid_fill(d_co, stride, axes, start, count)
            host = np.empty((3, stride), dtype=np.float32)
            engine.check(lib.sdfk_memcpy_d2h(host.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_co),
                                             host.nbytes), "d2h")
            np.testing.assert_array_equal(host[:, :count], co[:, start:start + count].astype(np.float32))
            for mode in (engine.MODE_SPECIALIZED, engine.MODE_INTERPRET):
                prog.eval_grid(axes, start, count, d_out, mode=mode)
                got = np.empty(count, dtype=np.float32)
                engine.check(lib.sdfk_memcpy_d2h(got.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out),
                                                 got.nbytes), "d2h")
                np.testing.assert_array_equal(got, want[start:start + count])
        finally:
            lib.sdfk_free(ctypes.c_void_p(d_co))
            lib.sdfk_free(ctypes.c_void_p(d_out))


@pytest.mark.parametrize("cfg,request_res", [("cfg2", 512), ("cfg3", 1024)])
def test_baseline_size_grids(cfg, request_res, engine):
    """BASELINE.json configs[1] (513^3) and configs[2] (1025^3) at FULL size, evaluated straight from the
    grid tables in x-slabs. Checked through (a) a random sample of points against the oracle and
    (b) slab-partition independence: re-evaluating a window with a different partition is identical."""
    from aegolius_amd.cores.helper_functions import grid_axes
    build, size = (scenes.cfg2_tree, (2, 2, 2)) if cfg == "cfg2" else (scenes.cfg3_chain, (4, 4, 4))
    axes64, res = grid_axes(size, (request_res,) * 3)
    axes = [a.astype(np.float32) for a in axes64]
    n = res[0] * res[1] * res[2]
    assert n == (request_res + 1) ** 3
    low = lower_geometry(build(ns))
    prog = engine.Program(low.code, low.params, low.tables, low.result_reg)
    lib = engine.lib()
    d_out = lib.sdfk_malloc(n * 4)
    assert d_out
    try:
        plane = res[1] * res[2]
        slab = 97 * plane
        for s in range(0, n, slab):
            c = min(slab, n - s)
            prog.eval_grid(axes, s, c, d_out + 4 * s)
        rng = np.random.default_rng(11)
        idx = np.sort(rng.choice(n, size=20000, replace=False))
        # pull the sampled values (contiguous windows of 1 float each would be slow: fetch planes lazily)
        got = np.empty(idx.size, dtype=np.float32)
        win = np.empty(plane, dtype=np.float32)
        cur = -1
        for k, i in enumerate(idx.tolist()):
            p = i // plane
            if p != cur:
                engine.check(lib.sdfk_memcpy_d2h(win.ctypes.data_as(ctypes.c_void_p),
                                                 ctypes.c_void_p(d_out + 4 * p * plane), win.nbytes), "d2h")
                cur = p
            got[k] = win[i - p * plane]
        ix, rem = idx // plane, idx % plane
        iy, iz = rem // res[2], rem % res[2]
        co = np.stack([axes[0][ix], axes[1][iy], axes[2][iz]]).astype(np.float64)
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(build(ns), co)
        check("tree_cfg3_mod_chain" if cfg == "cfg3" else "tree_cfg2_smooth_union10", got, ref)
        # partition independence on a window straddling slab boundaries
        w0, wn = 96 * plane + 12345, 2 * plane + 777
        d_w = lib.sdfk_malloc(wn * 4)
        try:
            prog.eval_grid(axes, w0, wn, d_w)
            a, b = np.empty(wn, dtype=np.float32), np.empty(wn, dtype=np.float32)
            engine.check(lib.sdfk_memcpy_d2h(a.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_w), a.nbytes), "d2h")
            engine.check(lib.sdfk_memcpy_d2h(b.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_out + 4 * w0),
                                             b.nbytes), "d2h")
            np.testing.assert_array_equal(a, b)
        finally:
            lib.sdfk_free(ctypes.c_void_p(d_w))
    finally:
        lib.sdfk_free(ctypes.c_void_p(d_out))


def test_sdf_function_called_directly(engine, golden_inputs):
    """`sdf_*(co, *params)` and modification closures are callable like the reference's functions."""
    out = ns.sdf_torus(golden_inputs, 0.6, 0.17)
    with np.errstate(all="ignore"):
        ref = sdf_oracle.PRIMS["sdf_torus"](golden_inputs, 0.6, 0.17)
    check("prim_torus", out, ref)
    b = ns.Box(0.6, 0.4, 0.3)
    f = b.rounding(0.05)
    out = f(golden_inputs, (0.6, 0.4, 0.3))
    with np.errstate(all="ignore"):
        ref = sdf_oracle.PRIMS["sdf_box"](golden_inputs, (0.6, 0.4, 0.3)) - 0.05
    check("mod_rounding", out, ref)


def test_create_on_a_tagged_grid_equals_create_on_the_plain_array(engine):
    """generate_grid's read-only tagged array takes the table-driven path; a plain copy takes the upload path."""
    for build, size, res in ((scenes.cfg2_tree, (2, 2, 2), (40, 52, 300)), (scenes.cfg3_chain, (4, 4, 4), (30, 30, 30)),
                             (scenes.cfg4_scene2d, (10, 10), (300, 500))):
        co, _ = ns.generate_grid(size, res)
        assert co.grid_axes is not None
        fast = build(ns).create(co)
        slow = build(ns).create(np.array(co))
        assert fast.dtype == np.float32
        np.testing.assert_array_equal(fast, slow)


def test_output_dtype_option_and_point_cloud(engine, golden_inputs):
    s = ns.Sphere(0.83)                      # no input point within rounding of the surface
    try:
        aegolius_amd.config.output_dtype = np.float64
        assert s.create(golden_inputs).dtype == np.float64
    finally:
        aegolius_amd.config.output_dtype = np.float32
    pts = s.point_cloud(golden_inputs)
    inside = np.linalg.norm(golden_inputs, axis=0) <= 0.83
    assert pts.shape == (3, int(inside.sum())) and np.all(pts[2] == 0)
    np.testing.assert_array_equal(pts[:2], golden_inputs[:2, inside])


def test_sharded_evaluation_matches_whole_grid(engine):
    """Slab-sharding on ONE device (each 'rank' = one slab): concatenated slabs == the whole-grid field."""
    import torch
    from aegolius_amd.distributed import _GpuSlabEvaluator, slab_bounds
    from aegolius_amd.cores.helper_functions import grid_axes
    tree = scenes.cfg5_tree(ns)
    size, resolution = (3, 3, 3), (40, 30, 52)
    co, res = ns.generate_grid(size, resolution)
    whole = tree.create(co)
    axes = [a.astype(np.float32) for a in grid_axes(size, resolution)[0]]
    ev = _GpuSlabEvaluator(tree)
    for world, u in ((2, 1), (3, 1), (8, 1), (2, res[2]), (3, res[2]), (8, res[2])):   # arbitrary cuts and whole rows
        parts = [ev(axes, *slab_bounds(whole.size, world, r, unit)) for r in range(world) for unit in (u,)]
        torch.cuda.synchronize()
        np.testing.assert_array_equal(torch.cat(parts).cpu().numpy(), whole)


def test_single_process_sharded_grid(engine):
    """sdfk_eval_grid_sharded: slabs of whole rows on a device list (here all on device 0), concurrent host
    threads; equals the unsharded field; bad device indices are refused."""
    from aegolius_amd.cores.helper_functions import grid_axes
    tree = scenes.cfg5_tree(ns)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    axes = [a.astype(np.float32) for a in grid_axes((3, 3, 3), (20, 30, 70))[0]]
    whole = prog.eval_grid_host(axes)
    for shards in (1, 3, 8):
        np.testing.assert_array_equal(prog.eval_grid_sharded(axes, shards, devices=[0] * shards), whole)
    np.testing.assert_array_equal(prog.eval_grid_sharded(axes, 2), whole)
    with pytest.raises(engine.SdfkError):
        prog.eval_grid_sharded(axes, 2, devices=[0, 99])
    # the field left on the devices: slabs land in place / by peer copies in one device buffer (sdfk_eval_grid_sharded_device)
    for shards in (1, 3, 8):
        field = prog.eval_grid_sharded_resident(axes, shards, devices=[0] * shards, gather_device=0)
        np.testing.assert_array_equal(field.numpy(), whole)
        field.free()
    with pytest.raises(engine.SdfkError):
        prog.eval_grid_sharded_resident(axes, 2, devices=[0, 0], gather_device=99)
    # a refused call leaves nothing behind in the runtime: the next launch check does not report ITS error
    np.testing.assert_array_equal(prog.eval_grid_host(axes), whole)
    # the device-to-device copy path (hipMemcpyPeerAsync), forced for the shards of the gather device itself
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests'); import numpy as np, scenes\n"
              "import aegolius_amd.cores as ns\nfrom aegolius_amd import _engine\nfrom aegolius_amd._lower import lower_geometry\n"
              "from aegolius_amd.cores.helper_functions import grid_axes\n"
              "prog = _engine.Program.from_lowered(lower_geometry(scenes.cfg5_tree(ns)))\n"
              "axes = [a.astype(np.float32) for a in grid_axes((3, 3, 3), (20, 30, 70))[0]]\n"
              "whole = prog.eval_grid_host(axes)\n"
              "for shards in (1, 3, 8):\n"
              "    assert np.array_equal(prog.eval_grid_sharded_resident(axes, shards, devices=[0] * shards).numpy(), whole)\n"
              "print('ok')\n" % (root, root))
    res = subprocess.run([sys.executable, "-c", script], env=dict(os.environ, SDFK_FORCE_PEER_COPY="1"), capture_output=True,
                         text=True, timeout=300)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stderr[-2000:]


CULL_SCENES = ["tree_cfg2_smooth_union10", "tree_cfg5_three_level", "tree_cfg4_union50_2d", "tree_pawn_3D",
               "tree_deep_right", "combine_SMOOTH_SUBTRACT2", "combine_SUBTRACT2", "combine_INTERSECT_nary",
               "combine_SMOOTH_INTERSECT2", "combine_modified_result", "alias_symmetry_in_child_not_visible"]


ROW_SHAPES = [((9, 37, 1025), 0), ((20, 33, 64), 0), ((7, 50, 40), 1), ((3, 16, 32), 3), ((5, 5, 333), 2)]


@pytest.mark.parametrize("name", CULL_SCENES)
def test_row_block_culling_is_bit_exact(name, engine):
    """sdfk_eval_device_rows (bricks of 32 points x 16 rows, several packed pairs per lane, per-lane transform
    bases) returns exactly what the plain specialised kernel returns: odd row lengths (rows only 4-byte
    aligned), row counts that are no multiple of the brick height, last chunks of 1 / 8 / 13 points,
    shifted pointers — and a row length that has nothing to do with the data."""
    low = lower_geometry(scenes.SCENES[name](ns))
    prog = engine.Program.from_lowered(low)
    for shape, mis in ROW_SHAPES:
        if "2d" in name:
            co, _ = ns.generate_grid((10, 10), (shape[0] * shape[1] - 1, shape[2] - 1))
        else:
            co, _ = ns.generate_grid((2.6, 2.6, 2.6), tuple(r - 1 for r in shape))
        co32 = co.astype(np.float32)
        n = co32.shape[1]
        # rows run along z (3-D) or along y (2-D): the first change of the next slower coordinate ends row 0
        slower = co32[0] if "2d" in name else co32[1]
        row_len = int(np.flatnonzero(slower != slower[0])[0])
        assert n % row_len == 0
        stride = n + 5
        plain = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_NOCULL)
        rows = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_SPECIALIZED, row_len=row_len)
        np.testing.assert_array_equal(rows, plain)
        if "2d" not in name:
            # the plane hint (row blocks never straddle two x-planes): the true plane height, slabs that start
            # anywhere inside a plane, and values that have nothing to do with the data
            p_rows = shape[1]
            for plane_rows, first in ((p_rows, 0), (7, 3), (1, 0), (16, 15), (10 ** 6, 5)):
                got = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_SPECIALIZED, row_len=row_len,
                                   plane_rows=plane_rows, first_row_in_plane=first)
                np.testing.assert_array_equal(got, plain)
            r_first = p_rows + 3 if shape[0] > 2 else 1                   # a slab of whole rows from mid-plane
            sl = slice(r_first * row_len, n - 2 * row_len)
            m = sl.stop - sl.start
            got = _device_eval(engine, prog, co32[:, sl], m, m + 3, mis, engine.MODE_SPECIALIZED, row_len=row_len,
                               plane_rows=p_rows, first_row_in_plane=r_first % p_rows)
            np.testing.assert_array_equal(got, plain[sl])
        # the kernel built for flat grids (x shared by a row, z = 0): on the 2-D scenes its fast path, on the 3-D
        # ones (z is not 0, x and y change along "rows") every brick must fall back to the general path
        flat = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_SPECIALIZED, row_len=row_len, flat=True)
        np.testing.assert_array_equal(flat, plain)
        # grid flavour (coordinates from the per-axis tables): the whole grid and a slab of whole rows
        axes = [a.astype(np.float32) for a in co.grid_axes]
        np.testing.assert_array_equal(prog.eval_grid_host(axes), plain)
        s0, cnt = 3 * row_len, (n // row_len - 5) * row_len
        np.testing.assert_array_equal(prog.eval_grid_host(axes, s0, cnt), plain[s0:s0 + cnt])
    # scattered points under a row hint: nothing is uniform, nothing may be skipped wrongly
    rng = np.random.default_rng(11)
    pts = rng.uniform(-1.5, 1.5, (3, 48 * 100)).astype(np.float32)
    pts[:, 1000:1100] = pts[:, 1000:1001]                   # a run of identical points
    pts[:2, 2000:2048] = pts[:2, 2000:2001]                 # x/y-constant stretch that is not a whole row
    plain = _device_eval(engine, prog, pts, pts.shape[1], pts.shape[1], 0, engine.MODE_NOCULL)
    rows = _device_eval(engine, prog, pts, pts.shape[1], pts.shape[1], 0, engine.MODE_SPECIALIZED, row_len=48)
    np.testing.assert_array_equal(rows, plain)


@pytest.mark.parametrize("shift", [1.0e3, 1.0e5])
@pytest.mark.parametrize("name", ["tree_cfg2_smooth_union10", "tree_cfg5_three_level"])
def test_culling_far_from_the_origin_is_bit_exact(name, shift, engine):
    """The whole scene moved to |p| ~ 1e3 and ~ 1e5: fp32 rounding of the operands now grows with the coordinates
    (about 6e-8 |p| per step), not with their values — the cull margin carries a term in |c| + rho for that
    (csrc/sdfk_codegen.cpp emit_probe). Both culling kernels still return the plain kernel's bits, and the field
    stays within the movement of the float64 oracle under a one-ulp change of its fp32 inputs."""
    tree = scenes.SCENES[name](ns)
    tree.move((shift, -shift, 0.5 * shift))
    low = lower_geometry(tree)
    assert len(low.cull_sites) > 0
    prog = engine.Program.from_lowered(low)
    co, _ = ns.generate_grid((2.6, 2.6, 2.6), (36, 40, 257))
    co = np.asarray(co) + np.asarray([[shift], [-shift], [0.5 * shift]])
    co32 = np.ascontiguousarray(co, dtype=np.float32)
    n = co32.shape[1]
    plain = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_NOCULL)
    bricks = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED)
    rows = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=257)
    np.testing.assert_array_equal(bricks, plain)
    np.testing.assert_array_equal(rows, plain)
    co64 = co32.astype(np.float64)
    with np.errstate(all="ignore"):
        ref = sdf_oracle.evaluate(tree, co64)
        moved = np.zeros_like(ref)
        for axis in range(3):                                   # the oracle's own sensitivity to an ulp of the input
            bump = co64.copy()
            bump[axis] = np.nextafter(co32[axis], np.float32(np.inf)).astype(np.float64)
            moved = np.maximum(moved, np.abs(sdf_oracle.evaluate(tree, bump) - ref))
    err = np.abs(plain.astype(np.float64) - ref)
    assert np.all(err <= 1e-6 * np.maximum(1.0, np.abs(ref)) + 8.0 * moved + 8.0 * np.spacing(np.float32(shift))), float(err.max())


def _signed_numpy(s, sep, crop):
    """`signed` / `signed_old` on a 3-D array, restated from the oracle (oracle/sdf_oracle._signed; C/modifications.py:163-275)"""
    boundary = s < sep
    interior = None
    for axis in (0, 1):
        b = np.moveaxis(boundary, axis, 0)
        chu, chuu = np.zeros(b.shape), np.zeros(b.shape)
        chu[1:] = b[1:] * ~b[:-1]
        chuu[:-1] = b[:-1] * ~b[1:]
        mark = np.cumsum(chu, axis=0)
        fmark = np.flip(np.cumsum(chuu, axis=0), axis=0) if crop else np.flip(np.cumsum(np.flip(chuu, axis=0), axis=0), axis=0)
        part = np.moveaxis(np.clip(mark % 2 + fmark % 2, 0, 1), 0, axis)
        interior = part if interior is None else interior * part
    interior = sdf_oracle._conv_averaging(interior, (2, 2, 1), 1)
    if crop:
        interior = np.pad(interior[1:-1, 1:-1, 1:-1], pad_width=1, mode="edge")
    return s * (1 - 2 * (interior > 0.5))


@pytest.mark.parametrize("shape", [(3, 3, 3), (5, 4, 33), (9, 7, 65), (4, 6, 64), (7, 5, 32), (6, 6, 97), (12, 9, 129),
                                   (3, 40, 10), (40, 3, 5), (2, 2, 2), (17, 33, 31), (8, 8, 257)])
@pytest.mark.parametrize("crop", [1, 0])
def test_signed_bit_planes_on_awkward_shapes(shape, crop, engine):
    """sdfk_grid_signed packs the boundary test 32 points to a word along the last axis: word edges (32, 33, 64, 65
    points), a last point alone in its word (33, 65, 97, 129, 257), rows shorter than a word, the crop's clamps on all
    three axes and grids too thin for the caller's scratch — on noise, where every scan line changes parity many times.
    Bit-exact against the restated reference arithmetic."""
    if crop and min(shape) < 3:
        pytest.skip("the reference's crop needs three points per axis")
    rng = np.random.default_rng(sum(shape) + crop)
    lib = engine.lib()
    for density in (0.5, 0.15, 0.9):
        s = rng.uniform(0.0, 1.0, shape)
        # blobs as well as noise: runs of boundary points along both scan axes
        s[rng.uniform(size=shape) < 0.3] = 0.0
        sep = density
        host = np.ascontiguousarray(s, dtype=np.float32).ravel()
        want = _signed_numpy(host.astype(np.float64).reshape(shape), np.float64(np.float32(sep)), bool(crop)).astype(np.float32).ravel()
        d = lib.sdfk_malloc(host.nbytes + 64)
        try:
            engine.check(lib.sdfk_memcpy_h2d(ctypes.c_void_p(d), host.ctypes.data_as(ctypes.c_void_p), host.nbytes), "h2d")
            engine.check(lib.sdfk_grid_signed(ctypes.c_void_p(d), shape[0], shape[1], shape[2], float(np.float32(sep)), crop, None, None), "signed")
            got = np.empty_like(host)
            engine.check(lib.sdfk_memcpy_d2h(got.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d), host.nbytes), "d2h")
        finally:
            lib.sdfk_free(ctypes.c_void_p(d))
        np.testing.assert_array_equal(got, want)
        assert (want < 0).any() or density < 0.2 or min(shape) < 4


def _ill_conditioned(ns):
    """scenes whose float64 reference itself moves by more than 1e-6 under a one-ulp change of its fp32 inputs"""
    out = {}
    out["neucircle_order_0.3_moved"] = ns.NEUCircle(0.5, 0.3)    # |x|^p + |y|^p with p < 1: unbounded slope at the axes
    out["neucircle_order_0.3_moved"].move((0.2, -0.1, 0.0))
    s = ns.Sphere(0.6)
    s.capped_exponential(1.5, 0.004)                              # value map of width 4e-3: slope 1 / 4e-3
    out["capped_exponential_narrow"] = s
    g = ns.Sphere(0.6)
    g.gaussian_boundary(2.0, 0.03)
    out["gaussian_boundary_narrow"] = g
    t = ns.Torus(0.6, 0.2)
    t.sigmoid_falloff(1.0, 0.003)
    out["sigmoid_falloff_narrow"] = t
    return out


@pytest.mark.parametrize("name", sorted(_ill_conditioned(ns)))
def test_ill_conditioned_scenes_stay_within_the_references_own_sensitivity(name, engine):
    """Fractional super-ellipse orders and narrow value maps amplify the fp32 rounding of the INPUT beyond 1e-6: there
    the bar is the movement of the float64 reference under a one-ulp change of its inputs (x8), and the plain 1e-6
    everywhere else. Bounded both ways: no point outside the combined bar, and the scene really is of that class
    (some point does need the sensitivity term — otherwise it belongs with the ordinary scenes)."""
    geo = _ill_conditioned(ns)[name]
    co = scenes.input_points()
    with np.errstate(all="ignore"):
        ref = np.asarray(sdf_oracle.evaluate(geo, co.copy()), dtype=np.float64)
        sens = scenes.input_sensitivity(lambda c: sdf_oracle.evaluate(_ill_conditioned(ns)[name], c), co)
    for mode in (engine.MODE_SPECIALIZED, engine.MODE_INTERPRET):
        prog = engine.Program.from_lowered(lower_geometry(geo))
        got = prog.eval_host(np.ascontiguousarray(co, dtype=np.float32), mode=mode).astype(np.float64)
        err = np.abs(got - ref)
        err[np.isnan(got) & np.isnan(ref)] = 0.0
        plain_bar = 1e-6 * np.maximum(1.0, np.abs(ref))
        assert np.all(err <= plain_bar + 8.0 * sens), (name, float(np.nanmax(err - plain_bar - 8.0 * sens)))
        assert np.mean(err > plain_bar) <= 0.10, (name, float(np.mean(err > plain_bar)))
    assert np.any(sens > 1e-6 * np.maximum(1.0, np.abs(ref))), name


@pytest.mark.parametrize("name", CULL_SCENES)
def test_brick_culling_is_bit_exact(name, engine):
    """The culling tile kernel (skips operand subtrees per 128-point brick) returns exactly what the
    plain specialised kernel returns — on a grid large enough for thousands of bricks, ragged end included."""
    low = lower_geometry(scenes.SCENES[name](ns))
    assert len(low.cull_sites) > 0
    prog = engine.Program.from_lowered(low)
    size = (10, 10) if "2d" in name else (2.6, 2.6, 2.6)
    co, _ = ns.generate_grid(size, (1400, 1100) if "2d" in name else (100, 112, 130))
    co32 = co.astype(np.float32)
    n = co32.shape[1] - 3                                   # ragged: not a multiple of 4 / 128 / 2048
    stride = (n + 255) // 256 * 256
    culled = _device_eval(engine, prog, co32[:, :n], n, stride, 0, engine.MODE_SPECIALIZED)
    plain = _device_eval(engine, prog, co32[:, :n], n, stride, 0, engine.MODE_NOCULL)
    np.testing.assert_array_equal(culled, plain)
    # the grid flavour of the culling kernel (coordinates expanded from the per-axis tables) agrees as well
    axes = [a.astype(np.float32) for a in co.grid_axes]
    for start, count in ((0, n), (12345, 700001)):
        np.testing.assert_array_equal(prog.eval_grid_host(axes, start, count), plain[start:start + count])
    # and the plain one is right (sampled against the oracle)
    idx = np.random.default_rng(5).choice(n, 4000, replace=False)
    with np.errstate(all="ignore"):
        ref = sdf_oracle.evaluate(scenes.SCENES[name](ns), co32[:, idx].astype(np.float64))
    check(name, plain[idx], ref)


def test_brick_masks_are_conservative(engine):
    """Every skip decision of the culling probe is justified point by point (float64 oracle): where the
    mask says 'second operand of site k is irrelevant on this brick', d_k - acc_{k-1} >= w holds at every
    point of the brick; where it says 'first operand irrelevant', acc_{k-1} - d_k >= w."""
    from aegolius_amd._ir import CombineSDF
    tree = scenes.cfg2_tree(ns)
    low = lower_geometry(tree)
    prog = engine.Program.from_lowered(low)
    co, _ = ns.generate_grid((2, 2, 2), (32, 32, 1024))      # rows of 1025 points: bricks are 1/8 of a row
    co32 = co.astype(np.float32)
    n = co32.shape[1]
    stride = (n + 255) // 256 * 256
    lib = engine.lib()
    nb = (n + 2047) // 2048 * 16
    d_co, d_m = lib.sdfk_malloc(3 * stride * 4), lib.sdfk_malloc(nb * 8)
    try:
        host = np.zeros((3, stride), dtype=np.float32)
        host[:, :n] = co32
        engine.check(lib.sdfk_memcpy_h2d(ctypes.c_void_p(d_co), host.ctypes.data_as(ctypes.c_void_p), host.nbytes), "h2d")
        engine.check(lib.sdfk_debug_brick_masks(prog.handle, ctypes.c_void_p(d_co), n, stride, ctypes.c_void_p(d_m),
                                                None), "masks")
        engine.check(lib.sdfk_sync(None), "sync")
        masks = np.empty(nb, dtype=np.uint64)
        engine.check(lib.sdfk_memcpy_d2h(masks.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_m), masks.nbytes), "d2h")
    finally:
        lib.sdfk_free(ctypes.c_void_p(d_co))
        lib.sdfk_free(ctypes.c_void_p(d_m))
    masks = masks[:(n + 127) // 128]
    # per-primitive fields of the left-deep chain, in chain order
    prims, node = [], tree
    while isinstance(node.modified_object, CombineSDF):
        a, b = node.modified_object.children
        prims.append(b)
        node = a
    prims.append(node)
    prims = prims[::-1]
    co64 = co32.astype(np.float64)
    d = [sdf_oracle.evaluate(p, co64) for p in prims]
    w = 0.1
    pad = (-n) % 128
    acc = d[0].copy()
    n_skipped = 0
    for k in range(1, 10):
        gap = np.concatenate([d[k] - acc, np.full(pad, np.inf)]).reshape(-1, 128)
        skip_b = ((masks >> np.uint64(2 * (k - 1) + 1)) & np.uint64(1)).astype(bool)
        skip_a = ((masks >> np.uint64(2 * (k - 1))) & np.uint64(1)).astype(bool)
        assert np.all(gap[skip_b].min(axis=1) >= w)
        gap_a = np.concatenate([acc - d[k], np.full(pad, np.inf)]).reshape(-1, 128)
        assert np.all(gap_a[skip_a].min(axis=1) >= w)
        assert not np.any(skip_a & skip_b)
        n_skipped += int(skip_b.sum()) + int(skip_a.sum())
        acc = sdf_oracle.smin_poly(acc, d[k], w, 3)
    assert n_skipped > 0.3 * masks.size * 9          # and the probe is not vacuous
    # 7 of every ~8 bricks lie inside one row (x/y-constant runs), the others straddle a row end
    assert 0.8 < ((masks >> np.uint64(63)) & np.uint64(1)).mean() < 0.95


def test_row_block_masks_are_conservative(engine):
    """The same justification for the row-block kernel (sdfk_eval_device_rows): bricks are 32-point windows
    (aligned in the flat array) of 16 consecutive rows; every skip bit is checked against the float64 oracle at
    every point of its brick, the probe must cull far more than the 128-point line bricks do, and all bricks of
    a regular grid must come out as 'uniform rows'."""
    from aegolius_amd._ir import CombineSDF
    tree = scenes.cfg2_tree(ns)
    low = lower_geometry(tree)
    prog = engine.Program.from_lowered(low)
    dz = 2.0 / 1024                                          # the spacing of the 1025^3 north-star grid, on a slab
    co, res = ns.generate_grid((8 * dz, 128 * dz, 2), (8, 128, 1024))
    co32 = co.astype(np.float32)
    n, L = co32.shape[1], 1025
    assert n % L == 0 and res[-1] == L
    R = n // L
    lib = engine.lib()
    nb, brows = ctypes.c_int64(0), ctypes.c_int(0)
    engine.check(lib.sdfk_debug_row_masks(prog.handle, ctypes.c_void_p(1), n, n, L, None, ctypes.byref(nb),
                                          ctypes.byref(brows), None), "size query")
    nb, brows = nb.value, brows.value
    nchunk = (L + 62) // 32
    assert nb == nchunk * ((R + brows - 1) // brows)
    d_co, d_m = lib.sdfk_malloc(3 * n * 4), lib.sdfk_malloc(nb * 24)
    try:
        engine.check(lib.sdfk_memcpy_h2d(ctypes.c_void_p(d_co), co32.ctypes.data_as(ctypes.c_void_p), co32.nbytes), "h2d")
        engine.check(lib.sdfk_debug_row_masks(prog.handle, ctypes.c_void_p(d_co), n, n, L, ctypes.c_void_p(d_m), None,
                                              None, None), "masks")
        engine.check(lib.sdfk_sync(None), "sync")
        words = np.empty((nb, 3), dtype=np.uint64)
        engine.check(lib.sdfk_memcpy_d2h(words.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_m), words.nbytes), "d2h")
    finally:
        lib.sdfk_free(ctypes.c_void_p(d_co))
        lib.sdfk_free(ctypes.c_void_p(d_m))
    masks, uniform = words[:, 0], words[:, 2]
    assert np.all(words[:, 1] == 0) and np.all(uniform == 1)
    # brick of every point
    r, z = np.divmod(np.arange(n, dtype=np.int64), L)
    brick = (r // brows) * nchunk + ((r * L + z) >> 5) - ((r * L) >> 5)
    assert brick.max() < nb
    prims, node = [], tree
    while isinstance(node.modified_object, CombineSDF):
        a, b = node.modified_object.children
        prims.append(b)
        node = a
    prims.append(node)
    prims = prims[::-1]
    co64 = co32.astype(np.float64)
    d = [sdf_oracle.evaluate(p, co64) for p in prims]
    w = 0.1
    acc = d[0].copy()
    alive = np.ones((10, nb), dtype=bool)                     # primitives a brick still evaluates
    for k in range(1, 10):
        skip_b = ((masks >> np.uint64(2 * (k - 1) + 1)) & np.uint64(1)).astype(bool)
        skip_a = ((masks >> np.uint64(2 * (k - 1))) & np.uint64(1)).astype(bool)
        alive[k] &= ~skip_b
        alive[:k] &= ~skip_a
        gap_b = np.full(nb, np.inf)
        np.minimum.at(gap_b, brick, d[k] - acc)
        gap_a = np.full(nb, np.inf)
        np.minimum.at(gap_a, brick, acc - d[k])
        assert np.all(gap_b[skip_b] >= w) and np.all(gap_a[skip_a] >= w)
        assert not np.any(skip_a & skip_b)
        acc = sdf_oracle.smin_poly(acc, d[k], w, 3)
    assert alive.sum(axis=0).min() >= 1
    assert alive.mean() < 0.45, alive.mean()                 # central slab of the scene, where the primitives crowd


def test_point_tree_equals_brute_force_scan(engine, golden_inputs):
    """Nearest-point distance through the box tree (P_NEARTREE, tables > 256 points) is bit-identical to the scan
    over all points (P_NEAREST2/3): clustered, uniform and curve-like clouds, queries inside, near and far away;
    and it agrees with the float64 oracle."""
    from aegolius_amd import _prims
    rng = np.random.default_rng(12)
    helix = np.stack([np.cos(np.linspace(0, 40, 20000)), np.sin(np.linspace(0, 40, 20000)), np.linspace(-1, 1, 20000)])
    clouds = [rng.normal(0, 0.5, (3, 16384)), rng.uniform(-1, 1, (3, 3000)), helix,
              np.concatenate([rng.normal((1, 1, 0), 0.05, (700, 3)), rng.normal((-1, 0, 0.5), 0.2, (900, 3))]).T]
    co = np.concatenate([golden_inputs, golden_inputs * 8.0, rng.normal(0, 0.5, (3, 3000))], axis=1)
    co = co.astype(np.float32).astype(np.float64)
    for pts in clouds:
        for cls, sub in ((ns.geom_3d.PointCloud3D, pts), (ns.PointCloud2D, pts[:2])):
            tree = cls(sub).create(co)
            old = _prims.TREE_THRESHOLD
            try:
                _prims.TREE_THRESHOLD = 1 << 30
                scan = cls(sub).create(co)
            finally:
                _prims.TREE_THRESHOLD = old
            np.testing.assert_array_equal(tree, scan)
    ref = sdf_oracle.evaluate(ns.geom_3d.PointCloud3D(clouds[0]), co)
    check("point_cloud_tree", ns.geom_3d.PointCloud3D(clouds[0]).create(co), ref)


def test_inlined_sincos_accuracy(engine):
    """The kernels' own sincos (Cody-Waite + minimax polynomials, |x| <= 8192, ocml beyond) through a twist of the
    point (1, 0, z): sdf_x gives cos(pitch*z), sdf_y gives sin(pitch*z). Within 2.5e-7 absolute (about 2 ulp at 1)
    of float64 over small, moderate and huge angles, quadrant boundaries included."""
    angles = np.concatenate([np.linspace(-7.0, 7.0, 20001), np.linspace(-8300.0, 8300.0, 20001),
                             np.arange(-64, 65) * (np.pi / 4), np.linspace(-3.0e5, 3.0e5, 2001)])
    co = np.zeros((3, angles.size))
    co[0] = 1.0
    co[2] = angles.astype(np.float32)                 # pitch 1: the angle is the fp32 z itself
    exact = co[2].astype(np.float64)
    for fn, truth in ((ns.sdf_x, np.cos(exact)), (ns.sdf_y, np.sin(exact))):
        g = ns.GenericGeometry(fn, 0.0)
        g.twist(1.0)
        got = g.create(co).astype(np.float64)
        assert np.abs(got - truth).max() <= 2.5e-7, np.abs(got - truth).max()


def test_inlined_atan2_accuracy(engine):
    """The kernels' own atan2 through bend(R = 1, angle = 2 pi): sdf_x of the bent point is atan2(x, 1 - y).
    All octants, axes, tiny and huge magnitudes, signed zeros; within 1.5 ulp of numpy's float64 arctan2."""
    rng = np.random.default_rng(8)
    theta = np.concatenate([np.linspace(-3.1, 3.1, 40001), np.arange(-12, 13) * (np.pi / 8) * 0.999999])
    rad = np.concatenate([10.0 ** rng.uniform(-6, 4, 40001), np.ones(25)])
    x = (rad * np.sin(theta)).astype(np.float32)
    u = (rad * np.cos(theta)).astype(np.float32)          # u = 1 - y
    extra_x = np.array([0.0, 0.0, 1.0, -1.0, 0.0, -0.0, 3e-30, -3e-30], dtype=np.float32)
    extra_u = np.array([1.0, -1.0, 0.0, 0.0, 0.0, 2.0, 1e-30, 1e30], dtype=np.float32)
    x, u = np.concatenate([x, extra_x]), np.concatenate([u, extra_u])
    co = np.zeros((3, x.size))
    co[0] = x
    co[1] = (np.float32(1.0) - u).astype(np.float32)
    g = ns.GenericGeometry(ns.sdf_x, 0.0)
    g.bend(1.0, 2 * np.pi)
    got = g.create(co).astype(np.float64)
    # what the kernel is asked: atan2(x, -(y - 1)) with y as stored (fp32)
    uu = -(co[1].astype(np.float32) - np.float32(1.0)).astype(np.float64)
    want = np.arctan2(co[0], uu)
    keep = np.abs(want) < 3.14                                    # (at |phi| = pi the bend switches to its rigid tail)
    err = np.abs(got - want)[keep]
    ulp = np.spacing(np.maximum(np.abs(want[keep]), 1e-30).astype(np.float32)).astype(np.float64)
    assert (err / ulp).max() <= 1.5, (err / ulp).max()


def test_very_large_trees_run_without_a_compile(engine, golden_inputs):
    """A union of 300 spheres (899 instructions): the default mode evaluates it at once — in round 1 on the interpreter
    kernel (no multi-minute hiprtc build of 300 inlined children), since round 3 table-driven ("chain mode": one function
    per kind of child, built in under a second while the interpreter serves the first call) — and matches the oracle. A
    big tree that is NO n-ary chain (100 pairwise smooth unions: 99 sites, the mask kernels take the 64 widest) is served by
    the interpreter kernel while its specialised kernel builds."""
    import time
    rng = np.random.default_rng(0)
    objs = []
    for _ in range(300):
        s = ns.Sphere(float(rng.uniform(0.03, 0.1)))
        s.move(tuple(rng.uniform(-1.5, 1.5, 3)))
        objs.append(s)
    u = ns.CombineGeometry("UNION").combine(*objs)
    assert lower_geometry(u).code.shape[0] > 600
    t0 = time.perf_counter()
    got = u.create(golden_inputs)
    assert time.perf_counter() - t0 < 20.0
    check("union_of_300_spheres", got, sdf_oracle.evaluate(u, golden_inputs))
    engine.lib().sdfk_jit_drain()
    np.testing.assert_array_equal(u.create(golden_inputs), got)          # the chain kernel by now: the same bits
    acc = objs[0]
    for o in objs[1:100]:               # (nested pairwise: a few Python frames per level in the lowering, as in the reference)
        acc = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(acc, o, parameters=0.05)
    t0 = time.perf_counter()
    smooth = acc.create(golden_inputs)
    assert time.perf_counter() - t0 < 20.0
    check("smooth_union_of_100_spheres", smooth, sdf_oracle.evaluate(acc, golden_inputs))


def test_chains_of_thousands_of_members(engine):
    """16,384 spheres: one chain (round 3 stopped at 4096, when every member's value lived in LDS; with candidate lists
    per cell a workgroup holds 128 survivors per brick whatever the chain's size) — culled row blocks, the un-culled chain
    kernel and the interpreter agree bit for bit on a small grid, and with the oracle on a sample. 32,768 members is the
    limit (the lowering keeps 32,767 cull sites); a few members more still run as a chain whose first member is a pair."""
    from aegolius_amd import workloads
    tree = workloads.sphere_union(ns, 16384, seed=12, radius=0.02)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    assert prog.chain_members == 16384
    assert engine.Program.from_lowered(lower_geometry(workloads.sphere_union(ns, 4300, seed=12, radius=0.03))).chain_members == 4300
    co, _ = ns.generate_grid((2, 2, 2), (20, 24, 64))
    co32 = co.astype(np.float32)
    n = co32.shape[1]
    row_len = int(np.flatnonzero(co32[1] != co32[1][0])[0])
    plain = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_NOCULL)
    np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=row_len), plain)
    np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n, 0, engine.MODE_INTERPRET), plain)
    pick = np.random.default_rng(1).choice(n, 1500, replace=False)
    ref = sdf_oracle.evaluate(tree, co32[:, pick].astype(np.float64))
    err, bad = violations(plain[pick], ref)
    assert not bad.any(), float(np.nanmax(err))


def test_level_loop_build_of_a_left_deep_chain(engine):
    """The loop over surviving levels (a bitset walked with ctz: -DSDFK_CHAIN_LOOP_MIN=n) is off by default since round 4 —
    the level-after-level form is faster on every chain that reaches it — but stays a build switch: both forms of a 20-level
    smooth chain agree with the interpreter kernel bit for bit."""
    from aegolius_amd import workloads
    co, _ = ns.generate_grid((2, 2, 2), (20, 24, 64))
    co32 = co.astype(np.float32)
    n = co32.shape[1]
    row_len = int(np.flatnonzero(co32[1] != co32[1][0])[0])
    low = lower_geometry(workloads.cfg2_tree(ns, seed=21, count=21))
    lib = engine.lib()
    fields = []
    try:
        for defs in (b"", b"-DSDFK_CHAIN_LOOP_MIN=12"):
            lib.sdfk_debug_set_rtc_defs(defs)
            prog = engine.Program.from_lowered(low)
            fields.append(_device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=row_len))
    finally:
        lib.sdfk_debug_set_rtc_defs(b"")
    interp = _device_eval(engine, engine.Program.from_lowered(low), co32, n, n, 0, engine.MODE_INTERPRET)
    np.testing.assert_array_equal(fields[0], interp)
    np.testing.assert_array_equal(fields[1], interp)


def test_big_trees_that_are_no_chains_run_on_specialised_kernels(engine):
    """A left-deep smooth union of 120 primitives (359 instructions: beyond SDFK_BIG_PROGRAM, so built without the two
    quadratic LLVM passes; 119 sites: four mask words per brick) — row blocks with the row length, line bricks without
    it, the grid flavour, the flag-writing build and the interpreter kernel agree bit for bit, and with the oracle on a
    sample."""
    from aegolius_amd import workloads
    co, _ = ns.generate_grid((2, 2, 2), (24, 24, 64))
    co32 = co.astype(np.float32)
    n = co32.shape[1]
    row_len = int(np.flatnonzero(co32[1] != co32[1][0])[0])
    for count, instructions in ((120, 359),):                    # (tests/fuzz_big_trees.py covers up to 170 primitives, tools/big_tree_bench.py 400)
        tree = workloads.cfg2_tree(ns, seed=40 + count, count=count)
        low = lower_geometry(tree)
        assert low.code.shape[0] == instructions and len(low.cull_sites) == count - 1
        prog = engine.Program.from_lowered(low)
        assert prog.chain_members == 0
        interp = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_INTERPRET)
        np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=row_len), interp)
        if count == 120:                                          # (no row length: line bricks — one build of that kind is enough)
            np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED), interp)
        pick = np.random.default_rng(count).choice(n, 800, replace=False)
        err, bad = violations(interp[pick], sdf_oracle.evaluate(tree, co32[:, pick].astype(np.float64)))
        assert not bad.any(), float(np.nanmax(err))
        if count == 120:
            # the other builds of the row-block kernel with four mask words: the grid flavour (coordinates from the axis
            # tables) and the flag-writing one behind point_cloud
            keep = aegolius_amd.config.mode
            try:
                aegolius_amd.config.mode = engine.MODE_SPECIALIZED
                np.testing.assert_array_equal(tree.create(co), interp)
                cloud = tree.point_cloud(co)
            finally:
                aegolius_amd.config.mode = keep
            np.testing.assert_array_equal(cloud[:2], np.asarray(co)[:2, interp <= 0])


def test_flat_big_tree_with_three_mask_words(engine):
    """The flat (2-D) builds of the row-block kernel with more than 64 sites: a left-deep smooth union of 90 circles and
    rectangles (89 sites: three mask words per brick) on a 601 x 501 grid — the grid flavour behind create(generate_grid(...))
    and the array flavour with the row-length hint against the interpreter kernel, bit for bit."""
    import torch
    rng = np.random.default_rng(3)
    acc = None
    for k in range(90):
        o = ns.Circle(float(rng.uniform(0.05, 0.2))) if k % 2 else ns.Rectangle(float(rng.uniform(0.1, 0.4)), float(rng.uniform(0.1, 0.4)))
        o.rotate(float(rng.uniform(0, np.pi)), (0, 0, 1))
        o.move((float(rng.uniform(-0.9, 0.9)), float(rng.uniform(-0.9, 0.9)), 0))
        acc = o if acc is None else ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(acc, o, parameters=0.05)
    low = lower_geometry(acc)
    assert len(low.cull_sites) == 89
    co, res = ns.generate_grid((2, 2), (600, 500))
    keep = aegolius_amd.config.mode
    try:
        aegolius_amd.config.mode = engine.MODE_INTERPRET
        interp = acc.create(np.array(co))
        aegolius_amd.config.mode = engine.MODE_SPECIALIZED
        np.testing.assert_array_equal(acc.create(co), interp)                      # flat grid flavour
    finally:
        aegolius_amd.config.mode = keep
    prog = engine.Program.from_lowered(low)
    c32 = torch.from_numpy(np.ascontiguousarray(np.asarray(co), dtype=np.float32)).cuda()
    n = c32.shape[1]
    out = torch.empty(n, dtype=torch.float32, device="cuda")
    prog.eval_device(c32.data_ptr(), n, n, out.data_ptr(), mode=engine.MODE_SPECIALIZED, row_len=int(res[1]), flat=True)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), interp)                        # flat array flavour


def test_sharded_evaluation_of_trees_with_conv_operators(engine):
    """Slabs of whole planes with a recomputed halo (evaluate_slab_staged): conv_averaging (iterated, even and odd
    kernels, nested under other modifications) and conv_edge_detection give, slab by slab, exactly the whole-grid
    field — 3-D and 2-D grids, 'ranks' emulated on one device; signed / user code are refused with the reason."""
    import torch
    from aegolius_amd.distributed import _GpuSlabEvaluator, slab_bounds
    from aegolius_amd.cores.helper_functions import grid_axes

    def tree3(res):
        a = ns.Box(0.9, 0.7, 0.5)
        a.rotate(0.6, (1, 1, 0))
        a.conv_averaging((4, 3, 2), 2, res)
        a.rounding(0.02)
        b = ns.Torus(0.5, 0.15)
        b.conv_edge_detection(res)
        b.conv_averaging(3, 1, res)
        return ns.CombineGeometry("UNION2").combine(a, b)

    def tree2(res):
        c = ns.Circle(0.8)
        c.conv_averaging((5, 3), 3, res)
        c.onion(0.05)
        return c
    for build, size, resolution in ((tree3, (2, 2, 2), (26, 12, 10)), (tree2, (3, 3), (40, 24))):
        co, res = ns.generate_grid(size, resolution)
        whole = np.asarray(build(resolution).create(co)).ravel()
        axes = [a.astype(np.float32) for a in grid_axes(size, resolution)[0]]
        plane = axes[1].size * axes[2].size
        ev = _GpuSlabEvaluator(build(resolution))
        assert ev.staged
        for world in (2, 3, 5):
            parts = [ev(axes, *slab_bounds(whole.size, world, r, plane)) for r in range(world)]
            torch.cuda.synchronize()
            np.testing.assert_array_equal(torch.cat(parts).cpu().numpy(), whole)
    # `signed` crosses every slab: a single rank (no process group) holds the whole grid and must reproduce the plain
    # evaluation; the exchange between ranks is covered by test_gpu_consumers.test_signed_shards_across_ranks
    s = ns.Sphere(0.5)
    s.boundary()
    s.signed((8, 8, 8))
    co, _ = ns.generate_grid((2, 2, 2), (8, 8, 8))
    whole = np.asarray(s.create(co)).ravel()
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (8, 8, 8))[0]]
    ev = _GpuSlabEvaluator(s)
    assert ev.staged
    np.testing.assert_array_equal(ev(axes, 0, whole.size).cpu().numpy(), whole)


def test_library_and_torch_share_one_hip_runtime(engine):
    """A fresh interpreter that touches libsdfk.so BEFORE importing torch must still leave torch a working GPU
    (PyTorch-ROCm bundles its own libamdhip64: the engine loads that copy instead of a second runtime)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, faulthandler; sys.path.insert(0, %r)\n"
            "faulthandler.dump_traceback_later(90, exit=True)\n"      # a stall says where (stderr) instead of hanging the suite
            "from aegolius_amd import _engine\n"
            "_engine.require_gpu()\n"
            "print('engine up', flush=True)\n"
            "import torch\n"
            "assert torch.cuda.is_available()\n"
            "print(float(torch.ones(8, device='cuda').sum()), flush=True)\n"
            "faulthandler.cancel_dump_traceback_later()\n" % root)
    try:
        res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=240)
    except subprocess.TimeoutExpired as exc:                      # (seen once in round 4: the child was silent for 7 minutes)
        pytest.fail("the fresh interpreter did not finish: stdout %r stderr %r" % (exc.stdout, (exc.stderr or b"")[-2000:]))
    assert res.returncode == 0 and res.stdout.strip().endswith("8.0"), (res.stdout, res.stderr[-2000:])


def test_largest_baseline_grid_2049_cubed_on_one_gpu(engine):
    """BASELINE configs[4] at full size on ONE device: 2049^3 = 8.6e9 points (flat indices beyond 2^32, 103 GB of
    coordinates + 34 GB of field resident in HBM), 20-primitive three-level tree through the row-block kernel;
    40,000 sampled points against the oracle, plus the first and last points."""
    import torch
    from aegolius_amd.cores.helper_functions import grid_axes
    free, _total = torch.cuda.mem_get_info()
    n1d = 2049
    n = n1d ** 3
    if free < 16 * n + (8 << 30):
        pytest.skip("needs 146 GB of free HBM")
    tree = scenes.cfg5_tree(ns)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    axes = [a.astype(np.float32) for a in grid_axes((3, 3, 3), (2048,) * 3)[0]]
    co = torch.empty((3, n), dtype=torch.float32, device="cuda")
    out = torch.empty((n,), dtype=torch.float32, device="cuda")
    engine.grid_fill(co.data_ptr(), n, axes, 0, n)
    prog.eval_device(co.data_ptr(), n, n, out.data_ptr(), mode=engine.MODE_SPECIALIZED, row_len=n1d)
    torch.cuda.synchronize()
    rng = np.random.default_rng(2049)
    idx = np.unique(np.concatenate([rng.integers(0, n, 40000), [0, n - 1, n1d - 1, n1d, 2 ** 32 - 1, 2 ** 32, 2 ** 33 + 7]]))
    got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
    ix, rem = np.divmod(idx, n1d * n1d)
    iy, iz = np.divmod(rem, n1d)
    pts = np.stack([axes[0][ix], axes[1][iy], axes[2][iz]]).astype(np.float64)
    np.testing.assert_array_equal(co[:, torch.from_numpy(idx).cuda()].cpu().numpy(), pts.astype(np.float32))
    del co
    with np.errstate(all="ignore"):
        check("tree_cfg5_three_level", got, sdf_oracle.evaluate(tree, pts))


def test_flat_baseline_grid_16385_squared(engine):
    """BASELINE configs[3] at FULL size: the 50-primitive n-ary UNION on the 16385 x 16385 flat grid (268 M points),
    resident array + row hint + flat hint (the kernel built for flat grids) and straight from the axis tables;
    20,000 sampled points against the oracle, the two routes and the un-culled kernel agree on a window."""
    import torch
    from aegolius_amd.cores.helper_functions import grid_axes
    tree = scenes.cfg4_scene2d(ns)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    axes = [a.astype(np.float32) for a in grid_axes((10, 10), (16384, 16384))[0]]
    n1d = int(axes[0].size)
    n = n1d * n1d
    assert n1d == 16385 and axes[2].size == 1
    co = torch.empty((3, n), dtype=torch.float32, device="cuda")
    out = torch.empty((n,), dtype=torch.float32, device="cuda")
    engine.grid_fill(co.data_ptr(), n, axes, 0, n)
    prog.eval_device(co.data_ptr(), n, n, out.data_ptr(), mode=engine.MODE_SPECIALIZED, row_len=n1d, flat=True)
    torch.cuda.synchronize()
    rng = np.random.default_rng(16385)
    idx = np.unique(np.concatenate([rng.integers(0, n, 20000), [0, n - 1, n1d - 1, n1d]]))
    got = out[torch.from_numpy(idx).cuda()].cpu().numpy()
    ix, iy = np.divmod(idx, n1d)
    pts = np.stack([axes[0][ix], axes[1][iy], np.zeros(idx.size, dtype=np.float32)]).astype(np.float64)
    with np.errstate(all="ignore"):
        check("tree_cfg4_union50_2d", got, sdf_oracle.evaluate(tree, pts))
    # a window of whole rows: axis-table route and un-culled kernel
    r0, rows = 5000, 40
    w0, wn = r0 * n1d, rows * n1d
    win = torch.empty((wn,), dtype=torch.float32, device="cuda")
    prog.eval_grid(axes, w0, wn, win.data_ptr(), mode=engine.MODE_SPECIALIZED)
    torch.cuda.synchronize()
    assert torch.equal(win, out[w0:w0 + wn])
    prog.eval_device(co.data_ptr() + 4 * w0, wn, n, win.data_ptr(), mode=engine.MODE_NOCULL)
    torch.cuda.synchronize()
    assert torch.equal(win, out[w0:w0 + wn])


@pytest.mark.parametrize("shape", [(513, 513), (700, 333), (64, 4100)])
def test_two_row_coordinates_equal_the_three_row_path(shape, engine):
    """sdfk_eval_device_rows2d_xy — x and y rows only, z = 0 by contract (12 B/point) — against the three-row calls on the
    same x, y with a row of zeros: the chain-mode flat kernel (50-child union), a mask kernel (9-child union below the
    chain threshold), a program without cull sites (plain kernel), with and without the row hint; equal as floats (the
    sign of a zero may differ between the flat and the general transform: the existing rule for flat grids)."""
    import torch
    n0, n1 = shape
    n = n0 * n1
    stride = (n + 255) // 256 * 256
    rng = np.random.default_rng(n0 * 7 + n1)
    ax0 = np.linspace(-5, 5, n0).astype(np.float32)
    ax1 = np.linspace(-5, 5, n1).astype(np.float32)
    co = torch.zeros((3, stride), dtype=torch.float32, device="cuda")
    co[0, :n] = torch.from_numpy(np.repeat(ax0, n1)).cuda()
    co[1, :n] = torch.from_numpy(np.tile(ax1, n0)).cuda()
    small = ns.CombineGeometry("UNION").combine(*[_placed2d(rng) for _ in range(9)])
    single = ns.Circle(0.8)
    single.onion(0.1)
    single.rotate(0.4, (0, 0, 1))
    single.move((0.3, -0.2, 0))
    outs = [torch.empty((stride,), dtype=torch.float32, device="cuda") for _ in range(2)]
    for tree in (scenes.cfg4_scene2d(ns), small, single):
        prog = engine.Program.from_lowered(lower_geometry(tree))
        for row_len in (n1, None):
            prog.eval_device(co.data_ptr(), n, stride, outs[0].data_ptr(), mode=engine.MODE_SPECIALIZED, row_len=row_len,
                             flat=row_len is not None)
            outs[1].fill_(float("nan"))
            prog.eval_device_xy(co.data_ptr(), n, stride, outs[1].data_ptr(), mode=engine.MODE_SPECIALIZED, row_len=row_len)
            torch.cuda.synchronize()
            assert torch.equal(outs[0][:n], outs[1][:n]), (shape, row_len)
        # un-culled two-row kernel too
        prog.eval_device_xy(co.data_ptr(), n, stride, outs[1].data_ptr(), mode=engine.MODE_NOCULL)
        torch.cuda.synchronize()
        assert torch.equal(outs[0][:n], outs[1][:n])


def _placed2d(rng):
    o = [ns.Circle(0.4), ns.Rectangle(0.8, 0.5), ns.NGon(0.4, 6)][int(rng.integers(0, 3))]
    o.rotate(float(rng.uniform(0, np.pi)), (0, 0, 1))
    o.move((float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)), 0))
    return o


def test_apply_and_generic_geometry_variants(engine, golden_inputs):
    """EuclideanTransform.apply / apply_ec_transforms (reference cores/transformations.py:232-264) and the
    GenericGeometry2D / 3D classes evaluate like the object protocol they are part of."""
    b = ns.Box(0.4, 0.3, 0.2)
    b.rotate(0.7, (1, 2, 3))
    b.move((0.1, -0.2, 0.3))
    b.set_scale(1.3)
    want = b.create(golden_inputs)
    carrier = ns.Sphere(9.0)                            # only its transform matters
    carrier.rotate(0.7, (1, 2, 3))
    carrier.move((0.1, -0.2, 0.3))
    carrier.set_scale(1.3)
    np.testing.assert_array_equal(carrier.apply(ns.sdf_box, golden_inputs, ((0.4, 0.3, 0.2),)), want)
    got = ns.EuclideanTransform.apply_ec_transforms(ns.sdf_box, golden_inputs, ((0.4, 0.3, 0.2),), b.rotation_matrix,
                                                    b.center, b.scale)
    np.testing.assert_array_equal(got, want)
    g3 = ns.geom_3d.GenericGeometry3D(ns.sdf_box, (0.4, 0.3, 0.2))
    g3.rotate(0.7, (1, 2, 3))
    g3.move((0.1, -0.2, 0.3))
    g3.set_scale(1.3)
    np.testing.assert_array_equal(g3.create(golden_inputs), want)
    g2 = ns.geom_2d.GenericGeometry2D(ns.sdf_circle, 0.5)
    np.testing.assert_array_equal(g2.propagate(golden_inputs, "ignored"), ns.Circle(0.5).create(golden_inputs))


def test_instancing_tree_picks_the_same_instance_as_the_scan(engine, golden_inputs, monkeypatch):
    """curve_instancing with more than 256 instances goes through the box tree; except on exact ties between two
    centres it lands on the instance the scan finds — among centres at the same fp32 distance the lowest index — bit for bit."""
    from aegolius_amd import _prims
    for name in ("mod_curve_instancing_many", "mod_fully_aligned_curve_instancing_many"):
        name = [k for k in scenes.SCENES if k.endswith(name[4:])][0]
        low = lower_geometry(scenes.SCENES[name](ns))
        assert any(_ops_name(w) == "CURVEINSTT" for w in low.code[:, 0])
        tree = scenes.SCENES[name](ns).create(golden_inputs.copy())
        monkeypatch.setattr(_prims, "TREE_THRESHOLD", 10 ** 9)
        low_scan = lower_geometry(scenes.SCENES[name](ns))
        assert any(_ops_name(w) == "CURVEINST" for w in low_scan.code[:, 0])
        scan = scenes.SCENES[name](ns).create(golden_inputs.copy())
        monkeypatch.undo()
        np.testing.assert_array_equal(tree, scan)           # ties included: the lowest index wins in both


def _ops_name(word):
    from aegolius_amd import _ops
    return _ops.OPS[int(word) & 255].name if hasattr(_ops, "OPS") else str(int(word) & 255)


def test_concurrent_first_use_of_new_tree_shapes(engine, golden_inputs):
    """Several threads evaluate tree shapes nobody has built yet, in AUTO mode: the first calls are served by the
    interpreter kernel while ONE background worker runs hiprtc and modules are loaded as they become ready — kernel
    launches, memory copies, hiprtc and hipModuleLoadData all in flight together (the situation in which an early
    version of the per-flavour JIT was suspected to hang; builds and module loads are serialised since). Every call
    returns, and a later call of the same tree (specialised kernel by then) gives the same bits."""
    import threading
    import time
    co = np.ascontiguousarray(golden_inputs)
    trees = [scenes.random_tree(ns, 770001 + k, depth=3) for k in range(8)]
    first, errors = {}, []

    def worker(ids):
        try:
            for k in ids:
                for _ in range(3):                                  # interpreter first, the specialised kernel later
                    out = trees[k].create(co)
                    first.setdefault(k, out)
                    assert np.array_equal(out, first[k], equal_nan=True), "tree %d changed bits between calls" % k
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))
    aegolius_amd.config.mode = engine.MODE_AUTO
    threads = [threading.Thread(target=worker, args=(ids,), daemon=True)
               for ids in ([0, 1, 2, 3], [3, 2, 1, 0], [4, 5, 6, 7], [7, 6, 5, 4])]
    t0 = time.time()
    for t in threads:
        t.start()
    for t in threads:
        t.join(max(1.0, 240.0 - (time.time() - t0)))
    assert not any(t.is_alive() for t in threads), "a first-use call did not return within 240 s"
    assert not errors, errors
    engine.lib().sdfk_jit_drain()                                   # every background build has finished
    aegolius_amd.config.mode = engine.MODE_SPECIALIZED
    for k, tree in enumerate(trees):
        np.testing.assert_array_equal(tree.create(co), first[k])


_IMPORT_DURING_BUILD = """
import os, sys, time
sys.path.insert(0, {root!r})
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
co, _ = ns.generate_grid((3, 3, 3), (64, 64, 64))
tree = workloads.cfg5_tree(ns)
first = tree.create(co)                       # AUTO: interpreter kernel now, the specialised kernel builds beside us
assert "torch" not in sys.modules
import torch                                  # dlopen of a dozen HIP libraries while the build is in flight
_engine.lib().sdfk_jit_drain()
import numpy as np
assert np.array_equal(tree.create(co), first)
print("builds", _engine.jit_stats()[0])
"""


_IMPORT_DURING_WAITED_BUILD = """
import os, sys, threading, time
sys.path.insert(0, {root!r})
os.environ["SDFK_ASYNC_JIT"] = "0"            # the call WAITS for its kernel (ctypes releases the GIL meanwhile)
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
co, _ = ns.generate_grid((3, 3, 3), (64, 64, 64))
tree = workloads.cfg5_tree(ns)
ns.Sphere(0.3).create(co)                      # device context first
assert "torch" not in sys.modules
done = []
def importer():
    time.sleep(0.3)                            # the build of the 20-primitive tree takes a couple of seconds
    import torch                               # dlopen of a dozen HIP libraries on ANOTHER thread while the caller waits
    done.append(torch.__version__)
t = threading.Thread(target=importer)
t.start()
first = tree.create(co)
t.join()
import numpy as np
assert done and np.array_equal(tree.create(co), first)
print("builds", _engine.jit_stats()[0])
"""


def test_importing_a_hip_library_while_a_call_waits_for_its_build_does_not_deadlock(engine):
    """The same lock inversion with the build the caller WAITS for (SDFK_ASYNC_JIT=0, MODE_SPECIALIZED): ctypes releases the
    GIL during the call, so another Python thread can import a HIP library meanwhile. Waited-for builds run in the compiler
    child process too (in-process hiprtc only when the helper is missing, with a note on stderr)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDFK_CACHE_DIR="off")
    res = subprocess.run([sys.executable, "-c", _IMPORT_DURING_WAITED_BUILD.format(root=root)], env=env, capture_output=True,
                         text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "builds" in res.stdout and "building inside this process" not in res.stderr


def test_importing_a_hip_library_during_a_background_build_does_not_deadlock(engine, tmp_path):
    """The round-2 "hang inside ROCm", reproduced 6 times out of 6 in round 3 (profiles/r03_hang_import_during_build.txt):
    hiprtc inside the process holds comgr's global mutex while `import torch` on the main thread runs HIP fat-binary
    registration under the loader lock. Background builds now run in a child process; this is the reproducer."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDFK_CACHE_DIR="off")
    res = subprocess.run([sys.executable, "-c", _IMPORT_DURING_BUILD.format(root=root)], env=env, capture_output=True,
                         text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    assert "builds 1" in res.stdout


def _chain_trees():
    from aegolius_amd import workloads
    rng = np.random.default_rng(5)

    def boxes(n):
        objs = []
        for _ in range(n):
            o = ns.Box(*rng.uniform(0.8, 1.6, 3))
            o.rotate(float(rng.uniform(0, np.pi)), rng.normal(0, 1, 3))
            o.move(rng.uniform(-0.2, 0.2, 3))
            objs.append(o)
        return ns.CombineGeometry("INTERSECT").combine(*objs)

    def modified_union(n, rescale):
        u = workloads.sphere_union(ns, n, seed=9, radius=0.08)
        u.onion(0.01)                                             # value modifications of the result stay in the chain
        if rescale:
            u.rescale(1.3)                                        # pushed into the members by the lowering: still a chain
        return u

    def union_with_a_smooth_pair(n):
        rng = np.random.default_rng(9)
        objs = []
        for _ in range(n):
            o = ns.Sphere(float(rng.uniform(0.04, 0.12)))
            o.move(rng.uniform(-0.9, 0.9, 3))
            objs.append(o)
        # one member is a combination of its own (a cull site inside a member): not a chain — the mask kernels
        objs[n // 2] = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(objs[n // 2], ns.Box(0.2, 0.1, 0.1), parameters=0.1)
        return ns.CombineGeometry("UNION").combine(*objs)
    return {"union_150_spheres": (workloads.sphere_union(ns, 150), False, True), "intersect_80_boxes": (boxes(80), False, True),
            "union_130_mixed_2d": (workloads.cfg4_scene2d(ns, seed=3, count=130), True, True),
            "union_70_onion": (modified_union(70, False), False, True),
            "union_70_onion_rescaled": (modified_union(70, True), False, True),
            "union_70_with_a_smooth_pair": (union_with_a_smooth_pair(70), False, False)}


@pytest.mark.parametrize("name", ["union_150_spheres", "intersect_80_boxes", "union_130_mixed_2d", "union_70_onion",
                                  "union_70_onion_rescaled", "union_70_with_a_smooth_pair"])
def test_chain_mode_is_bit_exact(name, engine):
    """n-ary UNION / INTERSECT of more than 64 children run TABLE-DRIVEN (one function per kind of child, loops over
    parameter-offset tables, a list of surviving children per brick instead of mask bits): the culled row-block kernel,
    the un-culled chain kernel and the interpreter give the same bits on awkward shapes, from arrays and from axis
    tables, and agree with the oracle."""
    tree, flat, chain = _chain_trees()[name]
    low = lower_geometry(tree)
    assert len(low.cull_sites) > 64
    prog = engine.Program.from_lowered(low)
    # (the last tree is the fallback: more than 64 sites but no chain — the mask kernels take the 64 widest sites)
    assert ("#define SDFK_CHAIN 1" in prog.source()) == chain
    for shape, mis in ROW_SHAPES[:4]:
        if flat:
            co, _ = ns.generate_grid((10, 10), (shape[0] * shape[1] - 1, shape[2] - 1))
        else:
            co, _ = ns.generate_grid((2.2, 2.2, 2.2), tuple(r - 1 for r in shape))
        co32 = co.astype(np.float32)
        n = co32.shape[1]
        slower = co32[0] if flat else co32[1]
        row_len = int(np.flatnonzero(slower != slower[0])[0])
        stride = n + 5
        interp = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_INTERPRET)
        plain = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_NOCULL)
        rows = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_SPECIALIZED, row_len=row_len, flat=flat)
        norows = _device_eval(engine, prog, co32, n, stride, mis, engine.MODE_SPECIALIZED)      # no line-brick flavour: plain
        np.testing.assert_array_equal(plain, interp)
        np.testing.assert_array_equal(rows, plain)
        np.testing.assert_array_equal(norows, plain)
        axes = [a.astype(np.float32) for a in co.grid_axes]
        np.testing.assert_array_equal(prog.eval_grid_host(axes), plain)
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(tree, co32.astype(np.float64))
        err, bad = violations(rows, ref)
        assert not bad.any(), (name, shape, float(np.nanmax(err)))
    # scattered points under a row hint
    rng = np.random.default_rng(12)
    pts = rng.uniform(-1.2, 1.2, (3, 48 * 64)).astype(np.float32)
    if flat:
        pts[2] = 0.0
    plain = _device_eval(engine, prog, pts, pts.shape[1], pts.shape[1], 0, engine.MODE_NOCULL)
    rows = _device_eval(engine, prog, pts, pts.shape[1], pts.shape[1], 0, engine.MODE_SPECIALIZED, row_len=48)
    np.testing.assert_array_equal(rows, plain)


def _cells_stats(engine, enable=1):
    rec = (ctypes.c_longlong * 8)()
    engine.lib().sdfk_debug_cells_stats(enable, rec)
    keys = ("fine_cells", "coarse_cells", "pool_used", "pool_cap", "sum_lists", "longest", "without_list", "empty")
    return dict(zip(keys, [int(v) for v in rec]))


@pytest.mark.parametrize("members", [100, 300])
def test_candidate_lists_are_exact_and_used(members, engine, monkeypatch):
    """Chains of more than 64 members probe, per brick, the CANDIDATE LIST of the brick's cell (pre-pass sdfk_spec_cells:
    100 members — one level of cells; 300 — a coarse level first). Bit for bit the un-culled chain kernel on: a whole
    grid with the plane hint, x-slabs that start in the middle of a plane, the same points with the planes in a shuffled
    order and with jittered coordinates (bricks that are NOT inside the sphere their cell derived from its corner points:
    they must fall back to every member), a pool too small for the lists (cells without a list), and from axis tables.
    The statistics show that lists exist, are short, and that the starved pool leaves cells without one."""
    from aegolius_amd import workloads
    tree = workloads.sphere_union(ns, members, seed=21, radius=0.07)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    assert prog.chain_members == members
    co, _ = ns.generate_grid((2.1, 2.1, 2.1), (36, 70, 130))
    co32 = co.astype(np.float32)
    n0, n1, n2 = [int(a.size) for a in co.grid_axes]
    n = co32.shape[1]
    _cells_stats(engine, 1)
    try:
        plain = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_NOCULL)
        rows = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1)
        np.testing.assert_array_equal(rows, plain)
        st = _cells_stats(engine)
        assert st["fine_cells"] > 0 and st["without_list"] == 0 and st["pool_used"] <= st["pool_cap"], st
        assert (st["coarse_cells"] > 0) == (members >= 128), st
        filled = st["fine_cells"] - st["empty"]
        assert 0 < st["sum_lists"] / filled < 0.6 * members, st          # the lists are much shorter than the chain
        # no plane hint: cells of rows x points only
        np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=n2), plain)
        # x-slabs of whole rows that start inside a plane
        for first_row, rows_n in ((17, 5 * n1 + 3), (n1 - 1, 2 * n1 + 1), (3, n1 - 10)):
            a, b = first_row * n2, (first_row + rows_n) * n2
            part = _device_eval(engine, prog, co32[:, a:b], b - a, b - a, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1,
                                first_row_in_plane=first_row % n1)
            np.testing.assert_array_equal(part, plain[a:b])
        # planes in a shuffled order / jittered coordinates: rows still have one x and one y, but a cell's corner points no
        # longer bound it
        rng = np.random.default_rng(members)
        order = rng.permutation(n0)
        idx = (order[:, None] * (n1 * n2) + np.arange(n1 * n2)[None, :]).ravel()
        shuffled = _device_eval(engine, prog, co32[:, idx], n, n, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1)
        np.testing.assert_array_equal(shuffled, plain[idx])
        jit = co32 + rng.normal(0, 0.05, co32.shape).astype(np.float32)
        np.testing.assert_array_equal(_device_eval(engine, prog, jit, n, n, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1),
                                      _device_eval(engine, prog, jit, n, n, 0, engine.MODE_NOCULL))
        # axis tables
        np.testing.assert_array_equal(prog.eval_grid_host([a.astype(np.float32) for a in co.grid_axes]), plain)
        # a pool that cannot hold the lists: cells without one, same field
        monkeypatch.setenv("SDFK_CELLS_POOL", "4")
        _cells_stats(engine)
        starved = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1)
        np.testing.assert_array_equal(starved, plain)
        assert _cells_stats(engine)["without_list"] > 0
        monkeypatch.delenv("SDFK_CELLS_POOL")
    finally:
        _cells_stats(engine, 0)
    with np.errstate(all="ignore"):
        pick = np.random.default_rng(3).choice(n, 3000, replace=False)
        ref = sdf_oracle.evaluate(tree, co32[:, pick].astype(np.float64))
    err, bad = violations(plain[pick], ref)
    assert not bad.any(), float(np.nanmax(err))


def test_survivor_counts_at_the_edges_of_the_brick_list(engine):
    """How many members survive on a brick decides which path evaluates it: a list of up to SDFK_ALIST_CAP = 192 records in
    LDS, the cell's whole candidate list beyond that, every member where there is no list. Three clusters of nearly
    coincident spheres — 10, 192 and 193 members, far apart — make bricks with exactly those survivor counts (read out
    through the -DSDFK_DEBUG_NALIVE build of the kernel), and a fold over more than 64 candidates (several batches);
    the field of every path equals the un-culled chain kernel's bit for bit."""
    rng = np.random.default_rng(44)
    objs = []
    for centre, count in (((-0.7, -0.7, -0.6), 10), ((0.7, 0.6, -0.5), 192), ((0.0, -0.5, 0.7), 193)):
        for k in range(count):
            o = ns.Sphere(0.2 + 1e-6 * k)
            o.move(tuple(np.asarray(centre) + rng.uniform(-1e-6, 1e-6, 3)))
            objs.append(o)
    tree = ns.CombineGeometry("UNION").combine(*objs)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    assert prog.chain_members == 395
    co, _ = ns.generate_grid((2.4, 2.4, 2.4), (48, 64, 128))
    co32 = co.astype(np.float32)
    n0, n1, n2 = [int(a.size) for a in co.grid_axes]
    n = co32.shape[1]
    plain = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_NOCULL)
    rows = _device_eval(engine, prog, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1)
    np.testing.assert_array_equal(rows, plain)
    lib = engine.lib()
    lib.sdfk_debug_set_rtc_defs(b"-DSDFK_DEBUG_NALIVE=1")
    try:
        dbg = engine.Program.from_lowered(lower_geometry(tree))
        how = _device_eval(engine, dbg, co32, n, n, 0, engine.MODE_SPECIALIZED, row_len=n2, plane_rows=n1)
    finally:
        lib.sdfk_debug_set_rtc_defs(b"")
    seen = set(np.unique(how).astype(np.int64).tolist())
    assert -1 not in seen, "a brick without a list on a regular grid"
    assert 10 in seen and 192 in seen, sorted(seen)[:20]               # lists in LDS, the second one exactly full
    over = sorted(v - 100000 for v in seen if v >= 100000)
    assert over and min(over) >= 193, (over[:5], sorted(seen)[-5:])    # 193 survivors: the cell's whole list runs
    # bricks far from every cluster keep few members; none keeps zero
    assert min(v for v in seen if 0 <= v < 100000) >= 1


def _clustered_scene(kind="UNION", groups=8, members=25, seed=3, nested=True):
    """`groups` rigidly placed (some rescaled) clusters of `members` primitives each, every cluster an n-ary hard
    combination of its own; nested: two clusters are themselves combined first (three levels)."""
    rng = np.random.default_rng(seed)
    out = []
    for k in range(groups):
        objs = []
        for j in range(members):
            o = ns.Sphere(float(rng.uniform(0.03, 0.08))) if (j + k) % 3 else ns.Box(*(float(x) for x in rng.uniform(0.04, 0.12, 3)))
            if j % 5 == 0:
                o.rounding(0.01)
            o.move(rng.uniform(-0.25, 0.25, 3))
            objs.append(o)
        g = ns.CombineGeometry(kind).combine(*objs)
        g.rotate(float(rng.uniform(0, 3)), tuple(rng.normal(size=3)))
        g.move(rng.uniform(-0.7, 0.7, 3))
        if k % 3 == 0:
            g.rescale(1.25)
        out.append(g)
    if nested:
        pair = ns.CombineGeometry(kind + "2").combine(out[0], out[1])
        pair.move((0.1, -0.05, 0.02))
        out = [pair] + out[2:]
    scene = ns.CombineGeometry(kind).combine(*out)
    scene.rotate(0.3, (1, 0, 0))                    # the whole scene moved and rescaled: pushed into the members too
    scene.move((0.05, 0.0, -0.02))
    scene.rescale(0.9)
    return scene


@pytest.mark.parametrize("kind", ["UNION", "INTERSECT"])
def test_nested_hard_unions_are_flattened_into_one_chain(kind, engine, monkeypatch):
    """UNION(move(UNION(a, b, ...)), ...) lowers to ONE n-ary chain — every member behind the transforms of the groups
    it belonged to, the two affine maps composed in float64 into one — and runs on the chain kernels: all kernels of the
    flattened program agree bit for bit, the field is the oracle's within 1e-6 and the nested program's within fp32
    rounding (min / max are exact, a positive scale commutes with them; only the composed transform rounds differently)."""
    tree = _clustered_scene(kind)
    low = lower_geometry(tree)
    prog = engine.Program.from_lowered(low)
    assert prog.chain_members == 200
    monkeypatch.setenv("SDFK_NO_FLATTEN", "1")
    nested = engine.Program.from_lowered(lower_geometry(tree))
    monkeypatch.delenv("SDFK_NO_FLATTEN")
    assert nested.chain_members == 0
    for shape, mis in ROW_SHAPES[:3]:
        co, _ = ns.generate_grid((2.4, 2.4, 2.4), tuple(r - 1 for r in shape))
        co32 = co.astype(np.float32)
        n = co32.shape[1]
        row_len = int(np.flatnonzero(co32[1] != co32[1][0])[0])
        old = _device_eval(engine, nested, co32, n, n + 5, mis, engine.MODE_NOCULL)
        want = _device_eval(engine, prog, co32, n, n + 5, mis, engine.MODE_NOCULL)
        for mode, hint in ((engine.MODE_SPECIALIZED, row_len), (engine.MODE_SPECIALIZED, None), (engine.MODE_INTERPRET, None)):
            np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n + 5, mis, mode, row_len=hint), want)
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(tree, co32.astype(np.float64))
        for field in (want, old):
            err, bad = violations(field, ref)
            assert not bad.any(), (kind, shape, float(np.nanmax(err)))
        assert float(np.max(np.abs(want - old))) < 2e-6


def _perforated_plate(holes=120, seed=5, moved=True):
    rng = np.random.default_rng(seed)
    plate = ns.Box(1.8, 1.6, 0.3)
    plate.rounding(0.02)
    cyl = []
    for k in range(holes):
        c = ns.Cylinder(float(rng.uniform(0.03, 0.07)), 1.0) if k % 4 else ns.Sphere(float(rng.uniform(0.05, 0.12)))
        c.move((float(rng.uniform(-.85, .85)), float(rng.uniform(-.75, .75)), 0.0 if k % 4 else float(rng.uniform(-.1, .1))))
        cyl.append(c)
    u = ns.CombineGeometry("UNION").combine(*cyl)
    if moved:
        u.move((0.01, -0.02, 0.0))
    tree = ns.CombineGeometry("SUBTRACT2").combine(plate, u)
    if moved:
        tree.rotate(0.25, (1, 0.2, 0))
        tree.rescale(1.1)
    return tree


@pytest.mark.parametrize("moved", [False, True])
def test_body_minus_a_large_union_runs_as_one_chain(moved, engine, monkeypatch):
    """SUBTRACT2(body, UNION(holes...)) = max(body, max_j -hole_j): lowered as ONE n-ary INTERSECT of the body and the
    negated holes (negation is exact), also when union and result carry transforms — chain kernels instead of a chain
    inside an operand. All kernels of the rewritten program agree bit for bit; oracle within 1e-6; the nested program
    (SDFK_NO_FLATTEN) within fp32 rounding (equal bits when no transform had to be composed)."""
    tree = _perforated_plate(moved=moved)
    prog = engine.Program.from_lowered(lower_geometry(tree))
    assert prog.chain_members == 121
    monkeypatch.setenv("SDFK_NO_FLATTEN", "1")
    nested = engine.Program.from_lowered(lower_geometry(tree))
    monkeypatch.delenv("SDFK_NO_FLATTEN")
    # (not rewritten, the union is still a chain of its own with plate and subtraction as the rest of the program — unless
    #  its members read moved coordinates)
    assert nested.chain_members == (0 if moved else 120)
    for shape, mis in ROW_SHAPES[:3]:
        co, _ = ns.generate_grid((2.4, 2.4, 1.0), tuple(r - 1 for r in shape))
        co32 = co.astype(np.float32)
        n = co32.shape[1]
        row_len = int(np.flatnonzero(co32[1] != co32[1][0])[0])
        old = _device_eval(engine, nested, co32, n, n + 5, mis, engine.MODE_NOCULL)
        want = _device_eval(engine, prog, co32, n, n + 5, mis, engine.MODE_NOCULL)
        for mode, hint in ((engine.MODE_SPECIALIZED, row_len), (engine.MODE_SPECIALIZED, None), (engine.MODE_INTERPRET, None)):
            np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n + 5, mis, mode, row_len=hint), want)
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(tree, co32.astype(np.float64))
        for field in (want, old):
            err, bad = violations(field, ref)
            assert not bad.any(), (moved, shape, float(np.nanmax(err)))
        if moved:
            assert float(np.max(np.abs(want - old))) < 2e-6
        else:
            np.testing.assert_array_equal(want, old)


def _rest_scenes():
    from aegolius_amd import workloads

    def union(n=100, seed=4):
        return workloads.sphere_union(ns, n, seed=seed, radius=0.08)
    out = {}
    out["clipped"] = ns.CombineGeometry("INTERSECT2").combine(union(), ns.Box(1.5, 1.5, 1.5))
    ground = ns.Box(2.0, 2.0, 0.2)
    ground.move((0, 0, -0.8))
    out["blended_with_ground"] = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(ground, union(), parameters=0.1)
    body = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(ns.Box(1.6, 1.6, 1.6), ns.Sphere(1.0), parameters=0.1)
    out["composite_body_minus_union"] = ns.CombineGeometry("SUBTRACT2").combine(body, union())
    t = ns.CombineGeometry("INTERSECT2").combine(union(), ns.Sphere(0.9))
    t.onion(0.01)
    out["clipped_onion"] = t
    u = union()
    u.move((0.05, 0, 0))
    t = ns.CombineGeometry("INTERSECT2").combine(u, ns.Box(1.5, 1.5, 1.5))
    t.rotate(0.3, (0, 1, 0))
    t.move((0.1, 0, 0))
    t.rescale(1.1)
    out["clipped_and_placed"] = t
    ground = ns.Box(2.0, 2.0, 0.2)
    ground.move((0, 0, -0.8))
    t = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(ground, union(), parameters=0.1)
    t.rotate(0.2, (1, 0, 0))
    t.onion(0.02)
    out["blended_placed_onion"] = t
    two = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(union(60, 4), union(30, 9), parameters=0.05)   # the larger one is the chain
    out["two_unions_blended"] = two
    return out


@pytest.mark.parametrize("name", ["clipped", "blended_with_ground", "composite_body_minus_union", "clipped_onion",
                                  "clipped_and_placed", "blended_placed_onion", "two_unions_blended"])
def test_chain_inside_a_larger_program(name, engine):
    """A large hard union that is an OPERAND — clipped by a box, blended with a ground plane, subtracted from a composite
    body, with value modifications on top, placed as a whole: the union runs as a chain (per-brick survivor lists), the rest
    of the program per point around its value (sdfk_chain_tail). All kernels agree bit for bit — the survivor list is
    exact whatever is done with the chain's value afterwards — and with the oracle within 1e-6."""
    tree = _rest_scenes()[name]
    low = lower_geometry(tree)
    prog = engine.Program.from_lowered(low)
    assert prog.chain_members == (60 if name == "two_unions_blended" else 100)
    for shape, mis in ROW_SHAPES[:3]:
        co, _ = ns.generate_grid((2.2, 2.2, 2.2), tuple(r - 1 for r in shape))
        co32 = co.astype(np.float32)
        n = co32.shape[1]
        row_len = int(np.flatnonzero(co32[1] != co32[1][0])[0])
        want = _device_eval(engine, prog, co32, n, n + 5, mis, engine.MODE_NOCULL)
        for mode, hint in ((engine.MODE_SPECIALIZED, row_len), (engine.MODE_SPECIALIZED, None), (engine.MODE_INTERPRET, None)):
            np.testing.assert_array_equal(_device_eval(engine, prog, co32, n, n + 5, mis, mode, row_len=hint), want)
        axes = [a.astype(np.float32) for a in co.grid_axes]
        np.testing.assert_array_equal(prog.eval_grid_host(axes), want)
        np.testing.assert_array_equal(prog.select_grid(axes, 0.0), np.flatnonzero(want <= 0))
        with np.errstate(all="ignore"):
            ref = sdf_oracle.evaluate(tree, co32.astype(np.float64))
        err, bad = violations(want, ref)
        assert not bad.any(), (name, shape, float(np.nanmax(err)))


def test_staged_operator_inside_a_flattened_union(engine, monkeypatch):
    """A grid-neighbourhood operator (staged evaluation: its position in the tree is the key of its stage) on ONE member
    of a moved 20-member union: the members are re-framed by the lowering, the stages still find their operator."""
    res = (32, 32, 32)
    co, _ = ns.generate_grid((2, 2, 2), res)

    def build():
        rng = np.random.default_rng(8)
        objs = []
        for k in range(20):
            o = ns.Sphere(float(rng.uniform(0.1, 0.2)))
            o.move(rng.uniform(-0.6, 0.6, 3))
            objs.append(o)
        blurred = ns.Box(0.5, 0.4, 0.3)
        blurred.conv_averaging((3, 3, 3), 1, res)
        objs[5] = blurred
        u = ns.CombineGeometry("UNION").combine(*objs)
        u.move((0.05, -0.02, 0.0))
        u.rescale(1.1)
        return u
    got = build().create(co)
    monkeypatch.setenv("SDFK_NO_FLATTEN", "1")
    nested = build().create(co)
    monkeypatch.delenv("SDFK_NO_FLATTEN")
    with np.errstate(all="ignore"):
        ref = sdf_oracle.evaluate(build(), np.asarray(co).astype(np.float32).astype(np.float64))
    for field in (got, nested):
        err, bad = violations(field.ravel(), ref.ravel())
        assert not bad.any(), float(np.nanmax(err))


_FORCED_CHAIN = """
import sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import numpy as np
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
from aegolius_amd._lower import lower_geometry
import test_gpu_parity as T
for tree, size, req, flat in [(workloads.cfg4_scene2d(ns, seed=7, count=c), (10, 10), (140, 75), True) for c in (9, 20, 50, 63, 64, 65)] + \
                             [(workloads.sphere_union(ns, c), (2, 2, 2), (20, 33, 64), False) for c in (8, 33, 64, 65)]:
    low = lower_geometry(tree)
    prog = _engine.Program.from_lowered(low)
    assert "#define SDFK_CHAIN 1" in prog.source()
    co, _ = ns.generate_grid(size, req)
    co32 = co.astype(np.float32)
    n = co32.shape[1]
    slower = co32[0] if flat else co32[1]
    row_len = int(np.flatnonzero(slower != slower[0])[0])
    interp = T._device_eval(_engine, prog, co32, n, n + 3, 0, _engine.MODE_INTERPRET)
    plain = T._device_eval(_engine, prog, co32, n, n + 3, 0, _engine.MODE_NOCULL)
    rows = T._device_eval(_engine, prog, co32, n, n + 3, 1, _engine.MODE_SPECIALIZED, row_len=row_len, flat=flat)
    assert np.array_equal(plain, interp) and np.array_equal(rows, plain), (len(low.cull_sites) + 1, flat)
print("ok")
"""


def test_chain_mode_with_few_children(engine):
    """Chain mode forced onto short chains (SDFK_CHAIN_MIN=8: 9 to 65 children, one mask word and two): the case in
    which a word written by one lane and read back by all without a barrier gave stale lists and a memory fault."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDFK_CHAIN_MIN="8")
    res = subprocess.run([sys.executable, "-c", _FORCED_CHAIN.format(root=root)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0 and res.stdout.strip().endswith("ok"), res.stderr[-3000:]
