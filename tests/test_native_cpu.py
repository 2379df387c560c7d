"""CPU (no GPU needed): the C-ABI library builds, loads, exports every symbol of include/sdfk.h,
validates programs, generates + hiprtc-compiles specialised kernels for gfx950, and the lowering
keeps the reference's aliasing rules. No compute calls."""
import os
import re
import sys

import numpy as np
import pytest

import scenes
import aegolius_amd.cores as ns
from aegolius_amd import _ops
from aegolius_amd._lower import lower_geometry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(built):
    header = open(os.path.join(ROOT, "include", "sdfk.h")).read()
    declared = set(re.findall(r"\b(sdfk_[a-z0-9_]+)\s*\(", header))
    assert declared == set(built.SIGNATURES), declared ^ set(built.SIGNATURES)
    lib = built.lib()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.sdfk_abi_version() == 1
    assert built.device_count() >= 0


def test_opcode_table_is_consistent(built):
    assert len(_ops.OPS) == len(set(o.name for o in _ops.OPS)) >= 80
    hdr = open(os.path.join(ROOT, "aegolius_amd", "csrc", "sdfk_device.h")).read()
    for o in _ops.OPS:
        assert re.search(r"\b%s\(" % o.func, hdr), o.func


def test_linspace_matches_numpy(built):
    for lo, hi, n in ((-1.0, 1.0, 129), (-2.0, 2.0, 1025), (-5.0, 5.0, 16385), (-1.5, 1.5, 2), (0.3, 0.3, 7),
                      (-0.7, 0.9, 1)):
        np.testing.assert_array_equal(built.linspace_f32(lo, hi, n), np.linspace(lo, hi, n).astype(np.float32))


@pytest.mark.parametrize("name", sorted(scenes.SCENES))
def test_every_scene_lowers_to_a_valid_program(name, built):
    obj = scenes.SCENES[name](ns)
    if name.startswith("host_"):                    # opaque user code: a stage plan instead of one program
        from aegolius_amd._eval import _plan_stages
        stages, low, _ = _plan_stages(lambda **kw: lower_geometry(obj, **kw))
        assert stages and any(_ops.OPS[w & 255].name == "V_FIELD" for w in low.code[:, 0])
        for stage in (st[0] for st in stages):
            if stage is not None:
                assert built.Program(stage.code, stage.params, stage.tables, stage.result_reg).handle
    else:
        low = lower_geometry(obj)
    assert low.fits_interpreter
    prog = built.Program(low.code, low.params, low.tables, low.result_reg)   # C++ validation
    assert prog.handle


@pytest.mark.parametrize("name", ["tree_cfg2_smooth_union10", "tree_cfg3_mod_chain", "tree_cfg4_union50_2d",
                                  "tree_cfg5_three_level", "prim_polygon_concave", "mod_fully_aligned_curve_instancing",
                                  "alias_rotsym_recover", "prim_quad", "tree_deep_right", "tree_pawn_3D",
                                  "combine_SMOOTH_SUBTRACT2", "combine_INTERSECT_nary", "combine_modified_result"])
def test_specialised_kernel_compiles_for_gfx950(name, built):
    low = lower_geometry(scenes.SCENES[name](ns))
    prog = built.Program.from_lowered(low)          # with the brick-culling kernel when the tree has sites
    src = prog.source()
    assert "sdfk_spec_v4" in src and "sdfk_point" in src
    assert prog.compile_check() > 1000


def _ops_of(low):
    return [_ops.OPS[w & 255].name for w in low.code[:, 0]]


def test_identity_transforms_cost_nothing():
    """Combine nodes carry an identity transform: (I·co)/1 - 0 is exact, so no instruction is emitted."""
    low = lower_geometry(scenes.cfg2_tree(ns))
    names = _ops_of(low)
    assert names.count("XFORM") == 10 and names.count("SMIN3") == 9 and "MOVC" not in names
    assert len(names) == 29 and low.n_vreg == 2


def test_topology_key_ignores_parameter_values():
    a = lower_geometry(scenes.cfg2_tree(ns, seed=1))
    b = lower_geometry(scenes.cfg2_tree(ns, seed=2))
    assert a.code.tobytes() == b.code.tobytes() and a.params.tobytes() != b.params.tobytes()


def test_in_place_modifications_write_through_only_where_the_reference_aliases():
    # symmetry below a displacement: in place on the SAME register the displacement function reads
    s = ns.Sphere(0.5)
    s.symmetry(0)
    s.displacement(ns.sdf_x, (0.0,))
    low = lower_geometry(s)
    sym = [w for w in low.code[:, 0] if _ops.OPS[w & 255].name == "SYMMETRY"][0]
    axis = [w for w in low.code[:, 0] if _ops.OPS[w & 255].name == "P_AXIS"][0]
    assert (sym >> 8) & 255 == (sym >> 16) & 255 == (axis >> 16) & 255
    # symmetry inside a combine child: the child owns a private copy, the sibling reads the original
    a = ns.Sphere(0.3)
    a.symmetry(0)
    u = ns.CombineGeometry("UNION2").combine(a, ns.Box(1, 1, 1))
    low = lower_geometry(u)
    sym = [w for w in low.code[:, 0] if _ops.OPS[w & 255].name == "SYMMETRY"][0]
    box = [w for w in low.code[:, 0] if _ops.OPS[w & 255].name == "P_BOX"][0]
    assert (sym >> 8) & 255 != (sym >> 16) & 255          # wrote a fresh register
    assert (box >> 16) & 255 == (sym >> 16) & 255          # sibling still reads the source register


def test_cull_sites_are_validated(built):
    low = lower_geometry(scenes.cfg2_tree(ns))
    assert low.cull_sites.shape == (9, 5) and np.all(low.cull_k >= 2.0 - 1e-6)
    prog = built.Program.from_lowered(low)
    src = prog.source()
    assert "sdfk_probe" in src and "sdfk_point_culled" in src and "sdfk_spec_t" in src
    assert prog.compile_check() > 1000
    bad = low.cull_sites.copy()
    bad[0, 2] += 1                                           # a-range no longer adjacent to the b-range
    with pytest.raises(built.SdfkError):
        built.Program(low.code, low.params, low.tables, low.result_reg, bad, low.cull_k)
    bad = low.cull_sites.copy()
    bad[0, 0] -= 1                                           # not a combiner
    with pytest.raises(built.SdfkError):
        built.Program(low.code, low.params, low.tables, low.result_reg, bad, low.cull_k)
    with pytest.raises(built.SdfkError):
        built.Program(low.code, low.params, low.tables, low.result_reg, low.cull_sites, low.cull_k * np.inf)
    # no Lipschitz bound -> no sites: twist / repetition / sign below a union
    a = ns.Box(0.5, 0.3, 0.2)
    a.twist(1.0)
    u = ns.CombineGeometry("UNION2").combine(a, ns.Sphere(0.3))
    assert len(lower_geometry(u).cull_sites) == 0


def test_point_tree_layout_and_validation(built):
    """P_NEARTREE: every point sits in exactly one leaf of the three-level tree, boxes are the exact bounds, indices stay
    inside the table;
    the library refuses a tree whose indices leave it."""
    from aegolius_amd import _prims
    rng = np.random.default_rng(4)
    pts = rng.normal(0, 1, (5000, 3)).astype(np.float32)
    table, n_root = _prims.build_point_tree(pts)
    seen = []
    assert n_root == 1                                          # 5000 points: 157 .. 313 leaves, <= 10 middle boxes
    for r in range(n_root):
        rlo, rhi, mfirst, nm = table[8 * r:8 * r + 3], table[8 * r + 3:8 * r + 6], int(table[8 * r + 6]), int(table[8 * r + 7])
        assert 1 <= nm <= _prims.TREE_LEAF
        for m in range(nm):
            mid = table[mfirst + 8 * m: mfirst + 8 * m + 8]
            lo, hi, first, nl = mid[0:3], mid[3:6], int(mid[6]), int(mid[7])
            assert 1 <= nl <= _prims.TREE_LEAF and np.all(lo >= rlo) and np.all(hi <= rhi)
            for l in range(nl):
                row = table[first + 8 * l: first + 8 * l + 8]
                pf, pn = int(row[6]), int(row[7])
                assert 1 <= pn <= _prims.TREE_LEAF
                p = table[pf:pf + 3 * pn].reshape(-1, 3)
                np.testing.assert_array_equal(p.min(axis=0), row[0:3])
                np.testing.assert_array_equal(p.max(axis=0), row[3:6])
                assert np.all(row[0:3] >= lo) and np.all(row[3:6] <= hi)
                seen.append(p)
    seen = np.concatenate(seen)
    assert seen.shape == pts.shape
    np.testing.assert_array_equal(np.sort(seen.view("f4,f4,f4").ravel()), np.sort(pts.view("f4,f4,f4").ravel()))
    # the leaf order of the points (what CURVEINSTT maps back to instance rows) is a permutation that reproduces the table
    table2, n_root2, point_base, order = _prims.build_point_tree(pts, with_order=True)
    np.testing.assert_array_equal(table2, table)
    assert n_root2 == n_root and np.array_equal(np.sort(order), np.arange(pts.shape[0]))
    np.testing.assert_array_equal(table[point_base:].reshape(-1, 3), pts[order])
    # degenerate inputs: one point; all points equal (every split is a tie); the argument checks of the C entry point
    one, n1 = _prims.build_point_tree(pts[:1])
    assert n1 == 1 and one.size == 8 + 8 + 8 + 3 and one[7] == 1 and one[15] == 1 and one[23] == 1
    same, ns_ = _prims.build_point_tree(np.repeat(pts[:1], 1000, axis=0))
    assert ns_ == 1 and same[8 * ns_ + 7] >= 1
    big, nb = _prims.build_point_tree(rng.normal(0, 1, (70000, 3)).astype(np.float32))
    assert 3 <= nb <= 5                                         # 2188 .. 4375 leaves -> 69 .. 137 middle boxes -> 3 .. 5 roots
    n_mid = int(sum(big[8 * r + 7] for r in range(nb)))
    n_leaf = int(sum(big[8 * nb + 8 * m + 7] for m in range(n_mid)))
    assert big.size == 8 * (nb + n_mid + n_leaf) + 3 * 70000
    assert int(sum(big[8 * (nb + n_mid) + 8 * l + 7] for l in range(n_leaf))) == 70000
    with pytest.raises(ValueError):
        built.point_tree(np.zeros((0, 3), dtype=np.float32), 32)
    with pytest.raises(built.SdfkError):
        built.point_tree(pts, 1)
    # through the public classes: large clouds lower to the tree, small ones to the scan
    names = lambda obj: [_ops.OPS[w & 255].name for w in lower_geometry(obj).code[:, 0]]      # noqa: E731
    assert "P_NEARTREE" in names(ns.geom_3d.PointCloud3D(pts.T.astype(np.float64)))
    assert "P_NEAREST3" in names(ns.geom_3d.PointCloud3D(pts[:100].T.astype(np.float64)))
    assert "P_NEARTREE" in names(ns.PointCloud2D(pts[:, :2].T.astype(np.float64)))
    low = lower_geometry(ns.geom_3d.PointCloud3D(pts.T.astype(np.float64)))
    assert built.Program(low.code, low.params, low.tables, low.result_reg).handle
    bad = low.tables.copy()
    bad[int(low.params[1]) + 6] = 1e7                       # first middle box of root box 0 far outside
    with pytest.raises(built.SdfkError):
        built.Program(low.code, low.params, bad, low.result_reg)


def test_header_is_valid_c_and_links(built, tmp_path):
    """include/sdfk.h compiles as C99 with -Wall -Werror and a plain-C program drives the library through it
    (create / source / hiprtc compile check / error reporting / linspace) — no Python, no C++ in the consumer."""
    import shutil
    import subprocess
    gcc = shutil.which("gcc")
    assert gcc, "gcc is part of the image"
    exe = tmp_path / "cabi_smoke"
    libdir = os.path.join(ROOT, "aegolius_amd")
    cmd = [gcc, "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cabi", "cabi_smoke.c"), "-o", str(exe), "-L", libdir, "-lsdfk",
           "-Wl,-rpath," + libdir, "-Wl,-rpath-link,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    env = dict(os.environ, LD_LIBRARY_PATH=libdir + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    run = subprocess.run([str(exe), str(_ops.BY_NAME["P_SPHERE"].code)], capture_output=True, text=True, env=env, timeout=300)
    assert run.returncode == 0 and "cabi ok" in run.stdout, (run.returncode, run.stdout, run.stderr)


def test_folded_affine_chains_keep_the_lipschitz_bound():
    """Consecutive in-place affine maps are merged into one instruction; the Lipschitz bound of the register must be
    the bound of the merged map, not the first map's factor applied twice (found by tests/fuzz_random_trees.py, seed
    9002 depth 4: a bound that is too small lets the culling kernels skip an operand they must evaluate)."""
    from aegolius_amd._lower import Lowerer

    def xform(k, shift):                      # q = k p - shift  (M = k I)
        return list((np.eye(3) * k).ravel()) + list(shift)
    L = Lowerer()
    L.emit("XFORM", 1, 0, params=xform(0.5, (0.1, 0.0, 0.0)))
    assert np.isclose(L.lip_c[1], 0.5)
    L.emit("XFORM", 1, 1, params=xform(0.5, (0.0, 0.2, 0.0)))          # merged into the first
    L.emit("CSCALE", 1, 1, params=[0.8])                                # and again
    L.emit("XLATE", 1, 1, params=[0.3, 0.0, 0.0])
    assert len(L.code) == 1 and np.isclose(L.lip_c[1], 0.5 * 0.5 * 0.8)
    np.testing.assert_allclose(np.asarray(L.params[:9]).reshape(3, 3), np.eye(3) * 0.2)
    # and through the public API: two nested contractions under a union give K = L_a + L_b of the true fields
    a = ns.Sphere(0.3)
    a.scale_sdf(0.5)                            # value scale 0.5, coordinates scaled by 2
    a.set_scale(0.25)                           # node: coordinates / 0.25, value * 0.25
    b = ns.Box(0.2, 0.2, 0.2)
    low = lower_geometry(ns.CombineGeometry("UNION2").combine(a, b))
    assert low.cull_sites.shape[0] == 1 and np.isclose(low.cull_k[0], 2.0, atol=1e-6)


def test_no_gpu_means_an_error_never_a_cpu_result(built):
    """The product path has no CPU evaluation: without an MI355X every entry point of the path raises."""
    if built.device_count() > 0:
        pytest.skip("a GPU is visible")
    import aegolius_amd.cores as ns
    co, res = ns.generate_grid((2, 2, 2), (4, 4, 4))
    s = ns.Sphere(0.5)
    for call in (lambda: s.create(co), lambda: s.propagate(co), lambda: s.create_resident(co), lambda: s.point_cloud(co),
                 lambda: ns.from_sdf(np.zeros(125), res), lambda: ns.sdf_sphere(np.asarray(co), 0.5)):
        with pytest.raises(built.SdfkError, match="no HIP device|no CPU path"):
            call()


_CACHE_SCRIPT = """
import json, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + "/tests")
import scenes
import aegolius_amd.cores as ns
from aegolius_amd import _engine
from aegolius_amd._lower import lower_geometry
prog = _engine.Program.from_lowered(lower_geometry(scenes.cfg2_tree(ns)))
size, seconds = prog.compile_flavour(_engine.FLAVOUR_ROWS_ARRAY)
again = prog.compile_flavour(_engine.FLAVOUR_ROWS_ARRAY)
other = _engine.Program.from_lowered(lower_geometry(scenes.cfg2_tree(ns))).compile_flavour(_engine.FLAVOUR_ROWS_ARRAY)
print(json.dumps(dict(size=size, seconds=seconds, again=again[1], other=other[1], builds=_engine.jit_stats()[0])))
"""


def test_code_objects_are_built_once_per_process_and_cached_on_disk(built, tmp_path):
    """One flavour of one tree shape: built by hiprtc once per process (a second program of the same shape and a second
    request find the code object), written to the on-disk cache (on by default, here pointed at a scratch directory),
    loaded from there by the next process without any hiprtc build; SDFK_CACHE_DIR=off writes nothing."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _CACHE_SCRIPT.format(root=root)

    def run(cache):
        env = dict(os.environ, SDFK_CACHE_DIR=cache)
        res = subprocess.run([sys.executable, "-c", script], env=env, capture_output=True, text=True, check=True)
        return json.loads(res.stdout.strip().splitlines()[-1])
    cold = run(str(tmp_path))
    files = list(tmp_path.iterdir())
    assert cold["builds"] == 1 and cold["again"] < 0.05 and cold["other"] < 0.05
    # the file is the code object + a 24-byte trailer (magic, length, checksum)
    assert len(files) == 1 and files[0].suffix == ".co" and files[0].stat().st_size == cold["size"] + 24
    warm = run(str(tmp_path))
    assert warm["builds"] == 0 and warm["size"] == cold["size"] and warm["seconds"] < 0.05
    assert len(list(tmp_path.iterdir())) == 1
    off = run("off")
    assert off["builds"] == 1 and len(list(tmp_path.iterdir())) == 1
    # a damaged file (truncated by a full disk, a flipped byte) is never handed to the loader: it is deleted, the
    # kernel is rebuilt and the cache rewritten
    blob = files[0].read_bytes()
    for damaged in (blob[:len(blob) // 2], blob[:100] + bytes([blob[100] ^ 0xFF]) + blob[101:], b"", blob[:-24]):
        files[0].write_bytes(damaged)
        again = run(str(tmp_path))
        assert again["builds"] == 1 and again["size"] == cold["size"]
        assert files[0].read_bytes() == blob


def test_on_disk_cache_is_size_capped(built, tmp_path):
    """SDFK_CACHE_MAX_MB bounds the directory: the oldest code objects go when a new one is written."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for k in range(6):                                             # 6 x 300 KB of stale entries, oldest first
        f = tmp_path / ("sdfk-%016x-1.co" % k)
        f.write_bytes(b"x" * 300_000)
        os.utime(f, (1_000_000 + k, 1_000_000 + k))
    env = dict(os.environ, SDFK_CACHE_DIR=str(tmp_path), SDFK_CACHE_MAX_MB="1")
    res = subprocess.run([sys.executable, "-c", _CACHE_SCRIPT.format(root=root)], env=env, capture_output=True, text=True,
                         check=True)
    assert json.loads(res.stdout.strip().splitlines()[-1])["builds"] == 1
    left = sorted(p.name for p in tmp_path.iterdir())
    total = sum(p.stat().st_size for p in tmp_path.iterdir())
    assert total <= (1 << 20) and any(not n.endswith("-1.co") for n in left)          # the new object stayed
    assert "sdfk-%016x-1.co" % 0 not in left                                         # the oldest went first


def test_background_builds_run_in_a_compiler_process(built):
    """What a background build does (sdfk_rtc_helper: hiprtc in a child process, never inside the caller — see
    profiles/r03_hang_import_during_build.txt) yields the same code object as the in-process build."""
    import ctypes
    import aegolius_amd.cores as ns
    from aegolius_amd import workloads
    from aegolius_amd._lower import lower_geometry
    assert os.access(os.path.join(os.path.dirname(built.LIB_PATH), "sdfk_rtc_helper"), os.X_OK)
    prog = built.Program.from_lowered(lower_geometry(workloads.cfg2_tree(ns, seed=77)))
    n = ctypes.c_size_t(0)
    built.check(built.lib().sdfk_debug_compile_external(prog.handle, built.FLAVOUR_ROWS_ARRAY, ctypes.byref(n)), "external")
    inproc, _seconds = prog.compile_flavour(built.FLAVOUR_ROWS_ARRAY)
    assert n.value == inproc > 10000
    # a source that does not compile: the helper's log comes back as the error
    bad = built.Program.from_lowered(lower_geometry(ns.Sphere(0.5)))
    rc = built.lib().sdfk_debug_compile_external(bad.handle, built.FLAVOUR_ROWS_ARRAY, ctypes.byref(n))
    assert rc == -2 and "no cull sites" in built.last_error()


def test_flag_writing_builds_are_code_objects_of_their_own(built):
    """Fused selection launches the `| FLAVOUR_FLAGS` build of a flavour (#define SDFK_FLAGS: one bit per point instead of
    the field). It compiles for every kind of kernel — plain, row blocks, flat row blocks, chain mode — and is a separate
    translation unit: the field kernels carry no flag code (as a run-time branch it cost cfg 5 ten percent)."""
    import aegolius_amd.cores as ns
    from aegolius_amd import workloads
    from aegolius_amd._lower import lower_geometry
    F = built.FLAVOUR_FLAGS
    tree = built.Program.from_lowered(lower_geometry(workloads.cfg2_tree(ns, seed=5)))
    field_size, _ = tree.compile_flavour(built.FLAVOUR_ROWS_ARRAY)
    flag_size, _ = tree.compile_flavour(built.FLAVOUR_ROWS_ARRAY | F)
    assert flag_size > 10000 and flag_size != field_size
    for flavour in (built.FLAVOUR_ROWS_GRID, built.FLAVOUR_PLAIN_ARRAY, built.FLAVOUR_PLAIN_GRID):
        assert tree.compile_flavour(flavour | F)[0] > 5000
    chain = built.Program.from_lowered(lower_geometry(workloads.cfg4_scene2d(ns)))
    assert chain.compile_flavour(built.FLAVOUR_ROWS2D_ARRAY | F)[0] > 10000
    plain = built.Program.from_lowered(lower_geometry(workloads.cfg3_chain(ns)))
    assert plain.compile_flavour(built.FLAVOUR_PLAIN_ARRAY | F)[0] > 5000
    with pytest.raises(built.SdfkError):
        tree.compile_flavour(built.FLAVOUR_TILE_ARRAY | F)


def test_nested_hard_unions_flatten_only_at_chain_size(built, monkeypatch):
    """Lowering: nested hard UNION / INTERSECT whose groups carry nothing but a transform become one n-ary chain when
    that reaches the size of the chain kernels (22 members); smaller trees, groups with a modification between the two
    combiners, groups of the other kind and smooth combiners keep their hierarchy."""
    import aegolius_amd.cores as ns
    from aegolius_amd._lower import lower_geometry

    def cluster(kind, count, dx):
        objs = []
        for j in range(count):
            o = ns.Sphere(0.05 + 0.001 * j)
            o.move((dx + 0.1 * j, 0.02 * j, -0.03 * j))
            objs.append(o)
        return ns.CombineGeometry(kind).combine(*objs)

    def members(tree):
        return built.Program.from_lowered(lower_geometry(tree)).chain_members

    a, b = cluster("UNION", 12, 0.0), cluster("UNION", 12, 2.0)
    b.rotate(0.4, (0, 1, 1))
    b.rescale(1.5)
    assert members(ns.CombineGeometry("UNION2").combine(a, b)) == 24
    monkeypatch.setenv("SDFK_NO_FLATTEN", "1")
    assert members(ns.CombineGeometry("UNION2").combine(a, b)) == 0
    monkeypatch.delenv("SDFK_NO_FLATTEN")
    assert members(ns.CombineGeometry("UNION2").combine(cluster("UNION", 8, 0.0), cluster("UNION", 8, 2.0))) == 0     # 16 < 22
    c = cluster("UNION", 12, 2.0)
    c.rounding(0.01)                                            # a value modification between the combiners
    assert members(ns.CombineGeometry("UNION2").combine(a, c)) == 0
    assert members(ns.CombineGeometry("UNION2").combine(a, cluster("INTERSECT", 12, 2.0))) == 0
    assert members(ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(a, b, parameters=0.1)) == 0
    whole = cluster("INTERSECT", 24, 0.0)                      # a transform of the whole combination goes into its members
    whole.rotate(0.2, (1, 0, 0))
    whole.move((0.1, 0.2, 0.3))
    whole.rescale(1.3)
    assert members(whole) == 24
    whole.onion(0.01)                                           # ... and a value modification on top stays the chain's tail
    assert members(whole) == 24
    # body minus a large union: an INTERSECT of the body and the negated members
    assert members(ns.CombineGeometry("SUBTRACT2").combine(ns.Box(1, 1, 1), cluster("UNION", 30, 0.0))) == 31
    assert members(ns.CombineGeometry("SUBTRACT2").combine(ns.Box(1, 1, 1), cluster("UNION", 8, 0.0))) == 0
    # ... minus an INTERSECT: not rewritten, but the intersection itself is a chain and box + subtraction the REST of the
    # program, evaluated per point around the chain's value (sdfk_codegen.cpp chain_analyse); so are a clipped union, a
    # union blended with something else, and either of them placed as a whole
    assert members(ns.CombineGeometry("SUBTRACT2").combine(ns.Box(1, 1, 1), cluster("INTERSECT", 30, 0.0))) == 30
    clipped = ns.CombineGeometry("INTERSECT2").combine(cluster("UNION", 40, 0.0), ns.Sphere(2.0))
    assert members(clipped) == 40
    clipped.rotate(0.3, (0, 1, 0))
    clipped.move((0.1, 0, 0))
    clipped.rescale(1.2)
    assert members(clipped) == 40
    blended = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(ns.Box(3, 3, 0.1), cluster("UNION", 25, 0.0), parameters=0.1)
    blended.onion(0.01)
    assert members(blended) == 25
    body = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(ns.Box(1, 1, 1), ns.Sphere(0.7), parameters=0.1)
    assert members(ns.CombineGeometry("SUBTRACT2").combine(body, cluster("UNION", 30, 0.0))) == 30
    twisted = cluster("UNION", 30, 0.0)
    twisted.twist(0.3)                                          # the members read warped coordinates: no chain
    assert members(twisted) == 0
    d = cluster("UNION", 10, 2.0)
    d.rescale(-1.0)                                             # a negative scale turns min into max: not flattened
    assert members(ns.CombineGeometry("UNION2").combine(a, d)) == 0


_BIG_SCRIPT = """
import json, sys
sys.path.insert(0, {root!r})
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
from aegolius_amd._lower import lower_geometry
low = lower_geometry(workloads.cfg2_tree(ns, seed=7, count={count}))
prog = _engine.Program.from_lowered(low)
size, seconds = prog.compile_flavour(_engine.FLAVOUR_PLAIN_ARRAY)
print(json.dumps(dict(instructions=int(low.code.shape[0]), size=size, builds=_engine.jit_stats()[0])))
"""


def test_big_programs_are_built_with_their_own_compiler_options(built, tmp_path):
    """From SDFK_BIG_PROGRAM instructions on (300) a program that is no chain is built without the two LLVM passes whose
    time grows with the square of the program (csrc/sdfk.hip rtc_options): the options reach hiprtc through the compiler
    child process, and they are part of the cache key — the same program under another threshold is another file."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(threshold):
        env = dict(os.environ, SDFK_CACHE_DIR=str(tmp_path), SDFK_BIG_PROGRAM=str(threshold))
        res = subprocess.run([sys.executable, "-c", _BIG_SCRIPT.format(root=root, count=101)], env=env, capture_output=True,
                             text=True, check=True)
        return json.loads(res.stdout.strip().splitlines()[-1])
    big = run(300)
    assert big["instructions"] == 302 and big["builds"] == 1 and big["size"] > 50000
    assert len(list(tmp_path.iterdir())) == 1
    again = run(300)
    assert again["builds"] == 0 and again["size"] == big["size"]                # same options: the cached file
    full = run(100000)
    assert full["builds"] == 1 and len(list(tmp_path.iterdir())) == 2         # full pipeline: another key, another file


_CANCEL_SCRIPT = """
import sys, threading, time
sys.path.insert(0, {root!r})
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
from aegolius_amd._lower import lower_geometry
prog = _engine.Program.from_lowered(lower_geometry(workloads.cfg2_tree(ns, seed=9, count=150)))
def build():
    try:
        prog.compile_flavour(_engine.FLAVOUR_ROWS_ARRAY)          # 20 s and more of hiprtc, in the compiler child process
        print("BUILT", flush=True)
    except Exception as exc:
        print("STOPPED", str(exc)[:120], flush=True)
threading.Thread(target=build, daemon=True).start()
time.sleep(2.0)
print("LEAVING", flush=True)
"""


def test_a_process_that_leaves_does_not_wait_for_a_running_build(built, tmp_path):
    """atexit -> sdfk_jit_cancel: the compiler child process of a build that is still running is killed and the
    interpreter exits at once — with the build tiers of round 4 a background build may take a minute."""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SDFK_CACHE_DIR="off")
    t0 = time.time()
    res = subprocess.run([sys.executable, "-c", _CANCEL_SCRIPT.format(root=root)], env=env, capture_output=True, text=True,
                         timeout=120)
    took = time.time() - t0
    assert "LEAVING" in res.stdout and "BUILT" not in res.stdout, res.stdout + res.stderr[-300:]
    assert took < 12.0, took                                       # (import + 2 s of sleep + the kill; the build alone: > 20 s)
