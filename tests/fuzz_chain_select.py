"""Test-infrastructure script (GPU, uses the oracle; not collected by pytest): differential fuzzing of the round-3 paths.

    python tests/fuzz_chain_select.py [first_seed] [count]

For each seed a random n-ary hard UNION / INTERSECT (9 .. 400 children of random kinds — 3-D or 2-D primitives with
random rounding / onion / rotation / position —, in a third of the larger scenes as nested groups with rigid transforms
of their own, optionally with a value modification on top) on a random grid shape
(odd / short / long rows, 2-D scenes on flat grids):
  * the table-driven chain kernel with per-brick survivor lists (row blocks; candidate lists per cell from 65 members
    on), its un-culled loop (MODE_NOCULL) and — up to 120 children, and for every program that is no chain — the
    interpreter kernel: bit for bit (programs that are no chain and lie beyond SDFK_SPECIALIZE_LIMIT = 1200 instructions
    are served by the interpreter kernel in the product: that kernel against the oracle);
  * the per-axis table flavour (sdfk_eval_grid_host) against the array flavour: bit for bit;
  * a sample of the field against the float64 oracle (1e-6, magnitude-aware as tests/test_gpu_parity.py);
  * fused selection (flag-writing kernels + compaction) for random thresholds against numpy.flatnonzero(field <= t) of
    the field the same kernels write: exact, grid and array flavours.
Below 17 children the program is an ordinary tree (mask kernels); from 17 on it is chain mode.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_scene(ns, rng):
    flat = rng.random() < 0.4
    count = int(rng.choice([9, 16, 17, 18, 33, 64, 65, 100, 129, 250, 400], p=[.08, .08, .12, .1, .12, .1, .1, .1, .08, .07, .05]))
    if flat:
        kinds = [lambda: ns.Circle(rng.uniform(0.1, 0.5)), lambda: ns.Rectangle(rng.uniform(0.2, 0.9), rng.uniform(0.2, 0.6)),
                 lambda: ns.NGon(rng.uniform(0.15, 0.45), int(rng.integers(3, 9))),
                 lambda: ns.Segment((rng.uniform(-.3, 0), rng.uniform(-.3, 0), 0), (rng.uniform(0, .3), rng.uniform(0, .3), 0))]
        extent = 4.0
    else:
        kinds = [lambda: ns.Sphere(rng.uniform(0.05, 0.3)), lambda: ns.Box(rng.uniform(.1, .5), rng.uniform(.1, .4), rng.uniform(.1, .3)),
                 lambda: ns.Cylinder(rng.uniform(.05, .25), rng.uniform(.1, .4)), lambda: ns.Torus(rng.uniform(.15, .3), rng.uniform(.03, .08)),
                 lambda: ns.Cone(rng.uniform(.2, .5), rng.uniform(.2, .6))]
        extent = 1.1
    use = [kinds[i] for i in rng.choice(len(kinds), size=int(rng.integers(1, len(kinds) + 1)), replace=False)]
    objs = []
    for k in range(count):
        o = use[k % len(use)]()
        r = rng.random()
        if r < 0.3:
            o.rounding(float(rng.uniform(0.01, 0.06)))
        elif r < 0.5:
            o.onion(float(rng.uniform(0.01, 0.04)))
        if rng.random() < 0.7:
            o.rotate(float(rng.uniform(0, np.pi)), (0, 0, 1) if flat else tuple(rng.normal(size=3)))
        o.move((float(rng.uniform(-extent, extent)), float(rng.uniform(-extent, extent)), 0.0 if flat else float(rng.uniform(-extent, extent))))
        objs.append(o)
    kind = "UNION" if rng.random() < 0.7 else "INTERSECT"
    if count >= 33 and rng.random() < 0.35:                    # nested: rigidly placed groups (flattened by the lowering)
        ngroups = int(rng.integers(2, 7))
        groups = []
        for part in np.array_split(np.arange(count), ngroups):
            g = ns.CombineGeometry(kind).combine(*[objs[i] for i in part])
            if rng.random() < 0.8:
                g.rotate(float(rng.uniform(0, np.pi)), (0, 0, 1) if flat else tuple(rng.normal(size=3)))
            if rng.random() < 0.8:
                g.move((float(rng.uniform(-.5, .5)), float(rng.uniform(-.5, .5)), 0.0 if flat else float(rng.uniform(-.5, .5))))
            if rng.random() < 0.3:
                g.rescale(float(rng.uniform(0.7, 1.4)))
            groups.append(g)
        if len(groups) > 2 and rng.random() < 0.5:              # a third level
            pair = ns.CombineGeometry(kind + "2").combine(groups[0], groups[1])
            pair.move((0.05, -0.1, 0.0))
            groups = [pair] + groups[2:]
        objs = groups
    tree = ns.CombineGeometry(kind).combine(*objs)
    if kind == "UNION" and rng.random() < 0.2:                  # a body minus the union: lowered as one INTERSECT chain
        if rng.random() < 0.5:
            tree.move((0.03, -0.02, 0.0))
        body = ns.Rectangle(1.6 * extent, 1.4 * extent) if flat else ns.Box(1.6 * extent, 1.5 * extent, 1.2 * extent)
        tree = ns.CombineGeometry("SUBTRACT2").combine(body, tree)
    if count >= 17 and rng.random() < 0.3:                      # the union as an OPERAND of a larger program (chain + rest)
        w = rng.random()
        other = ns.Circle(0.8 * extent) if flat else ns.Sphere(0.8 * extent)
        if w < 0.35:
            tree = ns.CombineGeometry("INTERSECT2").combine(tree, other)
        elif w < 0.7:
            tree = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(other, tree, parameters=0.1)
        else:
            body = ns.CombineGeometry("SMOOTH_UNION2_2").combine_parametric(other, ns.Rectangle(extent, extent) if flat else ns.Box(extent, extent, extent), parameters=0.2)
            tree = ns.CombineGeometry("SUBTRACT2").combine(body, tree)
        if rng.random() < 0.5:
            tree.rotate(float(rng.uniform(0, 1)), (0, 0, 1))
            tree.move((0.02, 0.03, 0.0))
    r = rng.random()
    if r < 0.15:
        tree.rounding(0.02)
    elif r < 0.25:
        tree.onion(0.05)
    elif r < 0.3:
        tree.rescale(1.3)
    if flat:
        shape = (int(rng.integers(20, 200)), int(rng.choice([33, 64, 65, 127, 130, 257, 513, 1000])))
        size = (2.4 * extent, 2.4 * extent)
    else:
        shape = (int(rng.integers(3, 40)), int(rng.integers(3, 50)), int(rng.choice([8, 31, 32, 33, 64, 65, 100, 129, 257])))
        size = (2.6 * extent,) * 3
    return tree, size, shape, flat, count


def device_eval(engine, prog, co32, mode, row_len=None, flat=False, misalign=0):
    """eval_device on raw HIP buffers (row-block kernel with a row-length hint; optionally a 4-byte-aligned shift)."""
    import ctypes
    lib, vp = engine.lib(), ctypes.c_void_p
    n = co32.shape[1]
    stride = (n + 63) // 64 * 64
    d_co = lib.sdfk_malloc((3 * stride + misalign + 4) * 4)
    d_out = lib.sdfk_malloc((n + misalign + 4) * 4)
    try:
        host = np.zeros((3, stride), dtype=np.float32)
        host[:, :n] = co32
        engine.check(lib.sdfk_memcpy_h2d(vp(d_co + 4 * misalign), host.ctypes.data_as(vp), host.nbytes), "h2d")
        prog.eval_device(d_co + 4 * misalign, n, stride, d_out + 4 * misalign, mode=mode, row_len=row_len, flat=flat)
        engine.check(lib.sdfk_sync(None), "sync")
        out = np.empty(n, dtype=np.float32)
        engine.check(lib.sdfk_memcpy_d2h(out.ctypes.data_as(vp), vp(d_out + 4 * misalign), out.nbytes), "d2h")
        return out
    finally:
        lib.sdfk_free(vp(d_co))
        lib.sdfk_free(vp(d_out))


SPECIALIZE_LIMIT = int(os.environ.get("SDFK_SPECIALIZE_LIMIT", "1200"))     # csrc/sdfk.hip, run(): the same default


def main(first=9000, count=40):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from oracle import sdf_oracle
    bad, t0, chain_seen = [], time.time(), 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        tree, size, shape, flat, children = random_scene(ns, rng)
        co, _ = ns.generate_grid(size, tuple(s - 1 for s in shape))
        axes = [a.astype(np.float32) for a in co.grid_axes]
        co32 = np.ascontiguousarray(np.asarray(co).astype(np.float32))
        n = co32.shape[1]
        row_len = int(axes[1].size if flat else axes[2].size)      # (generate_grid may return other sizes than asked for)
        shape = tuple(int(a.size) for a in axes)
        low = lower_geometry(tree)
        prog = _engine.Program.from_lowered(low)
        chain = prog.chain_members > 0
        chain_seen += chain
        msg = []
        if not chain and low.code.shape[0] > SPECIALIZE_LIMIT:
            # a program that is no chain and lies beyond SDFK_SPECIALIZE_LIMIT (csrc/sdfk.hip: the size up to which a
            # specialised kernel is built within the build budget): the product serves it from the interpreter kernel,
            # which is what is checked — against the oracle, every sampled point
            out = device_eval(_engine, prog, co32, _engine.MODE_INTERPRET)
            pick = rng.choice(n, size=min(n, 4000), replace=False)
            with np.errstate(all="ignore"):
                ref, mag = sdf_oracle.evaluate_with_magnitude(tree, co32[:, pick].astype(np.float64))
            err = np.abs(out[pick].astype(np.float64) - ref) / np.maximum(np.maximum(1.0, np.abs(ref)), mag)
            n_off = int((~(err <= 1e-6)).sum())
            if n_off:
                bad.append((seed, ["%d of %d sampled points off the oracle (interpreter)" % (n_off, pick.size)]))
            print("seed %d: %d children, NO chain, %d instr (beyond the specialisation limit): interpreter, max scaled err %.2e" % (
                seed, children, low.code.shape[0], np.nanmax(err)), flush=True)
            continue
        plain = device_eval(_engine, prog, co32, _engine.MODE_NOCULL)
        for hint, shift in ((row_len, 0), (row_len, 1), (None, 0)):   # row blocks (aligned / shifted pointers), line bricks or plain
            culled = device_eval(_engine, prog, co32, _engine.MODE_SPECIALIZED, row_len=hint, flat=flat and hint is not None, misalign=shift)
            if not np.array_equal(culled, plain, equal_nan=True):
                msg.append("culled (row_len %r, shift %d) != un-culled at %d points" % (hint, shift, int((culled != plain).sum())))
        if children <= 120 or not chain:                        # (every specialised NON-chain program meets the interpreter)
            interp = device_eval(_engine, prog, co32, _engine.MODE_INTERPRET)
            if not np.array_equal(interp, plain, equal_nan=True):
                msg.append("interpreter != specialised at %d points" % int((interp != plain).sum()))
        table = prog.eval_grid_host(axes, mode=_engine.MODE_SPECIALIZED)
        if not np.array_equal(table, plain, equal_nan=True):
            msg.append("table flavour != array flavour at %d points" % int((table != plain).sum()))
        pick = rng.choice(n, size=min(n, 4000), replace=False)
        with np.errstate(all="ignore"):
            ref, mag = sdf_oracle.evaluate_with_magnitude(tree, co32[:, pick].astype(np.float64))
        err = np.abs(plain[pick].astype(np.float64) - ref) / np.maximum(np.maximum(1.0, np.abs(ref)), mag)
        n_off = int((~(err <= 1e-6)).sum())
        if n_off:
            msg.append("%d of %d sampled points off the oracle (max %.2e)" % (n_off, pick.size, np.nanmax(err)))
        for thr in (0.0, float(np.quantile(plain, rng.uniform(0.01, 0.99))), float(plain.min()), float(plain.max()), -1e30):
            want = np.flatnonzero(plain <= np.float32(thr))
            got_grid = prog.select_grid(axes, thr)
            got_arr = prog.select_host(co32, thr)
            if not (np.array_equal(got_grid, want) and np.array_equal(got_arr, want)):
                msg.append("fused selection at %r: %d / %d indices, want %d" % (thr, got_grid.size, got_arr.size, want.size))
                break
        if msg:
            bad.append((seed, msg))
        print("seed %d: %d children%s on %s, %d instr, max scaled err %.2e %s" % (
            seed, children, " (chain mode)" if chain else "", "x".join(map(str, shape)), low.code.shape[0], np.nanmax(err),
            "  <-- " + "; ".join(msg) if msg else ""), flush=True)
    print("%d scenes (%d in chain mode) in %.0f s: %d failures" % (count, chain_seen, time.time() - t0, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*(int(a) for a in sys.argv[1:])))
