"""Test-infrastructure script (GPU, uses the oracle; not collected by pytest): big trees that are NOT chains — 66 to 170
primitives (more than the 64 sites two mask words hold: SDFK_NMASK = 3 .. 6 words per brick), random pairwise combiners
(smooth and hard unions, subtractions, intersections), left-deep or randomly nested, some primitives with modifications.
For each seed on a small grid: row blocks == line bricks == interpreter kernel (deeply nested trees, which need more registers
than it has: the un-culled specialised kernel) bit for bit, and the oracle on a sample.

    python tests/fuzz_big_trees.py [first_seed] [count]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

OPS = ("SMOOTH_UNION2", "SMOOTH_UNION2", "SMOOTH_UNION2_2", "UNION2", "SUBTRACT2", "INTERSECT2", "SMOOTH_SUBTRACT2", "SMOOTH_INTERSECT2")


def primitive(ns, rng):
    kind = int(rng.integers(0, 6))
    o = [lambda: ns.Sphere(float(rng.uniform(0.15, 0.4))), lambda: ns.Box(*(float(x) for x in rng.uniform(0.2, 0.6, 3))),
         lambda: ns.Cylinder(float(rng.uniform(0.1, 0.3)), float(rng.uniform(0.3, 0.8))),
         lambda: ns.Torus(float(rng.uniform(0.2, 0.4)), float(rng.uniform(0.05, 0.12))),
         lambda: ns.Cone(float(rng.uniform(0.4, 0.8)), float(rng.uniform(0.2, 0.6))),
         lambda: ns.Sphere(float(rng.uniform(0.1, 0.3)))][kind]()
    if rng.random() < 0.15:
        o.onion(float(rng.uniform(0.01, 0.05)))
    if rng.random() < 0.15:
        o.rounding(float(rng.uniform(0.01, 0.05)))
    o.rotate(float(rng.uniform(0, np.pi)), rng.normal(0, 1, 3))
    o.move(rng.uniform(-0.8, 0.8, 3))
    return o


def combine(ns, rng, a, b):
    op = OPS[int(rng.integers(0, len(OPS)))]
    if op in ("SUBTRACT2", "INTERSECT2", "SMOOTH_SUBTRACT2", "SMOOTH_INTERSECT2") and rng.random() < 0.7:
        op = "SMOOTH_UNION2"                                        # (mostly unions: the scene keeps some volume)
    c = ns.CombineGeometry(op)
    if op.startswith("SMOOTH"):
        return c.combine_parametric(a, b, parameters=float(rng.uniform(0.02, 0.2)))
    return c.combine(a, b)


def build(ns, seed):
    rng = np.random.default_rng(seed)
    n = int(rng.integers(66, 171))
    parts = [primitive(ns, rng) for _ in range(n)]
    if rng.random() < 0.5:                                          # left-deep
        acc = parts[0]
        for p in parts[1:]:
            acc = combine(ns, rng, acc, p)
        return acc, n, "left-deep"
    while len(parts) > 1:                                           # random nesting
        i = int(rng.integers(0, len(parts) - 1))
        parts[i:i + 2] = [combine(ns, rng, parts[i], parts[i + 1])]
    return parts[0], n, "nested"


def main(first=0, count=12):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from oracle import sdf_oracle
    grid, _ = ns.generate_grid((2.4, 2.4, 2.4), (24, 20, 64))
    grid32 = np.asarray(grid).astype(np.float32)
    n = grid32.shape[1]
    lib = _engine.lib()
    d_co, d_out = lib.sdfk_malloc(3 * n * 4), lib.sdfk_malloc(n * 4)
    vp = _engine._vp
    _engine.check(lib.sdfk_memcpy_h2d(vp(d_co), _engine._ptr(np.ascontiguousarray(grid32)), grid32.nbytes), "h2d")
    failures, t0 = 0, time.time()
    for seed in range(int(first), int(first) + int(count)):
        tree, prims, shape = build(ns, seed)
        low = lower_geometry(tree)
        prog = _engine.Program.from_lowered(low)
        fields = []
        reference = _engine.MODE_INTERPRET
        try:
            prog.eval_device(d_co, n, n, d_out, mode=reference)
        except _engine.SdfkError:                                   # deep nesting: more registers than the interpreter kernel has
            reference = _engine.MODE_NOCULL                         # -> the un-culled specialised kernel is the reference
        for mode, row_len in ((reference, None), (_engine.MODE_SPECIALIZED, 65), (_engine.MODE_SPECIALIZED, None)):
            prog.eval_device(d_co, n, n, d_out, mode=mode, row_len=row_len)
            _engine.check(lib.sdfk_sync(None), "sync")
            host = np.empty(n, dtype=np.float32)
            _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(host), vp(d_out), n * 4), "d2h")
            fields.append(host)
        msg = []
        if not np.array_equal(fields[0], fields[1], equal_nan=True):
            msg.append("row blocks != reference kernel at %d points" % int((fields[0] != fields[1]).sum()))
        if not np.array_equal(fields[0], fields[2], equal_nan=True):
            msg.append("line bricks != reference kernel at %d points" % int((fields[0] != fields[2]).sum()))
        pick = np.random.default_rng(seed).choice(n, 600, replace=False)
        with np.errstate(all="ignore"):
            ref, mag = sdf_oracle.evaluate_with_magnitude(tree, grid32[:, pick].astype(np.float64))
        err = np.abs(fields[0][pick].astype(np.float64) - ref) / np.maximum(np.maximum(1.0, np.abs(ref)), mag)
        if (~(err <= 1e-6)).any():
            msg.append("%d of 600 sampled points beyond 1e-6 of the oracle (max %.2e)" % (int((~(err <= 1e-6)).sum()), float(np.nanmax(err))))
        failures += bool(msg)
        print("seed %d: %d primitives %s, %d instr, %d sites, reference %s, max scaled err %.2e %s" % (
            seed, prims, shape, low.code.shape[0], len(low.cull_sites), "interpreter" if reference == _engine.MODE_INTERPRET else "un-culled",
            float(np.nanmax(err)), ("<-- " + "; ".join(msg)) if msg else ""), flush=True)
    print("%d trees in %.0f s: %d failures" % (int(count), time.time() - t0, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
