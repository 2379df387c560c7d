"""Test-infrastructure script (GPU, uses the oracle; not collected by pytest): differential fuzzing beyond the
48 golden random trees. For each seed: the tree on the shared point cloud against the float64 oracle
(magnitude-aware tolerance, as tests/test_gpu_parity.py), interpreter == specialised bit for bit, and — when the tree
has cull sites — row-block kernel == line-brick kernel == plain kernel bit for bit on a small grid.

    python tests/fuzz_random_trees.py [first_seed] [count] [depth]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(first=5000, count=60, depth=3):
    import scenes
    import aegolius_amd
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from oracle import sdf_oracle
    co = scenes.input_points()
    grid, _ = ns.generate_grid((2.6, 2.6, 2.6), (40, 36, 64))
    grid32 = np.asarray(grid).astype(np.float32)
    n = grid32.shape[1]
    lib = _engine.lib()
    d_co, d_out = lib.sdfk_malloc(3 * n * 4), lib.sdfk_malloc(n * 4)
    vp = _engine._vp
    _engine.check(lib.sdfk_memcpy_h2d(vp(d_co), _engine._ptr(np.ascontiguousarray(grid32)), grid32.nbytes), "h2d")
    bad, t0, sites_seen = [], time.time(), 0
    for seed in range(first, first + count):
        tree = scenes.random_tree(ns, seed, depth)
        with np.errstate(all="ignore"):
            ref, mag = sdf_oracle.evaluate_with_magnitude(tree, co)
        outs = []
        for mode in (_engine.MODE_SPECIALIZED, _engine.MODE_INTERPRET):
            aegolius_amd.config.mode = mode
            outs.append(tree.create(co.copy()).astype(np.float64))
        aegolius_amd.config.mode = 0
        err = np.abs(outs[0] - ref) / np.maximum(np.maximum(1.0, np.abs(ref)), mag)
        err[np.isnan(ref) & np.isnan(outs[0])] = 0
        n_bad = int((~(err <= 1e-6)).sum())
        msg = []
        if n_bad > max(1, int(0.005 * ref.size)):
            msg.append("%d points off (max %.2e)" % (n_bad, np.nanmax(err)))
        if not np.array_equal(outs[0], outs[1], equal_nan=True):
            msg.append("interpreter != specialised")
        low = lower_geometry(tree)
        if len(low.cull_sites):
            sites_seen += 1
            prog = _engine.Program.from_lowered(low)
            fields = []
            for mode, row_len in ((_engine.MODE_NOCULL, None), (_engine.MODE_SPECIALIZED, None), (_engine.MODE_SPECIALIZED, 65)):
                prog.eval_device(d_co, n, n, d_out, mode=mode, row_len=row_len)
                _engine.check(lib.sdfk_sync(None), "sync")
                host = np.empty(n, dtype=np.float32)
                _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(host), vp(d_out), n * 4), "d2h")
                fields.append(host)
            if not (np.array_equal(fields[0], fields[1], equal_nan=True) and np.array_equal(fields[0], fields[2], equal_nan=True)):
                msg.append("culled kernels differ from the plain kernel")
        if msg:
            bad.append((seed, msg))
        print("seed %d: %d instr, %d sites, max scaled err %.2e %s" % (seed, low.code.shape[0], len(low.cull_sites),
                                                                       np.nanmax(err), "  <-- " + "; ".join(msg) if msg else ""), flush=True)
    lib.sdfk_free(vp(d_co))
    lib.sdfk_free(vp(d_out))
    print("%d trees (%d with cull sites) in %.0f s: %d failures" % (count, sites_seen, time.time() - t0, len(bad)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*(int(a) for a in sys.argv[1:])))
