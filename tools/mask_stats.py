"""Developer tool (GPU): dump the brick-culling masks of the bench workload and print how much of the
tree each brick still evaluates."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(grid=512):
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    low = lower_geometry(scenes.cfg2_tree(ns))
    prog = _engine.Program.from_lowered(low)
    axes64, res = grid_axes((2, 2, 2), (grid,) * 3)
    axes = [a.astype(np.float32) for a in axes64]
    n = res[0] * res[1] * res[2]
    stride = (n + 255) // 256 * 256
    d_co = lib.sdfk_malloc(3 * stride * 4)
    nb = (n + 2047) // 2048 * 16
    d_m = lib.sdfk_malloc(nb * 8)
    _engine.grid_fill(d_co, stride, axes, 0, n)
    _engine.check(lib.sdfk_debug_brick_masks(prog.handle, ctypes.c_void_p(d_co), n, stride, ctypes.c_void_p(d_m), None),
                  "masks")
    _engine.check(lib.sdfk_sync(None), "sync")
    m = np.empty(nb, dtype=np.uint64)
    _engine.check(lib.sdfk_memcpy_d2h(m.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_m), m.nbytes), "d2h")
    m = m[:(n + 127) // 128]
    ns_sites = len(low.cull_sites)
    skip_a = np.stack([(m >> np.uint64(2 * k)) & np.uint64(1) for k in range(ns_sites)]).astype(bool)
    skip_b = np.stack([(m >> np.uint64(2 * k + 1)) & np.uint64(1) for k in range(ns_sites)]).astype(bool)
    # left-deep chain: primitive 0 lives in a-range of site 0; primitive k (k>=1) is b-range of site k-1
    alive = np.ones((ns_sites + 1, m.size), dtype=bool)
    for k in range(ns_sites):
        alive[k + 1] &= ~skip_b[k]
        alive[:k + 1] &= ~skip_a[k]
    print("grid %d^3: %d bricks, %d sites; x/y-constant runs: %.3f" % (res[0], m.size, ns_sites, ((m >> np.uint64(63)) & np.uint64(1)).mean()))
    print("skip_b rate per site:", np.round(skip_b.mean(axis=1), 3))
    print("skip_a rate per site:", np.round(skip_a.mean(axis=1), 3))
    print("primitive evaluations still needed: %.3f of un-culled" % alive.mean())
    print("bricks evaluating k primitives:", np.bincount(alive.sum(axis=0), minlength=ns_sites + 2) / m.size)


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
