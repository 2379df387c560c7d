"""Developer tool (GPU): how many children of a left-deep combiner chain a row-block brick still evaluates
(sdfk_debug_row_masks). Usage: row_mask_stats.py cfg4 4096   |   row_mask_stats.py cfg2 256"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(workload="cfg4", grid="4096"):
    import bench
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    _engine.require_gpu()
    geo, size, desc = bench.build_workload(workload, ns)
    low = lower_geometry(geo)
    prog = _engine.Program.from_lowered(low)
    axes = [a.astype(np.float32) for a in grid_axes(size, (int(grid),) * len(size))[0]]
    while len(axes) < 3:
        axes.append(np.zeros(1, dtype=np.float32))
    n = int(np.prod([a.size for a in axes]))
    L = axes[2].size if axes[2].size > 1 else axes[1].size
    d_co = lib.sdfk_malloc(3 * n * 4)
    _engine.grid_fill(d_co, n, axes, 0, n)
    nb, brows = ctypes.c_int64(0), ctypes.c_int(0)
    _engine.check(lib.sdfk_debug_row_masks(prog.handle, ctypes.c_void_p(d_co), n, n, L, None, ctypes.byref(nb), ctypes.byref(brows), None), "size")
    d_m = lib.sdfk_malloc(nb.value * 24)
    _engine.check(lib.sdfk_debug_row_masks(prog.handle, ctypes.c_void_p(d_co), n, n, L, ctypes.c_void_p(d_m), None, None, None), "masks")
    _engine.check(lib.sdfk_sync(None), "sync")
    words = np.empty((nb.value, 3), dtype=np.uint64)
    _engine.check(lib.sdfk_memcpy_d2h(words.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(d_m), words.nbytes), "d2h")
    ns_sites = min(len(low.cull_sites), 64)     # (the read-out kernel returns the first two mask words: sites 0 .. 63)
    bits = np.zeros((2 * ns_sites, nb.value), dtype=bool)
    for k in range(ns_sites):
        w = words[:, k // 32]
        bits[2 * k] = ((w >> np.uint64(2 * (k % 32))) & np.uint64(1)).astype(bool)
        bits[2 * k + 1] = ((w >> np.uint64(2 * (k % 32) + 1)) & np.uint64(1)).astype(bool)
    alive = np.ones((ns_sites + 1, nb.value), dtype=bool)          # left-deep chain: child 0 = a-range of site 0
    for k in range(ns_sites):
        alive[k + 1] &= ~bits[2 * k + 1]
        alive[:k + 1] &= ~bits[2 * k]
    cnt = alive.sum(axis=0)
    print("%s: %d points, %d bricks of %d rows, %d sites; uniform bricks %.3f" % (desc, n, nb.value, brows.value, ns_sites, (words[:, 2] != 0).mean()))
    print("children still evaluated per brick: mean %.2f, median %d, max %d of %d" % (cnt.mean(), np.median(cnt), cnt.max(), ns_sites + 1))
    print("histogram:", np.round(np.bincount(cnt, minlength=8)[:12] / nb.value, 3))


if __name__ == "__main__":
    main(*sys.argv[1:])
