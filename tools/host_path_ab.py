"""Developer tool (GPU): where the time of sdfk_eval_host goes (2.10 -> 1.66 Gpoints/s in the bench line).
One process, the north-star tree on whole x-planes of the 1025^3 grid (the bench's `host_path` extra), variants:
  fresh   the product call: `Program.eval_host` allocates its (M,) result with numpy.empty every call (first-touch page faults
          of 4 B/point inside the timed call)
  reuse   the same C entry point writing into ONE result buffer that has been touched before
  f64     float64 coordinates in (narrowed on the way into the pinned slots)
  torch   `fresh` again after `import torch` + a CUDA context + 8 GB allocated and released (what bench.py has around it)
    python tools/host_path_ab.py [planes]"""
import ctypes
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    planes = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    os.environ.setdefault("SDFK_HOST_TRACE", "1")             # one breakdown line per call on stderr
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, workloads
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (1024,) * 3)[0]]
    m = planes * 1025 * 1025
    hco = np.empty((3, m), dtype=np.float32)
    hco[0] = np.repeat(axes[0][:planes], 1025 * 1025)
    hco[1] = np.tile(np.repeat(axes[1], 1025), planes)
    hco[2] = np.tile(axes[2], planes * 1025)
    tree = workloads.build("cfg2", ns)[0]
    prog = _engine.Program.from_lowered(lower_geometry(tree))
    lib = _engine.lib()
    mode = _engine.MODE_SPECIALIZED
    prog.eval_host(hco, mode=mode)

    def report(tag, fn, reps=5):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        print("%-8s min %.1f ms  median %.1f ms  %.0f Mpoints/s  %.1f GB/s over PCIe" % (
            tag, ts[0] * 1e3, ts[len(ts) // 2] * 1e3, m / ts[0] / 1e6, 16.0 * m / ts[0] / 1e9), flush=True)

    report("fresh", lambda: prog.eval_host(hco, mode=mode))
    keep = []
    report("keep", lambda: keep.append(prog.eval_host(hco, mode=mode)))       # the previous result is NOT unmapped inside the call
    t0 = time.perf_counter()
    keep.clear()
    print("freeing 5 results: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)

    def alloc_touch_free():
        a = np.empty(m, dtype=np.float32)
        a[::1024] = 0.0
        del a
    report("alloc", alloc_touch_free)
    out = np.zeros(m, dtype=np.float32)

    def reuse():
        _engine.check(lib.sdfk_eval_host(prog._h, _engine._ptr(hco), 0, m, m, _engine._ptr(out), 0, mode), "sdfk_eval_host")
    report("reuse", reuse)
    report("fresh", lambda: prog.eval_host(hco, mode=mode))
    h64 = hco[:, :m // 2].astype(np.float64)
    o2 = np.zeros(m // 2, dtype=np.float32)

    def f64():
        _engine.check(lib.sdfk_eval_host(prog._h, _engine._ptr(h64), 1, m // 2, m // 2, _engine._ptr(o2), 0, mode), "sdfk_eval_host")
    f64()
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        f64()
        ts.append(time.perf_counter() - t0)
    print("f64      min %.1f ms  %.0f Mpoints/s (half the points, result buffer reused)" % (min(ts) * 1e3, m / 2 / min(ts) / 1e6), flush=True)
    print("-- one traced call each: fresh, reuse", flush=True)
    prog.eval_host(hco, mode=mode)
    reuse()
    import torch
    big = torch.empty(2 * 1024 ** 3, dtype=torch.float32, device="cuda")
    big.zero_()
    torch.cuda.synchronize()
    del big
    torch.cuda.empty_cache()
    print("torch threads", torch.get_num_threads(), flush=True)
    report("torch", lambda: prog.eval_host(hco, mode=mode))
    report("torch-r", reuse)


if __name__ == "__main__":
    main()
