"""Developer tool (GPU): time the grid-neighbourhood operators on a resident 513^3 field.
Algorithmic traffic: 8 B/point per pass (fp32 read + write); `signed` adds its byte-sized work arrays."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(res=512):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    s = ns.Sphere(0.6)
    s.boundary()
    prog = _engine.Program.from_lowered(lower_geometry(s))
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (res,) * 3)[0]]
    n0, n1, n2 = (a.size for a in axes)
    n = n0 * n1 * n2
    d_f = lib.sdfk_malloc(n * 4)
    d_s = lib.sdfk_malloc(n * 4)
    vp = ctypes.c_void_p
    out = {"grid": [n0, n1, n2], "points": n}

    def refill():
        prog.eval_grid(axes, 0, n, d_f)

    def timed(label, fn, passes, reps=3):
        best = 1e9
        for _ in range(reps):
            refill()
            e0, e1 = _engine.Event(), _engine.Event()
            e0.record(None)
            fn()
            e1.record(None)
            best = min(best, e0.elapsed_ms(e1))
        out[label] = {"ms": best, "passes": passes, "GB/s": 8.0 * n * passes / best / 1e6}

    timed("box_average_3x3x3", lambda: _engine.check(lib.sdfk_grid_box_average(vp(d_f), n0, n1, n2, 3, 3, 3, 1, vp(d_s), None), "box"), 1)
    timed("box_average_5x5x1_x4", lambda: _engine.check(lib.sdfk_grid_box_average(vp(d_f), n0, n1, n2, 5, 5, 1, 4, vp(d_s), None), "box"), 4)
    timed("edge_detect", lambda: _engine.check(lib.sdfk_grid_edge_detect(vp(d_f), n0, n1, n2, vp(d_s), None), "edge"), 1)
    sep = float(axes[2][1] - axes[2][0])
    timed("signed", lambda: _engine.check(lib.sdfk_grid_signed(vp(d_f), n0, n1, n2, sep, 1, vp(d_s), None), "signed"), 1)
    lib.sdfk_free(vp(d_f))
    lib.sdfk_free(vp(d_s))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
