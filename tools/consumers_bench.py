"""Developer tool (GPU): time the field consumers on a resident field (default 1025^3, BASELINE cfg 2's grid).
Algorithmic traffic per point: select = 4 B (the field is read once; 1/8 B of packed flags goes to scratch and back) + 8 B
per selected point;
gradient = 4 B read + 4 B written per component (16 B for a 3-D grid)."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(res=1024):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    _engine.require_gpu()
    s = ns.Sphere(0.7)
    prog = _engine.Program.from_lowered(lower_geometry(s))
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (res,) * 3)[0]]
    n0, n1, n2 = (a.size for a in axes)
    n = n0 * n1 * n2
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    stride = (n + 63) // 64 * 64
    d_f = lib.sdfk_malloc(n * 4)
    d_v = lib.sdfk_malloc(3 * stride * 4)
    d_s = lib.sdfk_malloc(lib.sdfk_field_select_scratch(n))
    prog.eval_grid(axes, 0, n, d_f)
    m = i64(0)
    _engine.check(lib.sdfk_field_select(vp(d_f), n, 0.0, None, 0, ctypes.byref(m), vp(d_s), None), "count")
    d_i = lib.sdfk_malloc(max(m.value, 1) * 8)
    out = {"grid": [n0, n1, n2], "points": n, "selected": m.value}

    def timed(label, fn, nbytes, reps=5):
        best = 1e9
        for _ in range(reps):
            e0, e1 = _engine.Event(), _engine.Event()
            e0.record(None)
            fn()
            e1.record(None)
            best = min(best, e0.elapsed_ms(e1))
        out[label] = {"ms": round(best, 4), "algorithmic_GB": nbytes / 1e9, "GB/s": round(nbytes / best / 1e6, 1),
                      "frac_of_8TB/s": round(nbytes / best / 1e6 / 8000, 3)}

    timed("select_count_only", lambda: _engine.check(
        lib.sdfk_field_select(vp(d_f), n, 0.0, None, 0, ctypes.byref(m), vp(d_s), None), "count"), 4.0 * n)
    def two_step():                                             # what DeviceField.select does: count, allocate, finish
        _engine.check(lib.sdfk_field_select(vp(d_f), n, 0.0, None, 0, ctypes.byref(m), vp(d_s), None), "count")
        _engine.check(lib.sdfk_field_select_finish(n, m.value, vp(d_i), m.value, vp(d_s), None), "finish")
    timed("select_count_then_finish", two_step, 4.0 * n + 8.0 * m.value)
    timed("select", lambda: _engine.check(
        lib.sdfk_field_select(vp(d_f), n, 0.0, vp(d_i), m.value, ctypes.byref(m), vp(d_s), None), "select"),
        4.0 * n + 8.0 * m.value)
    timed("gradient_direction", lambda: _engine.check(
        lib.sdfk_field_gradient(vp(d_f), n0, n1, n2, 3, 1, vp(d_v), stride, None), "gradient"), 16.0 * n)
    timed("gradient_raw", lambda: _engine.check(
        lib.sdfk_field_gradient(vp(d_f), n0, n1, n2, 3, 0, vp(d_v), stride, None), "gradient"), 16.0 * n)
    for d in (d_f, d_v, d_s, d_i):
        lib.sdfk_free(vp(d))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
