"""Developer tool (GPU): the register-carry gradient kernel with different segment lengths (planes per thread) and XCD run
lengths IN ONE PROCESS, so that the process's state (profiles/r04_gradient_states.json: one process in three runs the
kernel 18 % faster than the others, whatever it does) is the same for every setting. One JSON line per process.
    python tools/grad_seg_ab.py [grid] [setting ...]      setting = SEG or SEG:GROUP"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    res = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    settings = sys.argv[2:] or ["32", "8", "16", "4", "32"]
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    _engine.require_gpu()
    prog = _engine.Program.from_lowered(lower_geometry(ns.Sphere(0.7)))
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (res,) * 3)[0]]
    n0, n1, n2 = (a.size for a in axes)
    n = n0 * n1 * n2
    vp = ctypes.c_void_p
    stride = (n + 63) // 64 * 64
    d_f = lib.sdfk_malloc(n * 4)
    d_v = lib.sdfk_malloc(3 * stride * 4)
    prog.eval_grid(axes, 0, n, d_f)
    out = {}
    first = None
    for s in settings:
        seg, _, group = s.partition(":")
        os.environ["SDFK_GC_SEG"] = seg
        if group:
            os.environ["SDFK_GC_GROUP"] = group
        else:
            os.environ.pop("SDFK_GC_GROUP", None)
        ts = []
        for _ in range(6):
            e0, e1 = _engine.Event(), _engine.Event()
            e0.record(None)
            _engine.check(lib.sdfk_field_gradient(vp(d_f), n0, n1, n2, 3, 1, vp(d_v), stride, None), "gradient")
            e1.record(None)
            ts.append(e0.elapsed_ms(e1))
        host = np.empty(1 << 16, dtype=np.float32)
        _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(host), vp(d_v + 4 * (stride + n // 2)), host.nbytes), "d2h")
        first = host if first is None else first
        out.setdefault(s, []).append({"ms": round(min(ts[1:]), 3), "same_bits": bool(np.array_equal(host, first))})
    print(json.dumps(out))


if __name__ == "__main__":
    main()
