"""Developer tool (GPU): sdfk_field_gradient at 1025^3 with the output buffer shifted by a few offsets and row strides —
how much the time depends on where the buffers lie relative to each other (run with SDFK_GRADIENT_FLAT=1 for the flat kernel)."""
import ctypes
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
    prog = _engine.Program.from_lowered(lower_geometry(ns.Sphere(0.7)))
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (res,) * 3)[0]]
    n0, n1, n2 = (a.size for a in axes)
    n = n0 * n1 * n2
    vp = ctypes.c_void_p
    slack = 64 << 20
    d_f = lib.sdfk_malloc(n * 4 + slack)
    d_v = lib.sdfk_malloc(3 * (n + (1 << 22)) * 4 + slack)
    print("field at %#x, vectors at %#x" % (d_f, d_v))
    for f_off in (0, 4096, 1 << 20):
        prog.eval_grid(axes, 0, n, d_f + f_off)
        for v_off in (0, 1024, 4096, 65536, 1 << 20, (1 << 21) + 8192):
            for stride in ((n + 63) // 64 * 64, (n + 1023) // 1024 * 1024, (n + 1023) // 1024 * 1024 + 256 * 37):
                best = 1e9
                for _ in range(4):
                    e0, e1 = _engine.Event(), _engine.Event()
                    e0.record(None)
                    _engine.check(lib.sdfk_field_gradient(vp(d_f + f_off), n0, n1, n2, 3, 0, vp(d_v + v_off), stride, None), "gradient")
                    e1.record(None)
                    best = min(best, e0.elapsed_ms(e1))
                print("field +%-8d vectors +%-8d stride n+%-6d : %.3f ms" % (f_off, v_off, stride - n, best), flush=True)


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
