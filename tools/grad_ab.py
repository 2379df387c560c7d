"""Developer tool (GPU): flat and register-carry gradient kernels alternating inside one process (1025^3, raw gradient);
carryG = runs of G consecutive chunks per XCD (0: one contiguous eighth of the work per XCD). Times are constant inside a
process and differ between processes and boxes: run it several times."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(res=1024, rounds=1, ballast_gb=0):
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
    ballast = [lib.sdfk_malloc(1 << 30) for _ in range(int(ballast_gb))]   # other allocations of the process, made first
    d_f = lib.sdfk_malloc(n * 4)
    d_v = lib.sdfk_malloc(3 * stride * 4)
    prog.eval_grid(axes, 0, n, d_f)
    line = []
    for r in range(rounds):
        for tag in ("flat", "carry0", "carry8", "carry32"):
            if tag == "flat":
                os.environ["SDFK_GRADIENT_FLAT"] = "1"
            else:
                os.environ.pop("SDFK_GRADIENT_FLAT", None)
                os.environ["SDFK_GC_GROUP"] = tag[5:]
            ts = []
            for _ in range(3):
                e0, e1 = _engine.Event(), _engine.Event()
                e0.record(None)
                _engine.check(lib.sdfk_field_gradient(vp(d_f), n0, n1, n2, 3, 0, vp(d_v), stride, None), "gradient")
                e1.record(None)
                ts.append(e0.elapsed_ms(e1))
            line.append("%s %.2f" % (tag, min(ts)))
    print("  ".join(line))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
