"""Developer tool (GPU): a vector-field chain on a resident 1025^3 cloud through sdfk_vec_eval_device.
Chain: radial-cylindrical field, turned about z by a per-point angle, revolved about x with the same coordinates,
normalised. Algorithmic bytes per point: 12 (p) + 4 (the angle row) + 12 (the field) = 28."""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(res=1024):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, _vector
    from aegolius_amd.cores.helper_functions import grid_axes
    lib, vp = _engine.lib(), ctypes.c_void_p
    _engine.require_gpu()
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (res,) * 3)[0]]
    n = int(np.prod([a.size for a in axes]))
    stride = (n + 63) // 64 * 64
    d_p = lib.sdfk_malloc(3 * stride * 4)
    d_s = lib.sdfk_malloc(stride * 4)
    d_o = lib.sdfk_malloc(3 * stride * 4)
    _engine.grid_fill(d_p, stride, axes, 0, n)
    _engine.check(lib.sdfk_memcpy_d2d(vp(d_s), vp(d_p), n * 4), "d2d")       # any per-point angle: the x coordinate
    small = np.zeros((3, 8))
    f = ns.RadialCylindricalVectorField()
    f.rotate_phi(np.zeros(8))
    f.revolution_x(small)
    f.normalize()
    instr, _ = _vector.lower_only(f.vf, small, ())
    prog = (_vector.VecInstr * len(instr))()
    for k, (op, ka, kb, src, imm) in enumerate(instr):
        prog[k].op = op | ka << 8 | kb << 12
        prog[k].src[0], prog[k].src[1] = src
        for j in range(4):
            prog[k].imm[j] = imm[j]
    out = {"grid": [a.size for a in axes], "points": n, "instructions": len(instr)}
    for label, kind, nbytes, interpret in (("chain_to_vector", 0, 28.0 * n, 0), ("chain_to_length", 6, 20.0 * n, 0),
                                           ("chain_to_vector_interpreter_kernel", 0, 28.0 * n, 1)):
        lib.sdfk_vec_set_interpret(interpret)
        _engine.check(lib.sdfk_vec_eval_device(prog, len(instr), vp(d_p), n, stride, vp(d_s), 1, stride, kind, vp(d_o),
                                               stride, None), "sdfk_vec_eval_device")     # builds the kernel: not timed
        _engine.check(lib.sdfk_sync(None), "sync")
        best = 1e9
        for _ in range(5):
            e0, e1 = _engine.Event(), _engine.Event()
            e0.record(None)
            _engine.check(lib.sdfk_vec_eval_device(prog, len(instr), vp(d_p), n, stride, vp(d_s), 1, stride, kind, vp(d_o),
                                                   stride, None), "sdfk_vec_eval_device")
            e1.record(None)
            best = min(best, e0.elapsed_ms(e1))
        out[label] = {"ms": round(best, 4), "algorithmic_GB": nbytes / 1e9, "GB/s": round(nbytes / best / 1e6, 1),
                      "frac_of_8TB/s": round(nbytes / best / 1e6 / 8000, 3)}
    lib.sdfk_vec_set_interpret(0)
    for d in (d_p, d_s, d_o):
        lib.sdfk_free(vp(d))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*(int(a) for a in sys.argv[1:]))
