"""Developer tool (GPU): big trees that are NOT chains (left-deep SMOOTH_UNION2 chains of n primitives, the cfg2 recipe) on a
513^3 grid — what the build-time tiers of csrc/sdfk.hip mean at run time: the kernel AUTO mode ends up on (row blocks up
to SDFK_ROWS_LIMIT instructions, line bricks up to SDFK_SPECIALIZE_LIMIT, big programs built without CodeGenPrepare /
VectorCombine) against the interpreter kernel that served the same program before, bit for bit.
    python tools/big_tree_bench.py [primitives ...]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [100, 150, 200, 300, 400]
    import torch
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, workloads
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (512,) * 3)[0]]
    n = int(np.prod([a.size for a in axes]))
    stride = (n + 255) // 256 * 256
    co = torch.empty((3, stride), dtype=torch.float32, device="cuda")
    out = torch.empty((stride,), dtype=torch.float32, device="cuda")
    ref = torch.empty((stride,), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=st)
    for count in sizes:
        low = lower_geometry(workloads.cfg2_tree(ns, seed=100 + count, count=count))
        prog = _engine.Program.from_lowered(low)
        rec = {"primitives": count, "instructions": int(low.code.shape[0]), "cull_sites": int(len(low.cull_sites)), "grid": "513^3"}

        def timed(mode, dst, reps):
            step = lambda: prog.eval_device(co.data_ptr(), n, stride, dst.data_ptr(), stream=st, mode=mode,      # noqa: E731
                                            row_len=int(axes[2].size), plane_rows=int(axes[1].size))
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            first = time.perf_counter() - t0
            ts = []
            for _ in range(reps):
                e0, e1 = _engine.Event(), _engine.Event()
                e0.record(st); step(); e1.record(st)
                ts.append(e0.elapsed_ms(e1))
            return first, sorted(ts)[len(ts) // 2]
        first, ms = timed(_engine.MODE_SPECIALIZED, out, 5)            # waits for the build AUTO would run in the background
        rec["specialised"] = {"first_call_s": round(first, 2), "ms": round(ms, 3), "kernel": prog.last_kernel() if hasattr(prog, "last_kernel") else None}
        _, msi = timed(_engine.MODE_INTERPRET, ref, 2)
        rec["interpreter_ms"] = round(msi, 3)
        rec["speedup"] = round(msi / ms, 1)
        rec["bit_identical"] = bool(torch.equal(out[:n].view(torch.int32), ref[:n].view(torch.int32)))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
