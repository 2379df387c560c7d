"""Developer tool (GPU): the fused selection (sdfk_eval_device_select) on the north-star tree, resident 1025^3 grid — run
under rocprofv3 --kernel-trace --stats for the per-kernel split.   python tools/fused_select_bench.py [grid] [workload]"""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(grid=1024, workload="cfg2"):
    import torch
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, workloads
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    tree, size, _desc = workloads.build(workload, ns)
    prog = _engine.Program.from_lowered(lower_geometry(tree))
    axes = [a.astype(np.float32) for a in grid_axes(size, (int(grid),) * len(size))[0]]
    n = int(np.prod([a.size for a in axes]))
    flat = axes[2].size == 1
    row_len = int(axes[1].size if flat else axes[2].size)
    stride = (n + 255) // 256 * 256
    co = torch.empty((3, stride), dtype=torch.float32, device="cuda")
    out = torch.empty(stride, dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=st)
    vp = ctypes.c_void_p
    scratch = torch.empty(lib.sdfk_eval_select_scratch(n, row_len), dtype=torch.uint8, device="cuda")
    m = ctypes.c_int64(0)
    _engine.check(lib.sdfk_eval_device_select(prog.handle, vp(co.data_ptr()), n, stride, row_len, int(flat), 0.0, None, 0,
                                              ctypes.byref(m), vp(scratch.data_ptr()), vp(st), _engine.MODE_SPECIALIZED), "count")
    index = torch.empty(max(m.value, 1), dtype=torch.int64, device="cuda")
    for _ in range(5):
        prog.eval_device(co.data_ptr(), n, stride, out.data_ptr(), stream=st, mode=_engine.MODE_SPECIALIZED, row_len=row_len, flat=flat)
    best = 1e9
    for _ in range(5):
        e0, e1 = _engine.Event(), _engine.Event()
        e0.record(st)
        _engine.check(lib.sdfk_eval_device_select(prog.handle, vp(co.data_ptr()), n, stride, row_len, int(flat), 0.0,
                                                  vp(index.data_ptr()), index.numel(), ctypes.byref(m), vp(scratch.data_ptr()),
                                                  vp(st), _engine.MODE_SPECIALIZED), "select")
        e1.record(st)
        best = min(best, e0.elapsed_ms(e1))
    want = torch.nonzero(out[:n] <= 0).flatten()
    print("points", n, "selected", m.value, "fused ms", round(best, 3), "equals nonzero(field <= 0):", bool(torch.equal(want, index[:m.value])))


if __name__ == "__main__":
    main(*sys.argv[1:])
