import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
from aegolius_amd._lower import lower_geometry
from aegolius_amd.cores.helper_functions import grid_axes
def run(tree, size, req, flat, tag):
    low = lower_geometry(tree); prog = _engine.Program.from_lowered(low)
    axes = [a.astype(np.float32) for a in grid_axes(size, (req,) * len(size))[0]]
    n = int(np.prod([a.size for a in axes])); L = int(axes[2].size) if axes[2].size > 1 else int(axes[1].size)
    stride = (n + 255) // 256 * 256
    co = torch.empty((3, stride), dtype=torch.float32, device="cuda")
    a = torch.empty(stride, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=st)
    prog.eval_device(co.data_ptr(), n, stride, a.data_ptr(), stream=st, mode=_engine.MODE_NOCULL)
    prog.eval_device(co.data_ptr(), n, stride, b.data_ptr(), stream=st, mode=_engine.MODE_SPECIALIZED, row_len=L, flat=flat)
    torch.cuda.synchronize()
    print(tag, "chain", "#define SDFK_CHAIN 1" in prog.source(), "leaves", len(low.cull_sites)+1, "instrs", low.code.shape[0], "differ", int((a[:n]!=b[:n]).sum()), "of", n)
for cnt in (20, 50, 63, 64, 65, 70, 130):
    run(workloads.cfg4_scene2d(ns, seed=7, count=cnt), (10, 10), 1024, True, "2d mixed %d" % cnt)
for cnt in (20, 50, 64, 65, 100):
    run(workloads.sphere_union(ns, cnt), (2, 2, 2), 128, False, "3d spheres %d" % cnt)
