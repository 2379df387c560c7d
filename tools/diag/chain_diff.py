"""Developer aid (GPU): where does the culled chain kernel differ from the un-culled one? (SDFK_CHAIN_MIN=8 cfg4)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
from aegolius_amd._lower import lower_geometry
from aegolius_amd.cores.helper_functions import grid_axes
tree, size, desc = workloads.build("cfg4", ns)
low = lower_geometry(tree)
prog = _engine.Program.from_lowered(low)
print("chain", "#define SDFK_CHAIN 1" in prog.source())
for req in (1024, 4096, 16384):
    axes = [a.astype(np.float32) for a in grid_axes(size, (req,) * 2)[0]]
    n = int(np.prod([a.size for a in axes])); L = int(axes[1].size)
    stride = (n + 255) // 256 * 256
    co = torch.empty((3, stride), dtype=torch.float32, device="cuda")
    a = torch.empty(stride, dtype=torch.float32, device="cuda"); b = torch.empty_like(a)
    st = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=st)
    prog.eval_device(co.data_ptr(), n, stride, a.data_ptr(), stream=st, mode=_engine.MODE_NOCULL)
    prog.eval_device(co.data_ptr(), n, stride, b.data_ptr(), stream=st, mode=_engine.MODE_SPECIALIZED, row_len=L, flat=True)
    torch.cuda.synchronize()
    diff = torch.nonzero(a[:n] != b[:n]).flatten()
    print(req, "points", n, "differ", diff.numel(), "nan", int(torch.isnan(a[:n]).sum()), int(torch.isnan(b[:n]).sum()))
    if diff.numel():
        d = diff[:10].cpu().numpy()
        print("  idx", d, "row", d // L, "col", d % L)
        print("  plain", a[diff[:10]].cpu().numpy(), "culled", b[diff[:10]].cpu().numpy())
        rows = (diff // L).cpu().numpy(); cols = (diff % L).cpu().numpy()
        print("  rows range", rows.min(), rows.max(), "cols range", cols.min(), cols.max(), "distinct row-blocks", len(np.unique(rows // 16)), "windows", len(np.unique(cols // 32)))
