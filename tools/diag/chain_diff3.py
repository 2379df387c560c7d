import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import aegolius_amd.cores as ns
from aegolius_amd import _engine, workloads
from aegolius_amd._lower import lower_geometry
from aegolius_amd.cores.helper_functions import grid_axes
from oracle import sdf_oracle
cnt = int(sys.argv[1]); req = int(sys.argv[2])
tree = workloads.cfg4_scene2d(ns, seed=7, count=cnt)
low = lower_geometry(tree); prog = _engine.Program.from_lowered(low)
axes = [a.astype(np.float32) for a in grid_axes((10, 10), (req, req))[0]]
n = int(np.prod([a.size for a in axes])); L = int(axes[1].size)
stride = (n + 255) // 256 * 256
co = torch.empty((3, stride), dtype=torch.float32, device="cuda")
outs = {}
st = torch.cuda.current_stream().cuda_stream
_engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=st)
for key, mode, kw in (("interp", _engine.MODE_INTERPRET, {}), ("plain", _engine.MODE_NOCULL, {}), ("culled", _engine.MODE_SPECIALIZED, dict(row_len=L, flat=True)),
                      ("culled_noflat", _engine.MODE_SPECIALIZED, dict(row_len=L, flat=False))):
    o = torch.zeros(stride, dtype=torch.float32, device="cuda")
    prog.eval_device(co.data_ptr(), n, stride, o.data_ptr(), stream=st, mode=mode, **kw)
    torch.cuda.synchronize()
    outs[key] = o[:n].cpu().numpy()
    print(key, "done", flush=True)
cpu = co[:, :n].cpu().numpy().astype(np.float64)
ref = sdf_oracle.evaluate(tree, cpu)
print("chain", "#define SDFK_CHAIN 1" in prog.source(), "leaves", cnt)
for k, v in outs.items():
    print(k, "max err vs oracle", float(np.abs(v - ref).max()), "differs from interp at", int((v != outs["interp"]).sum()))
