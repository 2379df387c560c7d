"""Developer tool (GPU): the SAME workload through the libraries of different rounds on ONE box — each version in a process of
its own (its own package tree under tools/_build/<tag>/, built from `git archive <commit>`; "now" = this tree), the versions
alternating, so that box-to-box and drift effects cancel.   python tools/version_ab.py cfg5 [rounds] [tag ...]
Answers the round-3 verdict's question about cfg 5 (0.690 in round 2, 0.661 in round 3 on the driver's boxes)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, json
root, workload = sys.argv[1], sys.argv[2]
sys.path.insert(0, root)
import numpy as np, torch
import aegolius_amd.cores as ns
from aegolius_amd import _engine
from aegolius_amd._lower import lower_geometry
from aegolius_amd.cores.helper_functions import grid_axes
def prims(rng, count):
    makers = [lambda: ns.Sphere(0.3), lambda: ns.Box(0.5, 0.4, 0.3), lambda: ns.Cylinder(0.2, 0.6), lambda: ns.Torus(0.3, 0.1), lambda: ns.Cone(0.6, np.pi / 8)]
    out = []
    for k in range(count):
        o = makers[k % 5]()
        angle = float(rng.uniform(0, np.pi)); axis = rng.normal(0, 1, 3)
        o.rotate(angle, axis); o.move(rng.uniform(-0.7, 0.7, 3)); out.append(o)
    return out
if workload in ("cfg1", "cfg3"):
    from aegolius_amd import workloads
    tree, size = workloads.build(workload, ns)[:2]
elif workload == "cfg2":
    p = prims(np.random.default_rng(1234), 10); tree = p[0]
    for q in p[1:]: tree = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(tree, q, parameters=0.1)
    size = (2, 2, 2)
else:
    p = prims(np.random.default_rng(2049), 20); groups = []
    for g in range(5):
        acc = p[4 * g]
        for q in p[4 * g + 1:4 * g + 4]: acc = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(acc, q, parameters=0.1)
        groups.append(acc)
    tree = ns.CombineGeometry("SUBTRACT2").combine(ns.CombineGeometry("UNION").combine(*groups[:4]), groups[4]); size = (3, 3, 3)
axes = [a.astype(np.float32) for a in grid_axes(size, (1024,) * 3)[0]]
n = int(np.prod([a.size for a in axes])); stride = (n + 255) // 256 * 256
co = torch.empty((3, stride), dtype=torch.float32, device="cuda"); out = torch.empty((stride,), dtype=torch.float32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
_engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=st)
prog = _engine.Program.from_lowered(lower_geometry(tree))
step = lambda: prog.eval_device(co.data_ptr(), n, stride, out.data_ptr(), stream=st, mode=_engine.MODE_SPECIALIZED, row_len=int(axes[2].size))
for _ in range(8): step()
torch.cuda.synchronize()
ts = []
for _ in range(30):
    e0, e1 = _engine.Event(), _engine.Event(); e0.record(st); step(); e1.record(st); ts.append(e0.elapsed_ms(e1))
ts.sort()
print(json.dumps({"median_ms": ts[len(ts) // 2], "min_ms": ts[0], "checksum": float(out[:n].double().sum()), "bits": int(out[:n].view(torch.int32).to(torch.int64).sum())}))
'''


def main():
    workload = sys.argv[1] if len(sys.argv) > 1 else "cfg5"
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    tags = sys.argv[3:] or ["r02", "r03", "now"]
    res = {t: [] for t in tags}
    env = dict(os.environ, SDFK_CACHE_DIR="off")
    for r in range(rounds):
        for t in tags:
            root = ROOT if t == "now" else os.path.join(ROOT, "tools", "_build", t)
            p = subprocess.run([sys.executable, "-c", CHILD, root, workload], capture_output=True, text=True, env=env, timeout=300)
            line = [x for x in p.stdout.splitlines() if x.startswith("{")]
            if not line:
                print(t, "FAILED", p.stderr[-400:], flush=True)
                continue
            d = json.loads(line[-1])
            res[t].append(d)
            print("round %d %-9s median %.3f min %.3f ms  checksum %.9e  bit sum %d" % (r, t, d["median_ms"], d["min_ms"], d["checksum"], d.get("bits", 0)), flush=True)
    n = 1076890625
    for t in tags:
        if res[t]:
            m = sorted(x["median_ms"] for x in res[t])[len(res[t]) // 2]
            print("%-4s %s: median of medians %.3f ms = %.3f of 8 TB/s" % (t, workload, m, 16.0 * n / (m * 1e-3) / 8e12))


if __name__ == "__main__":
    main()
