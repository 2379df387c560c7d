"""Developer tool (GPU): an n-ary UNION of many spheres on a resident grid — chain mode (culled row blocks, un-culled)
against the interpreter kernel.   python tools/big_union_bench.py [--spheres 1000] [--grid 512] [--json out.json]
--body: a box MINUS that union (lowered as one n-ary INTERSECT of the box and the negated spheres).
--clip / --blend: the union as an operand of a larger program (INTERSECT2 with a sphere / SMOOTH_UNION2 with a slab, rotated):
the union runs as a chain, the rest of the program per point around its value.
--groups G: the same number of spheres as G rigidly placed clusters, each a UNION of its own (nested unions: flattened
into one chain by the lowering); the NESTED program (SDFK_NO_FLATTEN=1, what round 2 ran: the interpreter kernel beyond
the specialisation limit) is timed next to it (the two differ by fp32 rounding: the flattened program composes the group's
and the member's transforms into one map)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--spheres", type=int, default=1000)
    ap.add_argument("--grid", type=int, default=512)
    ap.add_argument("--json", default=None)
    ap.add_argument("--groups", type=int, default=0)
    ap.add_argument("--body", action="store_true", help="a block MINUS the union (porous block): one INTERSECT chain after lowering")
    ap.add_argument("--clip", action="store_true", help="the union clipped by a sphere (INTERSECT2): chain + rest of the program")
    ap.add_argument("--blend", action="store_true", help="the union blended with a ground slab (SMOOTH_UNION2): chain + rest")
    ap.add_argument("--no-interp", action="store_true", help="skip the interpreter kernel (seconds per launch for thousands of members)")
    args = ap.parse_args()
    import torch
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, workloads
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    if args.groups:
        tree = workloads.clustered_union(ns, args.groups, args.spheres // args.groups)
    else:
        tree = workloads.sphere_union(ns, args.spheres)
    if args.body:
        tree = ns.CombineGeometry("SUBTRACT2").combine(ns.Box(1.7, 1.7, 1.7), tree)
    if args.clip:
        tree = ns.CombineGeometry("INTERSECT2").combine(tree, ns.Sphere(0.85))
    if args.blend:
        slab = ns.Box(2.0, 2.0, 0.3)
        slab.move((0, 0, -0.7))
        tree = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(slab, tree, parameters=0.1)
        tree.rotate(0.2, (1, 0, 0))
    t0 = time.perf_counter()
    low = lower_geometry(tree)
    prog = _engine.Program.from_lowered(low)
    t_lower = time.perf_counter() - t0
    nested = None
    if args.groups:
        os.environ["SDFK_NO_FLATTEN"] = "1"
        nested = _engine.Program.from_lowered(lower_geometry(tree))
        del os.environ["SDFK_NO_FLATTEN"]
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (args.grid,) * 3)[0]]
    n = int(np.prod([a.size for a in axes]))
    stride = (n + 255) // 256 * 256
    dev = torch.device("cuda", 0)
    co = torch.empty((3, stride), dtype=torch.float32, device=dev)
    outs = {k: torch.empty((stride,), dtype=torch.float32, device=dev) for k in ("culled", "plain", "interp")}
    stream = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=stream)
    row_len = int(axes[2].size)
    what = ("UNION of %d clusters, each a UNION of %d spheres" % (args.groups, args.spheres // args.groups)) if args.groups \
        else "n-ary UNION of %d spheres" % args.spheres
    res = {"workload": ("a box minus the " if args.body else "") + what + (" clipped by a sphere" if args.clip else "")
           + (" blended with a ground slab, rotated" if args.blend else ""), "chain_members": prog.chain_members, "instructions": int(low.code.shape[0]),
           "cull_sites": int(len(low.cull_sites)), "grid": "%d^3" % axes[0].size, "points": n,
           "chain_mode": "#define SDFK_CHAIN 1" in prog.source(), "lower_and_program_s": t_lower}
    for key, mode, rows, reps in (("culled", _engine.MODE_SPECIALIZED, True, 5), ("plain", _engine.MODE_NOCULL, False, 2),
                                  ("interp", _engine.MODE_INTERPRET, False, 1))[:2 if args.no_interp else 3]:
        def step():
            prog.eval_device(co.data_ptr(), n, stride, outs[key].data_ptr(), stream=stream, mode=mode,
                             row_len=row_len if rows else None, plane_rows=int(axes[1].size) if rows else None)
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        best = 1e30
        for _ in range(reps):
            e0, e1 = _engine.Event(), _engine.Event()
            e0.record(stream)
            step()
            e1.record(stream)
            best = min(best, e0.elapsed_ms(e1))
        res[key] = {"ms": best, "first_call_s": first, "mpoints_per_s": n / best / 1e3,
                    "frac_of_hbm_roofline": 16.0 * n / (best * 1e-3) / 8e12}
        print(key, res[key], flush=True)
    if nested is not None:
        out_n = torch.empty((stride,), dtype=torch.float32, device=dev)
        t0 = time.perf_counter()
        nested.eval_device(co.data_ptr(), n, stride, out_n.data_ptr(), stream=stream, mode=_engine.MODE_AUTO, row_len=row_len)
        torch.cuda.synchronize()
        first = time.perf_counter() - t0
        e0, e1 = _engine.Event(), _engine.Event()
        e0.record(stream)
        nested.eval_device(co.data_ptr(), n, stride, out_n.data_ptr(), stream=stream, mode=_engine.MODE_AUTO, row_len=row_len)
        e1.record(stream)
        res["nested_program_auto_mode"] = {"ms": e0.elapsed_ms(e1), "first_call_s": first, "chain_members": nested.chain_members,
                                           # (the flattened program composes group and member transforms into one map)
                                           "max_abs_difference_to_flattened": float((out_n[:n] - outs["culled"][:n]).abs().max())}
        print("nested", res["nested_program_auto_mode"], flush=True)
    res["bit_identical"] = bool(torch.equal(outs["culled"][:n], outs["plain"][:n]) and
                                (args.no_interp or torch.equal(outs["plain"][:n], outs["interp"][:n])))
    print(json.dumps(res))
    if args.json:
        json.dump(res, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
