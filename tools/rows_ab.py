"""Developer tool (GPU): A/B timing of kernel variants of one workload on one resident grid, in ONE process.

    python tools/rows_ab.py [--workload cfg2] [--grid 1024] [--reps 10] VARIANT [VARIANT ...]

A VARIANT is a string of -D switches for the generated source ("" = the shipped kernel), optionally prefixed by
`mode=nocull:` / `mode=interpret:` / `norows:` / `noplanes:` / `grid:` (sdfk_eval_grid) / `extra=-mllvm,-disable-cgp:` (raw compiler options); every variant is built as its own code object
(sdfk_debug_set_rtc_defs), timed with HIP events on the launch stream, and — unless it contains ABLATE — compared
bit for bit with the un-culled kernel's field."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--grid", type=int, default=1024)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--json", default=None)
    ap.add_argument("variants", nargs="*", default=[""])
    args = ap.parse_args()
    os.environ.setdefault("SDFK_PLANE_BLOCKS", "1")        # (`noplanes:` variants simply pass no plane hint)
    import torch
    import bench
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    lib = _engine.lib()
    tree, size, desc = bench.build_workload(args.workload, ns)
    axes64, _res = grid_axes(size, (args.grid,) * len(size))
    axes = [a.astype(np.float32) for a in axes64]
    n = int(axes[0].size) * int(axes[1].size) * int(axes[2].size)
    row_len = int(axes[2].size) if axes[2].size > 1 else int(axes[1].size)
    stride = (n + 255) // 256 * 256
    dev = torch.device("cuda", 0)
    co = torch.empty((3, stride), dtype=torch.float32, device=dev)
    out = torch.empty((stride,), dtype=torch.float32, device=dev)
    ref = torch.empty((stride,), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    _engine.grid_fill(co.data_ptr(), stride, axes, 0, n, stream=stream)
    low = lower_geometry(tree)
    prog = _engine.Program.from_lowered(low)
    prog.eval_device(co.data_ptr(), n, stride, ref.data_ptr(), stream=stream, mode=_engine.MODE_NOCULL)
    torch.cuda.synchronize()
    results = []
    for var in args.variants:
        mode, rows, planes, defs = _engine.MODE_SPECIALIZED, True, True, var
        gridpath = False
        os.environ.pop("SDFK_RTC_EXTRA", None)
        while ":" in defs:
            head, defs = defs.split(":", 1)
            if head == "norows":
                rows = False
            elif head == "noplanes":
                planes = False
            elif head == "grid":                                # the 4 B/point path: coordinates from the axis tables
                gridpath = True
            elif head.startswith("extra="):                     # raw compiler options for this variant, "," for " "
                os.environ["SDFK_RTC_EXTRA"] = head[6:].replace(",", " ")
            elif head.startswith("mode="):
                mode = {"nocull": _engine.MODE_NOCULL, "interpret": _engine.MODE_INTERPRET}[head[5:]]
        defs = " ".join("-DSDFK_" + t for t in defs.split("+") if t and t != "base")
        lib.sdfk_debug_set_rtc_defs(defs.encode())
        p = _engine.Program.from_lowered(low)

        def step():
            if gridpath:
                return p.eval_grid(axes, 0, n, out.data_ptr(), stream=stream, mode=mode)
            p.eval_device(co.data_ptr(), n, stride, out.data_ptr(), stream=stream, mode=mode, row_len=row_len if rows else None,
                          flat=axes[2].size == 1, plane_rows=int(axes[1].size) if planes and axes[2].size > 1 else None)
        out.zero_()
        try:
            step()
            torch.cuda.synchronize()
        except Exception as exc:  # noqa: BLE001
            print("%-60s FAILED %r" % (var, exc), flush=True)
            continue
        exact = None
        if not any(k in defs for k in ("ABLATE_EVAL", "ABLATE_STORE", "ABLATE_EDGE", "ABLATE_BARRIER")):
            exact = bool(torch.equal(out[:n].view(torch.int32), ref[:n].view(torch.int32)) or
                         torch.equal(out[:n], ref[:n]))            # (flat grids: the sign of a zero may differ)
        # >= 30 ms of warm-up, then >= args.reps launches and >= 60 ms of them (sub-millisecond kernels: the clocks settle)
        w0, w1 = _engine.Event(), _engine.Event()
        w0.record(stream)
        step()
        w1.record(stream)
        one = max(w0.elapsed_ms(w1), 1e-3)
        for _ in range(int(min(200, 30.0 / one))):
            step()
        times = []
        for _ in range(int(max(args.reps, min(300, 60.0 / one)))):
            e0, e1 = _engine.Event(), _engine.Event()
            e0.record(stream)
            step()
            e1.record(stream)
            times.append(e0.elapsed_ms(e1))
        times.sort()
        rec = {"variant": var, "min_ms": times[0], "median_ms": times[len(times) // 2], "bit_exact": exact,
               "frac_of_8TBs": 16.0 * n / (times[len(times) // 2] * 1e-3) / 8e12}
        results.append(rec)
        print("%-60s min %.3f  median %.3f ms  frac %.3f  exact %s" % (var or "(shipped)", rec["min_ms"], rec["median_ms"],
                                                                        rec["frac_of_8TBs"], exact), flush=True)
    lib.sdfk_debug_set_rtc_defs(b"")
    if args.json:
        json.dump({"workload": desc, "points": n, "results": results}, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
