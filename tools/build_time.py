"""Developer tool (no GPU needed): hiprtc build time of the specialised kernels against program size, for NON-chain programs
(left-deep SMOOTH_UNION2 chains of n primitives: nothing the table-driven chain mode takes) — what SDFK_SPECIALIZE_LIMIT is
derived from.   python tools/build_time.py [n ...]  ->  one JSON line per size (instructions, seconds and bytes per flavour)"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [10, 30, 60, 100, 150, 200, 300, 400]
    os.environ["SDFK_CACHE_DIR"] = "off"
    import numpy as np
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, workloads
    from aegolius_amd._lower import lower_geometry
    for n in sizes:
        tree = workloads.cfg2_tree(ns, seed=100 + n, count=n)
        low = lower_geometry(tree)
        rec = {"primitives": n, "instructions": int(low.code.shape[0]), "cull_sites": int(len(low.cull_sites))}
        for name, fl in (("plain_array", _engine.FLAVOUR_PLAIN_ARRAY), ("rows_array", _engine.FLAVOUR_ROWS_ARRAY),
                         ("tile_array", _engine.FLAVOUR_TILE_ARRAY)):
            prog = _engine.Program.from_lowered(low)
            t0 = time.perf_counter()
            try:
                size, secs = prog.compile_flavour(fl)
                rec[name] = {"seconds": round(time.perf_counter() - t0, 2), "code_object_bytes": int(size)}
            except Exception as exc:  # noqa: BLE001
                rec[name] = {"error": repr(exc)[:200], "seconds": round(time.perf_counter() - t0, 2)}
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
