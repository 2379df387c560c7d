"""Developer tool (GPU): nearest-point distance to the clouds the reference's image examples use (tests/golden/image_clouds.npz:
64,691 / 201,874 / 332,281 / 1,735,884 points) on a flat grid over the cloud's extent — the box tree of §8(f).2 at one
hundred times the terrain cloud's size. `python tools/image_cloud_bench.py [grid] [check_grid]`; the result is compared with
the full scan on a small grid (bit for bit)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(res=2048, check_res=192):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, _prims
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    from example_scenes import image_cloud
    lib = _engine.lib()
    out = {}
    for name in ("lines", "shapes", "owl_exterior", "owl_interior"):
        cloud = image_cloud(name)
        span = float(np.abs(cloud[:2]).max()) * 2.2
        row = {"cloud_points": int(cloud.shape[1])}
        fields = {}
        for label, thr, r in (("box_tree", 256, res), ("box_tree_small", 256, check_res), ("full_scan_small", 1 << 30, check_res)):
            axes = [a.astype(np.float32) for a in grid_axes((span, span), (r, r))[0]]
            n = int(np.prod([a.size for a in axes]))
            _prims.TREE_THRESHOLD = thr
            t0 = time.perf_counter()
            low = lower_geometry(ns.PointCloud2D(cloud))
            row.setdefault("lowering_s", round(time.perf_counter() - t0, 3))
            prog = _engine.Program.from_lowered(low)
            d_out = lib.sdfk_malloc(n * 4)
            prog.eval_grid(axes, 0, n, d_out)                     # build + warm-up
            ts = []
            for _ in range(3 if label != "full_scan_small" else 1):
                t0 = time.perf_counter()
                prog.eval_grid(axes, 0, n, d_out)
                ts.append((time.perf_counter() - t0) * 1e3)
            host = np.empty(n, dtype=np.float32)
            _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(host), _engine._vp(d_out), n * 4), "d2h")
            lib.sdfk_free(_engine._vp(d_out))
            fields[label] = host
            row[label] = {"grid": [int(a.size) for a in axes[:2]], "ms": round(min(ts), 3), "mpoints_per_s": round(n / min(ts) / 1e3, 1)}
        _prims.TREE_THRESHOLD = 256
        row["bit_identical_to_the_scan"] = bool(np.array_equal(fields["box_tree_small"], fields["full_scan_small"]))
        out[name] = row
        print(name, json.dumps(row), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:]])
