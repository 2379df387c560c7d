"""Developer tool (GPU): nearest-point distance to a 16,384-point cloud on a 257^3 grid, box tree vs full scan."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(res=256, m=16384, cloud_kind="synthetic"):
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine, _prims
    from aegolius_amd._lower import lower_geometry
    from aegolius_amd.cores.helper_functions import grid_axes
    rng = np.random.default_rng(1)
    # a terrain-like sheet: z = f(x, y) sampled on a jittered lattice
    xy = rng.uniform(-1, 1, (2, m))
    cloud = np.stack([xy[0], xy[1], 0.3 * np.sin(3 * xy[0]) * np.cos(2 * xy[1])])
    if cloud_kind == "terrain":                                 # the reference's own cloud (tests/golden/terrain_lr_cloud.npz)
        cloud = np.load(os.path.join(ROOT, "tests", "golden", "terrain_lr_cloud.npz"))["cloud"]
        m = cloud.shape[1]
    axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (res,) * 3)[0]]
    n = axes[0].size * axes[1].size * axes[2].size
    lib = _engine.lib()
    d_out = lib.sdfk_malloc(n * 4)
    out = {"grid": [int(a.size) for a in axes], "cloud_points": int(m), "cloud": cloud_kind}
    fields = {}
    for label, thr in (("box_tree", 256), ("full_scan", 1 << 30)):
        _prims.TREE_THRESHOLD = thr
        prog = _engine.Program.from_lowered(lower_geometry(ns.geom_3d.PointCloud3D(cloud)))
        prog.eval_grid(axes, 0, n, d_out)                     # compile + warm-up
        t0 = time.perf_counter()
        prog.eval_grid(axes, 0, n, d_out)
        ms = (time.perf_counter() - t0) * 1e3
        host = np.empty(n, dtype=np.float32)
        _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(host), _engine._vp(d_out), n * 4), "d2h")
        fields[label] = host
        out[label] = {"ms": ms, "mpoints_per_s": n / ms / 1e3}
    out["bit_identical"] = bool(np.array_equal(fields["box_tree"], fields["full_scan"]))
    out["speedup"] = out["full_scan"]["ms"] / out["box_tree"]["ms"]

    # curve_instancing with 4,096 instances along a helix: nearest INSTANCE through the same tree
    def helix(t, r, p):
        return np.asarray((r * np.cos(t), r * np.sin(t), p * t))
    inst = {}
    for label, thr in (("box_tree", 256), ("full_scan", 1 << 30)):
        _prims.TREE_THRESHOLD = thr
        s = ns.Sphere(0.004)
        s.fully_aligned_curve_instancing(helix, (0.7, 0.004), (-200.0, 200.0, 4096))
        prog = _engine.Program.from_lowered(lower_geometry(s))
        prog.eval_grid(axes, 0, n, d_out)
        t0 = time.perf_counter()
        prog.eval_grid(axes, 0, n, d_out)
        ms = (time.perf_counter() - t0) * 1e3
        host = np.empty(n, dtype=np.float32)
        _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(host), _engine._vp(d_out), n * 4), "d2h")
        inst[label] = host
        out["instancing_4096_" + label] = {"ms": ms, "mpoints_per_s": n / ms / 1e3}
    out["instancing_points_that_differ"] = int(np.count_nonzero(inst["box_tree"] != inst["full_scan"]))
    out["instancing_speedup"] = out["instancing_4096_full_scan"]["ms"] / out["instancing_4096_box_tree"]["ms"]
    _prims.TREE_THRESHOLD = 256
    lib.sdfk_free(_engine._vp(d_out))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(*[int(a) if a.isdigit() else a for a in sys.argv[1:]])
