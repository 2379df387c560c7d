"""Developer tool: evaluate every scene on the GPU (both kernel flavours) and print the deviation
from the golden vectors. Test infrastructure (it may use the oracle), not collected by pytest: python tests/report_gpu_parity.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # tests/ -> repo root
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import scenes  # noqa: E402
import aegolius_amd  # noqa: E402
import aegolius_amd.cores as ns  # noqa: E402
from aegolius_amd import _engine  # noqa: E402


def grid_scene_counts(g):
    """off-point counts of the grid-neighbourhood scenes, judged exactly like tests/test_gpu_parity.py does"""
    from oracle import sdf_oracle
    counts = {}
    for name, (build, key) in scenes.GRID_SCENES.items():
        co, res = scenes.grid_inputs(ns, key)
        tagged, _ = ns.generate_grid(*scenes.GRIDS[key])
        ref = g["gridscene/" + name]
        magnitude = None
        if "edge_detection" in name:
            with np.errstate(all="ignore"):
                _, magnitude = sdf_oracle.evaluate_with_magnitude(build(ns, res), co)
        worst = 0
        for mode in (_engine.MODE_SPECIALIZED, _engine.MODE_INTERPRET):
            aegolius_amd.config.mode = mode
            for arr in (co.copy(), tagged):
                out = build(ns, res).create(arr).astype(np.float64).ravel()
                scale = np.maximum(1.0, np.abs(ref.ravel()))
                if magnitude is not None:
                    scale = np.maximum(scale, magnitude.ravel())
                err = np.abs(out - ref.ravel()) / scale
                err[np.isnan(ref.ravel()) & np.isnan(out)] = 0
                worst = max(worst, int((~(err <= 1e-6)).sum()))
        counts[name] = worst
        print("%-42s bad %4d" % (name, worst), flush=True)
    aegolius_amd.config.mode = 0
    return counts


def main():
    g = np.load(os.path.join(ROOT, "tests", "golden", "golden_scenes.npz"))
    co = g["inputs"].astype(np.float64)
    worst = []
    budget = {"scenes": {}, "grid_scenes": {}}
    t0 = time.time()
    for name, build in scenes.SCENES.items():
        ref = g["scene/" + name]
        row = [name]
        outs = []
        for mode in (_engine.MODE_INTERPRET, _engine.MODE_SPECIALIZED):
            aegolius_amd.config.mode = mode
            try:
                out = build(ns).create(co).astype(np.float64)
            except Exception as exc:  # noqa: BLE001
                row.append("EXC " + repr(exc)[:120])
                continue
            outs.append(out)
            both_nan = np.isnan(ref) & np.isnan(out)
            err = np.abs(out - ref) / np.maximum(1.0, np.abs(ref))
            err[both_nan] = 0
            nbad = int((~(err <= 1e-6)).sum())
            row.append("max %.2e bad %4d" % (np.nanmax(err), nbad))
            if not name.startswith("random_tree_"):
                budget["scenes"][name] = max(budget["scenes"].get(name, 0), nbad)
        if len(outs) == 2:
            same = np.array_equal(outs[0], outs[1], equal_nan=True)
            row.append("interp==spec" if same else "INTERP!=SPEC (%d)" % int((outs[0] != outs[1]).sum()))
        if name.startswith("random_tree_") and outs:
            from oracle import sdf_oracle
            _, mag = sdf_oracle.evaluate_with_magnitude(build(ns), co)
            e = np.abs(outs[-1] - ref)
            e2 = e / np.maximum(np.maximum(1.0, np.abs(ref)), mag)
            row.append("scaled max %.2e bad %d (magnitude max %.1f)" % (np.nanmax(e2), int((~(e2 <= 1e-6)).sum()),
                                                                       mag.max()))
            budget["scenes"][name] = int((~(e2 <= 1e-6)).sum())       # (the test scales random trees like this)
            for i in np.argsort(-e / np.maximum(1.0, np.abs(ref)))[:3]:
                row.append("[ref %.4f err %.2e mag %.2f]" % (ref[i], e[i], mag[i]))
        print("%-42s %s" % (row[0], " | ".join(row[1:])), flush=True)
    budget["grid_scenes"] = grid_scene_counts(g)
    # tests/test_gpu_parity.py::test_grid_operators_at_scale: signed + averaging of |sphere| on 129^3
    from oracle import sdf_oracle
    co_s, _res = ns.generate_grid((2, 2, 2), (128, 128, 128))

    def at_scale():
        sph = ns.Sphere(0.6)
        sph.boundary()
        sph.signed((128, 128, 128))
        sph.conv_averaging((3, 3, 3), 1, (128, 128, 128))
        return sph
    got = at_scale().create(co_s).astype(np.float64)
    ref_s = sdf_oracle.evaluate(at_scale(), np.asarray(co_s).astype(np.float32).astype(np.float64))
    budget["grid_operators_at_scale"] = int((~(np.abs(got - ref_s) / np.maximum(1.0, np.abs(ref_s)) <= 1e-6)).sum())
    print("grid_operators_at_scale bad", budget["grid_operators_at_scale"], flush=True)
    budget["scenes"] = {k: v for k, v in sorted(budget["scenes"].items()) if v}        # zero is the default
    budget["grid_scenes"] = {k: v for k, v in sorted(budget["grid_scenes"].items()) if v}
    if "--write-budget" in sys.argv:
        import json
        path = os.path.join(ROOT, "gpurun_out", "parity_budget.json")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(budget, f, indent=1, sort_keys=True)
        print("wrote", path)
    print("total %.1fs" % (time.time() - t0))


if __name__ == "__main__":
    main()
