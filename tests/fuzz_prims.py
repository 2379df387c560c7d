"""Test-infrastructure script (not collected by pytest): every geometry class with RANDOM constructor arguments.

  python tests/fuzz_prims.py reference [first] [count]   build container: float64 oracle vs the REAL reference
  python tests/fuzz_prims.py gpu [first] [count]         GPU box: GPU evaluation vs the oracle
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _helix(t, r, p):
    return np.asarray((r * np.cos(t), r * np.sin(t), p * t))


def _ellipse(t, a, b):
    return np.asarray((a * np.cos(t), b * np.sin(t)))


JUMPY = {"Cone", "OrientedInfiniteCone", "SolidAngle", "Triangle", "Sector", "InfiniteSector", "NGon", "Polygon"}


def build(ns, seed):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    v3 = lambda s=1.0: tuple(float(x) for x in rng.uniform(-s, s, 3))      # noqa: E731
    v2 = lambda s=1.0: np.asarray(rng.uniform(-s, s, 2))                    # noqa: E731
    g3, g2 = ns.geom_3d, ns.geom_2d
    kinds = ["X", "Y", "Z", "InfiniteCylinder", "Cylinder", "Sphere", "Box", "Plane", "OrientedPlane", "Line", "Triangle3D",
             "Quad", "Torus", "ChainLink", "Braid", "Arc3D", "Cone", "InfiniteCone", "OrientedInfiniteCone", "SolidAngle",
             "ParametricCurve3D", "SegmentedParametricCurve3D", "SegmentedLine3D", "PointCloud3D",
             "Circle", "NEUCircle", "NGon", "Rectangle", "RoundedRectangle", "Segment", "Triangle", "Sector",
             "InfiniteSector", "Arc", "Polygon", "ParametricCurve", "SegmentedParametricCurve", "SegmentedLine", "PointCloud2D"]
    k = kinds[seed % len(kinds)]
    if k in ("X", "Y", "Z"): obj = getattr(g3, k)(u(-0.5, 0.5))
    elif k == "InfiniteCylinder": obj = g3.InfiniteCylinder(u(0.1, 0.9))
    elif k == "Cylinder": obj = g3.Cylinder(u(0.1, 0.8), u(0.1, 1.5))
    elif k == "Sphere": obj = g3.Sphere(u(0.05, 1.2))
    elif k == "Box": obj = g3.Box(u(0.1, 1.5), u(0.1, 1.5), u(0.1, 1.5))
    elif k == "Plane": obj = g3.Plane(v3(), u(0.01, 0.5))
    elif k == "OrientedPlane": obj = g3.OrientedPlane(v3(), u(-0.5, 0.5))
    elif k == "Line": obj = g3.Line(v3(), v3())
    elif k == "Triangle3D": obj = g3.Triangle3D(v3(), v3(), v3())
    elif k == "Quad":
        a, e1, e2 = np.asarray(v3(0.5)), np.asarray(v3(0.8)), np.asarray(v3(0.8))
        obj = g3.Quad(a, a + e1, a + e1 + e2, a + e2)
    elif k == "Torus": obj = g3.Torus(u(0.3, 0.9), u(0.02, 0.25))
    elif k == "ChainLink": obj = g3.ChainLink(u(0.2, 0.6), u(0.03, 0.15), u(0.1, 1.0))
    elif k == "Braid": obj = g3.Braid(u(0.4, 1.6), u(0.1, 0.4), u(0.03, 0.12), u(1.0, 8.0))
    elif k == "Arc3D": obj = g3.Arc3D(u(0.3, 0.8), u(0.03, 0.2), u(-3.0, 1.0), u(1.0, 6.0))
    elif k == "Cone": obj = g3.Cone(u(0.3, 1.2), u(0.1, 1.2))
    elif k == "InfiniteCone": obj = g3.InfiniteCone(u(0.1, 1.4))
    elif k == "OrientedInfiniteCone": obj = g3.OrientedInfiniteCone(u(0.1, 1.4))
    elif k == "SolidAngle": obj = g3.SolidAngle(u(0.3, 1.0), u(-1.0, 0.5), u(0.6, 2.5))
    elif k == "ParametricCurve3D": obj = g3.ParametricCurve3D(_helix, (u(0.3, 0.8), u(0.02, 0.15)), (u(-3, 0), u(1, 6), int(rng.integers(5, 60))), closed=bool(rng.integers(0, 2)))
    elif k == "SegmentedParametricCurve3D": obj = g3.SegmentedParametricCurve3D(rng.uniform(-1, 1, (3, int(rng.integers(3, 9)))), (0, u(1.0, 2.0), int(rng.integers(4, 40))), closed=bool(rng.integers(0, 2)))
    elif k == "SegmentedLine3D": obj = g3.SegmentedLine3D(rng.uniform(-1, 1, (3, int(rng.integers(2, 9)))), closed=True)
    elif k == "PointCloud3D": obj = g3.PointCloud3D(rng.uniform(-1, 1, (3, int(rng.integers(1, 600)))))
    elif k == "Circle": obj = g2.Circle(u(0.05, 1.2))
    elif k == "NEUCircle": obj = g2.NEUCircle(u(0.2, 1.0), float(rng.choice([0.5, 1, 1.5, 2, 3, 7.5, np.inf])))
    elif k == "NGon": obj = g2.NGon(u(0.2, 1.0), int(rng.integers(3, 24)))
    elif k == "Rectangle": obj = g2.Rectangle(u(0.1, 1.6), u(0.1, 1.6))
    elif k == "RoundedRectangle": obj = g2.RoundedRectangle(u(0.6, 1.6), u(0.6, 1.6), tuple(float(x) for x in rng.uniform(0, 0.25, 4)))
    elif k == "Segment": obj = g2.Segment(v3(), v3())
    elif k == "Triangle": obj = g2.Triangle(v2(), v2(), v2())
    elif k == "Sector": obj = g2.Sector(u(0.3, 1.0), u(-1.0, 0.5), u(0.6, 2.5))
    elif k == "InfiniteSector": obj = g2.InfiniteSector(u(-1.0, 0.5), u(0.6, 2.5))
    elif k == "Arc": obj = g2.Arc(u(0.3, 0.9), u(-3.0, 1.0), u(1.0, 6.0))
    elif k == "Polygon":
        m = int(rng.integers(3, 9))
        ang = np.sort(rng.uniform(0, 2 * np.pi, m))
        rad = rng.uniform(0.3, 1.0, m)
        obj = g2.Polygon(np.stack([rad * np.cos(ang), rad * np.sin(ang), 0 * ang]))   # star-shaped: simple, maybe concave
    elif k == "ParametricCurve": obj = g2.ParametricCurve(_ellipse, (u(0.3, 0.9), u(0.2, 0.7)), (0, u(2.0, 6.28), int(rng.integers(5, 60))), closed=bool(rng.integers(0, 2)))
    elif k == "SegmentedParametricCurve": obj = g2.SegmentedParametricCurve(rng.uniform(-1, 1, (3, int(rng.integers(3, 9)))), (0, u(1.0, 2.0), int(rng.integers(4, 40))), closed=bool(rng.integers(0, 2)))
    elif k == "SegmentedLine": obj = g2.SegmentedLine(rng.uniform(-1, 1, (3, int(rng.integers(2, 9)))), closed=bool(rng.integers(0, 2)))
    else: obj = g2.PointCloud2D(rng.uniform(-1, 1, (3, int(rng.integers(1, 600)))))
    t = int(rng.integers(0, 4))
    if t >= 1: obj.move(v3(0.5))
    if t >= 2: obj.rotate(u(0, np.pi), rng.normal(0, 1, 3))
    if t == 3: obj.set_scale(u(0.6, 1.6))
    return obj, k


def main(which="reference", first=0, count=400):
    import scenes
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    co = scenes.input_points()
    if which == "reference":
        sys.path.insert(0, "/root/reference/Code/spomso")
        sys.dont_write_bytecode = True
        import spomso.cores as ref
    failures = 0
    for seed in range(int(first), int(first) + int(count)):
        kind = "?"
        try:
            with np.errstate(all="ignore"):
                o, kind = build(ns, seed)
                want = np.asarray(sdf_oracle.evaluate(o, co.copy()), dtype=np.float64)
                if which == "reference":
                    got = np.asarray(build(ref, seed)[0].create(co.copy()), dtype=np.float64)
                    tol, allowed = 1e-11, 0
                else:
                    got = build(ns, seed)[0].create(co.copy()).astype(np.float64)
                    tol, allowed = 1e-6, (max(1, int(0.005 * want.size)) if kind in JUMPY else 0)
            err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
            err[np.isnan(got) & np.isnan(want)] = 0
            off = ~(err <= tol)
            note = ""
            if which != "reference" and off.any():
                # points where the reference itself moves as much under a one-ulp change of its fp32 input
                sens = scenes.input_sensitivity(lambda c: sdf_oracle.evaluate(build(ns, seed)[0], c), co)
                conditioned = off & (np.abs(got - want) <= 8.0 * sens)
                note = " (%d more within 8x the input sensitivity)" % conditioned.sum() if conditioned.any() else ""
                off &= ~conditioned
            bad, worst = int(off.sum()), float(np.nanmax(err))
            ok = bad <= allowed
            kind += note
        except Exception as exc:  # noqa: BLE001
            ok, bad, worst = False, -1, float("nan")
            print("seed %d (%s) raised %r" % (seed, kind, exc))
            if which != "reference":
                # an input the reference refuses (e.g. a (3, 2) point array, which its constructor transposes): the
                # product has to refuse it with the same exception type
                try:
                    build(ns, seed)[0].create(co.copy())
                except type(exc):
                    ok, bad, kind = True, 0, kind + " (refused by both with %s)" % type(exc).__name__
                except Exception:  # noqa: BLE001
                    pass
        failures += not ok
        print("seed %d %s: %d off, worst %.2e %s" % (seed, kind, bad, worst, "" if ok else " <-- FAIL"), flush=True)
    print("%s: %d cases, %d failures" % (which, int(count), failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
