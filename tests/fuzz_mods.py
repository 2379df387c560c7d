"""Test-infrastructure script (not collected by pytest): random chains drawn from ALL pointwise modification methods
(domain warps, value maps, second-field operations, instancing, post-processing) on random primitives.

  * here, in the build container (no GPU): `python tests/fuzz_mods.py reference [first] [count]` compares the float64
    ORACLE with the REAL reference (/root/reference) — this validates the oracle far beyond the golden scenes;
  * on the GPU box: `python tests/fuzz_mods.py gpu [first] [count]` compares the GPU evaluation with the oracle.
Discontinuous results (sign, binarisation, repetition / instancing cell borders) are compared with the count rule of
tests/test_gpu_parity.py.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _curve(t, r, p):
    return np.asarray((r * np.cos(t), r * np.sin(t), p * t))


def build(ns, scenes, seed):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    obj = scenes._random_leaf(ns, rng)
    names = []
    for _ in range(int(rng.integers(1, 5))):
        m = int(rng.integers(0, 40))
        names.append(m)
        if m == 0: obj.elongation((u(0, 0.4), u(0, 0.3), u(0, 0.2)))
        elif m == 1: obj.rounding(u(0.0, 0.08))
        elif m == 2: obj.rounding_cs(u(0.01, 0.05), u(0.8, 1.5))
        elif m == 3: obj.boundary()
        elif m == 4: obj.invert()
        elif m == 5: obj.sign()
        elif m == 6: obj.recover_volume(ns.Sphere(u(0.4, 0.9)).propagate)
        elif m == 7: obj.define_volume(ns.sdf_sphere, (u(0.4, 0.9),))
        elif m == 8: obj.onion(u(0.01, 0.06))
        elif m == 9: obj.concentric(u(0.05, 0.2))
        elif m == 10: obj.revolution(u(0.2, 0.6))
        elif m == 11: obj.axis_revolution(u(0.2, 0.6), u(-1.0, 1.0))
        elif m == 12: obj.extrusion(u(0.2, 0.8))
        elif m == 13: obj.twist(u(-1.5, 1.5))
        elif m == 14: obj.bend(u(0.8, 2.0), u(0.3, 1.5))
        elif m == 15: getattr(obj, str(rng.choice(["shear_xz", "shear_yz", "shear_xy", "shear_zy", "shear_yx", "shear_zx"])))(u(-0.5, 0.5))
        elif m == 16:
            i = int(rng.integers(0, 3))
            obj.shear(u(-0.5, 0.5), i, (i + 1 + int(rng.integers(0, 2))) % 3)
        elif m == 17: obj.displacement(ns.sdf_y, (u(-0.2, 0.2),))
        elif m == 18: obj.infinite_repetition((u(0.8, 1.5), u(0.8, 1.5), u(0.8, 1.5)))
        elif m == 19: obj.finite_repetition((u(1.5, 2.5), u(1.5, 2.5), u(1.5, 2.5)), (int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 4))))
        elif m == 20: obj.finite_repetition_rescaled((u(1.5, 2.5),) * 3, (2, 3, 2), (u(0.8, 1.2),) * 3, (u(0.0, 0.2),) * 3)
        elif m == 21: obj.symmetry(int(rng.integers(0, 3)))
        elif m == 22: obj.mirror((u(-0.6, -0.1), u(-0.3, 0.3), u(-0.2, 0.2)), (u(0.1, 0.6), u(-0.3, 0.3), u(-0.2, 0.2)))
        elif m == 23: obj.rotational_symmetry(int(rng.integers(2, 9)), u(0.0, 0.6), u(0.0, 1.0))
        elif m == 24: obj.linear_instancing(int(rng.integers(2, 6)), (u(-0.8, -0.2), u(-0.3, 0.3), u(-0.2, 0.2)), (u(0.2, 0.8), u(-0.3, 0.3), u(-0.2, 0.2)))
        elif m == 25: obj.curve_instancing(_curve, (u(0.3, 0.7), u(0.02, 0.1)), (0.0, u(2.0, 6.0), int(rng.integers(3, 12))))
        elif m == 26: obj.aligned_curve_instancing(_curve, (u(0.3, 0.7), u(0.02, 0.1)), (0.0, u(2.0, 6.0), int(rng.integers(3, 12))))
        elif m == 27: obj.fully_aligned_curve_instancing(_curve, (u(0.3, 0.7), u(0.02, 0.1)), (0.0, u(2.0, 6.0), int(rng.integers(3, 12))))
        elif m == 28: obj.move_sdf((u(-0.3, 0.3), u(-0.3, 0.3), u(-0.3, 0.3)))
        elif m == 29: obj.scale_sdf(u(0.6, 1.5))
        elif m == 30:
            from scipy.spatial.transform import Rotation
            obj.rotate_sdf(Rotation.from_rotvec(rng.normal(0, 1, 3)).as_matrix())
        elif m == 31: obj.sigmoid_falloff(u(0.5, 2.0), u(0.1, 0.5))
        elif m == 32: obj.positive_sigmoid_falloff(u(0.5, 2.0), u(0.1, 0.5))
        elif m == 33: obj.capped_exponential(u(0.5, 2.0), u(0.1, 0.5))
        elif m == 34: obj.hard_binarization(u(-0.1, 0.1))
        elif m == 35: obj.linear_falloff(u(0.5, 2.0), u(0.1, 0.5))
        elif m == 36: obj.relu(u(0.2, 1.0))
        elif m == 37: obj.smooth_relu(u(0.05, 0.3), u(0.5, 1.5), u(0.005, 0.02))
        elif m == 38: obj.slowstart(u(0.05, 0.3), u(0.5, 1.5), u(0.005, 0.02), bool(rng.integers(0, 2)))
        else: getattr(obj, str(rng.choice(["gaussian_boundary", "gaussian_falloff"])))(u(0.5, 2.0), u(0.1, 0.5))
    scenes._random_place(obj, rng)
    return obj, names


def compare(tag, a, b, tol):
    err = np.abs(a - b) / np.maximum(1.0, np.abs(b))
    err[np.isnan(a) & np.isnan(b)] = 0
    bad = int((~(err <= tol)).sum())
    return bad, float(np.nanmax(err)) if err.size else 0.0


def main(which="reference", first=0, count=200):
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
        try:
            with np.errstate(all="ignore"):
                want = np.asarray(sdf_oracle.evaluate(build(ns, scenes, seed)[0], co.copy()), dtype=np.float64)
                if which == "reference":
                    obj, names = build(ref, scenes, seed)
                    got = np.asarray(obj.create(co.copy()), dtype=np.float64)
                    bad, worst = compare("ref", want, got, 1e-11)
                    ok = bad == 0
                else:
                    obj, names = build(ns, scenes, seed)
                    got = obj.create(co.copy()).astype(np.float64)
                    # judged against the largest intermediate of the tree at the point, like tests/test_gpu_parity.py: two
                    # displacements add fields of size 1-2 to a value near 0 (seed 99668: 21 points at 1.4-1.7e-6 of the
                    # RESULT, 2e-7 of the operands — the same with the round-3 library)
                    _, mag = sdf_oracle.evaluate_with_magnitude(build(ns, scenes, seed)[0], co.copy())
                    err = np.abs(got - want) / np.maximum(np.maximum(1.0, np.abs(want)), mag)
                    err[np.isnan(got) & np.isnan(want)] = 0
                    bad, worst = int((~(err <= 1e-6)).sum()), float(np.nanmax(err)) if err.size else 0.0
                    if bad > max(1, int(0.005 * want.size)):
                        # steep value maps (gaussian / exponential of a small width): discount the points where the
                        # reference itself moves as much under a one-ulp change of its fp32 input
                        sens = scenes.input_sensitivity(lambda c: sdf_oracle.evaluate(build(ns, scenes, seed)[0], c), co)
                        err = np.abs(got - want)
                        off = ~(err / np.maximum(1.0, np.abs(want)) <= 1e-6) & ~(err <= 8.0 * sens)
                        off &= ~(np.isnan(got) & np.isnan(want))
                        names = names + ["%d of %d off points within 8x the input sensitivity" % (bad - int(off.sum()), bad)]
                        bad = int(off.sum())
                    ok = bad <= max(1, int(0.005 * want.size))
        except Exception as exc:  # noqa: BLE001
            names, ok, bad, worst = "?", False, -1, float("nan")
            print("seed %d raised %r" % (seed, exc))
        failures += not ok
        print("seed %d mods %s: %d off, worst %.2e %s" % (seed, names, bad, worst, "" if ok else " <-- FAIL"), flush=True)
    print("%s: %d cases, %d failures" % (which, int(count), failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
