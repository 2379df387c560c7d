"""Test-infrastructure script (GPU, uses the oracle; not collected by pytest): random trees with grid-neighbourhood
operators and user callables against the float64 oracle on a small generate_grid cloud.

    python tests/fuzz_staged.py [first_seed] [count]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def build(ns, scenes, rng, res):
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    obj = scenes._random_leaf(ns, rng) if hasattr(scenes, "_random_leaf") else ns.Sphere(u(0.3, 0.7))
    kinds = []
    for _ in range(int(rng.integers(1, 4))):
        k = int(rng.integers(0, 7))
        kinds.append(k)
        if k == 0:
            ks = tuple(int(x) for x in rng.integers(1, 5, 3))
            obj.conv_averaging(ks, int(rng.integers(1, 4)), res)
        elif k == 1:
            obj.conv_averaging(int(rng.integers(2, 5)), 1, res)
        elif k == 2:
            obj.boundary()
            obj.signed(res)
        elif k == 3:
            obj.custom_post_process(scenes._user_tanh, (u(0.3, 1.0), u(0.2, 0.8)))
        elif k == 4:
            obj.rounding(u(0.0, 0.05))
            obj.onion(u(0.01, 0.05))
        elif k == 5:
            obj.displacement(scenes._user_ripple, (u(0.0, 0.05),))
        else:
            obj.boundary()
            obj.signed_old(res)
    scenes._random_place(obj, rng)
    if rng.uniform() < 0.5:
        other = ns.Box(u(0.3, 0.8), u(0.3, 0.8), u(0.3, 0.8))
        other.move((u(-0.4, 0.4), u(-0.4, 0.4), u(-0.4, 0.4)))
        obj = ns.CombineGeometry(str(rng.choice(["UNION2", "INTERSECT2", "SUBTRACT2"]))).combine(obj, other)
    return obj, kinds


def main(first=100, count=40):
    import scenes
    import aegolius_amd.cores as ns
    from oracle import sdf_oracle
    size, res = (2.0, 2.0, 2.0), (14, 12, 10)
    co, _ = ns.generate_grid(size, res)
    co64 = np.asarray(co).astype(np.float32).astype(np.float64)
    failures = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        obj, kinds = build(ns, scenes, rng, res)
        rng = np.random.default_rng(seed)
        obj_o, _ = build(ns, scenes, rng, res)                # a second instance for the oracle (same recipe)
        with np.errstate(all="ignore"):
            ref = np.asarray(sdf_oracle.evaluate(obj_o, co64)).ravel()
        got = np.asarray(obj.create(co)).astype(np.float64).ravel()
        err = np.abs(got - ref) / np.maximum(1.0, np.abs(ref))
        err[np.isnan(ref) & np.isnan(got)] = 0
        nbad = int((~(err <= 2e-6)).sum())
        jumpy = any(k in (2, 6) for k in kinds)
        ok = nbad <= (max(2, int(0.01 * ref.size)) if jumpy else 0)
        failures += not ok
        print("seed %d ops %s: max err %.2e, %d points off  %s" % (seed, kinds, np.nanmax(err), nbad, "ok" if ok else "FAIL"), flush=True)
    print("%d cases, %d failures" % (count, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(*(int(a) for a in sys.argv[1:])))
