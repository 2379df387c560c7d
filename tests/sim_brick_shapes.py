"""Test-infrastructure script (CPU, float64 oracle): how many primitives of the north-star tree survive the
culling probe per brick, for different brick shapes on the 1025^3 grid of size (2, 2, 2). This is the
estimate behind DESIGN.md §4.3 (128 points in a line: 3.5 of 10 alive; 32 x 16 block: 1.8).

    python tests/sim_brick_shapes.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(samples=200000, spacing=2.0 / 1024, half=1.0, w=0.1):
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd._ir import CombineSDF
    from oracle import sdf_oracle
    tree = scenes.cfg2_tree(ns)
    prims, node = [], tree
    while isinstance(node.modified_object, CombineSDF):
        a, b = node.modified_object.children
        prims.append(b)
        node = a
    prims.append(node)
    prims = prims[::-1]
    rng = np.random.default_rng(0)
    c = rng.uniform(-half, half, (3, samples))
    d = [sdf_oracle.evaluate(p, c) for p in prims]
    for shape in [(128, 1, 1), (64, 2, 1), (32, 4, 1), (32, 16, 1), (32, 4, 4), (16, 4, 4), (8, 4, 4)]:
        rho = 0.5 * np.sqrt((((np.array(shape) - 1) * spacing) ** 2).sum())
        thr = w + 2 * 1.0001 * rho + 1e-6                     # K = L_a + L_b = 2 for distance fields
        acc = d[0].copy()
        alive = np.ones((len(prims), samples), bool)
        for k in range(1, len(prims)):
            alive[k] &= ~(d[k] - acc > thr)
            alive[:k] &= ~(acc - d[k] > thr)
            acc = sdf_oracle.smin_poly(acc, d[k], w, 3)
        cnt = alive.sum(axis=0)
        print("brick %-12s %4d points  radius %.4f  primitives alive %.2f  histogram %s" % (
            shape, int(np.prod(shape)), rho, cnt.mean(), np.round(np.bincount(cnt, minlength=8)[:8] / samples, 3)))


if __name__ == "__main__":
    main()
