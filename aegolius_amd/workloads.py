"""The five BASELINE.json configs as scene builders (recipes: SURVEY.md §8(d)).

Every builder takes the namespace that provides the operator API — `aegolius_amd.cores` here, `spomso.cores` in the
golden-vector generators of the build container — so that one recipe drives this package, the reference and the
oracle alike. `bench.py`, `__graft_entry__.smoke()` and the tests (through `tests/scenes.py`) all build from here.
"""
import numpy as np


def _cfg2_prims(ns, rng, count):
    makers = [lambda: ns.Sphere(0.3), lambda: ns.Box(0.5, 0.4, 0.3), lambda: ns.Cylinder(0.2, 0.6),
              lambda: ns.Torus(0.3, 0.1), lambda: ns.Cone(0.6, np.pi / 8)]
    out = []
    for k in range(count):
        o = makers[k % 5]()
        angle = float(rng.uniform(0, np.pi))                 # draw order: angle, axis, move
        axis = rng.normal(0, 1, 3)
        o.rotate(angle, axis)
        o.move(rng.uniform(-0.7, 0.7, 3))
        out.append(o)
    return out


def cfg1_sphere(ns):
    """BASELINE configs[0]: a single sphere."""
    return ns.Sphere(0.5)


def cfg2_tree(ns, seed=1234, count=10, width=0.1):
    """BASELINE configs[1] / north-star: left-deep chain of SMOOTH_UNION2 over `count` primitives."""
    prims = _cfg2_prims(ns, np.random.default_rng(seed), count)
    acc = prims[0]
    for p in prims[1:]:
        acc = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(acc, p, parameters=width)
    return acc


def cfg3_chain(ns):
    """BASELINE configs[2]: deep modification chain."""
    b = ns.Box(0.6, 0.3, 0.2)
    b.elongation((0.4, 0, 0.1))
    b.twist(np.pi / 2)
    b.bend(1.5, np.pi / 3)
    b.infinite_repetition((2, 2, 2))
    b.rotate(0.3, (0, 1, 1))
    b.move((0.1, -0.2, 0.05))
    return b


def cfg4_scene2d(ns, seed=7, count=50):
    """BASELINE configs[3]: n-ary UNION of 2-D primitives with onion / rounding."""
    rng = np.random.default_rng(seed)
    makers = [lambda: ns.Circle(0.4), lambda: ns.Rectangle(0.8, 0.5), lambda: ns.NGon(0.4, 6),
              lambda: ns.RoundedRectangle(0.8, 0.6, (0.1, 0.05, 0.15, 0.0))]
    objs = []
    for k in range(count):
        o = makers[k % 4]()
        if k % 2 == 0:
            o.onion(0.03)
        else:
            o.rounding(0.05)
        o.rotate(float(rng.uniform(0, np.pi)), (0, 0, 1))
        o.move((float(rng.uniform(-4, 4)), float(rng.uniform(-4, 4)), 0))
        objs.append(o)
    return ns.CombineGeometry("UNION").combine(*objs)


def cfg5_tree(ns, seed=2049):
    """BASELINE configs[4]: 20 primitives, 3 levels."""
    prims = _cfg2_prims(ns, np.random.default_rng(seed), 20)
    groups = []
    for g in range(5):
        acc = prims[4 * g]
        for p in prims[4 * g + 1:4 * g + 4]:
            acc = ns.CombineGeometry("SMOOTH_UNION2").combine_parametric(acc, p, parameters=0.1)
        groups.append(acc)
    union = ns.CombineGeometry("UNION").combine(*groups[:4])
    return ns.CombineGeometry("SUBTRACT2").combine(union, groups[4])


def sphere_union(ns, count=1000, seed=31, radius=0.05, extent=0.9):
    """A flat n-ary UNION of `count` spheres scattered in [-extent, extent]^3: the deep-n-ary-tree stress case
    (beyond the per-tree specialisation limit when count is large)."""
    rng = np.random.default_rng(seed)
    objs = []
    for _ in range(count):
        o = ns.Sphere(float(radius * rng.uniform(0.5, 1.5)))
        o.move(rng.uniform(-extent, extent, 3))
        objs.append(o)
    return ns.CombineGeometry("UNION").combine(*objs)


def clustered_union(ns, groups=20, members=50, seed=77, radius=0.03, spread=0.15, extent=0.8):
    """`groups` rigidly placed copies of a cluster of `members` spheres, each cluster a UNION of its own, the clusters
    rotated / moved (every third one rescaled) and united: the nested form of a large scene (instances of a molecule)."""
    rng = np.random.default_rng(seed)
    cluster = [(float(radius * rng.uniform(0.6, 1.4)), rng.uniform(-spread, spread, 3)) for _ in range(members)]
    out = []
    for k in range(groups):
        objs = []
        for r, c in cluster:
            o = ns.Sphere(r)
            o.move(c)
            objs.append(o)
        g = ns.CombineGeometry("UNION").combine(*objs)
        g.rotate(float(rng.uniform(0, np.pi)), tuple(rng.normal(size=3)))
        g.move(rng.uniform(-extent, extent, 3))
        if k % 3 == 0:
            g.rescale(1.2)
        out.append(g)
    return ns.CombineGeometry("UNION").combine(*out)


# name -> (builder, grid size, description, per-axis request the BASELINE config names)
BASELINE = {
    "cfg1": (cfg1_sphere, (2, 2, 2), "cfg1: Sphere(0.5)", 128),
    "cfg2": (cfg2_tree, (2, 2, 2), "cfg2: 10-primitive left-deep SMOOTH_UNION2(0.1) chain, rng 1234", 512),
    "cfg3": (cfg3_chain, (4, 4, 4), "cfg3: Box + elongation/twist/bend/infinite_repetition", 1024),
    "cfg4": (cfg4_scene2d, (10, 10), "cfg4: 2-D n-ary UNION of 50 onion/rounded primitives, rng 7", 16384),
    "cfg5": (cfg5_tree, (3, 3, 3), "cfg5: 20-primitive 3-level tree, rng 2049", 2048),
}


def build(name, ns):
    """-> (tree, grid size, description) of a BASELINE config."""
    if name.startswith("hard") and name[4:].isdigit():          # developer workloads: the same primitives under pairwise hard unions
        count = int(name[4:])
        prims = _cfg2_prims(ns, np.random.default_rng(100 + count), count)
        acc = prims[0]
        for p in prims[1:]:
            acc = ns.CombineGeometry("UNION2").combine(acc, p)
        return acc, (2, 2, 2), "left-deep UNION2 chain of %d primitives" % count
    if name.startswith("tree") and name[4:].isdigit():          # developer workloads: the cfg2 recipe with n primitives
        count = int(name[4:])
        return cfg2_tree(ns, seed=100 + count, count=count), (2, 2, 2), "left-deep SMOOTH_UNION2 chain of %d primitives" % count
    builder, size, desc, _request = BASELINE[name]
    return builder(ns), size, desc
