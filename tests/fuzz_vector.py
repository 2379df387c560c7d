"""Differential fuzzer for vector-field chains (test infrastructure; not collected by pytest):

    python tests/fuzz_vector.py reference [first] [count]   build container: float64 oracle vs the REAL reference
    python tests/fuzz_vector.py gpu [first] [count]         GPU box: GPU evaluation (both kernels) vs the oracle

Random field definitions, 1-6 random modifications with random operand kinds (number, 3-vector, per-point, field,
per-point axes, same / other revolution cloud), random read-out. GPU deviations are compared with 1e-6 and, where a
chain is ill-conditioned at a point, with 4x what the float64 chain itself moves when every intermediate vector is
disturbed by two fp32 ulps (fp32_noise_model)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

READS = ("create", "create", "create", "x", "y", "z", "phi", "theta", "length")


def build(ns, aux, seed):
    rng = np.random.default_rng(seed)
    u = lambda a, b: float(rng.uniform(a, b))      # noqa: E731
    angle = lambda: (u(-3, 3) if rng.random() < 0.4 else aux[str(rng.choice(["alpha", "beta"]))])   # noqa: E731
    kind = int(rng.integers(0, 10))
    key = "p"
    if kind == 0: f, key = ns.CartesianVectorField(), "second"
    elif kind == 1: f, key = ns.CylindricalVectorField(), "comps"
    elif kind == 2: f, key = ns.SphericalVectorField(), "comps"
    elif kind == 3: f = ns.RadialSphericalVectorField()
    elif kind == 4: f = ns.RadialCylindricalVectorField()
    elif kind == 5: f = ns.VortexCylindricalVectorField()
    elif kind == 6: f = ns.AngledRadialCylindricalVectorField(angle())
    elif kind == 7: f = ns.AngledVortexCylindricalVectorField(angle())
    elif kind == 8: f = getattr(ns, str(rng.choice(["XVectorField", "YVectorField", "ZVectorField"])))()
    else: f, key = ns.CartesianVectorField(), "axes"
    names = [type(f).__name__]
    for _ in range(int(rng.integers(1, 7))):
        m = int(rng.integers(0, 13))
        if m in (0, 1):
            operand = [u(-1, 1), (u(-1, 1), u(-1, 1), u(-1, 1)), aux["second"], aux["scale"]][int(rng.integers(0, 4))]
            (f.add if m == 0 else f.subtract)(operand)
        elif m == 2:
            f.rescale([u(-2, 2), aux["scale"], aux["second"], np.asarray([[u(0.5, 2)], [u(0.5, 2)], [u(0.5, 2)]])][int(rng.integers(0, 4))])
        elif m == 3: f.rotate_phi(angle())
        elif m == 4: f.rotate_theta(angle())
        elif m == 5: f.rotate_x(angle())
        elif m == 6: f.rotate_y(angle())
        elif m == 7: f.rotate_z(angle())
        elif m == 8: f.rotate_axis((u(-1, 1), u(-1, 1), u(0.2, 1)) if rng.random() < 0.5 else aux["axes"], angle())
        elif m in (9, 10, 11):
            getattr(f, "revolution_" + "xyz"[m - 9])(aux[key] if rng.random() < 0.5 else aux["co2"])
        else: f.normalize()
        names.append(f.modifications[-1])
    return f, key, READS[int(rng.integers(0, len(READS)))], names


def fp32_noise_model(vo, field, p, out, want, trials=6, eps=1.2e-7):
    """How far the float64 chain moves when every intermediate vector is disturbed by a couple of fp32 ulps (what any
    fp32 evaluation does): the conditioning of the chain — planar normalisations of nearly axial vectors
    (rotate_theta), read-outs at their singular points (phi on the axis, theta at the poles)."""
    from aegolius_amd._vector import VecClosure, _leaf_name, as_closure
    rng = np.random.default_rng(7)
    closure = as_closure(field.vf)
    mods, inner = closure.mods, closure.leaf
    while isinstance(inner, VecClosure):
        mods, inner = inner.mods + mods, inner.leaf
    worst = np.zeros_like(want)

    def jitter(a):
        a = np.asarray(a, dtype=np.float64)
        scale = np.sqrt((a * a).sum(axis=0)) if a.ndim == 2 else np.abs(a)
        return a + eps * scale * rng.uniform(-1, 1, size=a.shape)

    def snap(a):
        # the other way a float64 chain is fragile: a component that is round-off (cos(pi/2) = 6e-17 in NumPy) where an
        # exact evaluation has 0 — the GPU's revolutions take cos / sin of atan2 without the angle and get the 0 —, and
        # a planar normalisation (rotate_theta) blows that round-off up to a unit direction (seed 96336: an axial vector
        # revolved by pi/2 twice; the reference's length 1 against |cos(angle)| for the exactly axial vector)
        a = np.array(a, dtype=np.float64)
        scale = np.sqrt((a * a).sum(axis=0)) if a.ndim == 2 else np.abs(a)
        a[np.abs(a) < 1e-12 * scale] = 0.0
        return a
    for trial in range(trials + 1):
        if trial == trials:
            jitter = snap                                      # noqa: F811  (the last trial: exact zeros instead of noise)
        with np.errstate(all="ignore"):
            v = jitter(vo.leaf(_leaf_name(inner), jitter(p), field._vf_parameters))
            for name, args in mods:
                v = jitter(vo.modify(v, name, args))
            if out == "vector":
                r = v
            elif out in "xyz":
                r = v["xyz".index(out)]
            elif out == "phi":
                r = np.arctan2(v[1], v[0])
            elif out == "theta":
                r = np.arccos(np.clip(v[2], -1, 1))
            else:
                r = np.sqrt((v * v).sum(axis=0))
            d = np.abs(r - want)
            if out == "phi":
                d = np.abs((r - want + np.pi) % (2 * np.pi) - np.pi)
            worst = np.fmax(worst, np.nan_to_num(d, nan=np.inf))
    return worst


def main(which="reference", first=0, count=300):
    import vector_scenes as vs
    import aegolius_amd.cores as ns
    from oracle import vector_oracle as vo
    aux = vs.inputs()
    if which == "reference":
        sys.path.insert(0, "/root/reference/Code/spomso")
        sys.dont_write_bytecode = True
        import contextlib
        import io
        import spomso.cores as ref
    else:
        from aegolius_amd import _engine
        lib = _engine.lib()
    failures = 0
    for seed in range(int(first), int(first) + int(count)):
        f, key, read, names = build(ns, aux, seed)
        out = "vector" if read == "create" else read
        with np.errstate(all="ignore"):
            want = vo.evaluate(f.vf, aux[key], f._vf_parameters, out)
        note = ""
        if which == "reference":
            g, gkey, gread, _ = build(ref, aux, seed)
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                got = np.asarray(getattr(g, gread)(aux[gkey].copy()), dtype=np.float64)
            err = np.abs(got - want) / np.maximum(1.0, np.abs(want))
            err[np.isnan(got) & np.isnan(want)] = 0
            bad = int((~(err <= 1e-11)).sum())
        else:
            got = getattr(f, read)(aux[key].copy())
            lib.sdfk_vec_set_interpret(1)
            same = np.array_equal(getattr(build(ns, aux, seed)[0], read)(aux[key].copy()), got, equal_nan=True)
            lib.sdfk_vec_set_interpret(0)
            err = np.abs(got.astype(np.float64) - want)
            off = ~(err <= 1e-6 * np.maximum(1.0, np.abs(want))) & ~(np.isnan(got) & np.isnan(want))
            if out == "phi":                                   # angles are compared on the circle
                err = np.abs((got.astype(np.float64) - want + np.pi) % (2 * np.pi) - np.pi)
                off = ~(err <= 1e-6 * np.maximum(1.0, np.abs(want))) & ~(np.isnan(got) & np.isnan(want))
            if off.any():
                slack = 4.0 * fp32_noise_model(vo, f, aux[key], out, want)
                note = " (%d within the chain's own fp32 conditioning)" % int((off & (err <= slack)).sum())
                off &= ~(err <= slack)
            bad = int(off.sum()) + (0 if same else 1)
            if not same:
                note += " KERNELS DIFFER"
        failures += bad > 0
        print("seed %d %s -> %s: %d off, worst %.2e%s %s" % (seed, "+".join(names), read, bad, float(np.nanmax(err)) if err.size else 0,
                                                              note, "" if bad == 0 else " <-- FAIL"), flush=True)
    print("%s: %d cases, %d failures" % (which, int(count), failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(*sys.argv[1:]))
