"""Differential fuzzer for the field consumers (GPU box; not collected by pytest):

    python tests/fuzz_consumers.py [first_seed] [count]

Random grid shapes (1-D to 3-D, extents 2 .. ~3000, long / short rows, sizes around the tile edges of both kernels)
and random fp32 fields with plateaus, NaN, inf and subnormals: sdfk_field_select against numpy.flatnonzero (exact),
sdfk_field_gradient against numpy.gradient (raw: exact after rounding to fp32; direction: 1e-6)."""
import sys
import os

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def random_shape(rng):
    nd = int(rng.integers(1, 4))
    budget = int(rng.choice([200, 5000, 70000, 600000, 3000000]))
    kind = rng.integers(0, 4)
    dims = []
    for a in range(nd):
        left = nd - a - 1
        hi = max(2, int(budget / (2 ** left)))
        if kind == 0 and a == nd - 1:
            d = int(rng.choice([2, 3, 4, 5, 7]))                  # very short rows
        elif kind == 1 and a == nd - 1:
            d = int(rng.choice([1023, 1024, 1025, 1026, 2049, 4097, 8191, 8193]))
        else:
            d = int(rng.integers(2, max(3, min(hi, 3000))))
        d = max(2, min(d, hi))
        dims.append(d)
        budget = max(2, budget // d)
    return tuple(dims)


def main(first=0, count=200):
    from aegolius_amd import DeviceField
    bad = 0
    for seed in range(first, first + count):
        rng = np.random.default_rng(seed)
        shape = random_shape(rng)
        n = int(np.prod(shape))
        scale = float(rng.choice([1.0, 1.0, 1e-3, 1e6, 1e-30, 1e30, 1e-41]))
        f = (rng.normal(size=n) * scale).astype(np.float32)
        f[rng.random(n) < rng.choice([0.0, 0.05, 0.4])] = np.float32(0.25 * scale)
        special = rng.random() < 0.3
        if special:
            f[rng.random(n) < 0.002] = np.nan
            f[rng.random(n) < 0.002] = np.inf
            f[rng.random(n) < 0.002] = -0.0
        dev = DeviceField.from_host(f)
        msg = []
        for thr in (0.0, float(np.float32(rng.normal() * scale))):
            with np.errstate(invalid="ignore"):
                want = np.flatnonzero(f <= np.float32(thr))
            got = dev.select(thr)
            if not np.array_equal(got, want):
                msg.append("select(%g): %d vs %d" % (thr, got.size, want.size))
        with np.errstate(all="ignore"):
            raw = np.asarray(np.gradient(f.astype(np.float64).reshape(shape))).reshape(len(shape), -1)
            m = np.linalg.norm(raw, axis=0)
            keep = ~(m == 0)
            unit = raw.copy()
            unit[:, keep] = unit[:, keep] / m[keep]
            raw32 = raw.astype(np.float32)
        got_raw = dev.gradient(shape, normalize=False)
        if not np.array_equal(got_raw, raw32, equal_nan=True):
            # a float64 difference beyond the fp32 range (|field| > 1.7e38) is the documented exception
            msg.append("raw gradient differs at %d values" % np.count_nonzero(~((got_raw == raw32) | (np.isnan(got_raw) & np.isnan(raw32)))))
        got = dev.gradient(shape)
        if not np.array_equal(np.isnan(got), np.isnan(unit)):
            msg.append("NaN pattern differs")
        else:
            ok = ~np.isnan(unit)
            err = np.abs(got[ok].astype(np.float64) - unit[ok]).max() if ok.any() else 0.0
            if err > 1e-6:
                msg.append("direction max err %.3g" % err)
            if got[:, ~keep].any():
                msg.append("zero vectors not preserved")
        dev.free()
        print("seed %d shape %r scale %g special %d %s" % (seed, shape, scale, special, "  <-- " + "; ".join(msg) if msg else "ok"))
        bad += bool(msg)
    print("%d cases, %d failures" % (count, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(*(int(a) for a in sys.argv[1:])))
