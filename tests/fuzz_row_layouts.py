"""Test-infrastructure script (GPU; not collected by pytest): the row-block kernel under random layouts — grid shapes
with short / odd / long rows, row counts around multiples of the brick height, shifted (4-byte aligned) pointers,
row-length hints that do not describe the data — against the plain kernel, bit for bit; and the table flavour
(sdfk_eval_grid_host) on random slabs of the same grids, whole rows and arbitrary ranges.

    python tests/fuzz_row_layouts.py [cases]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main(cases=80):
    import scenes
    import aegolius_amd.cores as ns
    from aegolius_amd import _engine
    from aegolius_amd._lower import lower_geometry
    lib = _engine.lib()
    vp = _engine._vp
    rng = np.random.default_rng(77)
    trees = [scenes.cfg2_tree(ns), scenes.cfg5_tree(ns), scenes.SCENES["tree_pawn_3D"](ns), scenes.random_tree(ns, 9002, 4)]
    progs = [_engine.Program.from_lowered(lower_geometry(t)) for t in trees]
    cap = 1 << 21
    d_co, d_out = lib.sdfk_malloc((3 * cap + 64) * 4), lib.sdfk_malloc((cap + 64) * 4)
    failures = 0
    for case in range(cases):
        L = int(rng.choice([32, 33, 40, 63, 64, 65, 96, 127, 129, 257, 1025]))
        rows = int(rng.integers(1, max(2, min(3000, cap // L))))
        n1 = int(rng.integers(1, min(rows, 40) + 1))
        n0 = max(1, rows // n1)
        ax = [np.linspace(-1.3, 1.3, m).astype(np.float32) if m > 1 else np.zeros(1, np.float32) for m in (n0, n1, L)]
        co = np.stack([np.repeat(ax[0], n1 * L), np.tile(np.repeat(ax[1], L), n0), np.tile(ax[2], n0 * n1)])
        n = co.shape[1]
        kind = int(rng.integers(0, 4))
        hint = L
        if kind == 1:
            hint = 2 * L if n % (2 * L) == 0 else L          # a hint that is not the row length
        elif kind == 2:
            co = co[:, rng.permutation(n)]                    # scattered points under a row hint
        elif kind == 3 and n % 32 == 0:
            hint = 32
        mis = int(rng.integers(0, 4))
        stride = n + int(rng.integers(0, 7))
        host = np.zeros((3, stride), dtype=np.float32)
        host[:, :n] = co
        _engine.check(lib.sdfk_memcpy_h2d(vp(d_co + 4 * mis), _engine._ptr(host), host.nbytes), "h2d")
        prog = progs[case % len(progs)]
        res = []
        for mode, rl in ((_engine.MODE_NOCULL, None), (_engine.MODE_SPECIALIZED, hint)):
            prog.eval_device(d_co + 4 * mis, n, stride, d_out + 4 * mis, mode=mode, row_len=rl)
            _engine.check(lib.sdfk_sync(None), "sync")
            out = np.empty(n, dtype=np.float32)
            _engine.check(lib.sdfk_memcpy_d2h(_engine._ptr(out), vp(d_out + 4 * mis), n * 4), "d2h")
            res.append(out)
        ok = np.array_equal(res[0], res[1], equal_nan=True)
        if kind in (0, 1, 3):                                   # a real grid: the table flavour on random slabs
            for _ in range(3):
                whole_rows = bool(rng.integers(0, 2))
                if whole_rows:
                    r0 = int(rng.integers(0, n0 * n1))
                    r1 = int(rng.integers(r0, n0 * n1)) + 1
                    start, count = r0 * L, (r1 - r0) * L
                else:
                    start = int(rng.integers(0, n))
                    count = int(rng.integers(1, n - start + 1))
                got = prog.eval_grid_host(ax, start, count)
                ok = ok and np.array_equal(got, res[0][start:start + count], equal_nan=True)
        failures += not ok
        print("case %d: grid %dx%dx%d hint %d kind %d misalign %d stride+%d tree %d  %s" % (
            case, n0, n1, L, hint, kind, mis, stride - n, case % len(progs), "ok" if ok else "MISMATCH at %d points" % int((res[0] != res[1]).sum())), flush=True)
    lib.sdfk_free(vp(d_co))
    lib.sdfk_free(vp(d_out))
    print("%d cases, %d failures" % (cases, failures))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main(*(int(a) for a in sys.argv[1:])))
