"""Golden-vector generator for the vector-field path — runs ONLY in the build container, where the reference is
mounted read-only at /root/reference. It imports the real SPOMSO implementation, runs every scene of
tests/vector_scenes.py on the seeded inputs and stores inputs + float64 outputs next to this script (data only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_vector_golden.py
"""
import contextlib
import io
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))     # the repository root: tests/scenes.py takes its recipes from aegolius_amd/workloads.py
sys.path.insert(0, "/root/reference/Code/spomso")
sys.dont_write_bytecode = True

import spomso.cores as ref  # noqa: E402  (the real reference)
import vector_scenes as vs  # noqa: E402


def main():
    aux = vs.inputs()
    out = {"input/" + k: v.astype(np.float32) for k, v in aux.items()}
    meta = {"numpy": np.__version__, "reference": "peterropac/Aegolius SPOMSO 1.4.0", "n_points": vs.N, "scenes": {},
            "functions": {}, "raising": {}}
    for name in vs.SCENES:
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            got = np.asarray(vs.run(ref, name, {k: v.copy() for k, v in aux.items()}), dtype=np.float64)
        out["scene/" + name] = got
        meta["scenes"][name] = {"shape": list(got.shape), "nan": int(np.isnan(got).sum())}
    for name, fn in vs.FUNCTIONS.items():
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            got = np.asarray(fn(ref, {k: v.copy() for k, v in aux.items()}), dtype=np.float64)
        out["function/" + name] = got
        meta["functions"][name] = {"shape": list(got.shape)}
    for name, (fn, exc) in vs.RAISING.items():
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                fn(ref, aux)
            meta["raising"][name] = None
        except Exception as e:  # noqa: BLE001
            meta["raising"][name] = type(e).__name__
            assert isinstance(e, exc), (name, e)
    np.savez_compressed(os.path.join(HERE, "vector_golden.npz"), **out)
    with open(os.path.join(HERE, "vector_golden_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("vector scenes: %d, functions: %d, raising: %r" % (len(meta["scenes"]), len(meta["functions"]), meta["raising"]))


if __name__ == "__main__":
    main()
