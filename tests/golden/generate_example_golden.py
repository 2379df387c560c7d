"""Fixture generator for the reference's own example scenes — runs ONLY in the build container (the reference is
mounted read-only at /root/reference; it never travels).

For every entry of tests/example_scenes.py:
  1. the example SCRIPT itself is executed against the real reference (read from /root/reference at run time, plot
     libraries replaced by stubs, its variant constants set as the entry says) and the array it computes is taken from
     its namespace;
  2. the entry's builder is evaluated by the real reference on the same full-size grid: the two arrays must be equal
     bit for bit — the builder IS the script's scene;
  3. the builder is evaluated by the real reference on the entry's reduced grid (fp32-rounded coordinates, the
     "identical grids" of the north star): stored as the fixture, together with how many points the float64 ORACLE
     disagrees on (exact ties: see tests/test_example_scenes.py).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_example_golden.py
"""
import contextlib
import io
import json
import os
import re
import sys
import time
import types
from unittest import mock

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/Code/spomso")
sys.dont_write_bytecode = True
EXAMPLES_DIR = "/root/reference/Code/examples/scalar"

import spomso.cores as ref  # noqa: E402  (the real reference)
import example_scenes  # noqa: E402


class _Anything:
    """Stands in for every plotting object: any attribute, call, item or unpacking (`fig, ax = plt.subplots()`) works."""

    def __getattr__(self, _name):
        return self

    def __call__(self, *_a, **_k):
        return self

    def __getitem__(self, _i):
        return self

    def __iter__(self):
        return iter((self, self))


def _stub_plot_modules():
    stubs = {}
    for name in ("matplotlib", "matplotlib.pyplot", "matplotlib.cm", "matplotlib.colors", "plotly", "plotly.graph_objects",
                 "plotly.subplots", "plotly.express", "mpl_toolkits", "mpl_toolkits.axes_grid1"):
        m = types.ModuleType(name)
        m.__getattr__ = lambda _attr: _Anything()
        stubs[name] = m
    return stubs


def run_script(relpath, overrides, cwd=None, base=EXAMPLES_DIR, hide=("show_3d",)):
    """Execute an example script of the reference (its text stays where it is) -> its namespace."""
    with open(os.path.join(base, relpath)) as f:
        text = f.read()
    for var, value in overrides.items():                        # variant constants: `name = <literal>` at module level
        text, n = re.subn(r"(?m)^%s\s*=.*$" % re.escape(var), "%s = %r" % (var, value), text, count=1)
        assert n == 1, (relpath, var)
    text = re.sub(r"(?m)^(%s)\s*=.*$" % "|".join(hide), r"\1 = False", text)
    space = {"__name__": "__example__"}
    here = os.getcwd()
    try:
        if cwd:
            os.chdir(os.path.join(base, cwd))           # (a script that finds the reference's data files from os.getcwd())
        with mock.patch.dict(sys.modules, _stub_plot_modules()), contextlib.redirect_stdout(io.StringIO()), \
                np.errstate(all="ignore"):
            exec(compile(text, relpath, "exec"), space)         # noqa: S102 (the reference's own example, build container only)
    finally:
        os.chdir(here)
    return space


def main():
    from oracle import sdf_oracle
    import aegolius_amd.cores as ns
    out, meta = {}, {"numpy": np.__version__, "scipy": scipy.__version__, "reference": "peterropac/Aegolius SPOMSO 1.4.0",
                     "scenes": {}}
    failures = []
    only = set(sys.argv[1:])                                     # names given: regenerate those, keep the other fixtures as they are
    if only:
        old = np.load(os.path.join(HERE, "example_scenes.npz"))
        out.update({k: old[k] for k in old.files if k not in only})
        with open(os.path.join(HERE, "example_scenes_meta.json")) as f:
            meta["scenes"].update({k: v for k, v in json.load(f)["scenes"].items() if k not in only})
    for name, e in example_scenes.EXAMPLES.items():
        if only and name not in only:
            continue
        t0 = time.time()
        try:
            space = run_script(e["script"], e["overrides"], e.get("cwd"))
            script_field = np.asarray(space[e["variable"]], dtype=np.float64)
            co_full, _res = ref.generate_grid(e["size"], e["full_res"])
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                mine_full = np.asarray(e["build"](ref).create(co_full.copy()), dtype=np.float64)
            same = script_field.shape == mine_full.shape and np.array_equal(script_field, mine_full, equal_nan=True)
            co, res = ref.generate_grid(e["size"], e["res"])
            co = co.astype(np.float32).astype(np.float64)          # identical grids: fp32-valued coordinates
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                field = np.asarray(e["build"](ref).create(co.copy()), dtype=np.float64)
                oracle = sdf_oracle.evaluate(e["build"](ns), co.copy())
        except Exception as exc:  # noqa: BLE001
            failures.append((name, repr(exc)))
            continue
        off = ~(np.abs(oracle - field) <= 1e-12 * np.maximum(1.0, np.abs(field)))
        out[name] = field
        meta["scenes"][name] = {"script": e["script"], "variable": e["variable"], "overrides": e["overrides"],
                                "size": list(e["size"]), "script_resolution": list(e["full_res"]),
                                "resolution": list(e["res"]), "returned_resolution": [int(x) for x in res],
                                "points": int(field.size), "builder_equals_script_bit_for_bit": bool(same),
                                "script_points": int(script_field.size),
                                "oracle_off_points": int(off.sum()),
                                "oracle_off_max": float(np.abs(oracle - field)[off].max()) if off.any() else 0.0,
                                "nan": int(np.isnan(field).sum())}
        print("%-36s script == builder: %-5s  fixture %7d points, oracle off at %d  (%.1f s)"
              % (name, same, field.size, int(off.sum()), time.time() - t0), flush=True)
        if not same:
            failures.append((name, "builder differs from the script"))
    np.savez_compressed(os.path.join(HERE, "example_scenes.npz"), **out)
    with open(os.path.join(HERE, "example_scenes_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for name, err in failures:
        print("  FAILED %-36s %s" % (name, err[:300]))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
