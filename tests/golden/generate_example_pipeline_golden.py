"""Fixture generator for the reference's own VECTOR example scripts — runs ONLY in the build container (the reference is
mounted read-only at /root/reference; it never travels).

For every entry of tests/example_pipelines.py:
  1. the example SCRIPT itself is executed against the real reference (plots stubbed, its variant constant set as the entry
     says); `final_field` and the six read-outs are taken from its namespace;
  2. the entry's workflow is walked by the real reference at the script's own resolution with a pass-through hook: all
     seven arrays must equal the script's bit for bit — the workflow IS the script's pipeline;
  3. the workflow is walked by the real reference on the entry's reduced grid with a RECORDING hook: every array is stored
     as computed (float64) and the walk continues from its fp32 rounding (the "identical grids" rule applied to every
     stage: `example_pipelines.continue_from`). tests/test_example_pipelines.py replays the stages against the oracle and the GPU path.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/generate_example_pipeline_golden.py
"""
import contextlib
import io
import json
import os
import sys
import time

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
sys.path.insert(0, "/root/reference/Code/spomso")
sys.dont_write_bytecode = True
EXAMPLES_DIR = "/root/reference/Code/examples"

import spomso.cores as ref  # noqa: E402  (the real reference)
import example_pipelines as evs  # noqa: E402
from generate_example_golden import run_script  # noqa: E402


def main():
    out, meta = {}, {"numpy": np.__version__, "scipy": scipy.__version__, "reference": "peterropac/Aegolius SPOMSO 1.4.0",
                     "scenes": {}}
    failures = []
    ev = evs.ProductEvaluator(ref)
    reshape = evs.mod(ref, "helper_functions").smarter_reshape
    for name, e in evs.EXAMPLES.items():
        t0 = time.time()
        if e["raises"] is not None:
            got = []
            for walk in (lambda: run_script(e["script"], e["overrides"], e["cwd"], base=EXAMPLES_DIR),
                         lambda: e["run"](ref, ev, lambda _stage, value: value, e["size"], e["res"], **e["variant"])):
                try:
                    walk()
                    got.append(None)
                except Exception as exc:  # noqa: BLE001
                    got.append(type(exc).__name__)
            ok = got == [e["raises"].__name__] * 2
            meta["scenes"][name] = {"script": e["script"], "overrides": e["overrides"], "raises": e["raises"].__name__,
                                    "script_and_workflow_raise_it": bool(ok)}
            print("%-40s script and workflow raise %s: %s" % (name, e["raises"].__name__, ok), flush=True)
            if not ok:
                failures.append((name, "expected %s, got %r" % (e["raises"].__name__, got)))
            continue
        try:
            space = run_script(e["script"], e["overrides"], e["cwd"], base=EXAMPLES_DIR,
                               hide=("show_3d", "show_field_3d", "show_field", "show_midplane"))
            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                full = e["run"](ref, ev, lambda _stage, value: value, e["size"], e["full_res"], **e["variant"])
            same = True
            assert set(full) == set(e["outputs"]), (set(full), set(e["outputs"]))
            for r, (variable, reshaped) in e["outputs"].items():
                theirs = np.asarray(space[variable], dtype=np.float64)
                mine = np.asarray(full[r], dtype=np.float64)
                if reshaped:
                    mine = np.asarray(reshape(mine, e["full_res"]), dtype=np.float64)   # the scripts keep these as grids
                same = same and theirs.shape == mine.shape and np.array_equal(theirs, mine, equal_nan=True)
            stages = {}

            def record(stage, value):
                value = np.asarray(value, dtype=np.float64)
                assert stage not in stages, stage
                stages[stage] = value                           # stored as computed; the walk goes on from its fp32 rounding
                return evs.continue_from(stage, value)

            with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
                e["run"](ref, ev, record, e["size"], e["res"], **e["variant"])
        except Exception as exc:  # noqa: BLE001
            failures.append((name, repr(exc)))
            continue
        for stage, value in stages.items():
            out[name + "/" + stage] = value
        meta["scenes"][name] = {"script": e["script"], "overrides": e["overrides"], "size": list(e["size"]),
                                "script_resolution": list(e["full_res"]), "resolution": list(e["res"]),
                                "stages": {k: list(v.shape) for k, v in stages.items()},
                                "workflow_equals_script_bit_for_bit": bool(same),
                                "nan": int(sum(np.isnan(v).sum() for v in stages.values()))}
        print("%-40s script == workflow: %-5s  %d stages, %d points  (%.1f s)"
              % (name, same, len(stages), stages["coor"].shape[1], time.time() - t0), flush=True)
        if not same:
            failures.append((name, "workflow differs from the script"))
    np.savez_compressed(os.path.join(HERE, "example_pipelines.npz"), **out)
    with open(os.path.join(HERE, "example_pipelines_meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    for name, err in failures:
        print("  FAILED %-36s %s" % (name, err[:300]))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
