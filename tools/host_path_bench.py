"""Developer tool (GPU): PCIe-inclusive throughput of sdfk_eval_host (pageable host array in, host array out) for the
north-star tree on whole x-planes of the 1025^3 grid, under SDFK_HOST_THREADS / SDFK_HOST_CHUNK settings given on the
command line as THREADS:CHUNK pairs (each in a fresh process, the settings are read once)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r"""
import sys, time
sys.path.insert(0, %r); sys.path.insert(0, %r + "/tests")
import numpy as np
import scenes
import aegolius_amd.cores as ns
from aegolius_amd import _engine
from aegolius_amd._lower import lower_geometry
from aegolius_amd.cores.helper_functions import grid_axes
axes = [a.astype(np.float32) for a in grid_axes((2, 2, 2), (1024,) * 3)[0]]
planes = 96
m = planes * 1025 * 1025
hco = np.empty((3, m), dtype=np.float32)
hco[0] = np.repeat(axes[0][:planes], 1025 * 1025); hco[1] = np.tile(np.repeat(axes[1], 1025), planes); hco[2] = np.tile(axes[2], planes * 1025)
prog = _engine.Program.from_lowered(lower_geometry(scenes.cfg2_tree(ns)))
prog.eval_host(hco, mode=_engine.MODE_SPECIALIZED)
best = 1e9
for _ in range(4):
    t0 = time.perf_counter(); out = prog.eval_host(hco, mode=_engine.MODE_SPECIALIZED); best = min(best, time.perf_counter() - t0)
print("%%.1f ms  %%.0f Mpoints/s  %%.1f GB/s over PCIe" %% (best * 1e3, m / best / 1e6, 16.0 * m / best / 1e9))
""" % (ROOT, ROOT)
for spec in sys.argv[1:] or ["8:8388608"]:
    t, c = spec.split(":")
    env = dict(os.environ, SDFK_HOST_THREADS=t, SDFK_HOST_CHUNK=c)
    res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print("threads %s chunk %s: %s" % (t, c, (res.stdout.strip().splitlines() or [res.stderr[-300:]])[-1]), flush=True)
