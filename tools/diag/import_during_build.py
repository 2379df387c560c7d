"""Reproducer (GPU box): a large dlopen (import torch) while the background hiprtc worker builds a kernel.
usage: import_during_build.py [warm]   — `warm`: run one synchronous hiprtc build before the asynchronous one"""
import ctypes
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ctypes.CDLL(os.path.join(ROOT, "gpurun_out", "bt_on_signal.so"))
os.environ["SDFK_CACHE_DIR"] = "off"
import aegolius_amd                                              # noqa: E402
import aegolius_amd.cores as ns                                  # noqa: E402
from aegolius_amd import _engine, workloads                      # noqa: E402

co, _ = ns.generate_grid((3, 3, 3), (64, 64, 64))
if "warm" in sys.argv:
    aegolius_amd.config.mode = _engine.MODE_SPECIALIZED
    ns.Sphere(0.3).create(co)
    aegolius_amd.config.mode = _engine.MODE_AUTO
print("pid", os.getpid(), flush=True)
t0 = time.time()
workloads.cfg5_tree(ns).create(co)                               # AUTO: served by the interpreter, build in the background
print("create returned after %.3f s; importing torch" % (time.time() - t0), flush=True)
import torch                                                     # noqa: E402
print("torch imported after %.3f s" % (time.time() - t0), flush=True)
_engine.lib().sdfk_jit_drain()
print("done %.3f s" % (time.time() - t0), flush=True)
