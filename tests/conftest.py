import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """libsdfk.so, built in-tree (hipcc cross-compiles for gfx950 without a GPU)."""
    import __graft_entry__
    __graft_entry__.build()
    from aegolius_amd import _engine
    return _engine


@pytest.fixture(scope="session")
def golden():
    data = np.load(os.path.join(GOLDEN_DIR, "golden_scenes.npz"))
    with open(os.path.join(GOLDEN_DIR, "golden_meta.json")) as f:
        meta = json.load(f)
    return data, meta


@pytest.fixture(scope="session")
def golden_inputs(golden):
    return golden[0]["inputs"].astype(np.float64)


@pytest.fixture(scope="session")
def engine(built):
    """The ctypes engine on a box with an MI355X (GPU tests only): there is no CPU evaluation path."""
    built.require_gpu()
    return built
