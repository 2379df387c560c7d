"""create()/propagate() back end: lower the tree, fetch (or build) the native program, run it.

There is exactly one evaluation path — libsdfk.so on an MI355X. No NumPy evaluation exists in this
package; without the extension or without a GPU these functions raise.
"""
import collections
import threading

import numpy as np

from . import _engine
from ._lower import lower_expression, lower_geometry


class Config:
    """Process-wide evaluation settings."""
    device = 0                    # HIP device ordinal used by create()
    mode = _engine.MODE_AUTO      # MODE_AUTO (specialised, hiprtc) | MODE_INTERPRET | MODE_SPECIALIZED
    output_dtype = np.float32     # the reference returns float64; set to np.float64 to get an upcast copy
    cache_size = 128
    grid_fast_path = True         # generate_grid tags its (read-only) result; create() then evaluates the grid
                                  # from the per-axis tables on the GPU instead of uploading 12 B/point


config = Config()

_cache = collections.OrderedDict()
_lock = threading.Lock()


def program_for(lowered):
    key = lowered.key()
    with _lock:
        prog = _cache.get(key)
        if prog is not None:
            _cache.move_to_end(key)
            return prog
    prog = _engine.Program.from_lowered(lowered)
    with _lock:
        _cache[key] = prog
        while len(_cache) > config.cache_size:
            _cache.popitem(last=False)
    return prog


def _run(lowered, co):
    axes = getattr(co, "grid_axes", None) if config.grid_fast_path else None
    if axes is not None:
        out = program_for(lowered).eval_grid_host(axes, device=config.device, mode=config.mode)
    else:
        out = program_for(lowered).eval_host(co, device=config.device, mode=config.mode)
    if config.output_dtype is not np.float32:
        out = out.astype(config.output_dtype)
    return out


def evaluate_geometry(node, co):
    return _run(lower_geometry(node), co)


def evaluate_expr(expr, co, params):
    return _run(lower_expression(expr, params), co)
