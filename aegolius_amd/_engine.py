"""ctypes binding of libsdfk.so (C-ABI declared in include/sdfk.h).

This is the only place the Python layer touches native code. There is no CPU fallback: if the
shared library is missing or no MI355X is visible, evaluation raises.
"""
import atexit
import ctypes
import importlib.util
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsdfk.so")

MODE_AUTO, MODE_INTERPRET, MODE_SPECIALIZED, MODE_NOCULL = 0, 1, 2, 3
(FLAVOUR_PLAIN_ARRAY, FLAVOUR_PLAIN_GRID, FLAVOUR_TILE_ARRAY, FLAVOUR_TILE_GRID, FLAVOUR_TILE_MASK, FLAVOUR_ROWS_ARRAY,
 FLAVOUR_ROWS_GRID, FLAVOUR_ROWS_MASK, FLAVOUR_ROWS2D_ARRAY, FLAVOUR_ROWS2D_GRID) = range(10)
FLAVOUR_FLAGS = 0x100      # OR-ed onto a PLAIN / ROWS / ROWS2D flavour: its flag-writing build (fused selection)
FLAVOUR_XY = 0x200         # OR-ed onto PLAIN_ARRAY / ROWS2D_ARRAY: the build for two-row coordinates (z = 0 by contract)

_c = ctypes
_vp, _i64, _int, _sz = _c.c_void_p, _c.c_int64, _c.c_int, _c.c_size_t
_fp = _c.POINTER(_c.c_float)

# name -> (restype, argtypes); every symbol declared in include/sdfk.h
SIGNATURES = {
    "sdfk_abi_version": (_int, []),
    "sdfk_device_count": (_int, []),
    "sdfk_last_error": (_c.c_char_p, []),
    "sdfk_program_create": (_vp, [_vp, _sz, _vp, _sz, _vp, _sz, _int]),
    "sdfk_program_destroy": (None, [_vp]),
    "sdfk_program_set_params": (_int, [_vp, _vp, _sz]),
    "sdfk_program_set_cull": (_int, [_vp, _vp, _sz, _vp]),
    "sdfk_program_source": (_c.c_char_p, [_vp]),
    "sdfk_program_compile_check": (_int, [_vp, _c.POINTER(_sz)]),
    "sdfk_program_chain_members": (_int, [_vp]),
    "sdfk_program_compile_flavour": (_int, [_vp, _int, _c.POINTER(_sz), _c.POINTER(_c.c_double)]),
    "sdfk_debug_compile_external": (_int, [_vp, _int, _c.POINTER(_sz)]),
    "sdfk_debug_jit_stats": (None, [_c.POINTER(_i64), _c.POINTER(_c.c_double)]),
    "sdfk_jit_drain": (None, []),
    "sdfk_jit_cancel": (None, []),
    "sdfk_debug_set_rtc_defs": (None, [_c.c_char_p]),
    "sdfk_eval_device": (_int, [_vp, _vp, _i64, _i64, _vp, _vp, _int]),
    "sdfk_eval_device_rows": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int]),
    "sdfk_eval_device_rows2d": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int]),
    "sdfk_eval_device_rows2d_xy": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _vp, _int]),
    "sdfk_debug_cells_stats": (None, [_int, _c.POINTER(_c.c_longlong)]),
    "sdfk_eval_device_rows3d": (_int, [_vp, _vp, _i64, _i64, _i64, _i64, _i64, _vp, _vp, _int]),
    "sdfk_debug_row_masks": (_int, [_vp, _vp, _i64, _i64, _i64, _vp, _c.POINTER(_i64), _c.POINTER(_int), _vp]),
    "sdfk_eval_grid_sharded": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _vp, _int]),
    "sdfk_eval_grid_sharded_device": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _int, _vp, _int, _vp, _int]),
    "sdfk_eval_device_aux": (_int, [_vp, _vp, _i64, _i64, _vp, _int, _i64, _vp, _vp, _int]),
    "sdfk_eval_grid_aux": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _int, _i64, _vp, _vp, _int]),
    "sdfk_field_min": (_int, [_vp, _i64, _c.POINTER(_c.c_float), _vp]),
    "sdfk_grid_box_average": (_int, [_vp, _i64, _i64, _i64, _int, _int, _int, _int, _vp, _vp]),
    "sdfk_grid_edge_detect": (_int, [_vp, _i64, _i64, _i64, _vp, _vp]),
    "sdfk_grid_signed": (_int, [_vp, _i64, _i64, _i64, _c.c_float, _int, _vp, _vp]),
    "sdfk_grid_boundary_mask": (_int, [_vp, _i64, _c.c_float, _vp, _vp]),
    "sdfk_grid_signed_slab": (_int, [_vp, _i64, _i64, _vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "sdfk_eval_host": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _int, _int]),
    "sdfk_eval_host_resident": (_int, [_vp, _vp, _int, _i64, _i64, _vp, _int, _int]),
    "sdfk_field_select_scratch": (_sz, [_i64]),
    "sdfk_field_select": (_int, [_vp, _i64, _c.c_float, _vp, _i64, _c.POINTER(_i64), _vp, _vp]),
    "sdfk_field_select_finish": (_int, [_i64, _i64, _vp, _i64, _vp, _vp]),
    "sdfk_field_gradient": (_int, [_vp, _i64, _i64, _i64, _int, _int, _vp, _i64, _vp]),
    "sdfk_eval_select_scratch": (_sz, [_i64, _i64]),
    "sdfk_eval_device_select": (_int, [_vp, _vp, _i64, _i64, _i64, _int, _c.c_float, _vp, _i64, _c.POINTER(_i64), _vp, _vp, _int]),
    "sdfk_eval_grid_select": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _c.c_float, _vp, _i64, _c.POINTER(_i64),
                                     _vp, _vp, _int]),
    "sdfk_eval_select_finish": (_int, [_vp, _i64, _i64, _int, _i64, _vp, _i64, _vp, _vp]),
    "sdfk_vec_eval_device": (_int, [_vp, _int, _vp, _i64, _i64, _vp, _int, _i64, _int, _vp, _i64, _vp]),
    "sdfk_vec_set_interpret": (None, [_int]),
    "sdfk_vec_source": (_c.c_char_p, [_vp, _int, _int, _int]),
    "sdfk_vec_compile_check": (_int, [_vp, _int, _int, _int, _c.POINTER(_sz)]),
    "sdfk_vec_eval_host": (_int, [_vp, _int, _vp, _int, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _int, _int, _vp, _int]),
    "sdfk_eval_grid": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _vp, _int]),
    "sdfk_eval_grid_host": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp, _int, _int]),
    "sdfk_set_default_mode": (None, [_int]),
    "sdfk_debug_brick_masks": (_int, [_vp, _vp, _i64, _i64, _vp, _vp]),
    "sdfk_linspace_f32": (_int, [_c.c_double, _c.c_double, _i64, _vp]),
    "sdfk_point_tree_build": (_i64, [_vp, _i64, _int, _vp, _i64, _vp, _vp, _vp]),
    "sdfk_grid_fill": (_int, [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _vp]),
    "sdfk_set_device": (_int, [_int]),
    "sdfk_malloc": (_vp, [_sz]),
    "sdfk_free": (_int, [_vp]),
    "sdfk_memcpy_h2d": (_int, [_vp, _vp, _sz]),
    "sdfk_memcpy_d2h": (_int, [_vp, _vp, _sz]),
    "sdfk_memcpy_d2d": (_int, [_vp, _vp, _sz]),
    "sdfk_sync": (_int, [_vp]),
    "sdfk_event_create": (_vp, []),
    "sdfk_event_destroy": (_int, [_vp]),
    "sdfk_event_record": (_int, [_vp, _vp]),
    "sdfk_event_elapsed_ms": (_int, [_vp, _vp, _fp]),
    "sdfk_stream_probe": (_int, [_vp, _i64, _i64, _vp, _vp]),
}


class SdfkError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own HIP runtime (torch/lib/libamdhip64.so, same SONAME as /opt/rocm's).
    Two HIP runtimes in one process fight over the device: whichever initialises second sees no GPU. If torch is
    installed but not imported yet, load ITS runtime (and hiprtc) first, so that libsdfk.so — and a later
    `import torch` — resolve to the same copy. No torch: the system ROCm is used."""
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    libdir = os.path.join(list(spec.submodule_search_locations)[0], "lib")
    for name in ("libamdhip64.so", "libhiprtc.so"):
        path = os.path.join(libdir, name)
        if os.path.exists(path):
            try:
                ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
            except OSError:
                return


def lib():
    """Load libsdfk.so once. Raises (never falls back) when the extension has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise SdfkError(
                        "aegolius_amd: %s is missing - build it with "
                        "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc, gfx950). "
                        "There is no CPU path." % LIB_PATH)
                _share_torch_hip_runtime()
                handle = ctypes.CDLL(LIB_PATH)
                for name, (res, args) in SIGNATURES.items():
                    fn = getattr(handle, name)
                    fn.restype, fn.argtypes = res, args
                if handle.sdfk_abi_version() != 1:
                    raise SdfkError("libsdfk.so ABI version mismatch")
                # no background kernel build survives the interpreter (and with it torch's HIP runtime): queued ones are
                # dropped, running compiler processes killed — a big tree's build takes up to a minute
                atexit.register(handle.sdfk_jit_cancel)
                _lib = handle
    return _lib


def last_error():
    msg = lib().sdfk_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


def check(rc, what):
    if rc != 0:
        raise SdfkError("%s failed (%d): %s" % (what, rc, last_error()))


def device_count():
    return lib().sdfk_device_count()


def require_gpu():
    if device_count() < 1:
        raise SdfkError("aegolius_amd: no HIP device visible - SDF evaluation runs on an MI355X only "
                        "(there is no CPU path)")


def jit_stats():
    """(hiprtc builds this process has run — cache hits excluded —, seconds they took)."""
    n, t = _i64(0), _c.c_double(0.0)
    lib().sdfk_debug_jit_stats(ctypes.byref(n), ctypes.byref(t))
    return n.value, t.value


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None and a.size else None


class Program:
    """Owning wrapper of an `sdfk_program*` (a lowered expression tree)."""

    def __init__(self, code, params, tables, result_reg, cull_sites=None, cull_k=None):
        self.code = np.ascontiguousarray(code, dtype=np.uint32).reshape(-1, 2)
        self.params = np.ascontiguousarray(params, dtype=np.float32).ravel()
        self.tables = np.ascontiguousarray(tables, dtype=np.float32).ravel()
        self.result_reg = int(result_reg)
        self._h = lib().sdfk_program_create(_ptr(self.code), self.code.shape[0], _ptr(self.params), self.params.size,
                                            _ptr(self.tables), self.tables.size, self.result_reg)
        if not self._h:
            raise SdfkError("sdfk_program_create rejected the program: " + last_error())
        if cull_sites is not None and len(cull_sites):
            sites = np.ascontiguousarray(cull_sites, dtype=np.uint32).reshape(-1, 5)
            k = np.ascontiguousarray(cull_k, dtype=np.float32).ravel()
            check(lib().sdfk_program_set_cull(self._h, _ptr(sites), sites.shape[0], _ptr(k)), "sdfk_program_set_cull")

    @classmethod
    def from_lowered(cls, low, cull=True):
        """Program of a LoweredProgram (aegolius_amd._lower), with its brick-culling sites."""
        if cull:
            return cls(low.code, low.params, low.tables, low.result_reg, low.cull_sites, low.cull_k)
        return cls(low.code, low.params, low.tables, low.result_reg)

    @property
    def handle(self):
        return self._h

    def set_params(self, params):
        p = np.ascontiguousarray(params, dtype=np.float32).ravel()
        check(lib().sdfk_program_set_params(self._h, _ptr(p), p.size), "sdfk_program_set_params")
        self.params = p

    def source(self):
        s = lib().sdfk_program_source(self._h)
        return s.decode() if s else None

    @property
    def chain_members(self):
        """Members of the n-ary chain when the program runs on the table-driven chain kernels, else 0."""
        return int(lib().sdfk_program_chain_members(self._h))

    def compile_check(self):
        n = _sz(0)
        check(lib().sdfk_program_compile_check(self._h, ctypes.byref(n)), "sdfk_program_compile_check")
        return n.value

    def compile_flavour(self, flavour):
        """Build (or fetch) one kernel flavour (FLAVOUR_*), GPU or not -> (code-object bytes, seconds)."""
        n, t = _sz(0), _c.c_double(0.0)
        check(lib().sdfk_program_compile_flavour(self._h, int(flavour), ctypes.byref(n), ctypes.byref(t)),
              "sdfk_program_compile_flavour")
        return n.value, t.value

    def eval_host(self, co, device=0, mode=MODE_AUTO):
        """co: (3, N) float32/float64 host array -> (N,) float32 field."""
        require_gpu()
        co = np.asarray(co)
        if co.ndim != 2 or co.shape[0] != 3:
            raise ValueError("coordinates must have shape (3, N); got %r" % (co.shape,))
        if co.dtype not in (np.float32, np.float64):
            co = co.astype(np.float64)
        co = np.ascontiguousarray(co)
        n = co.shape[1]
        out = np.empty(n, dtype=np.float32)
        check(lib().sdfk_eval_host(self._h, _ptr(co), 0 if co.dtype == np.float32 else 1, n, n, _ptr(out), device,
                                   mode), "sdfk_eval_host")
        return out

    def eval_device(self, d_co, n, row_stride, d_out, stream=None, mode=MODE_AUTO, row_len=None, flat=False,
                    plane_rows=None, first_row_in_plane=0):
        """Device pointers (ints). Asynchronous on `stream` (a hipStream_t as int, None = default).
        `row_len`: layout hint — the points are consecutive rows of that many points (the last grid
        dimension of a generate_grid array); speeds up brick culling, never changes the field.
        `flat`: the rows are those of a flat (two-size) grid: z = 0, rows along y.
        `plane_rows`, `first_row_in_plane`: 3-D grids — rows per plane (the second grid dimension) and where in its
        plane the array starts (x-slabs of whole rows): row blocks then never straddle two planes."""
        if row_len and plane_rows and not flat:
            check(lib().sdfk_eval_device_rows3d(self._h, _vp(d_co), n, row_stride, int(row_len), int(plane_rows),
                                                int(first_row_in_plane), _vp(d_out), _vp(stream or 0), mode),
                  "sdfk_eval_device_rows3d")
            return
        if row_len and flat:
            check(lib().sdfk_eval_device_rows2d(self._h, _vp(d_co), n, row_stride, int(row_len), _vp(d_out),
                                                _vp(stream or 0), mode), "sdfk_eval_device_rows2d")
            return
        if row_len:
            check(lib().sdfk_eval_device_rows(self._h, _vp(d_co), n, row_stride, int(row_len), _vp(d_out),
                                              _vp(stream or 0), mode), "sdfk_eval_device_rows")
            return
        check(lib().sdfk_eval_device(self._h, _vp(d_co), n, row_stride, _vp(d_out), _vp(stream or 0), mode),
              "sdfk_eval_device")

    def eval_device_xy(self, d_xy, n, row_stride, d_out, stream=None, mode=MODE_AUTO, row_len=None):
        """Two coordinate rows (x, y) of a flat grid, z = 0 by contract: 12 instead of 16 bytes per point."""
        check(lib().sdfk_eval_device_rows2d_xy(self._h, _vp(d_xy), n, row_stride, int(row_len or 0), _vp(d_out),
                                               _vp(stream or 0), mode), "sdfk_eval_device_rows2d_xy")

    def eval_grid(self, axes, start, count, d_out, stream=None, mode=MODE_AUTO):
        ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
        check(lib().sdfk_eval_grid(self._h, _ptr(ax[0]), ax[0].size, _ptr(ax[1]), ax[1].size, _ptr(ax[2]),
                                   ax[2].size, start, count, _vp(d_out), _vp(stream or 0), mode), "sdfk_eval_grid")

    def _select(self, n, first, device, row_len=0, mode=MODE_AUTO):
        """Two-step protocol of the fused selection: `first(d_scratch, byref(count))` evaluates into flags and counts;
        the indices follow from the flags. -> ascending int64 host array."""
        require_gpu()
        L = lib()
        check(L.sdfk_set_device(int(device)), "sdfk_set_device")
        m = _i64(0)
        d_scratch = L.sdfk_malloc(L.sdfk_eval_select_scratch(n, int(row_len)))
        d_index = None
        try:
            if not d_scratch:
                raise SdfkError("select: out of device memory")
            first(_vp(d_scratch), ctypes.byref(m))
            out = np.empty(m.value, dtype=np.int64)
            if m.value:
                d_index = L.sdfk_malloc(m.value * 8)
                if not d_index:
                    raise SdfkError("select: out of device memory")
                check(L.sdfk_eval_select_finish(self._h, n, int(row_len), mode, m.value, _vp(d_index), m.value, _vp(d_scratch),
                                                None), "sdfk_eval_select_finish")
                check(L.sdfk_memcpy_d2h(_ptr(out), _vp(d_index), out.size * 8), "sdfk_memcpy_d2h")
            return out
        finally:
            for d in (d_scratch, d_index):
                if d:
                    L.sdfk_free(_vp(d))

    def select_grid(self, axes, threshold=0.0, start=0, count=None, device=0, mode=MODE_AUTO):
        """numpy.flatnonzero(field <= threshold) of the grid spanned by three per-axis tables, WITHOUT a field: the
        evaluation kernels write one flag bit per point, the compaction reads the flags (sdfk_eval_grid_select)."""
        ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
        total = ax[0].size * ax[1].size * ax[2].size
        count = total - start if count is None else count

        def first(d_scratch, m):
            check(lib().sdfk_eval_grid_select(self._h, _ptr(ax[0]), ax[0].size, _ptr(ax[1]), ax[1].size, _ptr(ax[2]),
                                              ax[2].size, start, count, float(threshold), None, 0, m, d_scratch, None,
                                              mode), "sdfk_eval_grid_select")
        grow = ax[2].size if ax[2].size > 1 else ax[1].size
        return self._select(count, first, device, row_len=grow if start % grow == 0 and count % grow == 0 else 0, mode=mode)

    def select_host(self, co, threshold=0.0, device=0, mode=MODE_AUTO):
        """The same for a (3, N) host array (uploaded once as float32; the row-length hint is detected like create())."""
        require_gpu()
        co = np.asarray(co)
        if co.ndim != 2 or co.shape[0] != 3:
            raise ValueError("coordinates must have shape (3, N); got %r" % (co.shape,))
        co32 = np.ascontiguousarray(co, dtype=np.float32)
        n = co32.shape[1]
        if n == 0:
            return np.empty(0, dtype=np.int64)
        L = lib()
        check(L.sdfk_set_device(int(device)), "sdfk_set_device")
        stride = (n + 63) // 64 * 64
        d_co = L.sdfk_malloc(3 * stride * 4)
        if not d_co:
            raise SdfkError("select: out of device memory")
        try:
            for r in range(3):
                check(L.sdfk_memcpy_h2d(_vp(d_co + 4 * r * stride), _ptr(co32[r]), n * 4), "sdfk_memcpy_h2d")
            # rows: the index at which x or y first changes (3-D grids), else at which x first changes (flat grids)
            row_len, flat = 0, 0
            head = co32[:, :min(n, 1 << 22)]
            ch = np.flatnonzero((head[0] != head[0, 0]) | (head[1] != head[1, 0]))
            if ch.size and ch[0] >= 32 and n % int(ch[0]) == 0:
                row_len = int(ch[0])
            else:
                cx = np.flatnonzero(head[0] != head[0, 0])
                if cx.size and cx[0] >= 32 and n % int(cx[0]) == 0:
                    row_len = int(cx[0])
                    flat = int(co32[2, 0] == 0 and co32[2, row_len - 1] == 0 and co32[2, -1] == 0)

            def first(d_scratch, m):
                check(L.sdfk_eval_device_select(self._h, _vp(d_co), n, stride, row_len, flat, float(threshold), None, 0, m,
                                                d_scratch, None, mode), "sdfk_eval_device_select")
            return self._select(n, first, device, row_len=row_len, mode=mode)
        finally:
            L.sdfk_free(_vp(d_co))

    def eval_grid_host(self, axes, start=0, count=None, device=0, mode=MODE_AUTO):
        """Field of the grid spanned by three per-axis tables (flat index z fastest), straight to a host array."""
        require_gpu()
        ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
        total = ax[0].size * ax[1].size * ax[2].size
        count = total - start if count is None else count
        out = np.empty(count, dtype=np.float32)
        check(lib().sdfk_eval_grid_host(self._h, _ptr(ax[0]), ax[0].size, _ptr(ax[1]), ax[1].size, _ptr(ax[2]),
                                        ax[2].size, start, count, _ptr(out), device, mode), "sdfk_eval_grid_host")
        return out

    def eval_grid_sharded(self, axes, n_shards, devices=None, mode=MODE_AUTO):
        """Whole grid, cut into `n_shards` slabs of whole rows evaluated concurrently on `devices`
        (default: shard d on device d modulo the device count), field returned as one host array."""
        require_gpu()
        ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
        out = np.empty(ax[0].size * ax[1].size * ax[2].size, dtype=np.float32)
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        if dev is not None and dev.size != n_shards:
            raise ValueError("one device per shard")
        check(lib().sdfk_eval_grid_sharded(self._h, _ptr(ax[0]), ax[0].size, _ptr(ax[1]), ax[1].size, _ptr(ax[2]),
                                           ax[2].size, n_shards, _ptr(dev) if dev is not None else None, _ptr(out),
                                           mode), "sdfk_eval_grid_sharded")
        return out

    def eval_grid_sharded_resident(self, axes, n_shards, devices=None, gather_device=0, mode=MODE_AUTO):
        """The same partition, the field reassembled on `gather_device` by device-to-device copies (no host buffer)
        -> DeviceField on that device."""
        require_gpu()
        ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
        dev = None if devices is None else np.ascontiguousarray(devices, dtype=np.int32)
        if dev is not None and dev.size != n_shards:
            raise ValueError("one device per shard")
        field = DeviceField(ax[0].size * ax[1].size * ax[2].size, gather_device)
        check(lib().sdfk_eval_grid_sharded_device(self._h, _ptr(ax[0]), ax[0].size, _ptr(ax[1]), ax[1].size, _ptr(ax[2]),
                                                  ax[2].size, n_shards, _ptr(dev) if dev is not None else None,
                                                  int(gather_device), _vp(field.ptr), mode), "sdfk_eval_grid_sharded_device")
        return field

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            try:
                _lib.sdfk_program_destroy(h)
            except Exception:
                pass


class DeviceField:
    """An (N,) float32 scalar field resident in HBM (what `create_resident` returns): the consumers of the field run
    on it without the field ever crossing PCIe. Owns its device memory; `free()` (or garbage collection) releases it."""

    def __init__(self, n, device=0):
        require_gpu()
        self.n, self.device = int(n), int(device)
        check(lib().sdfk_set_device(self.device), "sdfk_set_device")
        self.ptr = lib().sdfk_malloc(max(self.n, 1) * 4)
        if not self.ptr:
            raise SdfkError("DeviceField: out of device memory: " + last_error())

    @classmethod
    def from_host(cls, field, device=0):
        host = np.ascontiguousarray(field, dtype=np.float32).ravel()
        self = cls(host.size, device)
        if host.size:
            check(lib().sdfk_memcpy_h2d(_vp(self.ptr), _ptr(host), host.size * 4), "sdfk_memcpy_h2d")
        return self

    def _live(self):
        if not self.ptr:
            raise SdfkError("DeviceField has been freed")
        check(lib().sdfk_set_device(self.device), "sdfk_set_device")

    def numpy(self):
        self._live()
        out = np.empty(self.n, dtype=np.float32)
        if self.n:
            check(lib().sdfk_memcpy_d2h(_ptr(out), _vp(self.ptr), self.n * 4), "sdfk_memcpy_d2h")
        return out

    def count(self, threshold=0.0):
        """Number of points with field <= threshold."""
        self._live()
        m = _i64(0)
        check(lib().sdfk_field_select(_vp(self.ptr), self.n, float(threshold), None, 0, ctypes.byref(m), None, None),
              "sdfk_field_select")
        return m.value

    def select(self, threshold=0.0):
        """Ascending int64 indices of the points with field <= threshold (numpy.flatnonzero(field <= threshold));
        only the indices cross PCIe."""
        self._live()
        L = lib()
        m = _i64(0)
        d_scratch = L.sdfk_malloc(L.sdfk_field_select_scratch(self.n))
        d_index = None
        try:
            if not d_scratch:
                raise SdfkError("select: out of device memory")
            check(L.sdfk_field_select(_vp(self.ptr), self.n, float(threshold), None, 0, ctypes.byref(m), _vp(d_scratch),
                                      None), "sdfk_field_select")
            out = np.empty(m.value, dtype=np.int64)
            if m.value:
                d_index = L.sdfk_malloc(m.value * 8)
                if not d_index:
                    raise SdfkError("select: out of device memory")
                check(L.sdfk_field_select_finish(self.n, m.value, _vp(d_index), m.value, _vp(d_scratch), None),
                      "sdfk_field_select_finish")
                check(L.sdfk_memcpy_d2h(_ptr(out), _vp(d_index), out.size * 8), "sdfk_memcpy_d2h")
            return out
        finally:
            for d in (d_scratch, d_index):
                if d:
                    L.sdfk_free(_vp(d))

    def gradient(self, shape, normalize=True):
        """numpy.gradient (unit spacing) of the field reshaped to `shape` (1 to 3 axes), every vector normalised
        unless its norm is 0 -> (len(shape), N) float32 host array."""
        self._live()
        shape = tuple(int(x) for x in shape)
        if not 1 <= len(shape) <= 3 or int(np.prod(shape)) != self.n:
            raise ValueError("cannot reshape a field of %d points to %r" % (self.n, shape))
        if min(shape) < 2:
            raise ValueError("Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements are required.")
        dims = (1,) * (3 - len(shape)) + shape
        L = lib()
        stride = (self.n + 63) // 64 * 64
        d_vec = L.sdfk_malloc(len(shape) * stride * 4)
        if not d_vec:
            raise SdfkError("gradient: out of device memory")
        try:
            check(L.sdfk_field_gradient(_vp(self.ptr), dims[0], dims[1], dims[2], len(shape), 1 if normalize else 0,
                                        _vp(d_vec), stride, None), "sdfk_field_gradient")
            out = np.empty((len(shape), self.n), dtype=np.float32)
            for r in range(len(shape)):
                check(L.sdfk_memcpy_d2h(_ptr(out[r]), _vp(d_vec + 4 * r * stride), self.n * 4), "sdfk_memcpy_d2h")
            return out
        finally:
            L.sdfk_free(_vp(d_vec))

    def gradient_resident(self, shape, normalize=True):
        """gradient() of a 3-D field, left on the device as a DeviceVectorField."""
        self._live()
        shape = tuple(int(x) for x in shape)
        if len(shape) != 3 or int(np.prod(shape)) != self.n:
            raise ValueError("gradient_resident takes a 3-D grid of %d points; got %r" % (self.n, shape))
        if min(shape) < 2:
            raise ValueError("Shape of array too small to calculate a numerical gradient, at least (edge_order + 1) elements are required.")
        out = DeviceVectorField(self.n, self.device)
        check(lib().sdfk_field_gradient(_vp(self.ptr), shape[0], shape[1], shape[2], 3, 1 if normalize else 0, _vp(out.ptr),
                                        out.stride, None), "sdfk_field_gradient")
        return out

    def free(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p and _lib is not None:
            _lib.sdfk_free(_vp(p))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceVectorField:
    """A (3, N) float32 vector field resident in HBM (what `VectorField.create_resident` returns): three rows of
    `stride` floats. Usable as the input, a second field or the revolution coordinates of another vector-field chain
    without crossing PCIe. Owns its device memory."""

    def __init__(self, n, device=0):
        require_gpu()
        self.n, self.device = int(n), int(device)
        self.stride = (self.n + 63) // 64 * 64
        check(lib().sdfk_set_device(self.device), "sdfk_set_device")
        self.ptr = lib().sdfk_malloc(max(self.stride, 64) * 3 * 4)
        if not self.ptr:
            raise SdfkError("DeviceVectorField: out of device memory: " + last_error())

    shape = property(lambda self: (3, self.n))

    @classmethod
    def from_host(cls, vec, device=0):
        host = np.ascontiguousarray(vec, dtype=np.float32)
        if host.ndim != 2 or host.shape[0] != 3:
            raise ValueError("a vector field has shape (3, N); got %r" % (host.shape,))
        self = cls(host.shape[1], device)
        for r in range(3):
            if self.n:
                check(lib().sdfk_memcpy_h2d(_vp(self.ptr + 4 * r * self.stride), _ptr(host[r]), self.n * 4), "sdfk_memcpy_h2d")
        return self

    def row_ptr(self, r):
        if not self.ptr:
            raise SdfkError("DeviceVectorField has been freed")
        return self.ptr + 4 * r * self.stride

    def numpy(self):
        check(lib().sdfk_set_device(self.device), "sdfk_set_device")
        out = np.empty((3, self.n), dtype=np.float32)
        for r in range(3):
            if self.n:
                check(lib().sdfk_memcpy_d2h(_ptr(out[r]), _vp(self.row_ptr(r)), self.n * 4), "sdfk_memcpy_d2h")
        return out

    def free(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p and _lib is not None:
            _lib.sdfk_free(_vp(p))

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def linspace_f32(lo, hi, n):
    out = np.empty(int(n), dtype=np.float32)
    check(lib().sdfk_linspace_f32(float(lo), float(hi), int(n), _ptr(out)), "sdfk_linspace_f32")
    return out


def point_tree(points32, leaf, with_order=False):
    """sdfk_point_tree_build: (m, 3) fp32 points -> (table, n_root, point_base, order or None). Host only."""
    pts = np.ascontiguousarray(points32, dtype=np.float32)
    if pts.ndim != 2 or pts.shape[1] != 3 or pts.shape[0] < 1:
        raise ValueError("points must have shape (M, 3), M >= 1; got %r" % (pts.shape,))
    m = int(pts.shape[0])
    half = max(1, int(leaf) // 2)
    cap = 3 * m + 24 * (m // half + 4)
    table = np.empty(cap, dtype=np.float32)
    order = np.empty(m, dtype=np.int64) if with_order else None
    n_root, point_base = _c.c_int64(0), _c.c_int64(0)
    used = lib().sdfk_point_tree_build(_ptr(pts), m, int(leaf), _ptr(table), cap, _ptr(order) if with_order else None,
                                       _c.byref(n_root), _c.byref(point_base))
    if used == -2:
        raise ValueError(last_error())
    if used < 0:
        raise SdfkError("sdfk_point_tree_build: " + last_error())
    return table[:used].copy(), int(n_root.value), int(point_base.value), order


def grid_fill(d_co, row_stride, axes, start, count, stream=None):
    ax = [np.ascontiguousarray(a, dtype=np.float32) for a in axes]
    check(lib().sdfk_grid_fill(_vp(d_co), row_stride, _ptr(ax[0]), ax[0].size, _ptr(ax[1]), ax[1].size, _ptr(ax[2]),
                               ax[2].size, start, count, _vp(stream or 0)), "sdfk_grid_fill")


class Event:
    def __init__(self):
        self._h = lib().sdfk_event_create()
        if not self._h:
            raise SdfkError("sdfk_event_create: " + last_error())

    def record(self, stream=None):
        check(lib().sdfk_event_record(self._h, _vp(stream or 0)), "sdfk_event_record")

    def elapsed_ms(self, stop):
        ms = _c.c_float(0)
        check(lib().sdfk_event_elapsed_ms(self._h, stop._h, ctypes.byref(ms)), "sdfk_event_elapsed_ms")
        return ms.value

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None:
            try:
                _lib.sdfk_event_destroy(h)
            except Exception:
                pass
