"""Grid builder and reshape helpers — API mirror of reference cores/helper_functions.py:10-148.

`generate_grid` returns the same `(co, resolution)` pair as the reference: a float64 (3, N)
point cloud (z fastest) plus the odd-converted resolution. For grids that should never exist on the
host, use `aegolius_amd.DeviceGrid`, which expands the per-axis tables directly in HBM.
"""
import numpy as np


def resolution_conversion(resolution):
    """Force an odd number of points along an axis (reference :10-20)."""
    return int(resolution if resolution % 2 == 1 else resolution + 1)


def grid_axes(size, resolution):
    """Per-axis float64 coordinate tables + converted resolution of generate_grid (reference :40-88).
    Missing axes (1-D / 2-D grids) are a single 0.0."""
    resolution = np.asarray(resolution)
    if resolution.size == 1:
        r0 = resolution_conversion(resolution)
        res = (r0, r0, r0)
    elif resolution.size == 2:
        r0, r1 = resolution_conversion(resolution[0]), resolution_conversion(resolution[1])
        res = (r0, r1, r0)           # reference quirk: the third entry repeats the first (:45-48)
    elif resolution.size == 3:
        res = tuple(resolution_conversion(r) for r in resolution)
    else:
        raise UnboundLocalError("resolution must have 1, 2 or 3 entries")
    size = np.asarray(size)
    if size.size not in (1, 2, 3):
        raise UnboundLocalError("size must have 1, 2 or 3 entries")
    axes = [np.linspace(-size[k] / 2, size[k] / 2, res[k]) for k in range(size.size)]
    axes += [np.zeros(1)] * (3 - size.size)
    return axes, res


class GridCoords(np.ndarray):
    """The (3, N) array returned by generate_grid, tagged with the per-axis tables it was built from.

    `create(co)` recognises the tag and evaluates the grid straight from the three small tables on the
    GPU (no 12 B/point upload, no device coordinate array). The tag is only trusted while the array
    cannot have changed: the array is handed out read-only, and every view, slice, copy or ufunc result is
    a plain untagged ndarray. `co.setflags(write=True)` (or `config.grid_fast_path = False`) opts out."""

    def __array_finalize__(self, obj):
        self._grid_axes = None          # views / copies / results never inherit the tag

    def __reduce__(self):               # pickles as a plain array
        return np.asarray(self).__reduce__()

    @property
    def grid_axes(self):
        """The float64 axis tables if this array is still guaranteed to be the untouched grid, else None."""
        if self._grid_axes is not None and not self.flags.writeable:
            return self._grid_axes
        return None


def generate_grid(size, resolution):
    """Grid of points centred at zero: (co (3, N) float64, (res0, res1, res2))."""
    from .._eval import config
    axes, res = grid_axes(size, resolution)
    n = [a.size for a in axes]
    shape = (3, n[0] * n[1] * n[2])
    co = GridCoords(shape, dtype=np.float64) if config.grid_fast_path else np.empty(shape)
    shaped = np.asarray(co).reshape(3, *n)
    shaped[0] = axes[0][:, None, None]
    shaped[1] = axes[1][None, :, None]
    shaped[2] = axes[2][None, None, :]
    if config.grid_fast_path:
        co._grid_axes = [a.copy() for a in axes]
        co.setflags(write=False)
    return co, res


def smarter_reshape(pattern, resolution):
    """(N,) field -> grid-shaped array (reference :96-148)."""
    n_ele = pattern.shape[0]
    resolution = np.asarray(resolution)
    bad = ValueError(f"Cannot reshape the pattern with shape {pattern.shape}")
    if resolution.size == 1:
        res = resolution_conversion(resolution)
        for dim in (1, 2, 3):
            if n_ele // res ** dim == 1:
                return pattern if dim == 1 else pattern.reshape((res,) * dim)
        raise bad
    if resolution.size == 2:
        r0, r1 = resolution_conversion(resolution[0]), resolution_conversion(resolution[1])
        div = n_ele // (r0 * r1)
        return pattern.reshape(r0, r1) if div == 1 else pattern.reshape(r0, r1, int(div))
    if resolution.size == 3:
        r = [resolution_conversion(x) for x in resolution]
        if n_ele // (r[0] * r[1] * r[2]) == 1:
            return pattern.reshape(*r)
        raise bad


def vector_smarter_reshape(pattern, resolution):
    """(3, N) vector field -> array of shape (3, *grid shape) (reference :151-166)."""
    return np.asarray([smarter_reshape(pattern[i], resolution) for i in range(3)])


def nd_vector_smarter_reshape(pattern, resolution):
    """(ND, N) vector field -> array of shape (ND, *grid shape) (reference :169-188)."""
    first = smarter_reshape(pattern[0], resolution)
    out = np.zeros((pattern.shape[0],) + first.shape)
    out[0] = first
    for i in range(1, pattern.shape[0]):
        out[i] = smarter_reshape(pattern[i], resolution)
    return out


def binning(pattern, bins, equal_width=True):
    """Discretise a field into `bins` levels between its minimum and maximum (reference :191-215)."""
    lo, span = np.amin(pattern), np.amax(pattern) - np.amin(pattern)
    v = (pattern - lo) / span
    if equal_width:
        u = (v * bins).astype(int) / (bins - 1)
    else:
        u = np.round(v * (bins - 1), 0) / (bins - 1)
    return u * span + lo
