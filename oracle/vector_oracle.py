"""CPU oracle for the vector-field path — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A float64 NumPy restatement of the reference's vector-field arithmetic (SPOMSO 1.4.0, `Code/spomso/spomso/cores/`,
abbreviated C/): the field definitions of C/vector_functions.py:15-127, the modifications of
C/vector_modification_functions.py:14-160 as chained by C/modifications.py:1666-1975, and the read-outs of
C/geom.py:256-362. It walks the `VecClosure` that `aegolius_amd` records (leaf + list of modifications) and never
touches its lowering or libsdfk.so. Only tests/, smoke() and bench.py's cpu_baseline leg may import this module.

Pinning: tests/test_vector_golden.py checks every function here against golden vectors produced by the REAL reference in
the build container (tests/golden/generate_vector_golden.py; tests/golden/vector_golden.npz) to <= 1e-12.
"""
import numpy as np

from aegolius_amd._vector import VecClosure, _leaf_name, as_closure


def unit(v):                                         # batch_normalize, C/vector_modification_functions.py:14-20
    v = np.array(v, dtype=np.float64)
    m = np.sqrt((v * v).sum(axis=0))
    nz = m != 0
    v[:, nz] = v[:, nz] / m[nz]
    return v


def _angle(a):
    return np.squeeze(np.asarray(a, dtype=np.float64))


def _planar(p):
    q = np.array(p, dtype=np.float64)
    q[2] = 0
    return unit(q)


def _turn_xy(v, a):                                  # rotate_vectors_phi / _z_axis, :44-52, :95-103
    s, c = np.sin(a), np.cos(a)
    return np.asarray([v[0] * c - v[1] * s, v[0] * s + v[1] * c, v[2] + 0 * s])


def leaf(name, p, params):
    p = np.asarray(p, dtype=np.float64)
    if name == "cartesian_define":                   # C/vector_functions.py:15-20
        return np.asarray((p[0], p[1], p[2]))
    if name == "spherical_define":                   # :23-32
        r, phi, theta = p
        return np.asarray((r * np.cos(phi) * np.sin(theta), r * np.sin(phi) * np.sin(theta), r * np.cos(theta)))
    if name == "cylindrical_define":                 # :35-43
        r, phi, z = p
        return np.asarray((r * np.cos(phi), r * np.sin(phi), z))
    if name == "radial_vector_field_spherical":      # :46-48
        return unit(p)
    if name == "radial_vector_field_cylindrical":    # :51-55
        return _planar(p)
    if name == "vortex_vector_field_cylindrical":    # :71-79
        v = _planar(p)
        return np.asarray([-v[1], v[0], v[2]])
    if name == "aar_vector_field_cylindrical":       # :82-94
        return _turn_xy(_planar(p), _angle(params[0]))
    if name == "aav_vector_field_cylindrical":       # :97-109
        v, a = _planar(p), _angle(params[0])
        s, c = np.sin(a), np.cos(a)
        return np.asarray([-v[0] * s - v[1] * c, v[0] * c - v[1] * s, v[2] + 0 * s])
    if name in ("x_vector_field", "y_vector_field", "z_vector_field"):   # :112-127
        v = np.zeros(p.shape)
        v["xyz".index(name[0])] = 1
        return v
    if name in ("hyperbolic_vector_field_cylindrical", "awn_vector_field_cylindrical"):   # :58-68: wrong arity inside
        raise TypeError("cylindrical_define() takes 1 positional argument but 3 were given")
    raise KeyError(name)


def modify(v, name, args):
    if name in ("add", "subtract"):                  # :23-36
        a = np.asarray(args[0], dtype=np.float64)
        sign = 1.0 if name == "add" else -1.0
        return (v.T + sign * a.reshape(3)).T if a.size == 3 else v + sign * a
    if name == "rescale":                            # :39-41
        return v * np.asarray(args[0], dtype=np.float64)
    if name in ("rotate_phi", "rotate_z"):
        return _turn_xy(v, _angle(args[0]))
    if name == "rotate_x":                           # :71-80
        s, c = np.sin(_angle(args[0])), np.cos(_angle(args[0]))
        return np.asarray([v[0] + 0 * s, v[1] * c - v[2] * s, v[1] * s + v[2] * c])
    if name == "rotate_y":                           # :83-92
        s, c = np.sin(_angle(args[0])), np.cos(_angle(args[0]))
        return np.asarray([v[0] * c - v[2] * s, v[1] + 0 * s, v[0] * s + v[2] * c])
    if name == "rotate_theta":                       # :55-68
        r = _planar(v)
        s, c = np.sin(_angle(args[0])), np.cos(_angle(args[0]))
        t = np.asarray([r[0] * v[2], r[1] * v[2], -r[0] * v[0] - r[1] * v[1]])
        return v * c + t * s
    if name == "rotate_axis":                        # :106-119 (axes are not normalised)
        ax = np.asarray(args[0], dtype=np.float64)
        ax = np.repeat(ax.reshape(3, 1), v.shape[1], axis=1) if ax.size == 3 else ax
        s, c = np.sin(_angle(args[1])), np.cos(_angle(args[1]))
        return v * c + s * np.cross(ax.T, v.T).T + (1 - c) * ax * (ax * v).sum(axis=0)
    if name in ("revolution_x", "revolution_y", "revolution_z"):   # :122-159
        r = np.asarray(args[0], dtype=np.float64)
        i, j = {"x": (1, 2), "y": (0, 2), "z": (0, 1)}[name[-1]]
        a = np.arctan2(r[j], r[i])
        s, c = np.sin(a), np.cos(a)
        out = np.array(v, dtype=np.float64)
        out[i] = v[i] * c - v[j] * s
        out[j] = v[i] * s + v[j] * c
        return out
    if name == "normalize":
        return unit(v)
    raise KeyError(name)


def evaluate(closure, p, params=(), out="vector"):
    """closure(p, *params) of the reference in float64; `out` = vector | x | y | z | phi | theta | length."""
    closure = as_closure(closure)
    mods = closure.mods
    inner = closure.leaf
    while isinstance(inner, VecClosure):
        mods = inner.mods + mods
        inner = inner.leaf
    name = _leaf_name(inner)
    if name == "from_sdf":
        from oracle.sdf_oracle import from_sdf
        v = from_sdf(np.asarray(p, dtype=np.float64), *params)
    elif name is not None:
        v = leaf(name, p, params)
    elif hasattr(getattr(inner, "__self__", None), "_vf_parameters"):      # other_field.propagate used as a leaf
        owner = inner.__self__
        v = evaluate(owner.vf, p, owner._vf_parameters)
    else:
        v = np.asarray(inner(p, *params), dtype=np.float64)
    for mod_name, args in mods:
        v = modify(v, mod_name, args)
    if out == "vector":
        return v
    if out in "xyz":
        return v["xyz".index(out)]
    if out == "phi":                                 # C/geom.py:307-324
        return np.arctan2(v[1], v[0])
    if out == "theta":                               # :326-343
        return np.arccos(v[2])
    if out == "length":                              # :345-362
        return np.sqrt((v * v).sum(axis=0))
    raise KeyError(out)
