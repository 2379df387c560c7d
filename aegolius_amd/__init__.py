"""aegolius_amd — MI355X-native SDF grid evaluation behind the SPOMSO geometry API.

    from aegolius_amd.cores import generate_grid, Sphere, Box, CombineGeometry
    co, res = generate_grid((2, 2, 2), (128, 128, 128))
    field = CombineGeometry("SMOOTH_UNION2").combine_parametric(Sphere(0.3), Box(.5, .4, .3),
                                                                parameters=0.1).create(co)

The geometry objects record a symbolic expression tree; `create()` lowers it to a register-machine
program and runs it as ONE fused per-point HIP kernel (libsdfk.so, gfx950). There is no CPU path.
"""
from ._eval import config  # noqa: F401
from ._engine import DeviceField, DeviceVectorField  # noqa: F401
from . import cores  # noqa: F401

__version__ = "0.1.0"
