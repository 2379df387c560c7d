"""Factory for the thin shape classes of geom_2d / geom_3d.

The reference spells every shape out as a class that stores its constructor arguments, binds an
`sdf_*` function and exposes read-only properties (reference cores/geom_3d.py, cores/geom_2d.py).
Here one declarative row per shape generates the equivalent class: constructor parameter names
(so keyword calls keep working), the tuple handed to the SDF, and the properties.
"""
import inspect

from .geom import GenericGeometry


def shape(name, sdf, fields, pack=None, props=None, doc=""):
    """fields: constructor parameter names, in order.
    pack(**fields) -> tuple of SDF parameters (default: the fields themselves).
    props: {property name: function(fields dict) -> value} (default: one property per field)."""
    params = [inspect.Parameter("self", inspect.Parameter.POSITIONAL_OR_KEYWORD)]
    params += [inspect.Parameter(f, inspect.Parameter.POSITIONAL_OR_KEYWORD) for f in fields]
    sig = inspect.Signature(params)

    def __init__(self, *args, **kwargs):
        try:
            bound = sig.bind(self, *args, **kwargs)
        except TypeError as exc:
            raise TypeError("%s.__init__(): %s" % (name, exc)) from None
        vals = {k: v for k, v in bound.arguments.items() if k != "self"}
        sdf_args = pack(**vals) if pack else tuple(vals[f] for f in fields)
        GenericGeometry.__init__(self, sdf, *sdf_args)
        self._fields = vals

    __init__.__signature__ = sig
    ns = {"__init__": __init__, "__doc__": doc, "__module__": __name__}
    for pname, getter in (props or {f: (lambda v, f=f: v[f]) for f in fields}).items():
        ns[pname] = property(lambda self, g=getter: g(self._fields))
    return type(name, (GenericGeometry,), ns)
