"""Symbolic expression nodes.

The reference builds its scene as nested Python closures `f(co, *params)` (reference
cores/modifications.py:55-63, cores/combine.py:129-138). Closures are opaque, so the same builder
calls here record small immutable expression objects instead; `_lower.py` flattens them into the
register-machine program that libsdfk.so runs on the GPU. Every expression is still *callable* with
the reference's signature `expr(co, *params) -> (N,) field`, because reference user code passes
these objects around as functions (e.g. `obj.sign(direct=True)` into `recover_volume`).
"""


import inspect


class SDFExpr:
    """Callable stand-in for one reference closure."""

    def __call__(self, co, *params):
        from ._eval import evaluate_expr
        return evaluate_expr(self, co, params)


class PrimSDF(SDFExpr):
    """One of the reference's `sdf_*` functions (cores/sdf_3D.py, cores/sdf_2D.py). `arg_names` are the reference's
    parameter names after `co` (sdf_sphere(co, radius), sdf_box(co, size) ...): the object reports that signature
    (inspect.signature) and takes the arguments by position or by keyword, like the reference's function."""

    def __init__(self, name, lower, doc="", arg_names=None):
        self.name = name
        self.lower = lower  # lower(L, vdst, creg, args)
        self.__name__ = self.__qualname__ = name
        self.__doc__ = doc
        if arg_names is not None:
            kind = inspect.Parameter.POSITIONAL_OR_KEYWORD
            self.__signature__ = inspect.Signature([inspect.Parameter(n, kind) for n in ("co",) + tuple(arg_names)])

    def __call__(self, *args, **kwargs):
        sig = getattr(self, "__signature__", None)
        if sig is None:
            if kwargs:
                raise TypeError("%s() takes no keyword arguments" % self.name)
            co, params = args[0], args[1:]
        else:
            try:
                bound = sig.bind(*args, **kwargs)
            except TypeError as exc:
                raise TypeError("%s() %s" % (self.name, exc)) from None
            co, params = bound.args[0], bound.args[1:]
        from ._eval import evaluate_expr
        return evaluate_expr(self, co, params)

    def __repr__(self):
        return "<sdf primitive %s>" % self.name


class ModSDF(SDFExpr):
    """A modification closure: captures the chain `inner` as it was when the method was called
    (reference: `geo_object = self.geo_object` at the top of every ModifyObject method)."""

    def __init__(self, name, args, inner, second=None, second_params=None):
        self.name = name
        self.args = args
        self.inner = inner
        self.second = second                # second field: interior / displacement function
        self.second_params = second_params  # None -> called with the geometry's own parameters

    def __repr__(self):
        return "<modification %s of %r>" % (self.name, self.inner)


class CombineSDF(SDFExpr):
    """`new_geo_object` of CombineGeometry.combine / combine_parametric (reference
    cores/combine.py:129-135, 154-160). The operation name is read from the owning CombineGeometry
    at evaluation time, as in the reference."""

    def __init__(self, owner, children, parametric, parameters=None):
        self.owner = owner
        self.children = tuple(children)
        self.parametric = parametric
        self.parameters = parameters

    def __repr__(self):
        return "<combine %s of %d objects>" % (self.owner.operation_type, len(self.children))


class NodeSDF(SDFExpr):
    """`obj.propagate` / `obj.create` used as an SDF (reference idiom GenericGeometry(obj.propagate),
    examples/scalar/3D/chip_3D.py:49-51): a live reference to another geometry object."""

    def __init__(self, obj):
        self.obj = obj

    def __repr__(self):
        return "<node %r>" % (self.obj,)


class UnsupportedSDF(SDFExpr):
    """An opaque Python callable `fn(co, *params)`. It cannot be fused into a GPU program: the staged evaluation
    (_eval._run_staged) brings the coordinates it is handed back to the host, calls it there and feeds its result
    to the rest of the tree as an auxiliary field. `why` is the message raised where that is not possible."""
    name = "python_callable"
    inner = None

    def __init__(self, fn, why):
        self.fn = fn
        self.why = why
        self.args = {}
