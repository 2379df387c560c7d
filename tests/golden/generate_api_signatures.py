"""Build container only: record the public names and call signatures of the reference's `spomso.cores` modules that
are in scope as JSON, so that the API-mirror test can run where the reference is absent.

    PYTHONPATH=/root/reference/Code/spomso python tests/golden/generate_api_signatures.py

Data only: module -> {name: {"kind": "function"|"class", "signature": [[parameter name, kind, default repr], ...], "methods": {name: signature},
"properties": [...]}}. No source text of the reference is stored."""
import importlib
import inspect
import json
import os
import sys

sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference/Code/spomso")

MODULES = ["combine", "geom", "geom_2d", "geom_3d", "geom_vector", "helper_functions", "modifications", "post_processing",
           "sdf_2D", "sdf_3D", "transformations", "triangulation_functions", "vector_functions",
           "vector_modification_functions"]


def sig(obj):
    """[[name, kind, default repr or None], ...] — annotations are not part of the contract"""
    try:
        params = inspect.signature(obj).parameters.values()
    except (TypeError, ValueError):
        return None
    return [[p.name, p.kind.name, None if p.default is inspect.Parameter.empty else repr(p.default)] for p in params]


def describe(mod):
    out = {}
    for name, obj in vars(mod).items():
        if name.startswith("_") or getattr(obj, "__module__", None) != mod.__name__:
            continue
        if inspect.isfunction(obj):
            out[name] = {"kind": "function", "signature": sig(obj)}
        elif inspect.isclass(obj):
            methods, props = {}, []
            for m, member in vars(obj).items():
                if m.startswith("_") and m != "__init__":
                    continue
                if isinstance(member, property):
                    props.append(m)
                elif inspect.isfunction(member):
                    methods[m] = sig(member)
            out[name] = {"kind": "class", "signature": sig(obj), "methods": methods, "properties": sorted(props),
                         "bases": [b.__name__ for b in obj.__mro__[1:-1]]}
    return out


def main():
    rec = {"package_all": None, "modules": {}}
    pkg = importlib.import_module("spomso.cores")
    rec["package_names"] = sorted(n for n in vars(pkg) if not n.startswith("_"))
    for m in MODULES:
        rec["modules"][m] = describe(importlib.import_module("spomso.cores." + m))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_api.json")
    with open(path, "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
    print(path, sum(len(v) for v in rec["modules"].values()), "names")


if __name__ == "__main__":
    main()
