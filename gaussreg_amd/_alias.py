"""Helpers for the drop-in alias packages at the repo root (`geotransformer/`, INTEGRATION.md route A).

The alias package shadows GaussReg's own `geotransformer` package when this repo comes first on `sys.path`.  GaussReg's scripts
also import parts of that package this repo does not replace (`geotransformer.engine`, `.datasets`, `.utils.torch`,
`.modules.registration`, `.modules.loss`, ...; model.py:1-16, demo.py:1-18).  `chain()` makes every alias package a
*front* of the same-named package further down `sys.path`: its `__path__` is extended with that package's directory, so a
sub-module this repo does not provide resolves to GaussReg's own file, and one it does provide resolves here.  Nothing of
GaussReg is copied or shipped; without it on the path the alias packages simply carry the hot-path surface only.
"""
import importlib
import importlib.util
import os
import pkgutil
import sys


def chain(namespace):
    """Call as `chain(globals())` at the top of an alias package's __init__.py."""
    seen, out = set(), []
    for d in pkgutil.extend_path(list(namespace["__path__"]), namespace["__name__"]):
        key = os.path.realpath(d)
        if key not in seen:           # '' and the absolute repo root both on sys.path name the same directory
            seen.add(key)
            out.append(d)
    namespace["__path__"] = out
    return out


def upstream_names(package, submodule, names, namespace):
    """`from <package>.<submodule> import <names>` out of the chained (GaussReg's own) package, if it is on the path.
    Names that cannot be imported are left undefined (hot-path-only installation)."""
    try:
        mod = importlib.import_module(package + "." + submodule)
    except ImportError:
        return False
    for n in names:
        if hasattr(mod, n):
            namespace[n] = getattr(mod, n)
    return True


def shadowed_module(alias_name, alias_file):
    """The module a sub-module alias of this repo shadows: the same file name in a LATER entry of the parent package's
    `__path__`, loaded under `<alias_name>__upstream` (None when there is none).  Lets an alias sub-module re-export the
    names it does not replace (e.g. `knn_partition` next to the HIP `point_to_node_partition`)."""
    parent, _, leaf = alias_name.rpartition(".")
    here = os.path.dirname(os.path.abspath(alias_file))
    for d in list(getattr(sys.modules.get(parent), "__path__", [])):
        if os.path.abspath(d) == here:
            continue
        cand = os.path.join(d, leaf + ".py")
        if os.path.isfile(cand):
            name = alias_name + "__upstream"
            if name in sys.modules:
                return sys.modules[name]
            spec = importlib.util.spec_from_file_location(name, cand)
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            try:
                spec.loader.exec_module(mod)
            except Exception:
                del sys.modules[name]
                raise
            return mod
    return None
