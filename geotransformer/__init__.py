"""Drop-in alias package: the names GaussReg's scripts import, backed by gaussreg_amd (MI355X).

The hot-path surface lives here: ``geotransformer.ext``, ``geotransformer.modules.ops``,
``geotransformer.modules.{geotransformer, kpconv, sinkhorn, transformer}`` and ``geotransformer.utils.data``.
Everything else of GaussReg's package (engine, datasets, utils.torch / open3d / registration, modules.registration /
loss / layers, transforms) is NOT here: put GaussReg's checkout BEHIND this repo on ``sys.path`` and those sub-modules
resolve to its own files (``gaussreg_amd/_alias.py``; INTEGRATION.md route A).
"""
from gaussreg_amd._alias import chain as _chain

_chain(globals())   # sub-modules this repo does not override resolve to GaussReg's own package, if on sys.path
