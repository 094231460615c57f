"""Drop-in alias package: the names GaussReg's scripts import, backed by gaussreg_amd (MI355X).

Only the hot-path surface exists here: ``geotransformer.ext``,
``geotransformer.modules.ops.{grid_subsample, radius_search, ...}``,
``geotransformer.modules.geotransformer.{SuperPointMatching, PointMatching}`` and
``geotransformer.utils.data`` (pyramid builder).  The rest of the reference package (engine,
datasets, model glue) runs unchanged on stock PyTorch-ROCm and is out of scope.
"""
