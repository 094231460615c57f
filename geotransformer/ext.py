"""``geotransformer.ext`` -- the reference's native module name (setup.py:10), re-exported from the
HIP-backed mirror so `importlib.import_module('geotransformer.ext')` (modules/ops/radius_search.py:4,
grid_subsample.py:4 in the reference) resolves to the MI355X kernels."""
from gaussreg_amd.ext import grid_subsampling, radius_neighbors, radius_neighbors_limited  # noqa: F401
