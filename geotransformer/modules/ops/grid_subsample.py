from gaussreg_amd.ops import grid_subsample  # noqa: F401  (modules/ops/grid_subsample.py:7-22)
