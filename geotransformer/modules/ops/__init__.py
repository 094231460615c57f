from gaussreg_amd.ops import grid_subsample, radius_search  # noqa: F401
