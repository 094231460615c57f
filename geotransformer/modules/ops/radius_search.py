from gaussreg_amd.ops import radius_search  # noqa: F401  (modules/ops/radius_search.py:7-27)
