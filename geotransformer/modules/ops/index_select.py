from gaussreg_amd.ops import index_select  # noqa: F401  (modules/ops/index_select.py:4-31)
