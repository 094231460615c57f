from gaussreg_amd.kpconv import maxpool, nearest_upsample  # noqa: F401  (modules/kpconv/functional.py:6-67)
