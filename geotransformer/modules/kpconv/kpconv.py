from gaussreg_amd.kpconv import KPConv  # noqa: F401  (modules/kpconv/kpconv.py:79-122)
