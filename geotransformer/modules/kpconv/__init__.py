from gaussreg_amd.kpconv import KPConv, maxpool, nearest_upsample  # noqa: F401
