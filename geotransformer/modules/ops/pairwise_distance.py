from gaussreg_amd.ops import pairwise_distance  # noqa: F401  (modules/ops/pairwise_distance.py:4-31)
