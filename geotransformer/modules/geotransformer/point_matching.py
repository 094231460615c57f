from gaussreg_amd.matching import PointMatching  # noqa: F401
