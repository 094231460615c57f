from gaussreg_amd.matching import PointMatching, SuperPointMatching  # noqa: F401
