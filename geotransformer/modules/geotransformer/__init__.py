from gaussreg_amd.matching import LocalGlobalRegistration, PointMatching, SuperPointMatching  # noqa: F401
