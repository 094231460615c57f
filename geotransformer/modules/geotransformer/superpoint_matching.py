from gaussreg_amd.matching import SuperPointMatching  # noqa: F401
