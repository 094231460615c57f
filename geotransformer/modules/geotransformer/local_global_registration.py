from gaussreg_amd.matching import LocalGlobalRegistration  # noqa: F401
