from gaussreg_amd.sinkhorn import LearnableLogOptimalTransport  # noqa: F401
