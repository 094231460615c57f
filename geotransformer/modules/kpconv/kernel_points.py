from gaussreg_amd.kpconv import load_kernels  # noqa: F401  (the 15-point disposition is built in: no open3d, no PLY)
