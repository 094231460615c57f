"""gaussreg_amd -- MI355X-native implementation of GaussReg's two data-parallel hot paths.

Host side: Python on PyTorch-ROCm (device memory, streams, torch.distributed only).
Device side: hand-written HIP kernels for gfx950 behind the C ABI in include/gaussreg_hip.h
(gaussreg_amd/lib/libgaussreg_hip.so, built by `python -m gaussreg_amd.build`).

There is no CPU fallback: every op raises if the HIP library or a GPU is missing.
"""
__version__ = "0.1.0"
