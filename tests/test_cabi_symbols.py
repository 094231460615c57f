"""CPU: the C-ABI library loads and exports every symbol include/gaussreg_hip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gaussreg_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gr_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "gr_radius_count" in syms and "gr_grid_subsample" in syms


def test_library_exports_every_declared_symbol():
    from gaussreg_amd import _lib, build
    build.build()  # hipcc cross-compiles for gfx950 without a GPU
    L = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(L, s), f"libgaussreg_hip.so does not export {s}"
    # and the ctypes table covers the header exactly
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    assert _lib.lib().gr_version() >= 1000


def test_workspace_size_queries_are_host_only():
    from gaussreg_amd import _lib
    L = _lib.lib()
    assert L.gr_radius_workspace_bytes(20000, 20000, 1) > 0
    assert L.gr_grid_subsample_workspace_bytes(20000, 1) > 0


def test_ops_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from gaussreg_amd import ext
    p = torch.rand(10, 3)
    l = torch.tensor([10])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.radius_neighbors(p, p, l, l, 0.1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ext.grid_subsampling(p, l, 0.1)
