"""Kernel times of radius_neighbors on the bench workload (8 x 200k-pt clouds), per path."""
import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import ext, synthetic, _lib
pts, lens = synthetic.cloud_200k(8, seed=0)
dp = pts.cuda()
L = _lib.lib()
for _ in range(3): ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
L.gr_timing_reset(); L.gr_timing_enable(1)
for _ in range(40): ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
torch.cuda.synchronize()
out = {}
for k in (b"radius_count", b"radius_fill"):
    t, c = ctypes.c_double(0), ctypes.c_int64(0)
    L.gr_timing_read(k, ctypes.byref(t), ctypes.byref(c))
    out[k.decode()] = round(t.value / max(c.value, 1), 4)
L.gr_timing_enable(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(100): ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
torch.cuda.synchronize()
out["e2e_ms"] = round((time.perf_counter() - t0) / 100 * 1e3, 4)
print(out, flush=True)
