"""Per-kernel timing of radius_neighbors (dev tool): python tools/bench_radius_kernels.py [B]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import ext, _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = _lib.lib()
g = torch.Generator().manual_seed(0)
pts = (torch.rand(200000 * B, 3, generator=g) * 10 ** (1 / 3)).float().cuda()
lens = torch.tensor([200000] * B)
for _ in range(3):
    nbr = ext.radius_neighbors(pts, pts, lens, lens, 0.0625)
L.gr_timing_enable(1)
L.gr_timing_reset()
torch.cuda.synchronize()
t = time.perf_counter()
N = 20
for _ in range(N):
    nbr = ext.radius_neighbors(pts, pts, lens, lens, 0.0625)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / N
print(f"B={B} total {dt*1e3:.3f} ms  {pts.shape[0]/dt/1e6:.1f} Mpts/s width {nbr.shape[1]}")
for name in ("radius_traverse", "radius_expand", "radius_count", "radius_fill"):
    tot, n = ctypes.c_double(0), ctypes.c_int64(0)
    L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(n))
    if n.value:
        print(f"  {name}: {tot.value/n.value:.4f} ms x{n.value}")
