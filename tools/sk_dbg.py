import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd.sinkhorn import LearnableLogOptimalTransport
from gaussreg_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
ot = LearnableLogOptimalTransport(100).to(dev)
P, K = 256, 128
sc = torch.randn(P, K, K, device=dev, generator=g)
rm = torch.rand(P, K, device=dev, generator=g) > 0.3
sm = torch.rand(P, K, device=dev, generator=g) > 0.3
o = ot(sc, rm, sm)
L = _lib.lib()
L.gr_timing_reset(); L.gr_timing_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): o = ot(sc, rm, sm)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20 * 1e3
t, c = ctypes.c_double(0), ctypes.c_int64(0)
L.gr_timing_read(b"sinkhorn", ctypes.byref(t), ctypes.byref(c))
inner = o[:, :-1, :-1]
m = rm[:, :, None] & sm[:, None, :]
print(f"sinkhorn 256x128x128 x100: wall {wall:.3f} ms, kernel {t.value / max(c.value, 1):.3f} ms; checksum {float(inner[m].double().sum()):.6f}")
