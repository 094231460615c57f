import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd.registration import registration_with_ransac_from_correspondences
from gaussreg_amd import _lib
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(0)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
src = torch.rand(C, 3, device=dev, generator=g) * 3
a = 0.5
R = torch.tensor([[1, 0, 0], [0, torch.cos(torch.tensor(a)), -torch.sin(torch.tensor(a))], [0, torch.sin(torch.tensor(a)), torch.cos(torch.tensor(a))]], device=dev)
ref = src @ R.T + torch.tensor([0.2, -0.1, 0.4], device=dev) + 0.005 * torch.randn(C, 3, device=dev, generator=g)
out = torch.rand(C, device=dev, generator=g) < 0.4
ref[out] = torch.rand(int(out.sum()), 3, device=dev, generator=g) * 3
T = registration_with_ransac_from_correspondences(src, ref, None, 0.05, 3, 10000, seed=1)
L = _lib.lib(); L.gr_timing_reset(); L.gr_timing_enable(1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): T = registration_with_ransac_from_correspondences(src, ref, None, 0.05, 3, 10000, seed=1)
torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 20 * 1e3
t, c = ctypes.c_double(0), ctypes.c_int64(0)
L.gr_timing_read(b"ransac", ctypes.byref(t), ctypes.byref(c))
print(f"RANSAC C={C}: wall {wall:.3f} ms, kernels {t.value / max(c.value, 1):.3f} ms, |R - R_gt| {float((T[:3, :3] - R).abs().max()):.2e}")
