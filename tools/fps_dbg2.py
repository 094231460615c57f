import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd.registration import farthest_point_sampling
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev).manual_seed(1)
B, n, k = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
pts = torch.rand(B * n, 3, device=dev, generator=g)
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    idx = farthest_point_sampling(pts, [n] * B, [k] * B)
    torch.cuda.synchronize()
    print(B, n, k, f"{(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
