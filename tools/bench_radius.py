"""Quick timing of HIP radius_neighbors / grid_subsampling (dev tool; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import ext

def timeit(fn, n=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n

for B in (1, 4, 16):
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(200000 * B, 3, generator=g) * 10 ** (1 / 3)).float().cuda()
    lens = torch.tensor([200000] * B)
    t = timeit(lambda: ext.radius_neighbors(pts, pts, lens, lens, 0.0625))
    nb = ext.radius_neighbors(pts, pts, lens, lens, 0.0625)
    by = 12 * pts.shape[0] * 2 + 8 * nb.numel()
    print(f"radius B={B}: {t*1e3:.3f} ms  {pts.shape[0]/t/1e6:.1f} Mpts/s  width={nb.shape[1]}  alg {by/t/1e9:.1f} GB/s")
    t = timeit(lambda: ext.grid_subsampling(pts, lens, 0.05))
    t2 = timeit(lambda: ext.grid_subsampling(pts, lens, 0.05, order='cell'))
    print(f"grid   B={B}: ref-order {t*1e3:.3f} ms ({pts.shape[0]/t/1e6:.1f} Mpts/s)  cell-order {t2*1e3:.3f} ms ({pts.shape[0]/t2/1e6:.1f} Mpts/s)")
