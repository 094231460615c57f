"""Dev tool: grid_subsampling of GL_CLOUDS x 200 k points in a loop (for rocprofv3 --kernel-trace; GL_ORDER = reference | cell).
tools/trace_gaps.py <dir> keys_kernel prints the timeline of one call."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import ext
B = int(os.environ.get("GL_CLOUDS", "16"))
order = os.environ.get("GL_ORDER", "cell")
g = torch.Generator().manual_seed(0)
pts = (torch.rand(200000 * B, 3, generator=g) * 10 ** (1 / 3)).float().cuda()
lens = torch.tensor([200000] * B)
for _ in range(6):
    ext.grid_subsampling(pts, lens, 0.05, order=order)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = int(os.environ.get("GL_ITERS", "200"))
for _ in range(N):
    ext.grid_subsampling(pts, lens, 0.05, order=order)
torch.cuda.synchronize()
print("%d x 200 k, %s order: %.4f ms per call" % (B, order, (time.perf_counter() - t0) / N * 1e3))
