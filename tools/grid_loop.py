"""Dev tool: grid_subsampling of 16 x 200 k points in a loop (for rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import ext
B = int(os.environ.get("GL_CLOUDS", "16"))
g = torch.Generator().manual_seed(0)
pts = (torch.rand(200000 * B, 3, generator=g) * 10 ** (1 / 3)).float().cuda()
lens = torch.tensor([200000] * B)
for _ in range(6):
    ext.grid_subsampling(pts, lens, 0.05, order="cell")
torch.cuda.synchronize()
