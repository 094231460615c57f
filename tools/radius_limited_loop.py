"""Dev tool: the `limited` radius workload in a loop (for rocprofv3 --kernel-trace + tools/trace_gaps.py)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import _lib, ext, synthetic
B = int(os.environ.get("BRF_CLOUDS", "8"))
_lib.lib().gr_radius_search_mode(int(os.environ.get("BRF_MODE", "0")))
pts, lens = synthetic.cloud_200k(B, seed=0)
d = pts.cuda()
for _ in range(12):
    out = ext.radius_neighbors_limited(d, d, lens, lens, 0.0625, 40)
torch.cuda.synchronize()
print(out.shape)
