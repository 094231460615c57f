"""Dev tool: the data pyramid of 64 pairs (128 clouds of 30 000 points) in a loop, for rocprofv3 --kernel-trace +
tools/trace_gaps.py <dir> bbox_partial (one radius binning = one search) or kernel stats."""
import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
from gaussreg_amd.data import precompute_data_stack_mode
dev = torch.device("cuda", 0)
Bp = 64
clouds = [pair_pipeline.synthetic_room_pair(b, 30000, dev)[0] for b in range(Bp)] + [pair_pipeline.synthetic_room_pair(b, 30000, dev)[1] for b in range(Bp)]
bp = torch.cat(clouds).contiguous()
bl = torch.tensor([30000] * (2 * Bp))
order = os.environ.get("PY_ORDER", "reference")
for _ in range(2):
    precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, [89, 30, 43, 49, 49], order=order)
torch.cuda.synchronize()
t0 = time.perf_counter()
N = int(os.environ.get("PY_ITERS", "5"))
for _ in range(N):
    d = precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, [89, 30, 43, 49, 49], order=order)
torch.cuda.synchronize()
print("pyramid of %d pairs, %s order: %.3f ms per call; level sizes %s" % (Bp, order, (time.perf_counter() - t0) / N * 1e3, [int(p.shape[0]) for p in d["points"]]))
