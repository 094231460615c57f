"""Batched data pyramid only (dev tool, profiled with tools/prof.sh): B pairs per stack-mode call."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
from gen_golden_ext import room_pair
from gaussreg_amd.data import precompute_data_stack_mode
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
ORDER = sys.argv[2] if len(sys.argv) > 2 else "cell"
limits = [89, 30, 43, 49, 49]
clouds = []
for b in range(B):
    r_, s_ = room_pair(30000, b)
    clouds += [r_, s_]
bp = torch.from_numpy(np.concatenate(clouds)).cuda()
bl = torch.tensor([30000] * (2 * B))
out = precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, limits, order=ORDER)
ts = []
for _ in range(7):
    torch.cuda.synchronize(); t = time.perf_counter()
    out = precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, limits, order=ORDER)
    torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
dt = sorted(ts)[len(ts) // 2]  # median: the caching allocator hiccups now and then at these sizes
print(f"HIP pyramid ({ORDER} order), {B} pairs per call: {dt*1e3:.1f} ms -> {B/dt:.0f} pairs/s; level sizes {[p.shape[0] for p in out['points']]}")
