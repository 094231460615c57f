"""C3 stand-in: the demo pyramid (4 grid_subsample + 13 radius_search, utils/data.py:13-77) on the
synthetic room pair of SURVEY.md App. D (2 x 30 000 points), HIP vs the reference core on one CPU core."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
from gen_golden_ext import room_pair
from gaussreg_amd.data import precompute_data_stack_mode
ref, src = room_pair(30000, 0)
pts = torch.from_numpy(np.concatenate([ref, src])).cuda()
lens = torch.tensor([30000, 30000])
limits = [89, 30, 43, 49, 49]
for order in ("reference", "cell"):
    for _ in range(2): out = precompute_data_stack_mode(pts, lens, 5, 0.025, 0.0625, limits, order=order)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(10): out = precompute_data_stack_mode(pts, lens, 5, 0.025, 0.0625, limits, order=order)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 10
    print(f"HIP pyramid (order={order}): {dt*1e3:.2f} ms/pair  -> {1/dt:.1f} pairs/s; level sizes {[p.shape[0] for p in out['points']]}")
from oracle import capi
if capi.have_ref():
    p, l = pts.cpu().numpy(), lens.numpy()
    t = time.perf_counter()
    P, L, v = [p], [l], 0.025
    for i in range(5):
        if i > 0:
            sp, sl = capi.ref_grid_subsampling(P[-1], L[-1], v); P.append(sp); L.append(sl)
        v *= 2
    r = 0.0625
    for i in range(5):
        capi.ref_radius_neighbors(P[i], P[i], L[i], L[i], r)
        if i < 4:
            capi.ref_radius_neighbors(P[i+1], P[i], L[i+1], L[i], r); capi.ref_radius_neighbors(P[i], P[i+1], L[i], L[i+1], 2*r)
        r *= 2
    dt = time.perf_counter() - t
    print(f"reference core, 1 CPU thread: {dt*1e3:.0f} ms/pair -> {1/dt:.2f} pairs/s")

# ---- batched: B pairs stacked in one stack-mode call (lengths has 2B entries), as C5 would shard per rank
for B in (8, 64):
    clouds = []
    for b in range(B):
        r_, s_ = room_pair(30000, b)
        clouds += [r_, s_]
    bp = torch.from_numpy(np.concatenate(clouds)).cuda()
    bl = torch.tensor([30000] * (2 * B))
    for order in ("cell",):
        out = precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, limits, order=order)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(3): out = precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, limits, order=order)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 3
        print(f"HIP pyramid, {B} pairs per call (order={order}): {dt*1e3:.1f} ms -> {B/dt:.0f} pairs/s")
