"""The 11 KPConv layers of KPConvFPN (backbone.py:12-45) at their real widths (a ResidualBlock convolves out_dim/4
channels) on the synthetic demo pyramid: HIP vs the reference's formulation in stock torch ops on the same GPU."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
from gen_golden_ext import room_pair
from gaussreg_amd.data import precompute_data_stack_mode
from gaussreg_amd.kpconv import KPConv


def timeit(fn, n=10, warm=2):
    """median of per-call times (the torch formulation's multi-GB temporaries make the allocator hiccup now and then)"""
    for _ in range(warm): fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) * 1e3)
    return sorted(ts)[len(ts) // 2]


ref, src = room_pair(30000, 0)
pts = torch.from_numpy(np.concatenate([ref, src])).cuda()
d = precompute_data_stack_mode(pts, torch.tensor([30000, 30000]), 5, 0.025, 0.0625, [89, 30, 43, 49, 49])
P, NB, SUB = d["points"], d["neighbors"], d["subsampling"]
g = torch.Generator(device="cuda").manual_seed(0)
kp = torch.randn(15, 3) * 0.03
# (name, query level, support level, neighbours, channels)
layers = [("1_1", 0, 0, NB[0], 4, 64), ("1_2", 0, 0, NB[0], 32, 32), ("2_1s", 1, 0, SUB[0], 32, 32), ("2_2", 1, 1, NB[1], 64, 64),
          ("2_3", 1, 1, NB[1], 64, 64), ("3_1s", 2, 1, SUB[1], 64, 64), ("3_2", 2, 2, NB[2], 128, 128), ("3_3", 2, 2, NB[2], 128, 128),
          ("4_1s", 3, 2, SUB[2], 128, 128), ("4_2", 3, 3, NB[3], 256, 256), ("4_3", 3, 3, NB[3], 256, 256)]
tot_h = tot_t = 0.0
for name, ql, sl, nb, cin, cout in layers:
    q, s = P[ql], P[sl]
    sigma = 0.05 * (2 ** sl)
    conv = KPConv(cin, cout, 15, 0.0625 * 2 ** sl, sigma, kernel_points=kp * 2 ** sl).cuda()
    f = torch.relu(torch.randn(s.shape[0], cin, device="cuda", generator=g))
    def ref_torch():
        s2 = torch.cat([s, torch.zeros_like(s[:1]) + 1e6], 0); nbp = s2[nb] - q[:, None]
        w = (1 - ((nbp[:, :, None] - conv.kernel_points) ** 2).sum(3).sqrt() / sigma).clamp(min=0).transpose(1, 2)
        f2 = torch.cat([f, torch.zeros_like(f[:1])], 0); nf = f2[nb]
        o = (torch.matmul(w, nf).permute(1, 0, 2) @ conv.weights).sum(0)
        num = (nf.sum(-1) > 0).sum(-1).clamp(min=1); return o / num[:, None]
    with torch.no_grad():
        th, tt = timeit(lambda: conv(f, q, s, nb)), timeit(ref_torch, 5, 1)
    tot_h += th; tot_t += tt
    print(f"encoder{name}: M={q.shape[0]:6d} N={s.shape[0]:6d} H={nb.shape[1]:3d} {cin:3d}->{cout:3d}: HIP {th:.3f} ms  torch {tt:.3f} ms")
print(f"all 11 KPConv layers: HIP {tot_h:.2f} ms  torch {tot_t:.2f} ms per pair")
