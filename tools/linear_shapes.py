"""Dev tool: which nn.Linear shapes the network path of configs[4] runs, how long each takes (synchronised per call) and
what that is in TFLOP/s and GB/s -- torch.nn.functional.linear is wrapped for one 32-pair batch."""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch.nn.functional as F
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
pairs = [pair_pipeline.synthetic_room_pair(i, 200000, dev) for i in range(32)]
reg = pair_pipeline.PairRegistrar(dev, features="model")
reg.register_pairs(pairs)
torch.cuda.synchronize()
real = F.linear
acc = collections.OrderedDict()
def timed(x, w, b=None):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = real(x, w, b)
    torch.cuda.synchronize()
    k = (tuple(x.shape), tuple(w.shape), x.is_contiguous())
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += time.perf_counter() - t0
    return y
F.linear = timed
torch.nn.functional.linear = timed
reg.register_pairs(pairs)
torch.cuda.synchronize()
F.linear = real
tot = sum(v[1] for v in acc.values())
print("linear calls %d, %.2f ms per batch of 32 pairs" % (sum(v[0] for v in acc.values()), tot * 1e3))
for k, (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:25]:
    xs, ws, contig = k
    rows = 1
    for d in xs[:-1]:
        rows *= d
    fl = 2.0 * rows * ws[0] * ws[1]
    by = 4.0 * (rows * (ws[0] + ws[1]) + ws[0] * ws[1])
    print("%-28s x W%-14s contig=%d  %3d calls  %8.1f us each  %6.1f TF  %7.0f GB/s" % (xs, ws, contig, n, t / n * 1e6, fl * n / t / 1e12, by * n / t / 1e9))
