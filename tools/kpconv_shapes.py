"""Dev tool: the KPConv calls of one 32-pair batch of the network path -- (points, neighbours, channels), time per call
(synchronised), TFLOP/s of the whole op (gather-multiply + the K Cin x Cout product)."""
import os, sys, time, collections, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline, kpconv
dev = torch.device("cuda", 0)
pairs = [pair_pipeline.synthetic_room_pair(i, 200000, dev) for i in range(32)]
reg = pair_pipeline.PairRegistrar(dev, features="model")
reg.register_pairs(pairs)
torch.cuda.synchronize()
real = kpconv.KPConv.forward
acc = collections.OrderedDict()
def timed(self, s_feats, q_points, s_points, neighbor_indices):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = real(self, s_feats, q_points, s_points, neighbor_indices)
    torch.cuda.synchronize()
    k = (s_feats.shape[0], q_points.shape[0], neighbor_indices.shape[1], s_feats.shape[1], y.shape[1])
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += time.perf_counter() - t0
    return y
kpconv.KPConv.forward = timed
reg.register_pairs(pairs)
torch.cuda.synchronize()
kpconv.KPConv.forward = real
print("KPConv calls %d, %.2f ms per batch of 32 pairs" % (sum(v[0] for v in acc.values()), sum(v[1] for v in acc.values()) * 1e3))
for (n, m, h, cin, cout), (c, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    g = 2.0 * m * h * 15 * cin
    p = 2.0 * m * 15 * cin * cout
    print("N %8d M %8d H %3d  %4d -> %4d  %2d calls %9.1f us each   gather %6.1f GF  product %7.1f GF  %6.1f TF" %
          (n, m, h, cin, cout, c, t / c * 1e6, g / 1e9, p / 1e9, (g + p) * c / t / 1e12))
