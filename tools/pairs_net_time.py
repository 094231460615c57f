"""Dev tool: pairs/s of the pair path WITH the network (PairRegistrar(features="model")), reference vs cell row order inside the
pipeline; the bench's `pairs.with_network` workload (64 pairs, batches of 32)."""
import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
pairs = [pair_pipeline.synthetic_room_pair(i, 200000, dev) for i in range(64)]
for order in ("reference", "cell"):
    reg = pair_pipeline.PairRegistrar(dev, features="model", order=order, profile="--sections" in sys.argv)
    reg.register_pairs(pairs[:32])
    torch.cuda.synchronize()
    reg.section_ms.clear()
    t0 = time.perf_counter()
    reg.register_many(pairs, 32)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(order, "order: %.2f pairs/s  %.3f ms per pair" % (64 / dt, dt / 64 * 1e3), {k: round(v / 64, 3) for k, v in reg.section_ms.items()})
    reg.close()
    del reg
    torch.cuda.empty_cache()
