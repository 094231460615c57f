"""Dev tool: configs[4] descriptor-mode throughput against the number of host threads / streams of the per-pair stage."""
import sys, os, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
pairs = [pair_pipeline.synthetic_room_pair(i, 200000, dev) for i in range(64)]
for S in (1, 2, 4, 6, 8, 12):
    reg = pair_pipeline.PairRegistrar(dev, pair_streams=S)
    reg.register_pairs(pairs[:8])
    torch.cuda.synchronize(); t = time.perf_counter()
    reg.register_pairs(pairs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"streams {S}: {len(pairs)/dt:.1f} pairs/s, {dt/len(pairs)*1e3:.2f} ms/pair", flush=True)
    reg.close()
