"""FPS timing: B clouds of 200k points -> 30k samples per call; checks the result against the 2-cloud call."""
import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
from gaussreg_amd.registration import farthest_point_sampling
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
clouds = []
for b in range(B // 2):
    r, s, _ = pair_pipeline.synthetic_room_pair(b, 200000, dev)
    clouds += [r, s]
big = torch.cat(clouds).contiguous()
lens = [200000] * B
ks = [30000] * B
idx = farthest_point_sampling(big, lens, ks)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    idx = farthest_point_sampling(big, lens, ks)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 3 * 1e3
chk = sum(int(i.sum()) for i in idx)
print(f"B={B}: {ms:.2f} ms per call, {ms / B:.3f} ms per cloud, checksum {chk}", flush=True)
