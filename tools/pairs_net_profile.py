"""Dev tool: the model-mode pair path on 16 pairs (for rocprofv3 --kernel-trace --stats)."""
import sys, os, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
pairs = [pair_pipeline.synthetic_room_pair(i, 60000, dev) for i in range(16)]
reg = pair_pipeline.PairRegistrar(dev, features="model", num_samples=30000)
reg.register_pairs(pairs[:2])
torch.cuda.synchronize()
for _ in range(2):
    reg.register_pairs(pairs)
torch.cuda.synchronize()
reg.close()
