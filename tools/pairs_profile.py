"""Dev tool: the descriptor-mode pair path, 64 pairs per call (for rocprofv3 --kernel-trace --stats): prints the wall time
per pair so that the kernel total of the trace can be set against it."""
import sys, os, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
# (the synthetic clouds are made on the CPU: their stock-PyTorch generator kernels would otherwise be 4 % of the trace)
cpu = torch.device("cpu")
pairs = [tuple(t.to(dev) for t in pair_pipeline.synthetic_room_pair(i, 200000, cpu)) for i in range(64)]
reg = pair_pipeline.PairRegistrar(dev)
reg.register_pairs(pairs[:4])
reg.register_pairs(pairs)
torch.cuda.synchronize()
t0 = time.perf_counter()
PASSES = 8   # the synthetic clouds above are made by stock PyTorch kernels: enough passes that they are < 2 % of the trace
for _ in range(PASSES):
    reg.register_pairs(pairs)
torch.cuda.synchronize()
print("wall ms per pair", (time.perf_counter() - t0) / (64 * PASSES) * 1e3)
reg.close()
