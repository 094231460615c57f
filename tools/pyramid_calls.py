"""Development tool: what each of the 17 calls of the 64-pair demo pyramid costs (CUDA events around every op)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import ops, pair_pipeline
from gaussreg_amd import data as D

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reg = pair_pipeline.PairRegistrar(dev)
pairs = [pair_pipeline.synthetic_room_pair(i, 60000, dev) for i in range(B)]
sampled = reg._sample(pairs, 24) if hasattr(reg, "_sample") else None
points = torch.cat(sampled, 0).contiguous()
lengths = torch.tensor([c.shape[0] for c in sampled], dtype=torch.int64)
log = []
orig_rs, orig_gs = D.radius_search, D.grid_subsample


def timed(name, fn):
    def w(*a, **k):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = fn(*a, **k)
        e.record()
        e.synchronize()
        shape = tuple(out.shape) if torch.is_tensor(out) else tuple(out[0].shape)
        log.append((name, s.elapsed_time(e), shape, k.get("neighbor_limit", a[5] if len(a) > 5 else None) if name == "radius_search" else None))
        return out
    return w


D.radius_search = timed("radius_search", orig_rs)
D.grid_subsample = timed("grid_subsample", orig_gs)
for rep in range(2):
    log.clear()
    D.precompute_data_stack_mode(points, lengths, pair_pipeline.NUM_STAGES, pair_pipeline.INIT_VOXEL, pair_pipeline.INIT_RADIUS,
                                 pair_pipeline.NEIGHBOR_LIMITS)
tot = sum(t for _, t, _, _ in log)
for name, t, shape, lim in log:
    print(f"{name:15s} {t:8.3f} ms  out {shape}  limit {lim}")
print(f"total {tot:.2f} ms for {B} pairs")
