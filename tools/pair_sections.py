"""Wall time per stage of the configs[4] pair path (16 pairs per batch), with a device synchronise around every stage."""
import json, os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
pairs = [pair_pipeline.synthetic_room_pair(b, 200000, dev) for b in range(B)]
reg = pair_pipeline.PairRegistrar(dev, pair_streams=1)
reg_out = reg.register_pairs(pairs)
torch.cuda.synchronize(); t0 = time.perf_counter()
reg.register_pairs(pairs)
torch.cuda.synchronize(); plain = (time.perf_counter() - t0) * 1e3
reg.profile = True
reg.register_pairs(pairs)
tot = sum(reg.section_ms.values())
print(json.dumps({"pairs": B, "unprofiled_ms_per_pair": round(plain / B, 3), "profiled_ms_per_pair": round(tot / B, 3),
                  "sections_ms_per_pair": {k: round(v / B, 3) for k, v in reg.section_ms.items()}}, indent=1))
for S in (1, 2, 4, 8):
    r2 = pair_pipeline.PairRegistrar(dev, pair_streams=S)
    a = r2.register_pairs(pairs)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    a = r2.register_pairs(pairs)
    torch.cuda.synchronize()
    print(f"pair_streams={S}: {(time.perf_counter() - t0) * 1e3 / B:.3f} ms per pair; equal to sequential: {bool(torch.equal(a, reg_out))}")
