import sys, time, torch
sys.path.insert(0, "/root/repo")
from gaussreg_amd import pair_pipeline
dev = torch.device("cuda", 0)
pairs = [pair_pipeline.synthetic_room_pair(i, 200000, dev) for i in range(32)]
for feat in ("descriptor", "model"):
    reg = pair_pipeline.PairRegistrar(dev, features=feat)
    reg.register_pairs(pairs[:2])
    torch.cuda.synchronize(); t = time.perf_counter()
    reg.register_pairs(pairs)
    torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(feat, f"{len(pairs)/dt:.1f} pairs/s, {dt/len(pairs)*1e3:.2f} ms/pair")
    reg.close()
    regp = pair_pipeline.PairRegistrar(dev, features=feat, profile=True)
    regp.register_pairs(pairs[:8])
    print({k: round(v / 8, 3) for k, v in regp.section_ms.items()})
    regp.close()
