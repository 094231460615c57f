"""Dev tool: the SuperPointMatching fast path against a float64 torch evaluation of the reference expression, with the header
of the call printed (candidates, thresholds, overflow)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import _lib
from gaussreg_amd.matching import SuperPointMatching
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
for (nr, ns, k) in ((767, 701, 256), (767, 767, 256), (100, 900, 64), (40, 3, 256), (1000, 1000, 256)):
    fr = torch.nn.functional.normalize(torch.randn(nr, 256, device=dev, generator=g), dim=1)
    fs = torch.nn.functional.normalize(torch.randn(ns, 256, device=dev, generator=g), dim=1)
    ri, si, sc = SuperPointMatching(k)(fr, fs)
    torch.cuda.synchronize()
    S = torch.exp(-(2.0 - 2.0 * fr.double() @ fs.double().T).clamp(min=0))
    score = (S / S.sum(1, keepdim=True)) * (S / S.sum(0, keepdim=True))
    w = torch.topk(score.flatten(), min(k, nr * ns))
    want = set(w.indices.tolist())
    got = set((ri * ns + si).tolist())
    hdr = list(_lib._ws_cache.values())[0][:32].view(torch.int32).tolist()
    print("   header nr ns prefix k_rem k n_cand tau_max overflow =", hdr, flush=True)
    print(nr, ns, k, "returned", len(ri), "set difference", len(want ^ got), "score err", float((sc.double() - w.values).abs().max() / w.values.max()), flush=True)
