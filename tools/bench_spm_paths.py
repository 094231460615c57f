import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from bench_extras import _ms, _mfma
from gaussreg_amd.matching import SuperPointMatching
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(0)
R = lambda *s: torch.randn(*s, device=dev, generator=g)
fr = torch.nn.functional.normalize(R(767, 256), dim=1); fs = torch.nn.functional.normalize(R(767, 256), dim=1)
spm = SuperPointMatching(256)
print("single 767:", _ms(lambda: spm(fr, fs), 50, 5), "ms")
npair = 64
fb = torch.nn.functional.normalize(R(npair * 2 * 767, 256), dim=1); nl = [767] * (2 * npair)
ms = _ms(lambda: spm.forward_batch(fb, nl), 20, 3)
print("batch 64x767:", ms, "ms", _mfma(ms, 2.0 * npair * 767 * 767 * 256))
