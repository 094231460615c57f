import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd.matching import SuperPointMatching
g = torch.Generator(device="cuda").manual_seed(0)
ref = torch.nn.functional.normalize(torch.randn(767, 256, device="cuda", generator=g), dim=1)
src = torch.nn.functional.normalize(torch.randn(767, 256, device="cuda", generator=g), dim=1)
spm = SuperPointMatching(256)
for _ in range(20): spm(ref, src)
torch.cuda.synchronize()
