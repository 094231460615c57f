import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from gaussreg_amd import ext, synthetic
pts, lens = synthetic.cloud_200k(8, seed=0)
dp = pts.cuda()
for _ in range(12): ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
torch.cuda.synchronize()
