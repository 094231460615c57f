import os, sys, json, torch
root = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, "tools"))
import bench_extras
out = bench_extras.config3(torch.device("cuda", 0))
print(json.dumps(out["raster_3M_1080p"]))
