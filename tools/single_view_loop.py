"""Dev tool: GaussianRasterizer.forward, one camera per call, in a loop (for rocprofv3 --kernel-trace + tools/trace_gaps.py)."""
import os, sys
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import synthetic
from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
P, W, H = 1_000_000, 640, 480
g = synthetic.gaussians_c2(P, seed=0, sh_degree=3)
cams = synthetic.camera_ring(4, W, H, seed=0)
t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
rast = [GaussianRasterizer(GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0,
        torch.from_numpy(c["viewmatrix"]), torch.from_numpy(c["projmatrix"]), 3, torch.from_numpy(c["campos"]), False, False))
        for c in cams]
for i in range(16):
    img, radii = rast[i % 4](t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
torch.cuda.synchronize()
print(img.shape)
if "--time" in sys.argv:  # views/s of the drop-in forward, one camera per call (the bench's `single_view` figure)
    import time
    n = 400
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        img, radii = rast[i % 4](t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("single view: %.4f ms per call = %.0f views/s" % (dt * 1e3, 1.0 / dt))
