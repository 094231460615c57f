"""Dev tool: the bench scene (1 M Gaussians, 640 x 480) rendered 300 times with frames in flight (one camera per call, then
8 cameras per call), every output compared bit for bit with the serial render of the same camera."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gaussreg_amd import synthetic
from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
P, W, H = 1_000_000, 640, 480
g = synthetic.gaussians_c2(P, seed=0, sh_degree=3)
cams = synthetic.camera_ring(8, W, H, seed=0)
t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
sets = [GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0, torch.from_numpy(c["viewmatrix"]),
                                      torch.from_numpy(c["projmatrix"]), 3, torch.from_numpy(c["campos"]), False, False) for c in cams]
rast = [GaussianRasterizer(s) for s in sets]
kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
os.environ["GR_RASTER_PIPELINE"] = "0"
want = [tuple(x.clone() for x in r(t["means3D"], None, t["opacities"], **kw)) for r in rast]
wantb = rasterize_views(sets, t["means3D"], t["opacities"], **kw)
wantb = (wantb[0].clone(), wantb[1].clone(), wantb[2])
torch.cuda.synchronize()
os.environ["GR_RASTER_PIPELINE"] = "1"
bad = 0
outs = []
for i in range(300):
    outs.append((i % 8, rast[i % 8](t["means3D"], None, t["opacities"], **kw)))
    if len(outs) == 20:
        for k, (img, radii) in outs:
            bad += int(not (torch.equal(img, want[k][0]) and torch.equal(radii, want[k][1])))
        outs = []
for i in range(30):
    img, radii, nr = rasterize_views(sets, t["means3D"], t["opacities"], **kw)
    bad += int(not (torch.equal(img, wantb[0]) and torch.equal(radii, wantb[1]) and nr == wantb[2]))
print("mismatching calls:", bad)
sys.exit(1 if bad else 0)
