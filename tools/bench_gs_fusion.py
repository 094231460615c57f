"""config[3] stand-in: fuse two 1.5 M-Gaussian scenes (gs_fusion.py:231-262 on the .ply wire format) and render the
3 M-Gaussian result at 1920x1080 (8 views per call).  Synthetic scenes; fusion timed on device-resident records."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np, torch
from gaussreg_amd import synthetic
from gaussreg_amd.gs_io import gaussian_fuse_records, split_records
from gaussreg_amd.rasterizer import GaussianRasterizationSettings, ViewBatch, rasterize_views


def records(P, seed):
    g = synthetic.gaussians_c2(P, seed=seed, sh_degree=3)
    rec = np.zeros((P, 62), np.float32)
    rec[:, 0:3] = g["means3D"]
    rec[:, 6:9] = g["shs"][:, 0, :]
    rec[:, 9:54] = g["shs"][:, 1:, :].transpose(0, 2, 1).reshape(P, 45)   # f_rest: channel-major (gs_fusion.py:180)
    rec[:, 54] = np.log(g["opacities"][:, 0] / (1 - g["opacities"][:, 0]))
    rec[:, 55:58] = np.log(g["scales"])
    rec[:, 58:62] = g["rotations"]
    return torch.from_numpy(rec).cuda()


P = 1_500_000
r1, r2 = records(P, 0), records(P, 1)
c, s = np.cos(0.3), np.sin(0.3)
T = np.eye(4); T[:3, :3] = 1.1 * np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]]); T[:3, 3] = [0.2, -0.1, 0.05]
for _ in range(2): fused = gaussian_fuse_records(r1, r2, T)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): fused = gaussian_fuse_records(r1, r2, T)
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"gaussian_fuse 2 x {P} records -> {fused.shape[0]} kept: {dt*1e3:.2f} ms ({2*P*248/dt/1e9:.0f} GB/s of input records)")
parts = split_records(fused)
W, H, V = 1920, 1080, 8
cams = synthetic.camera_ring(V, W, H, seed=0)
st = ViewBatch([GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], torch.zeros(3), 1.0, torch.from_numpy(cm["viewmatrix"]),
                                              torch.from_numpy(cm["projmatrix"]), 3, torch.from_numpy(cm["campos"]), False, False) for cm in cams])
def render():
    return rasterize_views(st, parts["means3D"], parts["opacities"], shs=parts["shs"], scales=parts["scales"], rotations=parts["rotations"])
for _ in range(2): render()
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): img, radii, nr = render()
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 5
print(f"render {fused.shape[0]} fused Gaussians at {W}x{H}, {V} views/call: {dt*1e3:.1f} ms -> {V/dt:.0f} views/s ({sum(nr)/V/1e6:.2f} M instances/view)")
