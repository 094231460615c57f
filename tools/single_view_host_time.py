"""Dev tool: where the HOST time of a one-camera forward() goes (the library call vs the Python around it)."""
import os, sys, time
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import synthetic, _lib
from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
P, W, H = 1_000_000, 640, 480
g = synthetic.gaussians_c2(P, seed=0, sh_degree=3)
cams = synthetic.camera_ring(4, W, H, seed=0)
t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
rast = [GaussianRasterizer(GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0,
        torch.from_numpy(c["viewmatrix"]), torch.from_numpy(c["projmatrix"]), 3, torch.from_numpy(c["campos"]), False, False))
        for c in cams]
L = _lib.lib()
real = L.gr_raster_forward
acc = {"c": 0.0, "n": 0}
class Wrap:
    def __getattr__(self, name):
        f = getattr(L, name)
        if name == "gr_raster_forward_finish":
            def timed_f(*a):
                t0 = time.perf_counter()
                r = f(*a)
                acc["f"] = acc.get("f", 0.0) + time.perf_counter() - t0
                return r
            return timed_f
        if name != "gr_raster_forward":
            return f
        def timed(*a):
            t0 = time.perf_counter()
            r = f(*a)
            acc["c"] += time.perf_counter() - t0
            acc["n"] += 1
            return r
        return timed
_lib.lib = lambda: Wrap()
def loop(n):
    for i in range(n):
        rast[i % 4](t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
loop(20)
torch.cuda.synchronize()
acc["c"] = 0.0; acc["n"] = 0; acc["f"] = 0.0
t0 = time.perf_counter()
loop(400)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 400
c, f = acc["c"] / acc["n"] * 1e6, acc["f"] / acc["n"] * 1e6
print("per call %.1f us; inside gr_raster_forward %.1f us (enqueue); inside gr_raster_forward_finish %.1f us (waiting for the counts); "
      "Python around them %.1f us" % (dt * 1e6, c, f, dt * 1e6 - c - f))
if "--profile" in sys.argv:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); loop(400); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)
