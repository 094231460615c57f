"""Development tool: what the blend kernel does per tile (GR_BLEND_STATS=1 build path)."""
import ctypes, os, sys
os.environ["GR_BLEND_STATS"] = "1"
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import _lib, synthetic
from gaussreg_amd import rasterizer as R
L = _lib.lib()
ctypes.CDLL(_lib.LIB_PATH)
V = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = synthetic.gaussians_c2(1000000, 0)
cams = synthetic.camera_ring(V, 640, 480, 0)
t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
st = [R.GaussianRasterizationSettings(480, 640, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0, torch.from_numpy(c["viewmatrix"]),
                                      torch.from_numpy(c["projmatrix"]), 3, torch.from_numpy(c["campos"]), False, False) for c in cams]
img, radii, nr = R.rasterize_views(st, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
out = (ctypes.c_ulonglong * 8)()
fn = L.gr_debug_blend_stats
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong)]
fn(out)
tiles, batches, loaded, cell_entries, steps, blended, maxhits, lane_entries = [int(out[i]) for i in range(8)]
R_ = sum(nr)
print(f"views {V}: instances {R_}, tiles {tiles}, batches/tile {batches/tiles:.2f}, entries loaded {loaded} = {loaded/R_:.3f} of all,"
      f" cell-list entries per loaded entry {cell_entries/max(loaded,1):.2f} (of 16), wave blend steps/tile {steps/tiles:.1f},"
      f" blended (entry,pixel) pairs {blended} = {blended/max(cell_entries*16,1):.3f} of cell-list entry x 16 px;"
      f" per (wave, batch): longest list walked {2*steps/max(4*batches,1):.1f} entries, most blends by one pixel {maxhits/max(4*batches,1):.1f};"
      f" hit rate of live lanes {blended/max(lane_entries,1):.3f}")
