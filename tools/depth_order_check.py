"""Dev tool: gr_raster_forward through the C-ABI on a caller-owned geometry buffer, then the depth order the frame left there
(gr_raster_debug_geom_layout) against numpy's stable argsort of the depth fields -- frame by frame, so the plain first frames
and the deferred ones (four-launch depth sort) are both seen.   python tools/depth_order_check.py [P W H V frames]"""
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import torch
from gaussreg_amd import _lib, synthetic
from gaussreg_amd.rasterizer import GaussianRasterizationSettings, ViewBatch


def run(P, W, H, V, frames, verbose=True, scene=None):
    L = _lib.lib()
    g = scene if scene is not None else synthetic.gaussians_c2(P, seed=0, sh_degree=3)
    cams = synthetic.camera_ring(max(V, 4), W, H, seed=0)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    sets = [GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0, torch.from_numpy(c["viewmatrix"]),
                                          torch.from_numpy(c["projmatrix"]), 3, torch.from_numpy(c["campos"]), False, False)
            for c in cams]
    gbytes = L.gr_raster_geom_bytes(P, V, W, H)
    off = (ctypes.c_int64 * 4)()
    assert L.gr_raster_debug_geom_layout(P, V, W, H, off) == 4
    geom = torch.zeros(gbytes + 256, dtype=torch.uint8, device="cuda")
    binb = torch.empty(1 << 28, dtype=torch.uint8, device="cuda")
    color = torch.empty((V, 3, H, W), dtype=torch.float32, device="cuda")
    radii = torch.empty((V, P), dtype=torch.int32, device="cuda")
    nr = (ctypes.c_int64 * (V + 1))()
    st = _lib.stream_ptr(torch.device("cuda"))
    bad_frames = 0
    for f in range(frames):
        vb = ViewBatch([sets[(f + i) % len(sets)] for i in range(V)])
        nr[V] = 0
        rc = L.gr_raster_forward(P, 16, _lib.ptr(t["means3D"]), _lib.ptr(t["shs"]), None, _lib.ptr(t["opacities"]),
                                 _lib.ptr(t["scales"]), _lib.ptr(t["rotations"]), None, vb.array, V, _lib.ptr(radii),
                                 _lib.ptr(geom), gbytes, _lib.ptr(binb), binb.numel(), _lib.ptr(color), 0, nr, st)
        torch.cuda.synchronize()
        gh = geom.cpu().numpy()
        field = gh[off[0]: off[0] + 4 * V * P].view(np.uint32).reshape(V, P)
        order = gh[off[1]: off[1] + 4 * V * P].view(np.int32).reshape(V, P)
        nvis = gh[off[3]: off[3] + 4 * V].view(np.int32)
        ok = True
        for v in range(V):
            vis = np.flatnonzero(field[v])
            want = vis[np.argsort(field[v][vis], kind="stable")]
            got = order[v][: nvis[v]]
            if nvis[v] != len(vis) or not np.array_equal(got, want):
                ok = False
                if verbose:
                    nb = int((got[: len(want)] != want[: len(got)]).sum()) if len(got) and len(want) else -1
                    print(f"frame {f} view {v}: rc {rc} nvis {nvis[v]} want {len(vis)} mismatches {nb}")
                    if len(got) == len(want):
                        w = np.flatnonzero(got != want)[:5]
                        print("   first at", w, "got", got[w], "want", want[w], "fields", field[v][got[w]], field[v][want[w]])
        bad_frames += 0 if ok else 1
        if verbose:
            img = color.cpu().numpy()
            if f % len(sets) == 0:
                same = None if f == 0 else int((img != first).sum())
                if f == 0:
                    first = img.copy()
            else:
                same = None
            print(f"frame {f}: rc {rc} ok {ok} rendered {[int(nr[v]) for v in range(V)]} lds_state "
                  f"{L.gr_raster_lds_atomics_lane_ordered()} pixels differing from frame 0: {same}")
    return bad_frames


if __name__ == "__main__":
    a = [int(x) for x in sys.argv[1:]]
    P, W, H, V, frames = (a + [3000, 64, 48, 1, 8][len(a):])[:5]
    sys.exit(1 if run(P, W, H, V, frames) else 0)
