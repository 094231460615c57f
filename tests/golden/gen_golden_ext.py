#!/usr/bin/env python3
"""Generate golden vectors for grid_subsampling / radius_neighbors FROM THE REFERENCE ITSELF.

Runs only where /root/reference is mounted (this build container): it drives the reference's
own C++ core (radius_neighbors_cpu / grid_subsampling_cpu, compiled in place by oracle/Makefile
into oracle/_ref/libgaussreg_ref.so -- no reference source is copied) on seeded inputs and stores
inputs + outputs as small .npz fixtures next to this script.  The fixtures are data; they are what
pins oracle/*.c(pp) and the HIP path on the GPU box, where the reference does not exist.

Cases:
  c1           torch.manual_seed(0); torch.rand(20000,3) -- BASELINE config C1 (inputs are
               regenerated from the seed in the tests; only outputs are stored)
  multibatch   lengths [1200, 800], q != s for the radius search
  noneighbor   radius so small that only self matches / nothing matches (width 1 and width 0)
  ties         duplicated points (tie order is traversal-dependent in the reference; the tests
               compare equal-distance runs as sets)
  pyramid      the 5-level demo pyramid (utils/data.py:13-77 call pattern) on a small room pair
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(HERE, "..", "..")))

from oracle import capi  # noqa: E402


def room_pair(n_per_cloud, seed=0):
    """SURVEY.md App. D generator, scaled: points on the faces + interior of a 4x3x2.5 m box."""
    rng = np.random.default_rng(seed)
    clouds = []
    ext = np.array([4.0, 3.0, 2.5])
    for _ in range(2):
        nf = n_per_cloud // 8
        pts = []
        for axis in range(3):
            for side in (0.0, 1.0):
                p = rng.random((nf, 3)) * ext
                p[:, axis] = side * ext[axis] + rng.normal(0.0, 0.01, nf)
                pts.append(p)
        pts.append(rng.random((n_per_cloud - 6 * nf, 3)) * ext)
        p = np.concatenate(pts, 0)
        p -= (p.max(0) + p.min(0)) / 2
        clouds.append(p.astype(np.float32))
    return clouds


def main():
    assert os.path.isdir("/root/reference"), "golden vectors can only be generated next to the reference"
    capi.build()
    import torch

    # ---- c1 (inputs reproducible from the torch seed; store a checksum of them too)
    torch.manual_seed(0)
    pts = torch.rand(20000, 3).numpy()
    lens = np.array([20000], np.int64)
    sp, sl = capi.ref_grid_subsampling(pts, lens, 0.05)
    nb = capi.ref_radius_neighbors(pts, pts, lens, lens, 0.0625)
    assert sp.shape[0] == 7366 and nb.shape == (20000, 39), (sp.shape, nb.shape)
    np.savez_compressed(os.path.join(HERE, "ext_c1.npz"), points_sum=np.float64(pts.astype(np.float64).sum()),
                        first_points=pts[:4], s_points=sp, s_lengths=sl, neighbors=nb.astype(np.int32),
                        voxel=np.float32(0.05), radius=np.float32(0.0625))

    # ---- multibatch, q != s
    rng = np.random.default_rng(1)
    s = rng.random((2000, 3)).astype(np.float32)
    q = rng.random((700, 3)).astype(np.float32)
    sl_ = np.array([1200, 800], np.int64)
    ql_ = np.array([300, 400], np.int64)
    nb = capi.ref_radius_neighbors(q, s, ql_, sl_, 0.11)
    sp, spl = capi.ref_grid_subsampling(s, sl_, 0.07)
    np.savez_compressed(os.path.join(HERE, "ext_multibatch.npz"), q=q, s=s, q_lengths=ql_, s_lengths=sl_,
                        radius=np.float32(0.11), neighbors=nb.astype(np.int32), voxel=np.float32(0.07),
                        s_points=sp, sub_lengths=spl)

    # ---- no neighbour / self only
    p = rng.random((500, 3)).astype(np.float32)
    l1 = np.array([500], np.int64)
    nb_self = capi.ref_radius_neighbors(p, p, l1, l1, 1e-4)
    far = p + np.float32(10.0)
    nb_none = capi.ref_radius_neighbors(far, p, l1, l1, 0.05)
    np.savez_compressed(os.path.join(HERE, "ext_noneighbor.npz"), p=p, far=far, lengths=l1,
                        nb_self=nb_self.astype(np.int32), nb_none_shape=np.array(nb_none.shape, np.int64),
                        r_self=np.float32(1e-4), r_none=np.float32(0.05))

    # ---- ties (duplicates)
    base = rng.random((50, 3)).astype(np.float32)
    dup = np.concatenate([base, base], 0)
    l2 = np.array([100], np.int64)
    nb = capi.ref_radius_neighbors(dup, dup, l2, l2, 0.3)
    sp, spl = capi.ref_grid_subsampling(dup, l2, 0.2)
    np.savez_compressed(os.path.join(HERE, "ext_ties.npz"), p=dup, lengths=l2, radius=np.float32(0.3),
                        neighbors=nb.astype(np.int32), voxel=np.float32(0.2), s_points=sp, sub_lengths=spl)

    # ---- pyramid on a small room pair: call pattern of precompute_data_stack_mode
    ref_c, src_c = room_pair(3000, seed=0)
    points = np.concatenate([ref_c, src_c], 0)
    lengths = np.array([ref_c.shape[0], src_c.shape[0]], np.int64)
    out = {"points0": points, "lengths0": lengths}
    voxel, radius = 0.025 * 4, 0.0625 * 4   # coarser start so 6000 points give a meaningful 5-level pyramid
    out["voxel0"], out["radius0"] = np.float32(voxel), np.float32(radius)
    # utils/data.py:23-28: voxel_size doubles every iteration INCLUDING i == 0, so level i >= 1 is
    # subsampled with voxel0 * 2**i
    pl, ll = [points], [lengths]
    v = voxel
    for i in range(5):
        if i > 0:
            sp, sl2 = capi.ref_grid_subsampling(pl[-1], ll[-1], v)
            pl.append(sp)
            ll.append(sl2)
        v *= 2
    r = radius
    for i in range(5):
        out[f"points{i}"], out[f"lengths{i}"] = pl[i], ll[i]
        out[f"neighbors{i}"] = capi.ref_radius_neighbors(pl[i], pl[i], ll[i], ll[i], r).astype(np.int32)
        if i < 4:
            out[f"subsampling{i}"] = capi.ref_radius_neighbors(pl[i + 1], pl[i], ll[i + 1], ll[i], r).astype(np.int32)
            out[f"upsampling{i}"] = capi.ref_radius_neighbors(pl[i], pl[i + 1], ll[i], ll[i + 1], r * 2).astype(np.int32)
        r *= 2
    np.savez_compressed(os.path.join(HERE, "ext_pyramid.npz"), **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
