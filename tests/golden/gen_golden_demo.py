#!/usr/bin/env python3
"""Golden vectors AT DEMO SHAPES, generated from the reference itself in this container:
the reference's C++ core (oracle/_ref, compiled where it lies) for the 5-level pyramid of a 2 x 30 000-point pair, and
the reference's imported Python modules (torch CPU) for PointMatching at 256 x 128 x 128, log-Sinkhorn at
256 x 128 x 128, KPConv at a real stage shape with radius_search-produced neighbours, LocalGlobalRegistration with
correspondence_limit, calibrate_neighbors_stack_mode and index_select.

The inputs are large, so they are NOT stored: the tests regenerate them from the same seeds (numpy default_rng /
torch.manual_seed on the CPU -- both deterministic for a given library version; an fp64 checksum of every regenerated
input is stored and asserted first).  Outputs are stored as: exact data where small (bit-packed correspondence
matrices, gathered correspondences, transforms), per-level shapes + 64-bit position-weighted checksums + sampled rows
for the big integer tensors, and sampled rows / whole sample matrices for the float tensors -- each float sample in two
versions: the reference's own fp32 result and the same module evaluated in fp64, so a test can assert
|hip - f64| <= |ref32 - f64| + eps instead of a loose tolerance.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)

from oracle import capi  # noqa: E402
from gen_golden_ext import room_pair  # noqa: E402

LIMITS = [89, 30, 43, 49, 49]


from demo_inputs import checksum_i64, tie_rows  # noqa: E402


def ref_pyramid(points, lengths):
    """utils/data.py:13-77 with the reference C++ core (full widths, then the column truncation of radius_search.py:25-26)."""
    pts, lens = [points], [lengths]
    voxel = 0.025
    for i in range(1, 5):
        voxel *= 2
        p, l = capi.ref_grid_subsampling(pts[-1], lens[-1], voxel)
        pts.append(p)
        lens.append(l)
    nb, sub, up = [], [], []
    rad = 0.0625
    for i in range(5):
        nb.append(capi.ref_radius_neighbors(pts[i], pts[i], lens[i], lens[i], rad)[:, :LIMITS[i]])
        if i < 4:
            sub.append(capi.ref_radius_neighbors(pts[i + 1], pts[i], lens[i + 1], lens[i], rad)[:, :LIMITS[i]])
            up.append(capi.ref_radius_neighbors(pts[i], pts[i + 1], lens[i], lens[i + 1], 2 * rad)[:, :LIMITS[i + 1]])
        rad *= 2
    return pts, lens, nb, sub, up


def import_reference():
    import torch
    sys.path[:] = [p for p in sys.path if os.path.abspath(p or ".") != ROOT]
    sys.path.insert(0, "/root/reference")
    for name in ("ipdb", "IPython", "open3d", "coloredlogs", "easydict", "plyfile", "fpsample", "cv2"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["IPython"].embed = lambda *a, **k: None
    ext = types.ModuleType("geotransformer.ext")
    sys.modules["geotransformer.ext"] = ext
    torch.Tensor.cuda = lambda self, *a, **k: self
    import geotransformer
    assert geotransformer.__file__.startswith("/root/reference")
    return torch, ext


def main():
    assert os.path.isdir("/root/reference")
    capi.build()
    out = {}
    # ------------------------------------------------------------------ pyramid at 2 x 30 000 points (reference C++ core)
    ref, src = room_pair(30000, 0)
    points = np.concatenate([ref, src]).astype(np.float32)
    lengths = np.array([30000, 30000], np.int64)
    out["pyr_points_sum"] = np.float64(points.astype(np.float64).sum())
    pts, lens, nb, sub, up = ref_pyramid(points, lengths)
    rng = np.random.default_rng(7)
    for i in range(5):
        out[f"pyr_len_{i}"] = lens[i]
        out[f"pyr_pts_sum_{i}"] = np.float64(pts[i].astype(np.float64).sum())
        out[f"pyr_pts_bits_{i}"] = checksum_i64(pts[i].view(np.uint32))
        rows = np.sort(rng.choice(pts[i].shape[0], size=min(256, pts[i].shape[0]), replace=False))
        out[f"pyr_rows_{i}"] = rows
        out[f"pyr_pts_rows_{i}"] = pts[i][rows]
        for name, lst in (("nb", nb), ("sub", sub), ("up", up)):
            if i < len(lst):
                a = lst[i]
                out[f"pyr_{name}_shape_{i}"] = np.array(a.shape, np.int64)
                out[f"pyr_{name}_sum_{i}"] = checksum_i64(a)
                out[f"pyr_{name}_setsum_{i}"] = checksum_i64(np.sort(a, axis=1))  # invariant to the order inside a row
                qi, si = {"nb": (i, i), "sub": (i + 1, i), "up": (i, i + 1)}[name]
                tr = tie_rows(pts[qi], pts[si], a)
                keep = np.ones(a.shape[0], bool)
                keep[tr] = False
                out[f"pyr_{name}_tierows_{i}"] = tr.astype(np.int32)
                out[f"pyr_{name}_notie_sum_{i}"] = checksum_i64(a[keep])   # exact order on every row without a tie
                r = np.sort(rng.choice(a.shape[0], size=min(256, a.shape[0]), replace=False))
                out[f"pyr_{name}_rows_{i}"] = r
                out[f"pyr_{name}_vals_{i}"] = a[r].astype(np.int32)
    print("pyramid levels", [int(l.sum()) for l in lens], "widths", [a.shape[1] for a in nb])

    # ------------------------------------------------------------------ reference Python modules
    torch, ext = import_reference()
    torch.set_num_threads(8)
    from geotransformer.modules.geotransformer.point_matching import PointMatching
    from geotransformer.modules.geotransformer.local_global_registration import LocalGlobalRegistration
    from geotransformer.modules.sinkhorn import LearnableLogOptimalTransport
    import geotransformer.modules.kpconv.kpconv as kp_mod
    from geotransformer.modules.ops.index_select import index_select

    # ---- a10 at the demo shape: 256 patches of 128 x 128
    import demo_inputs
    pmi = demo_inputs.point_matching_inputs()
    score, rk, sk, rp, sp_, rki, ski, gs = (pmi[k] for k in ("score", "ref_masks", "src_masks", "ref_points", "src_points",
                                                               "ref_idx", "src_idx", "global_scores"))
    P, K = 256, 128
    out["pm_score_sum"] = np.float64(score.double().sum().item())
    pm = PointMatching(k=3, mutual=True, confidence_threshold=0.05, use_dustbin=False, use_global_score=False)
    corr = pm.compute_correspondence_matrix(torch.exp(score), rk, sk)
    a, b, c, d, e = pm(rp, sp_, rk, sk, rki, ski, score, gs)
    out.update(pm_corr_bits=np.packbits(corr.numpy()), pm_out_ref_points=a.numpy(), pm_out_src_points=b.numpy(),
               pm_out_ref_idx=c.numpy().astype(np.int32), pm_out_src_idx=d.numpy().astype(np.int32), pm_out_scores=e.numpy())
    print("point matching:", int(corr.sum()), "correspondences")

    # ---- Sinkhorn at 256 x 128 x 128, fp32 (the reference as is) and fp64 (same module on doubles)
    sc, rm, cm = demo_inputs.sinkhorn_inputs()
    out["sk_scores_sum"] = np.float64(sc.double().sum().item())
    ot = LearnableLogOptimalTransport(100)
    with torch.no_grad():
        ot.alpha.fill_(0.61)
        o32 = ot(sc, rm, cm)
        ot64 = LearnableLogOptimalTransport(100).double()
        ot64.alpha.data.fill_(float(np.float32(0.61)))
        o64 = ot64(sc.double(), rm, cm)
    pick = np.array([0, 77, 130, 255])
    out.update(sk_alpha=np.float32(0.61), sk_pick=pick, sk_out32=o32.numpy()[pick], sk_out64=o64.numpy()[pick],
               sk_sum32=o32.double().sum(dim=(1, 2)).numpy(), sk_sum64=o64.sum(dim=(1, 2)).numpy())

    # ---- KPConv at a real stage shape: level-1 points of the pair above, level-1 neighbours (limit 30), 64 -> 64
    base = demo_inputs.K015
    radius1, sigma1 = 0.125, 0.1  # stage 2 of the backbone: init_radius * 2, init_sigma * 2 (config.py:78-83, backbone.py)
    kp_mod.load_kernels = lambda radius, k, dimension=3, fixed='center': (base * radius).astype(np.float32)
    p1 = torch.from_numpy(pts[1])
    nb1 = torch.from_numpy(nb[1].astype(np.int64))
    Cin, Cout = 64, 64
    feats, w = demo_inputs.kpconv_inputs(p1.shape[0], Cin, Cout)
    conv = kp_mod.KPConv(Cin, Cout, 15, radius=radius1, sigma=sigma1, bias=False)
    with torch.no_grad():
        conv.weights.copy_(w)
        y32 = conv(feats, p1, p1, nb1)
        conv64 = kp_mod.KPConv(Cin, Cout, 15, radius=radius1, sigma=sigma1, bias=False).double()
        conv64.weights.data.copy_(w.double())
        conv64.kernel_points.data.copy_(conv.kernel_points.double())
        y64 = conv64(feats.double(), p1.double(), p1.double(), nb1)
    rows = np.sort(np.random.default_rng(5).choice(p1.shape[0], size=2048, replace=False))
    out.update(kp_feats_sum=np.float64(feats.double().sum().item()), kp_w_sum=np.float64(w.double().sum().item()),
               kp_radius=np.float32(radius1), kp_sigma=np.float32(sigma1), kp_kernel_points=conv.kernel_points.numpy(),
               kp_rows=rows, kp_out32=y32.numpy()[rows], kp_out64=y64.numpy()[rows],
               kp_colsum32=y32.double().sum(0).numpy(), kp_colsum64=y64.sum(0).numpy())
    print("kpconv", tuple(y32.shape), "max |f32 - f64| =", float((y32.double() - y64).abs().max()))

    # ---- LocalGlobalRegistration with correspondence_limit (local_global_registration.py:145-152)
    ref_k, src_k, rk2, sk2, lscore, gsc = demo_inputs.lgr_limit_inputs()
    lgr = LocalGlobalRegistration(3, 0.1, mutual=True, confidence_threshold=0.05, use_dustbin=False, use_global_score=True,
                                  correspondence_threshold=3, correspondence_limit=500, num_refinement_steps=5)
    a, b, c, T = lgr(ref_k, src_k, rk2, sk2, lscore, gsc)
    out.update(lgrl_score_sum=np.float64(lscore.double().sum().item()), lgrl_out_ref=a.numpy(), lgrl_out_src=b.numpy(),
               lgrl_out_scores=c.numpy(), lgrl_out_transform=T.numpy(), lgrl_limit=np.int64(500))
    print("lgr with limit 500:", a.shape[0], "correspondences")

    # ---- index_select (modules/ops/index_select.py:4-31)
    g = torch.Generator().manual_seed(3)
    data = torch.randn(500, 7, generator=g)
    idx2 = torch.randint(0, 500, (40, 9), generator=g)
    out.update(is_data=data.numpy(), is_idx=idx2.numpy().astype(np.int32), is_out0=index_select(data, idx2, dim=0).numpy(),
               is_out1=index_select(data.t().contiguous(), idx2, dim=1).numpy())

    # ---- calibrate_neighbors_stack_mode (utils/data.py:192-217), reference collate on the reference ext
    def ref_radius(q, s, ql, sl, r):
        return torch.from_numpy(capi.ref_radius_neighbors(q.numpy(), s.numpy(), ql.numpy(), sl.numpy(), float(r)))

    def ref_grid(p, l, v):
        a_, b_ = capi.ref_grid_subsampling(p.numpy(), l.numpy(), float(v))
        return [torch.from_numpy(a_), torch.from_numpy(b_)]

    ext.radius_neighbors = ref_radius
    ext.grid_subsampling = ref_grid
    # grid_subsample.py / radius_search.py look the functions up on the module object at call time
    import geotransformer.utils.data as data_mod

    class TinySet:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            r_, s_ = room_pair(4000, 100 + i)
            return {"ref_points": r_, "src_points": s_, "ref_feats": np.ones((r_.shape[0], 1), np.float32),
                    "src_feats": np.ones((s_.shape[0], 1), np.float32)}

    limits = data_mod.calibrate_neighbors_stack_mode(TinySet(), data_mod.registration_collate_fn_stack_mode, 4, 0.025, 0.0625,
                                                     keep_ratio=0.8, sample_threshold=2000)
    out["calib_limits"] = np.asarray(limits, np.int64)
    print("calibrated limits", limits)

    path = os.path.join(HERE, "demo_shapes.npz")
    np.savez_compressed(path, **out)
    print("demo_shapes.npz", os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
