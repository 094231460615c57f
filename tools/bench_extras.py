"""The SURVEY 8(d) kernels that are not the two headline kernels, timed for bench.py's "extras" key.

Every entry: {"ms": median milliseconds per call (torch.cuda events on the stream the C ABI launches on -- the library
launches on torch's current stream), "bytes" or "flops": the ALGORITHMIC figure of SURVEY 8(d), "bound", "frac": fraction
of that bound's peak (HBM 8 TB/s, fp32 MFMA 157.3 TFLOP/s)}.  Demo shapes of SURVEY App. D unless noted."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_GBS = 8000.0
MFMA_TF = 157.3


def _ms(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        b.synchronize()
        ts.append(a.elapsed_time(b))
    return float(sorted(ts)[len(ts) // 2])


def _hbm(ms, nbytes, **kw):
    d = {"ms": round(ms, 4), "bytes": int(nbytes), "bound": "hbm", "GB/s": round(nbytes / ms / 1e6, 1),
         "frac": round(nbytes / ms / 1e6 / HBM_GBS, 4)}
    d.update(kw)
    return d


def _mfma(ms, flops, **kw):
    d = {"ms": round(ms, 4), "flops": float(flops), "bound": "mfma_f32", "TFLOP/s": round(flops / ms / 1e9, 2),
         "frac": round(flops / ms / 1e9 / MFMA_TF, 4)}
    d.update(kw)
    return d


def run(dev):
    from gaussreg_amd import ext, ops, pair_pipeline, synthetic
    from gaussreg_amd.data import precompute_data_stack_mode
    from gaussreg_amd.embedding import GeometricStructureEmbedding
    from gaussreg_amd.kpconv import KPConv
    from gaussreg_amd.matching import LocalGlobalRegistration, PointMatching, SuperPointMatching
    from gaussreg_amd.registration import farthest_point_sampling
    from gaussreg_amd.sinkhorn import LearnableLogOptimalTransport
    out = {}
    g = torch.Generator(device=dev).manual_seed(0)
    R = lambda *s: torch.randn(*s, device=dev, generator=g)  # noqa: E731
    with torch.no_grad():
        # ---- grid_subsample, 200 k points (12 N + 12 M bytes), both row orders
        pts, lens = synthetic.cloud_200k(1, seed=0)
        dp = pts.to(dev)
        for order in ("reference", "cell"):
            sp, _ = ext.grid_subsampling(dp, lens, 0.05, order=order)
            ms = _ms(lambda: ext.grid_subsampling(dp, lens, 0.05, order=order))
            out[f"grid_subsample_200k_{order}_order"] = _hbm(ms, 12 * dp.shape[0] + 12 * sp.shape[0], Mpts_per_s=round(0.2 / ms * 1e3, 1))
        # ---- the same, 64 clouds of 200 k points per call (launch overheads amortised: what is left is the sort + the cell passes)
        pts64, lens64 = synthetic.cloud_200k(64, seed=1)
        dp64 = pts64.to(dev)
        for order in ("reference", "cell"):
            sp, _ = ext.grid_subsampling(dp64, lens64, 0.05, order=order)
            ms = _ms(lambda: ext.grid_subsampling(dp64, lens64, 0.05, order=order), 5, 1)
            out[f"grid_subsample_64x200k_{order}_order"] = _hbm(ms, 12 * dp64.shape[0] + 12 * sp.shape[0], Mpts_per_s=round(12.8 / ms * 1e3, 1),
                                                                note="algorithmic bytes 12 N + 12 M; the implementation is a bucket sort of every cloud's voxel keys (one pass over "
                                                                     "the top nine bits + every bucket finished in LDS), run heads counted per workgroup, one gather of "
                                                                     "the points; the measured traffic ratio is in profiles/*_pmc_ops.json")
        del pts64, dp64, sp
        # ---- the data pyramid, 64 pairs of 2 x 30 000 points per call
        Bp = 64
        clouds = []
        for b in range(Bp):
            r_, s_, _ = pair_pipeline.synthetic_room_pair(b, 30000, dev)
            clouds += [r_]
        for b in range(Bp):
            clouds += [pair_pipeline.synthetic_room_pair(b, 30000, dev)[1]]
        bp = torch.cat(clouds).contiguous()
        bl = torch.tensor([30000] * (2 * Bp))
        for order in ("reference", "cell"):
            ms = _ms(lambda: precompute_data_stack_mode(bp, bl, 5, 0.025, 0.0625, [89, 30, 43, 49, 49], order=order), 5, 1)
            out[f"pyramid_64pairs_{order}_order"] = {"ms": round(ms, 3), "pairs_per_s": round(Bp / ms * 1e3, 1)}
        d = precompute_data_stack_mode(bp[:60000].contiguous(), torch.tensor([30000, 30000]), 5, 0.025, 0.0625, [89, 30, 43, 49, 49])
        del bp, clouds
        # ---- point_to_node_partition (12 (N+M) + 8 N + M (1 + 128*9) bytes)
        nf, nc = d["lengths"][1].tolist(), d["lengths"][-1].tolist()
        pf, pc = d["points"][1][:nf[0]].contiguous(), d["points"][-1][:nc[0]].contiguous()
        ms = _ms(lambda: ops.point_to_node_partition(pf, pc, 128), 20)
        N, M = pf.shape[0], pc.shape[0]
        out["point_to_node_partition"] = _hbm(ms, 12 * (N + M) + 8 * N + M * (1 + 128 * 9), shape=[N, M, 128])
        # ---- correspondence matrix / PointMatching.forward (4 P K^2 + 2 P K in, P K^2 out)
        P, K = 256, 128
        score = torch.log_softmax(R(P, K, K) * 3, 2)
        rm = torch.rand(P, K, device=dev, generator=g) > 0.3
        sm = torch.rand(P, K, device=dev, generator=g) > 0.3
        rp, sp_ = R(P, K, 3), R(P, K, 3)
        ri = torch.randint(0, 30000, (P, K), device=dev)
        si = torch.randint(0, 30000, (P, K), device=dev)
        gs = torch.rand(P, device=dev)
        pm = PointMatching(3)
        ms = _ms(lambda: pm(rp, sp_, rm, sm, ri, si, score, gs), 20)
        out["point_matching_forward"] = _hbm(ms, 4 * P * K * K + 2 * P * K + P * K * K, shape=[P, K, K])
        expm = torch.exp(score)
        ms = _ms(lambda: pm.compute_correspondence_matrix(expm, rm, sm), 20)
        out["corr_matrix"] = _hbm(ms, 4 * P * K * K + 2 * P * K + P * K * K, shape=[P, K, K])
        # ---- pairwise_distance on fp32 MFMA: demo size (launch-bound) and a batched-size launch
        for n in (767, 8192):
            x, y = R(n, 256), R(n, 256)
            ms = _ms(lambda: ops.pairwise_distance(x, y), 20)
            out[f"pairwise_distance_{n}x{n}x256"] = _mfma(ms, 2.0 * n * n * 256)
        # ---- SuperPointMatching 767 x 767 x 256
        fr = torch.nn.functional.normalize(R(767, 256), dim=1)
        fs = torch.nn.functional.normalize(R(767, 256), dim=1)
        spm = SuperPointMatching(256)
        ms = _ms(lambda: spm(fr, fs), 20)
        out["superpoint_matching_767"] = _mfma(ms, 2.0 * 767 * 767 * 256, bytes=4 * 256 * (767 + 767) + 256 * 20)
        # ---- the same at the configs[4] shape: 64 pairs x (767 + 767) superpoints through gr_superpoint_matching_batch (ONE set of
        #      launches, grid.z = pair), and the distance kernel's own share of it (HIP events around that launch: "spm_distance")
        import ctypes
        from gaussreg_amd import _lib
        L = _lib.lib()

        def kernel_ms(name, fn, n=10):
            L.gr_timing_enable(1)
            fn()
            L.gr_timing_reset()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
            L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(cnt))
            L.gr_timing_enable(0)
            L.gr_timing_reset()
            return tot.value / max(cnt.value, 1)
        npair = 64
        fb = torch.nn.functional.normalize(R(npair * 2 * 767, 256), dim=1)
        nl = [767] * (2 * npair)
        ms = _ms(lambda: spm.forward_batch(fb, nl), 10)
        kms = kernel_ms("spm_distance", lambda: spm.forward_batch(fb, nl))
        fl = 2.0 * npair * 767 * 767 * 256
        out["superpoint_matching_batch_64x767"] = _mfma(ms, fl, pairs=npair, distance_kernel=_mfma(kms, fl),
                                                        note="whole op = mask compaction, distance + exp (fp32 MFMA, 128 x 128 tiles), row / column "
                                                             "sums, normalisation, 3-pass radix select of the top 256, sort -- per pair, grid.z = pair")
        xb, yb = R(npair, 767, 256), R(npair, 767, 256)
        ms = _ms(lambda: ops.pairwise_distance(xb, yb), 10)
        kms = kernel_ms("pairwise_distance", lambda: ops.pairwise_distance(xb, yb))
        out["pairwise_distance_batch_64x767x767x256"] = _mfma(ms, fl, distance_kernel=_mfma(kms, fl), note="(B, N, C) x (B, M, C) in one launch, grid.z = matrix")
        x8, y8 = R(8192, 256), R(8192, 256)
        kms = kernel_ms("pairwise_distance", lambda: ops.pairwise_distance(x8, y8))
        out["pairwise_distance_8192x8192x256"]["distance_kernel"] = _mfma(kms, 2.0 * 8192 * 8192 * 256)
        del xb, yb, x8, y8, fb
        # ---- KPConv: the backbone's 11 layers at their real widths on the demo pyramid, in the reference's row order (the API
        #      boundary's order) and in cell order (rows sorted by voxel key + spatially sorted input: what the network path runs on)
        kp = torch.randn(15, 3) * 0.03
        for order in ("reference", "cell"):
            if order == "reference":
                dd = d
            else:
                pp = d["points"][0]
                dd = precompute_data_stack_mode(pair_pipeline.spatial_sort(pp, torch.tensor([30000, 30000]), 0.05), torch.tensor([30000, 30000]), 5,
                                                0.025, 0.0625, [89, 30, 43, 49, 49], order="cell")
            Pl, NB, SUB = dd["points"], dd["neighbors"], dd["subsampling"]
            layers = [(0, 0, NB[0], 4, 64), (0, 0, NB[0], 32, 32), (1, 0, SUB[0], 32, 32), (1, 1, NB[1], 64, 64), (1, 1, NB[1], 64, 64),
                      (2, 1, SUB[1], 64, 64), (2, 2, NB[2], 128, 128), (2, 2, NB[2], 128, 128), (3, 2, SUB[2], 128, 128),
                      (3, 3, NB[3], 256, 256), (3, 3, NB[3], 256, 256)]
            tot, flops, gbytes = 0.0, 0.0, 0.0
            for ql, sl, nb, cin, cout in layers:
                q, s = Pl[ql], Pl[sl]
                conv = KPConv(cin, cout, 15, 0.0625 * 2 ** sl, 0.05 * 2 ** sl, kernel_points=kp * 2 ** sl).to(dev)
                f = torch.relu(torch.randn(s.shape[0], cin, device=dev, generator=g))
                tot += _ms(lambda: conv(f, q, s, nb), 5, 1)
                Mq, Hn = q.shape[0], nb.shape[1]
                flops += 2.0 * Mq * 15 * cin * cout + 2.0 * Mq * Hn * 15 * cin
                gbytes += Mq * Hn * (8 + 4 * cin) + 4 * Mq * cout
            name = "kpconv_backbone_11_layers" + ("" if order == "reference" else "_cell_order")
            out[name] = {"ms": round(tot, 3), "flops": flops, "gather_bytes": int(gbytes), "bound": "hbm (gather)", "row_order": order,
                         "TFLOP/s": round(flops / tot / 1e9, 2), "frac": round(gbytes / tot / 1e6 / HBM_GBS, 4)}
        # ---- log-Sinkhorn 256 x 128 x 128, 100 iterations (LDS resident: 2 passes over HBM)
        ot = LearnableLogOptimalTransport(100).to(dev)
        sc = R(P, K, K)
        ms = _ms(lambda: ot(sc, rm, sm), 10)
        out["sinkhorn_256x128x128_100it"] = _hbm(ms, 4 * P * K * K + 4 * P * (K + 1) * (K + 1), lds_passes=200)
        # ---- geometric structure embedding N = 767: function tables (default; bound by the 4 N^2 C bytes written) and the
        #      sinusoid -> matrix-core GEMM it replaced (308 GFLOP per cloud)
        gse = GeometricStructureEmbedding(256, 0.2, 15, 3).to(dev)
        pcs = pc[None].contiguous()
        ms = _ms(lambda: gse(pcs), 5, 1)
        n = pcs.shape[1]
        out["geo_embedding"] = _hbm(ms, 4.0 * n * n * 256, shape=[n, 256, 3],
                                    note="function tables + 4-point interpolation (gr_geo_embedding_table); bytes = the (N, N, C) result")
        gse_g = GeometricStructureEmbedding(256, 0.2, 15, 3, mode="gemm").to(dev)
        ms = _ms(lambda: gse_g(pcs), 5, 1)
        # the GEMM kernel runs SIX bf16 MFMAs per fp32 product on the bf16 matrix pipe (dense peak 2 500 TFLOP/s): `frac` is
        # quoted against the pipe it uses; the fp32-equivalent rate is a second figure, not a fraction of the fp32 pipe
        fe = 2.0 * n * n * 256 * 256 * 4
        out["geo_embedding_gemm"] = {"ms": round(ms, 4), "flops_fp32_equivalent": fe, "flops_bf16_executed": 6.0 * fe, "bound": "mfma_bf16",
                                     "TFLOP/s_bf16_executed": round(6.0 * fe / ms / 1e9, 2), "peak_TFLOP/s": 2500.0,
                                     "frac": round(6.0 * fe / ms / 1e9 / 2500.0, 4), "TFLOP/s_fp32_equivalent": round(fe / ms / 1e9, 2),
                                     "shape": [n, 256, 3], "note": "mode='gemm': split-bf16 x6 MFMA; bound by LDS operand traffic, not the matrix pipe"}
        del gse_g
        # ---- fused RPE attention (everything after the projections in one kernel; the embedding is the only N*M*C stream)
        from gaussreg_amd import _lib
        from gaussreg_amd.rpe_attention import RPEMultiHeadAttention
        import ctypes
        emb = gse(pcs)
        att = RPEMultiHeadAttention(256, 4).to(dev)
        xin = R(1, n, 256)
        ms_mod = _ms(lambda: att(xin, xin, xin, emb), 10, 2)
        Lh = _lib.lib()
        Lh.gr_timing_reset()
        Lh.gr_timing_enable(1)
        for _ in range(5):
            att(xin, xin, xin, emb)
        torch.cuda.synchronize()
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        Lh.gr_timing_read(b"rpe_attention", ctypes.byref(tot), ctypes.byref(cnt))
        Lh.gr_timing_enable(0)
        Lh.gr_timing_reset()
        out["rpe_attention_fused"] = _hbm(tot.value / max(cnt.value, 1), 4.0 * n * n * 256 + 4.0 * 4 * n * n + 3 * 4.0 * n * 256,
                                          module_ms_incl_projections=round(ms_mod, 4), shape=[n, 256, 4])
        del emb
        # ---- the whole network (gaussreg_amd.model.GeoTransformer, random weights) on the 2 x 30 000-pt demo pyramid
        from gaussreg_amd.model import GeoTransformer, make_cfg
        torch.manual_seed(0)
        net = GeoTransformer(make_cfg()).to(dev).eval()
        dd = dict(d)
        dd["features"] = torch.rand(d["points"][0].shape[0], 4, device=dev, generator=g)
        ms = _ms(lambda: net(dd), 5, 2)
        out["model_forward_2x30k"] = {"ms": round(ms, 3), "superpoints": [int(x) for x in d["lengths"][-1].tolist()],
                                      "note": "pyramid given; KPConvFPN + GeometricTransformer + matching + Sinkhorn + LGR + RANSAC"}
        del net, dd
        # ---- FPS 200 k -> 30 k, two clouds per call
        r_, s_, _ = pair_pipeline.synthetic_room_pair(0, 200000, dev)
        big = torch.cat([r_, s_]).contiguous()
        ms = _ms(lambda: farthest_point_sampling(big, [200000, 200000], [30000, 30000]), 3, 1)
        out["fps_2x200k_to_30k"] = {"ms": round(ms, 3), "ms_per_cloud": round(ms / 2, 3)}
        # ---- LocalGlobalRegistration (256 patches, 5 refinement steps)
        lgr = LocalGlobalRegistration(3, 0.1, True, 0.05, False, False, 3, None, 5)
        m2 = ot(sc, rm, sm)[:, :-1, :-1]
        rp2 = R(P, K, 3)
        sp2 = rp2 + 0.01 * R(P, K, 3)
        ms = _ms(lambda: lgr(rp2, sp2, rm, sm, m2, gs), 10)
        out["local_global_registration"] = {"ms": round(ms, 4)}
        del rp2, sp2, m2, sc
        # ---- configs[3]: gs_fusion.py render + fuse -- two 1.5 M-Gaussian scenes fused on the .ply wire format
        #      (gs_fusion.py:231-262), the ~3 M-Gaussian result rendered at 1920 x 1080, 8 cameras per call
        out.update(config3(dev))
    return out


def config3(dev, P=2_550_000, W=1920, H=1080, V=8):
    """P per input scene: the nearest-centre filter of gaussian_fuse keeps ~59 % of two overlapping synthetic scenes, so
    2 x 2.55 M inputs give the ~3 M-Gaussian fused scene configs[3] names."""
    import ctypes
    import numpy as np
    from gaussreg_amd import _lib, synthetic
    from gaussreg_amd.gs_io import gaussian_fuse_records, split_records
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings, ViewBatch, rasterize_views

    def records(seed):
        g = synthetic.gaussians_c2(P, seed=seed, sh_degree=3)
        rec = np.zeros((P, 62), np.float32)
        rec[:, 0:3] = g["means3D"]
        rec[:, 3:6] = 1.0                                                      # input normals: the fused file carries zeros
        rec[:, 6:9] = g["shs"][:, 0, :]
        rec[:, 9:54] = g["shs"][:, 1:, :].transpose(0, 2, 1).reshape(P, 45)   # f_rest: channel-major (gs_fusion.py:180)
        rec[:, 54] = np.log(g["opacities"][:, 0] / (1 - g["opacities"][:, 0]))
        rec[:, 55:58] = np.log(g["scales"])
        rec[:, 58:62] = g["rotations"]
        return torch.from_numpy(rec).to(dev)

    out = {}
    r1, r2 = records(0), records(1)
    c, s = np.cos(0.3), np.sin(0.3)
    T = np.eye(4)
    T[:3, :3] = 1.1 * np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
    T[:3, 3] = [0.2, -0.1, 0.05]
    ms = _ms(lambda: gaussian_fuse_records(r1, r2, T), 5, 2)
    fused = gaussian_fuse_records(r1, r2, T)
    kept = int(fused.shape[0])
    # algorithmic bytes: both inputs read once (248 B per record), the kept records written once
    out["gs_fuse_2x2p55M"] = _hbm(ms, 248.0 * (2 * P + kept), kept=kept, inputs=[P, P])
    del r1, r2
    parts = split_records(fused)
    del fused
    cams = synthetic.camera_ring(V, W, H, seed=0)
    st = ViewBatch([GaussianRasterizationSettings(H, W, cm["tanfovx"], cm["tanfovy"], torch.zeros(3), 1.0,
                                                  torch.from_numpy(cm["viewmatrix"]), torch.from_numpy(cm["projmatrix"]), 3,
                                                  torch.from_numpy(cm["campos"]), False, False) for cm in cams])
    last = {}

    def render():
        last["r"] = rasterize_views(st, parts["means3D"], parts["opacities"], shs=parts["shs"], scales=parts["scales"],
                                    rotations=parts["rotations"])

    L = _lib.lib()
    render()
    L.gr_timing_enable(1)
    L.gr_timing_reset()
    n = 5
    torch.cuda.synchronize()
    L.gr_timing_reset()
    import time
    t0 = time.perf_counter()
    for _ in range(n):
        render()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    k = {}
    for name in ("raster_preprocess", "raster_depth_sort", "raster_bin", "raster_blend"):
        tot, cnt = ctypes.c_double(0), ctypes.c_int64(0)
        L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(cnt))
        k[name] = round(tot.value / n, 4)
    L.gr_timing_enable(0)
    L.gr_timing_reset()
    R = float(sum(last["r"][2]))
    blend_bytes = R * 40.0 + 12.0 * H * W * V
    out["raster_3M_1080p"] = {"ms": round(ms, 3), "views_per_call": V, "views_per_s": round(V / ms * 1e3, 1), "gaussians": kept,
                              "instances_per_view": round(R / V, 1), "kernels_ms_per_call": k, "blend_bytes": blend_bytes,
                              "bound": "hbm", "frac": round(blend_bytes / (k["raster_blend"] / 1e3) / 1e9 / HBM_GBS, 4),
                              "note": "configs[3] stand-in: the fused scene of gs_fuse_2x2p55M, bit-exact blend mode"}
    return out


if __name__ == "__main__":
    import json
    print(json.dumps(run(torch.device("cuda", 0)), indent=1))
