"""One operator per process, for the FETCH_SIZE / WRITE_SIZE counter passes of the operators bench.py's `extras` prices against the
HBM roofline (tools/pmc_ops.sh runs it under rocprofv3 --pmc, tools/pmc_ops_summary.py sums the counters of ALL kernels of the
process and divides by the number of operator calls printed here).

    python tools/pmc_ops.py <op> [calls]
    ops: grid_ref_64x200k grid_cell_64x200k grid_ref_1x200k kpconv_2_2_x32 gs_fuse_2x2p55M fps_2x200k radius_8x200k radius_limited_8x200k

Prints one JSON line: {"op", "calls", "algorithmic_bytes_per_call", ...}.  Every call of the process is counted (no separate
warm-up: the first call's traffic is the same traffic)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)


def main():
    op = sys.argv[1]
    calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    from gaussreg_amd import ext, pair_pipeline, synthetic
    dev = torch.device("cuda", 0)
    info = {"op": op, "calls": calls}
    with torch.no_grad():
        if op.startswith("grid_"):
            order = "reference" if "_ref_" in op else "cell"
            B = 64 if "64x" in op else 1
            pts, lens = synthetic.cloud_200k(B, seed=1 if B > 1 else 0)
            dp = pts.to(dev)
            torch.cuda.synchronize()
            for _ in range(calls):
                sp, _ = ext.grid_subsampling(dp, lens, 0.05, order=order)
            info["algorithmic_bytes_per_call"] = 12 * dp.shape[0] + 12 * sp.shape[0]
            info["rows_in"], info["rows_out"] = int(dp.shape[0]), int(sp.shape[0])
        elif op == "radius_8x200k":
            pts, lens = synthetic.cloud_200k(8, seed=0)
            dp = pts.to(dev)
            for _ in range(calls):
                nb = ext.radius_neighbors(dp, dp, lens, lens, 0.0625)
            info["algorithmic_bytes_per_call"] = 24 * dp.shape[0] + 8 * dp.shape[0] * nb.shape[1]
            info["width"] = int(nb.shape[1])
        elif op == "radius_limited_8x200k":
            from gaussreg_amd.ops import radius_search
            pts, lens = synthetic.cloud_200k(8, seed=0)
            dp = pts.to(dev)
            for _ in range(calls):
                nb = radius_search(dp, dp, lens, lens, 0.0625, 40)
            info["algorithmic_bytes_per_call"] = 24 * dp.shape[0] + 8 * dp.shape[0] * nb.shape[1]
            info["width"] = int(nb.shape[1])
        elif op == "kpconv_2_2_x32":
            # encoder2_2 (64 -> 64 channels, level 1) over a batch of 32 pairs: the shape of one KPConv call of register_many
            from gaussreg_amd.data import precompute_data_stack_mode
            from gaussreg_amd.kpconv import KPConv
            Bp = 32
            clouds = [pair_pipeline.synthetic_room_pair(b, 30000, dev)[0] for b in range(Bp)] + \
                     [pair_pipeline.synthetic_room_pair(b, 30000, dev)[1] for b in range(Bp)]
            bp = torch.cat(clouds).contiguous()
            d = precompute_data_stack_mode(bp, torch.tensor([30000] * (2 * Bp)), 5, 0.025, 0.0625, [89, 30, 43, 49, 49])
            q = d["points"][1]
            nb = d["neighbors"][1]
            g = torch.Generator(device=dev).manual_seed(0)
            conv = KPConv(64, 64, 15, 0.125, 0.1, kernel_points=torch.randn(15, 3) * 0.06).to(dev)
            f = torch.relu(torch.randn(q.shape[0], 64, device=dev, generator=g))
            torch.cuda.synchronize()
            info["note"] = "counters include the pyramid build of the batch (one call, before the KPConv calls): use the per-kernel rows"
            for _ in range(calls):
                out = conv(f, q, q, nb)
            M, H = q.shape[0], nb.shape[1]
            info["algorithmic_bytes_per_call"] = M * H * (8 + 4 * 64) + 4 * M * 64
            info["points"], info["neighbors"] = int(M), int(H)
            info["intermediate_bytes_if_materialised"] = 4 * M * 15 * 64
            del out
        elif op == "gs_fuse_2x2p55M":
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from gaussreg_amd.gs_io import gaussian_fuse_records
            P = 2_550_000

            def records(seed):
                g = synthetic.gaussians_c2(P, seed=seed, sh_degree=3)
                rec = np.zeros((P, 62), np.float32)
                rec[:, 0:3] = g["means3D"]
                rec[:, 3:6] = 1.0
                rec[:, 6:9] = g["shs"][:, 0, :]
                rec[:, 9:54] = g["shs"][:, 1:, :].transpose(0, 2, 1).reshape(P, 45)
                rec[:, 54] = np.log(g["opacities"][:, 0] / (1 - g["opacities"][:, 0]))
                rec[:, 55:58] = np.log(g["scales"])
                rec[:, 58:62] = g["rotations"]
                return torch.from_numpy(rec).to(dev)
            r1, r2 = records(0), records(1)
            c, s = np.cos(0.3), np.sin(0.3)
            T = np.eye(4)
            T[:3, :3] = 1.1 * np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])
            T[:3, 3] = [0.2, -0.1, 0.05]
            torch.cuda.synchronize()
            for _ in range(calls):
                fused = gaussian_fuse_records(r1, r2, T)
            info["algorithmic_bytes_per_call"] = 248 * (2 * P + int(fused.shape[0]))
            info["kept"] = int(fused.shape[0])
        elif op == "fps_2x200k":
            from gaussreg_amd.registration import farthest_point_sampling
            r_, s_, _ = pair_pipeline.synthetic_room_pair(0, 200000, dev)
            big = torch.cat([r_, s_]).contiguous()
            torch.cuda.synchronize()
            for _ in range(calls):
                farthest_point_sampling(big, [200000, 200000], [30000, 30000])
            # exact FPS reads every point once per accepted sample in its definition (12 N S bytes); the bucketed form's
            # lower bound is not known in closed form -- the ratio is reported against 12 N + 4 S (inputs once, indices once)
            info["algorithmic_bytes_per_call"] = 2 * (12 * 200000 + 4 * 30000)
        else:
            raise SystemExit("unknown op " + op)
    torch.cuda.synchronize()
    print("PMC_OP " + json.dumps(info), flush=True)


if __name__ == "__main__":
    main()
