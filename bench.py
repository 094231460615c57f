#!/usr/bin/env python3
"""bench.py -- the driver contract.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline metric (BASELINE.json): GS views/s at 1 M Gaussians, 640x480 (configs[1]) -- a "step" is one
pass of the rasterizer forward over one batch of `--views` cameras of the same synthetic 1 M-Gaussian
scene (inputs resident in HBM before the timed region; cameras are 200-byte host structs).
The second half of the metric, radius_neighbors Mpts/s on 200 k-point clouds, is measured the same
way (its own warmup + timed steps) and reported under "radius_neighbors" in the same JSON line.

Multi-GPU: one process per GPU, views / clouds are sharded per rank (independent units, no data-path
collective), barrier + synchronize on both sides of the timed region, MAX over ranks of the elapsed
time (RCCL all_reduce), value = total units over all ranks / that time ("scaling": "weak").

`roofline`: dominant rasterizer kernel (the per-tile blend) -- algorithmic bytes per launch / average
launch duration measured with HIP events on the launch stream inside the timed region
(gr_timing_* hooks of the C ABI).  `cpu_baseline`: the oracle (kind "port") timed on one host core
on a bounded sample of the same workload, rank 0 at N=1 only.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--views", type=int, default=32, help="cameras per step (per GPU)")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--clouds", type=int, default=8, help="200k-point clouds per radius step (per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-radius", action="store_true")
    return ap.parse_args()


def pmc_traffic(key, ok, units=None):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (separate FETCH_SIZE / WRITE_SIZE
    passes, FETCH doubled per MI355X_MICROARCH.md); only valid for the configuration it was taken on
    (`units` = views or clouds per launch must match what the summary records)."""
    if not ok:
        return None
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_hbm.json"))
        with open(os.path.join(ROOT, "profiles", cands[-1])) as fh:
            doc = json.load(fh)
        if units is not None and doc.get("units_per_launch", {}).get(key) != units:
            return None
        return doc["traffic_bytes_per_launch"].get(key)
    except Exception:
        return None


def timing_read(L, name):
    tot = ctypes.c_double(0)
    n = ctypes.c_int64(0)
    L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(n))
    return tot.value, n.value


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend="nccl", rank=rank, world_size=world)  # "nccl" is RCCL on ROCm

    from gaussreg_amd import _lib, ext, synthetic
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings, ViewBatch, rasterize_views
    L = _lib.lib()

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        barrier()
        return max_over_ranks(time.perf_counter() - t0)

    # ------------------------------------------------------------------ rasterizer (headline)
    P, W, H, V = args.gaussians, args.width, args.height, args.views
    g = synthetic.gaussians_c2(P, seed=0, sh_degree=3)             # same scene on every rank
    cams = synthetic.camera_ring(V, W, H, seed=rank)               # different cameras per rank
    t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    settings = [GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0,
                                              torch.from_numpy(c["viewmatrix"]), torch.from_numpy(c["projmatrix"]), 3,
                                              torch.from_numpy(c["campos"]), False, False) for c in cams]
    settings = ViewBatch(settings)  # cameras marshalled once, like the other inputs
    last = {}

    def raster_step():
        img, radii, nr = rasterize_views(settings, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                         rotations=t["rotations"])
        last["nr"] = nr
        last["img"] = img

    raster_step()  # allocate / page in before anything is timed
    L.gr_timing_reset()
    L.gr_timing_enable(1)
    for _ in range(args.warmup):
        raster_step()
    barrier()
    L.gr_timing_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        raster_step()
    barrier()
    elapsed = max_over_ranks(time.perf_counter() - t0)
    blend_ms, blend_n = timing_read(L, "raster_blend")
    sort_ms, sort_n = timing_read(L, "raster_sort")
    pre_ms, pre_n = timing_read(L, "raster_preprocess")
    L.gr_timing_enable(0)
    L.gr_timing_reset()
    views_per_s = world * V * args.steps / elapsed
    R_total = float(sum(last["nr"]))
    # blend: algorithmic bytes per launch: per instance id 4 + xy 8 + conic/opacity 16 + rgb 12,
    # plus the image write 12*H*W per view
    blend_bytes = R_total * (4 + 8 + 16 + 12) + 12.0 * H * W * V
    blend_avg_s = blend_ms / max(blend_n, 1) / 1e3
    roofline = {"kernel": "raster_blend", "bound": "hbm",
                "achieved": round(blend_bytes / blend_avg_s / 1e9, 2) if blend_avg_s > 0 else None,
                "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(blend_bytes / blend_avg_s / 1e9 / HBM_PEAK_GBS, 4) if blend_avg_s > 0 else None,
                "traffic": pmc_traffic("raster_blend", (P, W, H) == (1_000_000, 640, 480), V),
                "bytes_per_launch": blend_bytes, "avg_launch_ms": round(blend_avg_s * 1e3, 4),
                "other_kernels_ms": {"raster_preprocess": round(pre_ms / max(pre_n, 1), 4),
                                     "raster_sort": round(sort_ms / max(sort_n, 1), 4)}}

    # ------------------------------------------------------------------ radius_neighbors (2nd half of the metric)
    radius = None
    if not args.no_radius:
        B = args.clouds
        pts, lens = synthetic.cloud_200k(B, seed=rank)
        dpts = pts.to(dev)
        out = {}

        def radius_step():
            out["nb"] = ext.radius_neighbors(dpts, dpts, lens, lens, 0.0625)

        radius_step()
        L.gr_timing_enable(1)
        for _ in range(args.warmup):
            radius_step()
        barrier()
        L.gr_timing_reset()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            radius_step()
        barrier()
        r_elapsed = max_over_ranks(time.perf_counter() - t0)
        fill_ms, fill_n = timing_read(L, "radius_fill")
        cnt_ms, cnt_n = timing_read(L, "radius_count")
        L.gr_timing_enable(0)
        L.gr_timing_reset()
        nq = dpts.shape[0]
        width = out["nb"].shape[1]
        fill_bytes = 12.0 * nq + 12.0 * nq + 8.0 * nq * width      # 12 Nq + 12 Ns + 8 Nq W
        fill_avg_s = fill_ms / max(fill_n, 1) / 1e3
        radius = {"metric": "radius_neighbors throughput, 200k-pt clouds", "value": round(world * nq * args.steps / r_elapsed / 1e6, 2),
                  "unit": "Mpts/s", "ms_per_step": round(r_elapsed / args.steps * 1e3, 4),
                  "config": {"workload": f"{B} x 200k-pt clouds per GPU per step, r=0.0625, self-search, width {width}"},
                  "roofline": {"kernel": "radius_fill", "bound": "hbm",
                               "achieved": round(fill_bytes / fill_avg_s / 1e9, 2) if fill_avg_s > 0 else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(fill_bytes / fill_avg_s / 1e9 / HBM_PEAK_GBS, 4) if fill_avg_s > 0 else None,
                               "traffic": pmc_traffic("radius_fill", True, B), "bytes_per_launch": fill_bytes,
                               "avg_launch_ms": round(fill_avg_s * 1e3, 4),
                               "other_kernels_ms": {"radius_count": round(cnt_ms / max(cnt_n, 1), 4)}}}

    # ------------------------------------------------------------------ CPU baseline (rank 0, N=1 only)
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import capi
        c = cams[0]
        tc = time.perf_counter()
        capi.rasterize_forward(g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
                               viewmatrix=c["viewmatrix"], projmatrix=c["projmatrix"], campos=c["campos"],
                               bg=np.zeros(3, np.float32), W=W, H=H, tanfovx=c["tanfovx"], tanfovy=c["tanfovy"],
                               sh_degree=3)
        dt = time.perf_counter() - tc
        cpu_baseline = {"value": round(1.0 / dt, 4), "unit": "views/s", "cores": 1, "kind": "port",
                        "sample": f"1 view of the same {P}-Gaussian {W}x{H} scene through oracle/rasterizer_oracle.c "
                                  f"({dt:.2f} s; the rasterizer has no reference implementation in the GaussReg tree)"}
        if radius is not None:
            p1, l1 = synthetic.cloud_200k(1, seed=0)
            kind = "reference" if capi.have_ref() else "port"
            fn = capi.ref_radius_neighbors if kind == "reference" else capi.radius_neighbors
            tc = time.perf_counter()
            fn(p1.numpy(), p1.numpy(), l1.numpy(), l1.numpy(), 0.0625)
            dt = time.perf_counter() - tc
            radius["cpu_baseline"] = {"value": round(0.2 / dt, 4), "unit": "Mpts/s", "cores": 1, "kind": kind,
                                      "sample": "one 200k-pt cloud, single thread ("
                                                + ("reference C++ core compiled in oracle/_ref" if kind == "reference"
                                                   else "oracle/radius_neighbors_oracle.c") + f", {dt:.2f} s)"}

    if rank == 0:
        line = {"metric": "GS views/sec @1M pts 640x480 (+ radius_neighbors Mpts/sec, see radius_neighbors)",
                "value": round(views_per_s, 2), "unit": "views/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "config": {"workload": f"diff_gaussian_rasterization forward: {P} synthetic Gaussians (SH deg 3), "
                                       f"{W}x{H}, {V} views per step per GPU (configs[1])",
                           "gaussians": P, "width": W, "height": H, "views_per_step": V,
                           "instances_per_view": round(R_total / V, 1), "parallelism": f"per-view sharding x{world}"},
                "roofline": roofline, "cpu_baseline": cpu_baseline, "radius_neighbors": radius}
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
