#!/usr/bin/env python3
"""bench.py -- the driver contract.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no torchrun environment re-launches itself under torch.distributed.run (one rank per GPU, RCCL);
it refuses to run if the box has fewer than N GPUs -- it never reports fewer ranks than asked for.

Headline metric (BASELINE.json): GS views/s at 1 M Gaussians, 640x480 (configs[1]) -- a "step" is one pass of the
rasterizer forward over one batch of `--views` cameras of the same synthetic 1 M-Gaussian scene (inputs resident in HBM
before the timed region; cameras are 200-byte host structs), through the DEFAULT path of the wrapper: every call ordered on
the caller's stream, like upstream's.  The opt-in `static_scene=True` mode (two calls in flight on internal streams,
gaussreg_amd/rasterizer.py _FramePipe) is timed separately and reported as `config.static_scene_views_per_s`.
The driver's record keeps the scalars of `config`, `roofline` and `cpu_baseline` and the tail of the line: the second half of
the metric (radius_neighbors Mpts/s and its roofline fractions) and the one-camera figures are therefore ALSO scalars there
(`roofline.radius_*`, `config.single_view_*`), and the line ends with a short `summary`.
The same JSON line also carries, each timed the same way
(own warmup, barrier + synchronize on both sides, MAX over ranks):
    "single_view"       views/s through diff_gaussian_rasterization.GaussianRasterizer.forward, ONE camera per call
                        (the drop-in boundary number)
    "radius_neighbors"  the second half of the metric: Mpts/s on 200 k-point clouds (bare ext.radius_neighbors), plus the
                        limited-width radius_search path the data pyramid uses
    "pairs"             configs[4]: batch coarse registration, `--pairs` synthetic scene pairs PER GPU (weak scaling; 128 per
                        GPU = 1024 over 8), per-rank batches through FPS -> pyramid -> matching ops -> LGR -> RANSAC,
                        4x4 transforms + errors gathered with ONE all_gather (gaussreg_amd/sharding.py)
    "extras"            the other SURVEY 8(d) kernels: {ms, algorithmic bytes or flops, fraction of the bound}
    "cpu_baseline"      the oracle / the compiled reference core on one host core, bounded samples, rank 0 at N=1 only

Multi-GPU: units (views, clouds, pairs) are sharded per rank with no data-path collective; value = units of all ranks /
MAX-over-ranks elapsed ("scaling": "weak").
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

try:
    _INITIAL_AFFINITY = sorted(os.sched_getaffinity(0))  # before any rank binding: the CPU-baseline workers get this mask back
except AttributeError:
    _INITIAL_AFFINITY = None

HBM_PEAK_GBS = 8000.0   # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s peak, ~6.3 TB/s achievable)
FP32_MFMA_PEAK_TF = 157.3


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)   # (a 120 ms timed region: the round-4 review found 57 ms thin)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--views", type=int, default=32, help="cameras per step (per GPU)")
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--clouds", type=int, default=8, help="200k-point clouds per radius step (per GPU)")
    ap.add_argument("--pairs", type=int, default=128, help="scene pairs per GPU in the configs[4] workload (0 = skip)")
    ap.add_argument("--pair-batch", type=int, default=64, help="pairs per pyramid call")
    ap.add_argument("--pair-points", type=int, default=200_000, help="points per cloud before FPS")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-radius", action="store_true")
    ap.add_argument("--no-radius-limited", action="store_true", help="skip the width-limited radius_search timing")
    ap.add_argument("--no-single-view", action="store_true")
    ap.add_argument("--no-static-scene", action="store_true",
                    help="skip the second, static_scene=True pass over the headline workload (tools/prof_all.sh: the rocprofv3 "
                         "averages of that command are then those of the default path alone)")
    ap.add_argument("--no-extras", action="store_true")
    ap.add_argument("--dry", action="store_true",
                    help="first contact with a multi-GPU node: initialise the process group, run the collectives the bench uses "
                         "(barrier, MAX / SUM all_reduce, sharding.gather_rows on a (3, 20) and on an empty device tensor), print "
                         "one JSON line and exit -- no kernels, seconds per rank (tools/first_8gpu_run.md)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend; gloo exists for the CPU suite's test of --dry (same code path, CPU tensors)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# process group / timing harness (device-agnostic: the CPU test drives it over gloo with stub workloads)
class Harness:
    def __init__(self, rank, world, device, sync=None, affinity=None):
        self.rank, self.world, self.device = rank, world, device
        self._sync = sync if sync is not None else (lambda: None)
        self.affinity = affinity  # where this rank's host thread was pinned (gaussreg_amd/affinity.py), None = not bound

    @staticmethod
    def from_env(backend="nccl"):
        import torch
        import torch.distributed as dist
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # one process per GPU: pin this one to the CPUs of its GPU's NUMA node BEFORE the library maps its pinned pages (the
        # mailbox page a rank spin-polls, the staging buffers) so that first touch puts them there.  Single-rank runs keep the
        # whole machine (the CPU-baseline leg uses every core).
        aff = None
        if world > 1:
            from gaussreg_amd import affinity
            aff = affinity.bind_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        if backend == "nccl":
            if not torch.cuda.is_available():
                raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for the product path)")
            if local_rank >= torch.cuda.device_count():
                raise SystemExit(f"rank {rank}: local GPU {local_rank} does not exist ({torch.cuda.device_count()} visible)")
            torch.cuda.set_device(local_rank)
            device = torch.device("cuda", local_rank)
            sync = lambda: torch.cuda.synchronize(device)  # noqa: E731
        else:
            device, sync = torch.device("cpu"), None
        if world > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group(backend=backend, rank=rank, world_size=world)  # "nccl" is RCCL on ROCm
        return Harness(rank, world, device, sync, aff)

    def barrier(self):
        import torch.distributed as dist
        self._sync()
        if self.world > 1:
            dist.barrier()
        self._sync()

    def max_over_ranks(self, x):
        if self.world == 1:
            return x
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, x):
        if self.world == 1:
            return x
        import torch
        import torch.distributed as dist
        t = torch.tensor([x], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    def timed(self, fn, steps, warmup, after_warmup=None):
        """W untimed steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; MAX over ranks."""
        for _ in range(warmup):
            fn()
        self.barrier()
        if after_warmup is not None:
            after_warmup()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def timed3(self, fn, steps, warmup, after_warmup=None, repeats=3):
        """`timed` three times over (sections whose timed region is under ~100 ms): (median, {min, max, spread}) -- spread =
        (max - min) / median of the elapsed times, so a line read once is not one unlucky sample."""
        ts = sorted(self.timed(fn, steps, warmup if i == 0 else 1, after_warmup=after_warmup) for i in range(repeats))
        med = ts[len(ts) // 2]
        return med, {"min_s": ts[0], "max_s": ts[-1], "spread": round((ts[-1] - ts[0]) / med, 4) if med > 0 else None}

    def close(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.destroy_process_group()


def throughput_line(h, metric, unit, units_per_step_per_rank, steps, warmup, elapsed, extra=None):
    """The contract's whole-job aggregate: units of ALL ranks / MAX-over-ranks time."""
    line = {"metric": metric, "value": round(h.world * units_per_step_per_rank * steps / elapsed, 2), "unit": unit,
            "n_gpus": h.world, "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None}
    if extra:
        line.update(extra)
    return line


def self_spawn(args, argv):
    """`python bench.py --gpus N` outside torchrun: become `python -m torch.distributed.run ... bench.py --gpus N`."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {have} GPU(s) visible on this box; refusing to report a "
                         f"{args.gpus}-GPU number from fewer devices")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def dry_run(h, backend):
    """`--dry`: every collective of the bench on tiny tensors, checked on every rank.  The only thing `backend` changes is
    where the tensors live (Harness.from_env)."""
    import torch
    from gaussreg_amd import sharding
    rank, world, dev = h.rank, h.world, h.device
    t0 = time.perf_counter()
    h.barrier()
    checks = {}
    checks["max_over_ranks"] = h.max_over_ranks(float(rank)) == float(world - 1)
    checks["sum_over_ranks"] = h.sum_over_ranks(float(rank + 1)) == world * (world + 1) / 2.0
    # the configs[4] gather: (3, 20) rows per rank, row value = 100 * rank + row
    local = (torch.arange(3, dtype=torch.float32, device=dev).reshape(3, 1) + 100.0 * rank).expand(3, 20).contiguous()
    rows = sharding.gather_rows(local, [3] * world)
    want = torch.cat([torch.arange(3, dtype=torch.float32) + 100.0 * r for r in range(world)])
    checks["gather_rows_3x20"] = tuple(rows.shape) == (3 * world, 20) and rows.device == local.device and \
        torch.equal(rows[:, 0].cpu(), want) and torch.equal(rows[:, 19].cpu(), want)
    # an idle rank (fewer pairs than GPUs): the odd ranks contribute nothing; counts learnt by the collective itself
    part = local if rank % 2 == 0 else local[:0]
    rows2 = sharding.gather_rows(part)
    want2 = torch.cat([torch.arange(3, dtype=torch.float32) + 100.0 * r for r in range(0, world, 2)])
    checks["gather_rows_ragged_with_empty"] = tuple(rows2.shape) == (3 * ((world + 1) // 2), 20) and torch.equal(rows2[:, 0].cpu(), want2)
    empty = sharding.gather_rows(local[:0], [0] * world)
    checks["gather_rows_all_empty"] = tuple(empty.shape) == (0, 20)
    h.barrier()
    ok = all(checks.values())
    n_ok = h.sum_over_ranks(1.0 if ok else 0.0)
    # where every rank's host thread sits (numa node of its GPU, the CPUs it was pinned to)
    placement = [h.affinity]
    if world > 1:
        import torch.distributed as dist
        placement = [None] * world
        dist.all_gather_object(placement, h.affinity)
    line = {"dry": True, "backend": backend, "n_gpus": world, "device": str(dev), "ok": bool(n_ok == world), "ranks_ok": int(n_ok),
            "checks": checks, "placement": placement, "seconds": round(time.perf_counter() - t0, 3)}
    if not ok:
        print(f"rank {rank}: --dry checks failed: {checks}", file=sys.stderr, flush=True)
    return line


# ----------------------------------------------------------------------------------------------------------------------
def newest_profile(suffix):
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix))
        return cands[-1] if cands else None
    except Exception:
        return None


def pmc_traffic(key, ok, units=None):
    """HBM bytes per launch from the newest COMMITTED rocprofv3 --pmc summary (profiles/*_pmc_hbm.json: separate FETCH_SIZE /
    WRITE_SIZE passes, FETCH doubled per MI355X_MICROARCH.md) -- not measured in this run; only valid for the configuration
    it was taken on.  The file it came from is reported next to it (`traffic_source`)."""
    if not ok:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", newest_profile("_pmc_hbm.json"))) as fh:
            doc = json.load(fh)
        if units is not None and doc.get("units_per_launch", {}).get(key) != units:
            return None
        return doc["traffic_bytes_per_launch"].get(key)
    except Exception:
        return None


def sq_valu_busy(kernel_substr, ok):
    """VALU-issue time / kernel duration of a kernel, from the newest committed SQ-counter summary (profiles/*_sq_counters
    .json: SQ_INSTS_VALU x 4 cycles / 1024 SIMDs / 2.4 GHz over the profiled duration) -- what actually bounds the kernels the
    HBM roofline above is quoted for.  None when the summary is missing or was taken on another configuration."""
    if not ok:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", newest_profile("_sq_counters.json"))) as fh:
            doc = json.load(fh)
        for k in doc["kernels"]:
            if kernel_substr in k["kernel"]:
                return round(k["valu_issue_us_at_2p4GHz"] / k["profiled_duration_us_avg"], 3)
    except Exception:
        pass
    return None


def sq_insts(kernel_substr, counter):
    """A raw per-launch SQ counter of a kernel from the newest committed SQ-counter summary (None if absent)."""
    try:
        with open(os.path.join(ROOT, "profiles", newest_profile("_sq_counters.json"))) as fh:
            doc = json.load(fh)
        for k in doc["kernels"]:
            if kernel_substr in k["kernel"]:
                return float(k["per_launch"][counter])
    except Exception:
        pass
    return None


def cpu_identity():
    """CPU model and core counts of this host ("state the core count and CPU model")."""
    model, phys = None, set()
    try:
        pid = cid = None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("model name") and model is None:
                    model = ln.split(":", 1)[1].strip()
                elif ln.startswith("physical id"):
                    pid = ln.split(":", 1)[1].strip()
                elif ln.startswith("core id"):
                    cid = ln.split(":", 1)[1].strip()
                elif not ln.strip():
                    if pid is not None and cid is not None:
                        phys.add((pid, cid))
                    pid = cid = None
    except Exception:
        pass
    try:
        usable = len(os.sched_getaffinity(0))
    except Exception:
        usable = os.cpu_count() or 1
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "physical_cores": len(phys) or None, "cores_available": usable}


def all_cores(task, units_per_task, unit, workers, tasks_per_worker=1, lead_s=12.0):
    """`workers` PROCESSES (tools/cpu_worker.py), each `tasks_per_worker` units of `task`, started together (the reference's
    model: one DataLoader worker process per core, config.py:40 num_workers).  Returns units/s over the common interval."""
    start = time.time() + lead_s   # the workers import numpy / torch first and then wait for this instant
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "cpu_worker.py"), task, str(tasks_per_worker),
                               repr(start), str(i)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True,
                              preexec_fn=(lambda: os.sched_setaffinity(0, _INITIAL_AFFINITY)) if _INITIAL_AFFINITY else None,
                              env=dict(os.environ, OMP_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES=""))
             for i in range(workers)]
    t_first, t_last, ok = None, None, 0
    for pr in procs:
        out, _ = pr.communicate()
        for ln in out.splitlines():
            if ln.startswith("DONE "):
                _, a, b = ln.split()
                t_first = float(a) if t_first is None else min(t_first, float(a))
                t_last = float(b) if t_last is None else max(t_last, float(b))
                ok += 1
    if not ok:
        return {"error": "no CPU worker finished"}
    late = max(0.0, t_first - start)
    dt = t_last - start
    return {"value": round(ok * tasks_per_worker * units_per_task / dt, 4), "unit": unit, "cores": ok, "tasks": ok * tasks_per_worker,
            "seconds": round(dt, 2), "workers_late_s": round(late, 2)}


def timing_read(L, name):
    tot = ctypes.c_double(0)
    n = ctypes.c_int64(0)
    L.gr_timing_read(name.encode(), ctypes.byref(tot), ctypes.byref(n))
    return tot.value, n.value


def hbm_roofline(kernel, bytes_per_launch, total_ms, launches, traffic=None, **more):
    avg_s = total_ms / max(launches, 1) / 1e3
    ach = bytes_per_launch / avg_s / 1e9 if avg_s > 0 else None
    r = {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 2) if ach else None, "peak": HBM_PEAK_GBS,
         "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4) if ach else None, "traffic": traffic,
         "traffic_source": ("profiles/" + str(newest_profile("_pmc_hbm.json")) + " (committed PMC summary, not this run)") if traffic else None,
         "bytes_per_launch": bytes_per_launch, "avg_launch_ms": round(avg_s * 1e3, 4)}
    r.update(more)
    return r


def per_step_ms(L, names, steps):
    """{timer name: total milliseconds per STEP} (a timer may fire several times per step)."""
    out = {}
    for n in names:
        ms, cnt = timing_read(L, n)
        if cnt:
            out[n] = round(ms / steps, 4)
    return out


def run_gpu(h, args):
    import numpy as np
    import torch
    from gaussreg_amd import _lib, ext, sharding, synthetic
    from gaussreg_amd.ops import radius_search
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, ViewBatch, rasterize_views
    L = _lib.lib()
    dev, rank, world = h.device, h.rank, h.world

    # ------------------------------------------------------------------ rasterizer (headline)
    P, W, H, V = args.gaussians, args.width, args.height, args.views
    g = synthetic.gaussians_c2(P, seed=0, sh_degree=3)             # same scene on every rank
    cams = synthetic.camera_ring(V, W, H, seed=rank)               # different cameras per rank
    t = {k: torch.from_numpy(v).to(dev) for k, v in g.items()}
    settings_list = [GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0,
                                                   torch.from_numpy(c["viewmatrix"]), torch.from_numpy(c["projmatrix"]), 3,
                                                   torch.from_numpy(c["campos"]), False, False) for c in cams]
    settings = ViewBatch(settings_list)  # cameras marshalled once, like the other inputs
    last = {}

    def raster_step():
        img, radii, nr = rasterize_views(settings, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                         rotations=t["rotations"])
        last["nr"] = nr
        last["img"] = img
        last["radii"] = radii

    raster_step()  # allocate / page in before anything is timed
    L.gr_timing_reset()
    L.gr_timing_enable(1)
    elapsed = h.timed(raster_step, args.steps, args.warmup, after_warmup=L.gr_timing_reset)
    blend_ms, blend_n = timing_read(L, "raster_blend")
    raster_kernels = per_step_ms(L, ["raster_preprocess", "raster_depth_sort", "raster_bin", "raster_sort", "raster_blend"],
                                 args.steps)
    # the opt-in static-scene mode: two calls in flight (gaussreg_amd/rasterizer.py _FramePipe); the blend then shares the chip
    # with the next step's preprocess / sort / binning
    def static_step():
        img, radii, nr = rasterize_views(settings, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                         rotations=t["rotations"], static_scene=True)
        last["img"] = img

    st_elapsed, st_ms, st_n, st_kernels = None, 0.0, 0, None
    if not args.no_static_scene:
        static_step()
        st_elapsed = h.timed(static_step, args.steps, args.warmup, after_warmup=L.gr_timing_reset)
        st_ms, st_n = timing_read(L, "raster_blend")
        st_kernels = per_step_ms(L, ["raster_preprocess", "raster_depth_sort", "raster_bin", "raster_sort", "raster_blend"],
                                 args.steps)
    L.gr_timing_enable(0)
    L.gr_timing_reset()
    R_total = float(sum(last["nr"]))
    # blend: algorithmic bytes per launch: per instance id 4 + xy 8 + conic/opacity 16 + rgb 12,
    # plus the image write 12*H*W per view
    blend_bytes = R_total * (4 + 8 + 16 + 12) + 12.0 * H * W * V
    line = throughput_line(
        h, "GS views/sec @1M pts 640x480 (+ radius_neighbors Mpts/sec, see radius_neighbors)", "views/s", V, args.steps,
        args.warmup, elapsed,
        {"dtype": "f32", "data": "synthetic",
         "config": {"workload": f"diff_gaussian_rasterization forward: {P} synthetic Gaussians (SH deg 3), {W}x{H}, "
                                f"{V} views per step per GPU (configs[1])",
                    "gaussians": P, "width": W, "height": H, "views_per_step": V,
                    "instances_per_view": round(R_total / V, 1), "parallelism": f"per-view sharding x{world}"},
         "roofline": hbm_roofline("raster_blend", blend_bytes, blend_ms, blend_n,
                                  pmc_traffic("raster_blend", (P, W, H) == (1_000_000, 640, 480), V),
                                  kernels_ms_per_step=raster_kernels,
                                  static_scene={"avg_launch_ms": round(st_ms / max(st_n, 1), 4), "kernels_ms_per_step": st_kernels,
                                                "note": "static_scene=True: two steps in flight, the blend runs next to the following "
                                                        "step's preprocess, sort and binning"},
                                  valu_busy=sq_valu_busy("blend_kernel<false", (P, W, H) == (1_000_000, 640, 480)),
                                  valu_busy_source="profiles/" + str(newest_profile("_sq_counters.json")))})
    # share of the (Gaussian, view) pairs that survive the culling: the others cost 12 bytes of preprocess output (radius,
    # depth field, rectangle -- their 64-byte record is never written) and are dropped by the depth sort's first pass
    line["config"]["visible_fraction"] = round(float((last["radii"] > 0).float().mean().item()), 4)
    for kname, kms in raster_kernels.items():  # scalars: the driver's flattened record drops nested objects
        line["roofline"]["ms_" + kname[7:]] = kms
    line["config"]["static_scene_views_per_s"] = round(world * V * args.steps / st_elapsed, 2) if st_elapsed else None
    line["config"]["static_scene_ms_per_step"] = round(st_elapsed / args.steps * 1e3, 4) if st_elapsed else None

    # the architecturally guaranteed ordering (explicit ballot ranking in the depth sort and the tile scatter instead of the
    # probed lane order of ds_add_rtn): same images, its cost in the record
    old_rank = L.gr_raster_ballot_ranking(1)
    raster_step()
    b_elapsed = h.timed(raster_step, args.steps, args.warmup)
    L.gr_raster_ballot_ranking(old_rank)
    line["config"]["ballot_ranking_views_per_s"] = round(world * V * args.steps / b_elapsed, 2)
    line["config"]["ballot_ranking_cost"] = round(b_elapsed / elapsed - 1.0, 4)
    line["config"]["lds_atomics_lane_ordered"] = int(L.gr_raster_lds_atomics_lane_ordered())

    # ------------------------------------------------------------------ the boundary: one camera per forward() call
    if not args.no_single_view:
        rast = [GaussianRasterizer(s) for s in settings_list[: min(V, 8)]]
        k = {"i": 0}

        def single_step():
            r = rast[k["i"] % len(rast)]
            k["i"] += 1
            last["sv"] = r(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])

        single_step()
        n_sv = max(args.steps, 200)
        # throughput with the per-kernel event timers OFF (ten event records per frame are a measurable share of a 0.2 ms
        # frame); the per-kernel figures come from a second, instrumented pass of the same loop
        L.gr_timing_enable(0)
        sv_elapsed, sv_spread = h.timed3(single_step, n_sv, max(args.warmup, 3))
        L.gr_timing_enable(1)
        L.gr_timing_reset()
        h.timed(single_step, n_sv, 1, after_warmup=L.gr_timing_reset)
        sv_kernels = per_step_ms(L, ["raster_preprocess", "raster_depth_sort", "raster_bin", "raster_sort", "raster_blend"], n_sv)
        L.gr_timing_enable(0)
        L.gr_timing_reset()
        line["single_view"] = {"value": round(world * n_sv / sv_elapsed, 2), "unit": "views/s",
                               "ms_per_view": round(sv_elapsed / n_sv * 1e3, 4), "frames": n_sv,
                               "api": "diff_gaussian_rasterization.GaussianRasterizer.forward, one camera per call, one frame at a "
                                      "time on the caller's stream (the default), same 1M-Gaussian scene",
                               "kernels_ms_per_view": sv_kernels}
        # opt-in: GaussianRasterizer(settings, static_scene=True) keeps two one-camera frames in flight; the drain at the closing
        # synchronise is one un-overlapped blend (200 frames: < 1 %)
        rast_s = [GaussianRasterizer(s, static_scene=True) for s in settings_list[: min(V, 8)]]

        def single_static():
            r = rast_s[k["i"] % len(rast_s)]
            k["i"] += 1
            last["sv"] = r(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])

        single_static()
        svs, svs_spread = h.timed3(single_static, n_sv, max(args.warmup, 3))
        line["single_view"]["static_scene"] = {"value": round(world * n_sv / svs, 2), "ms_per_view": round(svs / n_sv * 1e3, 4),
                                               "spread": svs_spread["spread"]}
        line["single_view"]["spread"] = sv_spread["spread"]
        line["config"]["single_view_views_per_s"] = line["single_view"]["value"]
        line["config"]["single_view_spread"] = sv_spread["spread"]
        line["config"]["single_view_static_scene_views_per_s"] = line["single_view"]["static_scene"]["value"]
        line["config"]["single_view_static_scene_spread"] = svs_spread["spread"]
        line["config"]["single_view_static_scene_min_views_per_s"] = round(world * n_sv / svs_spread["max_s"], 2)
        # one camera per call with the guaranteed ordering
        old_rank = L.gr_raster_ballot_ranking(1)
        single_step()
        svb, svb_spread = h.timed3(single_step, n_sv, max(args.warmup, 3))
        L.gr_raster_ballot_ranking(old_rank)
        line["config"]["single_view_ballot_ranking_views_per_s"] = round(world * n_sv / svb, 2)
        line["config"]["single_view_ballot_ranking_cost"] = round(svb / sv_elapsed - 1.0, 4)
        # the opt-in fast-exponential blend (1e-5 relative of the bit-exact image, tests/test_gpu_rasterizer_fast.py)
        rast_f = [GaussianRasterizer(s, fast_exp=True) for s in settings_list[: min(V, 8)]]

        def single_fast():
            r = rast_f[k["i"] % len(rast_f)]
            k["i"] += 1
            last["sv"] = r(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])

        single_fast()
        svf, _ = h.timed3(single_fast, n_sv, max(args.warmup, 3))
        line["single_view"]["fast_exp"] = {"value": round(world * n_sv / svf, 2), "ms_per_view": round(svf / n_sv * 1e3, 4)}
        line["config"]["single_view_fast_exp_views_per_s"] = line["single_view"]["fast_exp"]["value"]

        def fast_step():
            img, radii, nr = rasterize_views(settings, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                             rotations=t["rotations"], fast_exp=True)
            last["imgf"] = img

        fast_step()
        L.gr_timing_enable(1)
        f_elapsed = h.timed(fast_step, args.steps, args.warmup, after_warmup=L.gr_timing_reset)
        fb_ms, fb_n = timing_read(L, "raster_blend")
        L.gr_timing_enable(0)
        L.gr_timing_reset()
        line["config"]["fast_exp_views_per_s"] = round(world * V * args.steps / f_elapsed, 2)
        line["fast_exp"] = {"value": round(world * V * args.steps / f_elapsed, 2), "unit": "views/s",
                            "ms_per_step": round(f_elapsed / args.steps * 1e3, 4),
                            "blend_avg_launch_ms": round(fb_ms / max(fb_n, 1), 4),
                            "blend_frac_of_hbm_peak": round(blend_bytes / (fb_ms / max(fb_n, 1) / 1e3) / 1e9 / HBM_PEAK_GBS, 4),
                            "note": "opt-in GR_RASTER_FAST_EXP / fast_exp=True: v_exp_f32 in the blend; image within 1e-6 + 1e-5 rel "
                                    "of the float64 renderer on >= 99.9 % of the pixels (tests/test_gpu_rasterizer_fast.py); the "
                                    "headline `value` is the bit-exact mode"}

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
        # a step is 0.6 ms: at least 50 of them (and 10 to warm up -- the section follows the one-camera loops, which leave
        # the chip mostly idle) so that the timed region is tens of milliseconds
        n_r, w_r = max(args.steps, 50), max(args.warmup, 10)
        RT = ["radius_bin", "radius_tq", "radius_expand", "radius_count", "radius_fill", "radius_fused"]
        L.gr_timing_enable(1)
        r_elapsed, r_spread = h.timed3(radius_step, n_r, w_r)   # median of three timed regions (each ~30 ms)
        h.timed(radius_step, n_r, 1, after_warmup=L.gr_timing_reset)  # a fourth, instrumented pass for the per-kernel figures
        tq_ms, tq_n = timing_read(L, "radius_tq")
        ex_ms, _ = timing_read(L, "radius_expand")
        fill_ms, fill_n = timing_read(L, "radius_fill")
        rk = per_step_ms(L, RT, n_r)
        L.gr_timing_reset()
        nq = dpts.shape[0]
        width = out["nb"].shape[1]
        fill_bytes = 12.0 * nq + 12.0 * nq + 8.0 * nq * width      # 12 Nq + 12 Ns + 8 Nq W
        step_s = r_elapsed / n_r
        # the search proper (everything behind the binning): one thread per query (radius_tq.hpp) leaves compact rows, the
        # expand kernel widens them once the host knows the width; count + fill where that kernel gave up (dense clouds)
        if tq_n:
            s_name, s_ms, s_n = "radius_tq + radius_expand", tq_ms + ex_ms, tq_n
            s_kernels = ("tq_kernel<32, false", "tq_expand_kernel")
        else:
            s_name, s_ms, s_n = "radius_fill", fill_ms, fill_n
            s_kernels = ("traverse_kernel<128, false, true>", "traverse_kernel<128, true, true>")
        radius = {"metric": "radius_neighbors throughput, 200k-pt clouds", "value": round(world * nq * n_r / r_elapsed / 1e6, 2), "steps": n_r,
                  "unit": "Mpts/s", "ms_per_step": round(step_s * 1e3, 4),
                  "config": {"workload": f"{B} x 200k-pt clouds per GPU per step, r=0.0625, self-search, width {width}"},
                  "roofline": hbm_roofline(s_name, fill_bytes, s_ms, s_n, pmc_traffic("radius_search", True, B),
                                           kernels_ms_per_step=rk, valu_busy=sq_valu_busy(s_kernels[0], True),
                                           valu_busy_source="profiles/" + str(newest_profile("_sq_counters.json")),
                                           end_to_end_frac=round(fill_bytes / step_s / 1e9 / HBM_PEAK_GBS, 4))}
        # the second half of the metric as scalars the driver's flattened record keeps
        rl = line["roofline"]
        rl["radius_mpts_per_s"] = radius["value"]
        rl["radius_ms_per_step"] = radius["ms_per_step"]
        rl["radius_spread"] = r_spread["spread"]
        rl["radius_end_to_end_frac"] = radius["roofline"]["end_to_end_frac"]
        rl["radius_search_frac"] = radius["roofline"]["frac"]
        rl["radius_search_avg_launch_ms"] = radius["roofline"]["avg_launch_ms"]
        for k, v in rk.items():
            rl["radius_ms_" + k[7:]] = v
        # the same call on count + fill (gr_radius_search_mode 0: the default of rounds 1 - 5)
        old_mode = L.gr_radius_search_mode(0)
        radius_step()
        c_elapsed = h.timed(radius_step, n_r, w_r, after_warmup=L.gr_timing_reset)
        radius["count_fill"] = {"ms_per_step": round(c_elapsed / n_r * 1e3, 4), "kernels_ms_per_step": per_step_ms(L, RT, n_r),
                                "end_to_end_frac": round(fill_bytes / (c_elapsed / n_r) / 1e9 / HBM_PEAK_GBS, 4)}
        rl["radius_count_fill_end_to_end_frac"] = radius["count_fill"]["end_to_end_frac"]
        L.gr_radius_search_mode(old_mode)
        # the width-limited path the data pyramid calls (radius_search with neighbor_limit, utils/data.py:35-67)
        if not args.no_radius_limited:
            lim = 40

            def limited_step():
                out["nbl"] = radius_search(dpts, dpts, lens, lens, 0.0625, lim)

            limited_step()
            l_elapsed, l_spread = h.timed3(limited_step, n_r, w_r)
            h.timed(limited_step, n_r, 1, after_warmup=L.gr_timing_reset)
            lk = per_step_ms(L, RT, n_r)
            ltq_ms, ltq_n = timing_read(L, "radius_tq")
            lw = out["nbl"].shape[1]
            lbytes = 24.0 * nq + 8.0 * nq * lw
            radius["limited"] = {"value": round(world * nq * n_r / l_elapsed / 1e6, 2), "unit": "Mpts/s",
                                 "ms_per_step": round(l_elapsed / n_r * 1e3, 4), "neighbor_limit": lim, "width": lw,
                                 "end_to_end_frac": round(lbytes / (l_elapsed / n_r) / 1e9 / HBM_PEAK_GBS, 4),
                                 "kernels_ms_per_step": lk,
                                 "mode": "one thread per query, one kernel (default)" if ltq_n else "count + fill",
                                 "roofline": hbm_roofline("radius_tq", lbytes, ltq_ms, ltq_n) if ltq_n else None}
            rl["radius_limited_mpts_per_s"] = radius["limited"]["value"]
            rl["radius_limited_ms_per_step"] = radius["limited"]["ms_per_step"]
            rl["radius_limited_spread"] = l_spread["spread"]
            rl["radius_limited_end_to_end_frac"] = radius["limited"]["end_to_end_frac"]
            if ltq_n:
                rl["radius_limited_search_frac"] = radius["limited"]["roofline"]["frac"]
                rl["radius_limited_ms_tq"] = lk.get("radius_tq")
                rl["radius_limited_ms_bin"] = lk.get("radius_bin")
            # what bounds the search kernel: VALU issue.  Wave-level VALU instructions per query (committed SQ-counter
            # summary, same configuration) against the ~50 the tests + stores alone need
            vq = sq_insts("tq_kernel<32, true", "SQ_INSTS_VALU")
            if vq and B == 8:
                per_q = vq / float(nq)
                radius["roofline_valu_issue"] = {"bound": "valu_issue", "achieved": round(per_q, 1), "floor": 50.0,
                                                 "unit": "wave-level VALU instructions per query (tq_kernel, limit 40)", "frac": round(50.0 / per_q, 4),
                                                 "valu_busy": sq_valu_busy("tq_kernel<32, true", True),
                                                 "source": "profiles/" + str(newest_profile("_sq_counters.json")) + " (committed, not this run)"}
                rl["radius_valu_per_query"] = round(per_q, 1)
                rl["radius_valu_floor_per_query"] = 50.0
            # the same call on count + fill and through the three-threads-per-query single-pass kernel (modes 0 and 1)
            for mode, key in ((0, "count_fill"), (1, "single_pass")):
                old_mode = L.gr_radius_search_mode(mode)
                limited_step()
                s_elapsed = h.timed(limited_step, n_r, w_r, after_warmup=L.gr_timing_reset)
                L.gr_radius_search_mode(old_mode)
                radius["limited"][key] = {
                    "value": round(world * nq * n_r / s_elapsed / 1e6, 2), "ms_per_step": round(s_elapsed / n_r * 1e3, 4),
                    "end_to_end_frac": round(lbytes / (s_elapsed / n_r) / 1e9 / HBM_PEAK_GBS, 4), "kernels_ms_per_step": per_step_ms(L, RT, n_r)}
            rl["radius_limited_count_fill_end_to_end_frac"] = radius["limited"]["count_fill"]["end_to_end_frac"]
        L.gr_timing_enable(0)
        L.gr_timing_reset()
        del out, dpts
    line["radius_neighbors"] = radius

    # ------------------------------------------------------------------ configs[4]: pair-sharded coarse registration
    if args.pairs > 0:
        from gaussreg_amd import pair_pipeline
        n_total = args.pairs * world
        a, b = sharding.shard_bounds(n_total, rank, world)
        reg = pair_pipeline.PairRegistrar(dev)
        pairs = [pair_pipeline.synthetic_room_pair(i, args.pair_points, dev) for i in range(a, b)]
        reg.register_many(pairs, args.pair_batch)  # warm-up at the timed sizes: the caching allocator then holds every block the
        # timed pass asks for (a first-time hipMalloc of a 64-pair buffer costs more than the kernels that fill it)
        rows = []

        def pairs_pass():
            rows.clear()
            rows.append(reg.register_many(pairs, args.pair_batch))  # FPS over all clouds of the rank (25 per call), then blocks

        L.gr_timing_enable(1)
        p_elapsed = h.timed(pairs_pass, 1, 0, after_warmup=L.gr_timing_reset)
        pk = per_step_ms(L, ["fps", "radius_bin", "radius_tq", "radius_expand", "radius_count", "radius_fill", "radius_fused", "sinkhorn", "lgr", "ransac"], 1)
        L.gr_timing_enable(0)
        L.gr_timing_reset()
        local = torch.cat(rows, 0)
        reg.close()
        counts = [sharding.shard_bounds(n_total, r, world)[1] - sharding.shard_bounds(n_total, r, world)[0] for r in range(world)]
        allres = sharding.gather_rows(local, counts)          # ONE all_gather of (pairs, 20) floats (RCCL over xGMI)
        rre, rte = allres[:, 16], allres[:, 17]
        ok = (rre < 5.0) & (rte < 0.1)
        line["pairs"] = {"value": round(n_total / p_elapsed, 2), "unit": "pairs/s", "pairs_total": n_total,
                         "pairs_per_gpu": args.pairs, "pair_batch": args.pair_batch, "ms_per_pair_per_gpu": round(p_elapsed / args.pairs * 1e3, 3),
                         "config": "configs[4] stand-in: synthetic room pairs (2 x %d pts) -> FPS 30k -> 5-level pyramid -> "
                                   "point_to_node -> SuperPointMatching -> Sinkhorn -> LocalGlobalRegistration -> RANSAC; "
                                   "features are synthetic position descriptors (no pretrained weights offline)" % args.pair_points,
                         "gathered_rows": int(allres.shape[0]), "standin_success_rate": round(float(ok.float().mean()), 4),
                         "standin_success_rate_note": "NOT registration accuracy: the stand-in descriptors are built from coordinates "
                                                      "mapped through the ground-truth transform (pair_pipeline.py:5-14); it only shows that the "
                                                      "geometric stages (matching ops, Sinkhorn, LGR, RANSAC) are wired correctly",
                         "median_rre_deg": round(float(rre.median()), 4), "median_rte_m": round(float(rte.median()), 5),
                         "kernel_ms_total_per_gpu": pk}
        line["config"]["pairs_per_s"] = line["pairs"]["value"]
        # ---- where a pair's time goes: one more pass over the first block of pairs with a device synchronise around every stage
        #      (slower than the timed pass: nothing overlaps; the SHARES are what this is for).  `standin_descriptors` is harness
        #      work (random Fourier features in place of the learned features), everything else is the reference's pipeline.
        try:
            regp = pair_pipeline.PairRegistrar(dev, profile=True)
            nprof = min(len(pairs), args.pair_batch)
            regp.register_pairs(pairs[:nprof])
            regp.section_ms.clear()
            regp.register_pairs(pairs[:nprof])
            tot_ms = sum(regp.section_ms.values())
            line["pairs"]["stage_share"] = {k: round(v / tot_ms, 4) for k, v in sorted(regp.section_ms.items(), key=lambda kv: -kv[1])}
            line["pairs"]["stage_ms_per_pair_synchronised"] = {k: round(v / nprof, 4) for k, v in regp.section_ms.items()}
            line["pairs"]["stage_note"] = ("standin_descriptors = harness work (one HIP launch per call, csrc/standin.hip); patch_scores = the "
                                           "128 x 128 x 256 contraction of model.py:186-190 (rocBLAS through torch.bmm, as in the reference); "
                                           "patch_features / metrics = index bookkeeping in stock PyTorch, as in the reference")
            regp.close()
            del regp
        except Exception as e:
            line["pairs"]["stage_share"] = {"error": repr(e)}
        # ---- the same workload with the real network (random seeded weights: timing does not need a checkpoint; the estimates
        #      are not scored).  Fewer pairs: 18 ms of network per pair
        try:
            n_net = min(len(pairs), 64)
            regm = pair_pipeline.PairRegistrar(dev, features="model")
            net_batch = min(args.pair_batch, 32)
            torch.cuda.empty_cache()                     # the earlier sections' cached blocks are of other sizes
            regm.register_pairs(pairs[:net_batch])       # warm-up at the timed batch size (allocator, lazy kernels)

            def net_pass():
                regm.register_many(pairs[:n_net], net_batch)

            m_elapsed = h.timed(net_pass, 1, 0)
            regm.close()
            line["pairs"]["with_network"] = {
                "value": round(world * n_net / m_elapsed, 2), "unit": "pairs/s", "pairs_per_gpu": n_net, "pair_batch": net_batch,
                "ms_per_pair_per_gpu": round(m_elapsed / n_net * 1e3, 3),
                "config": "same pairs; KPConvFPN (one pass per batch, GroupNorm per pair) + GeometricTransformer (padded batches of "
                          "16 pairs: embedding and self-attention per cloud, the rest over the batch) + backbone features in the "
                          "patch scores instead of the synthetic descriptors; 28 M seeded random parameters"}
            line["config"]["pairs_with_network_per_s"] = line["pairs"]["with_network"]["value"]
            del regm
        except Exception as e:  # never takes the headline down with it
            line["pairs"]["with_network"] = {"error": repr(e)}
        del pairs, rows

    # ------------------------------------------------------------------ the other SURVEY 8(d) kernels
    if not args.no_extras and rank == 0:
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import bench_extras
            line["extras"] = bench_extras.run(dev)
        except Exception as e:  # an extra must never take the headline down with it
            line["extras"] = {"error": repr(e)}
    h.barrier()

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
        ident = cpu_identity()
        workers = max(1, min(ident["cores_available"], ident["physical_cores"] or ident["cores_available"], 64))
        cpu_baseline = {"value": round(1.0 / dt, 4), "unit": "views/s", "cores": 1, "kind": "port",
                        "sample": f"1 view of the same {P}-Gaussian {W}x{H} scene through oracle/rasterizer_oracle.c "
                                  f"({dt:.2f} s; the rasterizer has no reference implementation in the GaussReg tree)"}
        cpu_baseline.update(ident)
        kind = "reference" if capi.have_ref() else "port"
        if radius is not None:
            p1, l1 = synthetic.cloud_200k(1, seed=0)
            fn = capi.ref_radius_neighbors if kind == "reference" else capi.radius_neighbors
            tc = time.perf_counter()
            fn(p1.numpy(), p1.numpy(), l1.numpy(), l1.numpy(), 0.0625)
            dt = time.perf_counter() - tc
            radius["cpu_baseline"] = {"value": round(0.2 / dt, 4), "unit": "Mpts/s", "cores": 1, "kind": kind,
                                      "sample": "one 200k-pt cloud, single thread ("
                                                + ("reference C++ core compiled in oracle/_ref" if kind == "reference"
                                                   else "oracle/radius_neighbors_oracle.c") + f", {dt:.2f} s)"}
            radius["cpu_baseline"].update(ident)
            ac = all_cores("radius200k", 0.2, "Mpts/s", workers, tasks_per_worker=8)
            ac["sample"] = (f"{ac.get('tasks')} searches of independent 200k-pt clouds, 8 per worker process, one process per core "
                            f"on {workers} cores, same {kind} core")
            radius["cpu_baseline"]["all_cores"] = ac
            cpu_baseline["radius_1core_mpts_per_s"] = radius["cpu_baseline"]["value"]
            cpu_baseline["radius_all_cores_mpts_per_s"] = ac.get("value")
            cpu_baseline["radius_all_cores"] = ac.get("cores")
            cpu_baseline["radius_kind"] = kind
        if "pairs" in line:
            # the reference's per-pair CPU work on this path: the collate pyramid (utils/data.py:13-77) on one core
            from gaussreg_amd import pair_pipeline
            ref_p, src_p, _ = pair_pipeline.synthetic_room_pair(0, 30000, torch.device("cpu"))
            pts = np.concatenate([ref_p.numpy(), src_p.numpy()])
            lens = np.array([30000, 30000], np.int64)
            gs = capi.ref_grid_subsampling if kind == "reference" else capi.grid_subsampling
            rn = capi.ref_radius_neighbors if kind == "reference" else capi.radius_neighbors

            def one_pyramid(_i=0):
                plist, llist, voxel, rad = [pts], [lens], 0.025, 0.0625
                for i in range(1, 5):
                    voxel *= 2
                    p2, l2 = gs(plist[-1], llist[-1], voxel)
                    plist.append(p2)
                    llist.append(l2)
                for i in range(5):
                    rn(plist[i], plist[i], llist[i], llist[i], rad)
                    if i < 4:
                        rn(plist[i + 1], plist[i], llist[i + 1], llist[i], rad)
                        rn(plist[i], plist[i + 1], llist[i], llist[i + 1], 2 * rad)
                    rad *= 2

            tc = time.perf_counter()
            one_pyramid()
            dt = time.perf_counter() - tc
            line["pairs"]["cpu_baseline"] = {"value": round(1.0 / dt, 4), "unit": "pairs/s", "cores": 1, "kind": kind,
                                             "scope": "pyramid_only",
                                             "sample": f"the collate pyramid (4 grid_subsample + 13 radius_search) of ONE 2x30000-pt "
                                                       f"pair on one core ({dt:.2f} s); FPS, the network, matching and RANSAC are NOT "
                                                       f"included -- compare with pairs.pyramid_only_gpu, not with pairs.value"}
            line["pairs"]["cpu_baseline"].update(ident)
            ac = all_cores("pyramid", 1.0, "pairs/s", workers, tasks_per_worker=4)
            ac["sample"] = f"{ac.get('tasks')} pyramids, 4 per worker process, one process per core on {workers} cores (pyramid only)"
            line["pairs"]["cpu_baseline"]["all_cores"] = ac
            ex = line.get("extras") or {}
            for kname, kval in ex.items():
                if "pyramid" in kname and isinstance(kval, dict) and "ms" in kval:
                    line["pairs"]["pyramid_only_gpu"] = {"extras_key": kname, **{kk: kval[kk] for kk in kval if kk in ("ms", "pairs", "pairs_per_s", "ms_per_pair")}}
                    break
    line["cpu_baseline"] = cpu_baseline
    return order_line(line)


def order_line(line):
    """Key order of the printed line: the contract's fields first, the long sections in the middle, and a short `summary` of
    the figures that answer the metric LAST (the driver keeps the line's tail and the scalars of config / roofline / cpu_baseline)."""
    head = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]
    out = {k: line[k] for k in head if k in line}
    for k in ("extras", "pairs", "fast_exp", "single_view", "radius_neighbors"):
        if k in line:
            out[k] = line[k]
    for k in line:
        if k not in out:
            out[k] = line[k]
    cfg, rl, cb = line.get("config") or {}, line.get("roofline") or {}, line.get("cpu_baseline") or {}
    ex = line.get("extras") or {}

    def frac(name):
        v = ex.get(name)
        return v.get("frac") if isinstance(v, dict) else None

    def ms(name):
        v = ex.get(name)
        return v.get("ms") if isinstance(v, dict) else None
    out["summary"] = {
        "views_per_s": line.get("value"), "views_per_s_static_scene": cfg.get("static_scene_views_per_s"),
        "blend_frac_of_hbm": rl.get("frac"),
        "single_view_views_per_s": cfg.get("single_view_views_per_s"),
        "single_view_static_scene_views_per_s": cfg.get("single_view_static_scene_views_per_s"),
        "radius_mpts_per_s": rl.get("radius_mpts_per_s"), "radius_end_to_end_frac": rl.get("radius_end_to_end_frac"),
        "radius_search_frac": rl.get("radius_search_frac"), "radius_limited_end_to_end_frac": rl.get("radius_limited_end_to_end_frac"),
        "radius_valu_per_query": rl.get("radius_valu_per_query"),
        "radius_cpu_1core_mpts_per_s": cb.get("radius_1core_mpts_per_s"), "radius_cpu_all_cores_mpts_per_s": cb.get("radius_all_cores_mpts_per_s"),
        "pairs_per_s": cfg.get("pairs_per_s"), "pairs_with_network_per_s": cfg.get("pairs_with_network_per_s"),
        "grid_subsample_64x200k_ms": [ms("grid_subsample_64x200k_reference_order"), ms("grid_subsample_64x200k_cell_order")],
        "grid_subsample_200k_ms": ms("grid_subsample_200k_reference_order"),
        "spm_batch_64x767_frac": frac("superpoint_matching_batch_64x767"), "spm_767_ms": ms("superpoint_matching_767"),
        "kpconv_backbone_frac": frac("kpconv_backbone_11_layers"), "gs_fuse_frac": frac("gs_fuse_2x2p55M"),
        "fps_2x200k_ms": ms("fps_2x200k_to_30k"), "n_gpus": line.get("n_gpus")}
    return out


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        if args.backend != "nccl":
            raise SystemExit("--backend gloo: launch the ranks yourself (torch.distributed.run); the self-spawn is for GPUs")
        return self_spawn(args, argv)
    if env_world is not None and int(env_world) != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} launched with WORLD_SIZE={env_world}: they must agree")
    h = Harness.from_env(args.backend)
    if args.dry:
        line = dry_run(h, args.backend)
        if h.rank == 0:
            print(json.dumps(line), flush=True)
        h.close()
        return 0 if line["ok"] else 1
    if args.backend != "nccl":
        raise SystemExit("bench.py measures the HIP path on MI355X GPUs; --backend gloo exists for --dry only")
    line = run_gpu(h, args)
    if h.rank == 0:
        print(json.dumps(line), flush=True)
    h.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
