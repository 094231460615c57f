"""Multi-GPU sharding of the two hot paths: independent units, one process per GPU, one gather.

SURVEY.md section 8(e): scene pairs (registration_collate_fn_stack_mode, batch_size 1 -- config.py:39,48) and
rendered views are independent, so rank r of W owns a contiguous block of the work list, runs it on
its own GPU with no data-path collective, and the small per-unit results (a 4x4 transform + a few
scalars per pair, or counters / optionally images per view) are collected with ONE all_gather
(RCCL over xGMI on GPUs -- torch.distributed backend "nccl"; gloo in the CPU tests).
"""
from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_items: int, rank: int, world_size: int):
    """Contiguous block partition; the first (n % W) ranks get one extra unit."""
    base, extra = divmod(n_items, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_rows(local: torch.Tensor, counts: Sequence[int] = None) -> torch.Tensor:
    """All-gather a (n_local, ...) tensor along dim 0 (ragged across ranks) -> every rank gets the
    concatenation in rank order.  One collective; payload is tiny by design."""
    rank, w = world()
    if w == 1:
        return local
    if local.is_cuda and dist.get_backend() == "gloo":
        # gloo has no all_gather on device tensors (the CPU tests and a one-GPU box with two processes): stage through the host
        return gather_rows(local.cpu(), counts).to(local.device)
    n_local = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    if counts is None:
        all_n = [torch.zeros_like(n_local) for _ in range(w)]
        dist.all_gather(all_n, n_local)
        counts = [int(t.item()) for t in all_n]
    mx = max(counts) if counts else 0
    if mx == 0:  # every rank knows it: no zero-byte collective (RCCL need not accept one)
        return local[:0]
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(w)]
    dist.all_gather(bufs, pad)
    return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)


def run_sharded(units: Sequence, fn: Callable[[object], torch.Tensor]) -> torch.Tensor:
    """Apply `fn` (unit -> 1-D result tensor of fixed length) to this rank's block of `units`, then
    gather: returns (len(units), result_len) on every rank, rows in the order of `units`."""
    rank, w = world()
    a, b = shard_bounds(len(units), rank, w)
    rows: List[torch.Tensor] = [fn(units[i]).reshape(1, -1) for i in range(a, b)]
    if rows:
        local = torch.cat(rows, dim=0)
    else:
        # result length unknown on an idle rank: learn it from rank 0's first row
        local = None
    if w > 1:
        width = torch.tensor([local.shape[1] if local is not None else 0], dtype=torch.int64,
                             device=(local.device if local is not None else _default_device()))
        dist.all_reduce(width, op=dist.ReduceOp.MAX)
        if local is None:
            local = torch.zeros((0, int(width.item())), dtype=torch.float32, device=width.device)
    elif local is None:
        local = torch.zeros((0, 0))
    counts = [shard_bounds(len(units), r, w)[1] - shard_bounds(len(units), r, w)[0] for r in range(w)]
    return gather_rows(local, counts)


def _default_device():
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def render_views_sharded(settings, means3D, opacities, gather_images=False, **kw):
    """Render this rank's block of `settings` (a list of GaussianRasterizationSettings) with the
    batched HIP rasterizer; returns (local_images (v_local,3,H,W), gathered per-view instance counts
    (V,), and -- if gather_images -- all images (V,3,H,W) on every rank)."""
    from .rasterizer import rasterize_views
    rank, w = world()
    a, b = shard_bounds(len(settings), rank, w)
    imgs, radii, nr = rasterize_views(settings[a:b], means3D, opacities, **kw) if b > a else (None, None, [])
    dev = means3D.device
    local = torch.tensor(nr, dtype=torch.int64, device=dev).reshape(-1, 1)
    counts = [shard_bounds(len(settings), r, w)[1] - shard_bounds(len(settings), r, w)[0] for r in range(w)]
    all_nr = gather_rows(local, counts).reshape(-1)
    all_imgs = None
    if gather_images:
        H, W = settings[0].image_height, settings[0].image_width
        loc = imgs if imgs is not None else torch.zeros((0, 3, H, W), dtype=torch.float32, device=dev)
        all_imgs = gather_rows(loc, counts)
    return imgs, all_nr, all_imgs
