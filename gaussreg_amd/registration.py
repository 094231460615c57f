"""Correspondence-based RANSAC with a similarity model on the GPU -- the step GaussReg delegates to Open3D
(geotransformer/utils/open3d.py:169-198, called from model.py:209-215).  PARITY UNPINNED (Open3D is not in
the reference tree, and its sampler is unseeded): behaviour is judged on registration error, see tests."""
import torch

from . import _lib


@torch.no_grad()
def registration_with_ransac_from_correspondences(src_points, ref_points, correspondences=None, distance_threshold=0.05,
                                                  ransac_n=3, num_iterations=10000, with_scaling=True, refine=True,
                                                  seed=0, return_stats=False):
    """Same positional signature as the reference wrapper (open3d.py:169-176); returns a (4,4) float32 tensor
    (the reference returns a float64 numpy array).  `correspondences` (C,2) selects rows of src / ref."""
    dev = _lib.require_gpu()
    L = _lib.lib()
    s = torch.as_tensor(src_points, dtype=torch.float32)
    r = torch.as_tensor(ref_points, dtype=torch.float32)
    s = (s if s.is_cuda else s.to(dev)).contiguous()
    dev = s.device
    r = r.to(dev).contiguous()
    if correspondences is not None:
        c = torch.as_tensor(correspondences, dtype=torch.int64, device=dev)
        s, r = s[c[:, 0]].contiguous(), r[c[:, 1]].contiguous()
    n = s.shape[0]
    out = torch.empty((4, 4), dtype=torch.float32, device=dev)
    stats = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, L.gr_ransac_workspace_bytes(int(num_iterations)))
        _lib.check(L.gr_ransac_similarity(_lib.ptr(s), _lib.ptr(r), n, int(ransac_n), int(num_iterations), int(seed),
                                          float(distance_threshold), int(bool(with_scaling)), int(bool(refine)),
                                          _lib.ptr(out), _lib.ptr(stats), _lib.ptr(ws), ws.numel(),
                                          _lib.stream_ptr(dev)))
    return (out, stats) if return_stats else out


@torch.no_grad()
def registration_with_ransac_batch(src_points, ref_points, row_offsets, fallback_transforms=None, distance_threshold=0.05,
                                   ransac_n=3, num_iterations=10000, with_scaling=True, refine=True, seed=0,
                                   return_stats=False):
    """registration_with_ransac_from_correspondences for a batch of scene pairs in one call (gr_ransac_similarity_seg):
    row i of src_points corresponds to row i of ref_points; pair b owns rows [row_offsets[b], row_offsets[b+1]) (int32
    tensor on the device, B + 1 entries) and uses seed + b.  A pair with fewer than ransac_n rows keeps
    fallback_transforms[b] (model.py:209-220; default identity).  Returns (B,4,4) float32 on the device, no host
    synchronisation."""
    L = _lib.lib()
    s = src_points.contiguous()
    r = ref_points.contiguous()
    dev = s.device
    off = row_offsets.to(device=dev, dtype=torch.int32).contiguous()
    B = off.shape[0] - 1
    out = torch.empty((B, 4, 4), dtype=torch.float32, device=dev)
    stats = torch.zeros((B, 2), dtype=torch.int32, device=dev)
    fb = None if fallback_transforms is None else fallback_transforms.to(device=dev, dtype=torch.float32).contiguous()
    if B > 0:
        with torch.cuda.device(dev):
            ws = _lib.workspace(dev, L.gr_ransac_seg_workspace_bytes(int(num_iterations), B))
            _lib.check(L.gr_ransac_similarity_seg(_lib.ptr(s), _lib.ptr(r), _lib.ptr(off), B, int(ransac_n),
                                                  int(num_iterations), int(seed), float(distance_threshold),
                                                  int(bool(with_scaling)), int(bool(refine)), _lib.ptr(fb), _lib.ptr(out),
                                                  _lib.ptr(stats), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
    return (out, stats) if return_stats else out


@torch.no_grad()
def farthest_point_sampling(points, lengths, num_samples, start_indices=None, gather=False):
    """Exact FPS in stack mode (stand-in for fpsample.bucket_fps_kdline_sampling, demo.py:46; parity unpinned).
    points (N,3); lengths / num_samples: per-cloud sizes (lists or 1-D tensors).  Returns a list of int64
    CUDA tensors with LOCAL indices, one per cloud, in sampling order (first = start index, default 0).
    `gather=True`: returns the sampled POINTS instead, a list of (k_b, 3) views of one stacked tensor (one row gather for
    the whole call instead of one indexing launch per cloud)."""
    dev = _lib.require_gpu()
    L = _lib.lib()
    p = torch.as_tensor(points, dtype=torch.float32)
    p = (p if p.is_cuda else p.to(dev)).contiguous()
    dev = p.device
    lens = [int(x) for x in (lengths.tolist() if hasattr(lengths, "tolist") else lengths)]
    ks = [int(x) for x in (num_samples.tolist() if hasattr(num_samples, "tolist") else num_samples)]
    st = None if start_indices is None else _lib.host_i64([int(x) for x in start_indices])
    out = torch.empty((sum(ks),), dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, L.gr_fps_workspace_bytes(p.shape[0], len(lens)))
        _lib.check(L.gr_fps(_lib.ptr(p), _lib.host_i64(lens), _lib.host_i64(ks), st, p.shape[0], len(lens), _lib.ptr(out),
                            _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
    if gather:
        from .ops import index_select
        base = torch.tensor([0] + lens[:-1], dtype=torch.int64).cumsum(0).to(dev)        # first row of every cloud in the stack
        glob = out + torch.repeat_interleave(base, torch.tensor(ks, device=dev))
        return list(torch.split(index_select(p, glob, 0), ks))
    return list(torch.split(out, ks))
