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
