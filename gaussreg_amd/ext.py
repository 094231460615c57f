"""Mirror of the reference's native module ``geotransformer.ext`` (setup.py:10, pybind.cpp:6-18).

Same two functions, same positional signatures, same error texts -- backed by the HIP kernels:

    radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius) -> LongTensor (Nq, max_count)
        reference: geotransformer/extensions/cpu/radius_neighbors/radius_neighbors.cpp:5-68
    grid_subsampling(points, lengths, voxel_size) -> [s_points (M,3) f32, s_lengths (B,) i64]
        reference: geotransformer/extensions/cpu/grid_subsampling/grid_subsampling.cpp:5-62

Differences from the reference (documented in INTEGRATION.md): tensors may live on the GPU (the
reference insists on CPU tensors); CPU tensors are accepted, computed on the GPU and returned on
the CPU; outputs follow the input's device like the reference's `at::device(points.device())`.
Nothing is ever computed on the CPU.
"""
import ctypes

import torch

from . import _lib

_ORDER = {"reference": 0, "cell": 1}


def _check_float(x, name):
    if x.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")  # torch_helper.h:31-35


def _check_long(x, name):
    if x.dtype != torch.int64:
        raise RuntimeError(f"{name} must be an long tensor")  # torch_helper.h:25-29 (sic)


def _check_contig(x, name):
    if not x.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")  # torch_helper.h:12-13


def _check_points(x, name):
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if x.dim() != 2 or x.shape[1] != 3:
        raise RuntimeError(f"{name} must have shape (N, 3)")


class SupportGrid:
    """Cell grid of one support cloud, kept between searches that use the SAME supports, lengths and radius with
    different queries (the data pyramid does three per level, geotransformer/utils/data.py:44-75): the first search
    bins the supports, the following ones only bin their queries (gr_radius_count_cached).  Owns its workspace, sized
    for the largest query set announced in `max_queries`.  The support tensor must not be modified in between."""

    def __init__(self, max_queries):
        self.max_queries = int(max_queries)
        self.ws = None
        self.sig = (ctypes.c_int64 * 4)()
        self.key = None          # (data_ptr, ns, nb, radius): what the grid was built for
        self._keep = None        # keeps the support tensor alive

    def workspace(self, L, dev, nq, ns, nb):
        need = L.gr_radius_workspace_bytes(max(nq, self.max_queries), ns, nb)
        if self.ws is None or self.ws.numel() < need or self.ws.device != dev:
            self.ws = torch.empty((need,), dtype=torch.uint8, device=dev)
            self.key = None      # a new buffer holds no grid
        return self.ws


def _radius(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid, checks):
    dev = _lib.require_gpu()
    L = _lib.lib()
    if checks:
        for t, n in ((q_points, "q_points"), (s_points, "s_points")):
            _check_points(t, n)
            _check_float(t, n)
        for t, n in ((q_lengths, "q_lengths"), (s_lengths, "s_lengths")):
            _check_long(t, n)
        for t, n in ((q_points, "q_points"), (s_points, "s_points"), (q_lengths, "q_lengths"), (s_lengths, "s_lengths")):
            _check_contig(t, n)
        if q_lengths.numel() != s_lengths.numel():
            raise RuntimeError("q_lengths and s_lengths must have the same number of batch elements")
    out_device = q_points.device
    same = s_points is q_points or (s_points.data_ptr() == q_points.data_ptr() and s_points.shape == q_points.shape)
    q = q_points if q_points.is_cuda else q_points.to(dev)
    s = q if same else (s_points if s_points.is_cuda else s_points.to(dev))
    dev = q.device
    ql, sl = q_lengths.tolist(), s_lengths.tolist()
    nq, ns, nb = q.shape[0], s.shape[0], len(ql)
    hq, hs = _lib.host_i64(ql), _lib.host_i64(sl)
    with torch.cuda.device(dev):
        info = (ctypes.c_int64 * 4)()
        st = _lib.stream_ptr(dev)
        if grid is None:
            ws = _lib.workspace(dev, L.gr_radius_workspace_bytes(nq, ns, nb))
            _lib.check(L.gr_radius_count(_lib.ptr(q), _lib.ptr(s), hq, hs, nq, ns, nb, float(radius), _lib.ptr(ws),
                                         ws.numel(), info, st))
        else:
            ws = grid.workspace(L, dev, nq, ns, nb)
            key = (s.data_ptr(), ns, nb, float(radius), tuple(sl))
            reuse = 1 if (grid.key == key and ns > 0 and nq > 0) else 0
            _lib.check(L.gr_radius_count_cached(_lib.ptr(q), _lib.ptr(s), hq, hs, nq, ns, nb, float(radius), _lib.ptr(ws),
                                                ws.numel(), info, grid.sig, reuse, st))
            grid.key = key if (ns > 0 and nq > 0 and nb > 0) else None
            grid._keep = s
        width = int(info[0])
        if neighbor_limit is not None and neighbor_limit > 0:
            width = min(width, int(neighbor_limit))
        out = torch.empty((nq, width), dtype=torch.int64, device=dev)
        if nq > 0 and width > 0:
            _lib.check(L.gr_radius_fill(_lib.ptr(q), _lib.ptr(s), nq, ns, nb, float(radius), width, info,
                                        _lib.ptr(out), _lib.ptr(ws), ws.numel(), st))
    return out if out_device.type == "cuda" else out.to(out_device)


_WIDTH_HINT = {}  # (radius, neighbor_limit) -> largest neighbour count of the last search of that call site


def _radius_limited(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid, contiguous=True):
    """radius_search with neighbor_limit > 0: the (nq, limit) rows are allocated before anything is counted and ONE
    kernel searches, ranks and writes them (gr_radius_search); the read-back of max_count only decides whether the
    reference would have returned fewer columns (radius_search.py:25-26 keeps min(max_count, limit))."""
    dev = _lib.require_gpu()
    L = _lib.lib()
    out_device = q_points.device
    same = s_points is q_points or (s_points.data_ptr() == q_points.data_ptr() and s_points.shape == q_points.shape)
    q = q_points if q_points.is_cuda else q_points.to(dev)
    s = q if same else (s_points if s_points.is_cuda else s_points.to(dev))
    dev = q.device
    ql, sl = q_lengths.tolist(), s_lengths.tolist()
    nq, ns, nb = q.shape[0], s.shape[0], len(ql)
    limit = int(neighbor_limit)
    hq, hs = _lib.host_i64(ql), _lib.host_i64(sl)
    with torch.cuda.device(dev):
        info = (ctypes.c_int64 * 6)()
        st = _lib.stream_ptr(dev)
        if grid is None:
            ws = _lib.workspace(dev, L.gr_radius_workspace_bytes(nq, ns, nb))
            sig, reuse = None, 0
        else:
            ws = grid.workspace(L, dev, nq, ns, nb)
            key = (s.data_ptr(), ns, nb, float(radius), tuple(sl))
            reuse = 1 if (grid.key == key and ns > 0 and nq > 0) else 0
            sig = grid.sig
        # Row stride of the search.  The reference's limit is an upper bound chosen by calibration and can be far above
        # what a level ever returns (the demo pyramid: limit 89, largest count 18): rows of `limit` columns would be 5 x the
        # bytes, written by the kernel and read again by the truncating copy.  A call site (radius, limit) therefore
        # remembers the largest count it has seen and searches with rows of that width + a margin (a multiple of 8: whole
        # 64-byte sectors per row piece, streaming stores); a call whose largest count does not fit is repeated at the full
        # limit -- the result is the same tensor either way.
        hint = _WIDTH_HINT.get((float(radius), limit))
        stride = limit if hint is None else min(limit, max(8, (hint + max(2, hint // 4) + 7) // 8 * 8))
        while True:
            out = torch.empty((nq, stride), dtype=torch.int64, device=dev)
            _lib.check(L.gr_radius_search(_lib.ptr(q), _lib.ptr(s), hq, hs, nq, ns, nb, float(radius), stride, _lib.ptr(out),
                                          _lib.ptr(ws), ws.numel(), info, sig, reuse, st))
            if grid is not None:
                grid.key = key if (ns > 0 and nq > 0 and nb > 0) else None
                grid._keep = s
                reuse = 1 if grid.key is not None else 0  # (a repeat finds the supports binned)
            if int(info[0]) <= stride or stride == limit:
                break
            stride = limit
        if nq > 0 and ns > 0 and nb > 0:
            _WIDTH_HINT[(float(radius), limit)] = int(info[0])
            if len(_WIDTH_HINT) > 4096:
                _WIDTH_HINT.clear()
        width = min(int(info[0]), limit)
        if width < stride:
            # (contiguous=False: the column slice of the searched rows, row stride `stride` -- what the reference itself returns
            # when it truncates, radius_search.py:26; saves a copy of the whole result where nobody needs it dense)
            out = out[:, :width].contiguous() if contiguous else out[:, :width]
    return out if out_device.type == "cuda" else out.to(out_device)


def radius_neighbors(q_points, s_points, q_lengths, s_lengths, radius, grid=None):
    """`grid`: optional SupportGrid shared by consecutive searches over the same supports and radius."""
    return _radius(q_points, s_points, q_lengths, s_lengths, radius, None, grid, True)


def radius_neighbors_limited(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid=None, two_pass=False,
                             contiguous=True):
    """radius_neighbors + the column truncation of modules/ops/radius_search.py:25-26 done inside the
    kernel: only min(max_count, neighbor_limit) columns are ever written (contiguous result).  With a positive limit the
    search is ONE pass (gr_radius_search); `two_pass=True` forces the count + fill pair the bare radius_neighbors uses."""
    if neighbor_limit is not None and neighbor_limit > 0 and not two_pass:
        return _radius_limited(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid, contiguous)
    return _radius(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid, False)


def grid_subsampling(points, lengths, voxel_size, order="reference"):
    """`order="reference"` (default) reproduces the reference's row order bit for bit;
    `order="cell"` keeps everything on the device (rows sorted by voxel key)."""
    dev = _lib.require_gpu()
    L = _lib.lib()
    _check_points(points, "points")
    _check_float(points, "points")
    _check_long(lengths, "lengths")
    _check_contig(points, "points")
    _check_contig(lengths, "lengths")
    out_device = points.device
    p = points if points.is_cuda else points.to(dev)
    dev = p.device
    lens = lengths.tolist()
    n, nb = p.shape[0], len(lens)
    hl = _lib.host_i64(lens)
    out_l = (ctypes.c_int64 * max(nb, 1))()
    total = ctypes.c_int64(0)
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, L.gr_grid_subsample_workspace_bytes(n, nb))
        out = torch.empty((n, 3), dtype=torch.float32, device=dev)
        _lib.check(L.gr_grid_subsample(_lib.ptr(p), hl, n, nb, float(voxel_size), _ORDER[order], _lib.ptr(out),
                                       out_l, ctypes.byref(total), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev)))
    s_points = out[: total.value]
    if total.value * 4 < 3 * n:
        s_points = s_points.clone()  # do not pin the (n, 3) allocation behind a view of less than three quarters of it: the
                                     # pyramid keeps every level's output (the copy costs 20 - 40 us per call)
    s_lengths = torch.tensor([out_l[b] for b in range(nb)], dtype=torch.int64, device=lengths.device)
    if out_device.type != "cuda":
        s_points = s_points.to(out_device)
    return [s_points, s_lengths]
