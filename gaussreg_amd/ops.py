"""Mirror of ``geotransformer.modules.ops`` for the hot-path operators (HIP-backed).

reference: geotransformer/modules/ops/grid_subsample.py:7-22, radius_search.py:7-27.
"""
from . import ext


def grid_subsample(points, lengths, voxel_size, order="reference"):
    """Grid subsampling in stack mode -> (s_points (M,3), s_lengths (B,)).  grid_subsample.py:7-22."""
    s_points, s_lengths = ext.grid_subsampling(points, lengths, voxel_size, order=order)
    return s_points, s_lengths


def radius_search(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid=None, contiguous=True):
    """Radius search in stack mode -> (N, min(max_count, neighbor_limit)) int64, padded with M.

    radius_search.py:7-27.  The reference materialises the full width and returns a NON-contiguous
    column slice (radius_search.py:26); here the truncation happens inside the fill kernel, so only
    the kept columns are ever written and the result is contiguous (what KPConv's index_select
    needs once the CPU->GPU copy that used to re-densify it is gone -- SURVEY.md section 8b).
    `grid` (optional ext.SupportGrid): reuse the support binning across searches over the same supports and radius.
    `contiguous=False`: where the rows were searched wider than the result, return the column slice (as the reference does)
    instead of a dense copy."""
    for t, n in ((q_points, "q_points"), (s_points, "s_points")):
        ext._check_points(t, n)
        ext._check_float(t, n)
        ext._check_contig(t, n)
    for t, n in ((q_lengths, "q_lengths"), (s_lengths, "s_lengths")):
        ext._check_long(t, n)
        ext._check_contig(t, n)
    return ext.radius_neighbors_limited(q_points, s_points, q_lengths, s_lengths, radius, neighbor_limit, grid=grid,
                                        contiguous=contiguous)


# ---------------------------------------------------------------------------------------------
# dense / partition operators (reference: modules/ops/pairwise_distance.py, pointcloud_partition.py)
import ctypes  # noqa: E402

import torch  # noqa: E402

from . import _lib  # noqa: E402


def _cuda_f32(t, name):
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be a float tensor")
    dev = _lib.require_gpu()
    return (t if t.is_cuda else t.to(dev)).contiguous()


def pairwise_distance(x, y, normalized=False, channel_first=False):
    """pairwise_distance.py:4-31.  (*, N, C) x (*, M, C) -> (*, N, M); the leading dims are one launch (grid.z)."""
    if channel_first:
        x, y = x.transpose(-1, -2), y.transpose(-1, -2)
    out_device = x.device
    lead = x.shape[:-2]
    if y.shape[:-2] != lead:
        raise RuntimeError("pairwise_distance: x and y must have the same leading (batch) dimensions")
    xs = _cuda_f32(x, "x").reshape(-1, x.shape[-2], x.shape[-1])
    ys = _cuda_f32(y, "y").reshape(-1, y.shape[-2], y.shape[-1])
    L = _lib.lib()
    B, N, C = xs.shape
    M = ys.shape[1]
    out = torch.empty((B, N, M), dtype=torch.float32, device=xs.device)
    with torch.cuda.device(xs.device):
        ws = _lib.workspace(xs.device, L.gr_pairwise_distance_batch_workspace_bytes(B, N, M))
        _lib.check(L.gr_pairwise_distance_batch(_lib.ptr(xs), _lib.ptr(ys), B, N, M, C, int(bool(normalized)), _lib.ptr(out),
                                                _lib.ptr(ws), ws.numel(), _lib.stream_ptr(xs.device)))
    out = out.reshape(*lead, N, M)
    return out if out_device.type == "cuda" else out.to(out_device)


@torch.no_grad()
def point_to_node_partition(points, nodes, point_limit, return_count=False):
    """pointcloud_partition.py:61-111 without the (M, N) matrix.  Returns
    (point_to_node (N,), [node_sizes (M,)], node_masks (M,), node_knn_indices (M,K), node_knn_masks (M,K))."""
    out_device = points.device
    p = _cuda_f32(points, "points")
    nd = _cuda_f32(nodes, "nodes")
    L = _lib.lib()
    N, M, K = p.shape[0], nd.shape[0], int(point_limit)
    dev = p.device
    p2n = torch.empty((N,), dtype=torch.int64, device=dev)
    masks = torch.empty((M,), dtype=torch.bool, device=dev)
    knn_idx = torch.empty((M, K), dtype=torch.int64, device=dev)
    knn_masks = torch.empty((M, K), dtype=torch.bool, device=dev)
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, L.gr_point_to_node_workspace_bytes(N, M))
        _lib.check(L.gr_point_to_node_partition(_lib.ptr(p), N, _lib.ptr(nd), M, K, _lib.ptr(p2n), _lib.ptr(masks),
                                                _lib.ptr(knn_idx), _lib.ptr(knn_masks), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr(dev)))
    res = [p2n, masks, knn_idx, knn_masks]
    if return_count:
        res.insert(1, torch.bincount(p2n, minlength=M))
    if out_device.type != "cuda":
        res = [r.to(out_device) for r in res]
    return tuple(res)


@torch.no_grad()
def point_to_node_partition_batch(points, point_lengths, nodes, node_lengths, point_limit):
    """point_to_node_partition (pointcloud_partition.py:61-111) for every (cloud, superpoints) pair of a stack in one call
    (gr_point_to_node_partition_batch): `points` (sum N_c, 3) / `nodes` (sum M_c, 3) with per-cloud lengths.  Returns
    (point_to_node (sum N,), node_masks (sum M,), node_knn_indices (sum M, K), node_knn_masks (sum M, K)); indices are LOCAL
    to their cloud, exactly what the single-cloud call returns for it."""
    p = _cuda_f32(points, "points")
    nd = _cuda_f32(nodes, "nodes")
    L = _lib.lib()
    K = int(point_limit)
    dev = p.device
    po, no = [0], [0]
    for n in point_lengths:
        po.append(po[-1] + int(n))
    for m in node_lengths:
        no.append(no[-1] + int(m))
    if len(po) != len(no) or po[-1] != p.shape[0] or no[-1] != nd.shape[0]:
        raise ValueError("lengths do not match the stacked tensors")
    nclouds = len(po) - 1
    p2n = torch.empty((po[-1],), dtype=torch.int64, device=dev)
    masks = torch.empty((no[-1],), dtype=torch.bool, device=dev)
    knn_idx = torch.empty((no[-1], K), dtype=torch.int64, device=dev)
    knn_masks = torch.empty((no[-1], K), dtype=torch.bool, device=dev)
    h_po, h_no = _lib.host_i64(po), _lib.host_i64(no)
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, L.gr_point_to_node_batch_workspace_bytes(h_po, h_no, nclouds))
        _lib.check(L.gr_point_to_node_partition_batch(_lib.ptr(p), h_po, _lib.ptr(nd), h_no, nclouds, K, _lib.ptr(p2n),
                                                      _lib.ptr(masks), _lib.ptr(knn_idx), _lib.ptr(knn_masks), _lib.ptr(ws),
                                                      ws.numel(), _lib.stream_ptr(dev)))
    return p2n, masks, knn_idx, knn_masks


_gather_flags = {}


def index_select(data, index, dim):
    """Advanced index select (modules/ops/index_select.py:4-31): `index` may have any shape; the `dim`-th dimension of
    `data` is replaced by index.shape.  The case the backbone runs (dim 0 of a 2-D fp32 feature / point tensor:
    kpconv.py:93-105, functional.py:17,63) is a HIP row gather; other dims / dtypes are reshaped onto it or, for
    non-float data, use torch's own gather on the device."""
    if not isinstance(index, torch.Tensor) or index.dtype != torch.int64:
        raise RuntimeError("index must be a LongTensor")
    nd = data.dim()
    dim = dim + nd if dim < 0 else dim
    if not (0 <= dim < nd):
        raise IndexError("dim out of range")
    if data.dtype != torch.float32:
        out = data.index_select(dim, index.reshape(-1))
    else:
        dev = _lib.require_gpu()
        out_device = data.device
        d = data if data.is_cuda else data.to(dev)
        dev = d.device
        # bring `dim` to the front and flatten the rest: (a_dim, rest)
        moved = d.movedim(dim, 0).contiguous()
        n = moved.shape[0]
        rest = moved.shape[1:]
        c = 1
        for r in rest:
            c *= int(r)
        flat = moved.reshape(n, max(c, 1))
        idx = index.to(dev).reshape(-1).contiguous()
        m = idx.shape[0]
        res = torch.empty((m, max(c, 1)), dtype=torch.float32, device=dev)
        key = (dev.type, dev.index)
        flag = _gather_flags.get(key)
        if flag is None:
            flag = torch.zeros(1, dtype=torch.int32, device=dev)
            _gather_flags[key] = flag
        if m > 0 and c > 0:
            L = _lib.lib()
            eager = m <= 65536
            if eager:
                # a small gather is checked eagerly like torch's: its own flag word, so an earlier LARGE gather's pending
                # error is neither raised here nor lost (that one poisons its rows with NaN and is reported by
                # gather_error_pending)
                own = torch.zeros(1, dtype=torch.int32, device=dev)
            with torch.cuda.device(dev):
                _lib.check(L.gr_gather_rows(_lib.ptr(flat), n, max(c, 1), _lib.ptr(idx), m, _lib.ptr(res),
                                            _lib.ptr(own if eager else flag), _lib.stream_ptr(dev)))
            if eager and int(own.item()) != 0:
                raise IndexError("index out of range in index_select")
        out = res.reshape((m,) + tuple(rest)).movedim(0, dim)
        if out_device.type != "cuda":
            out = out.to(out_device)
        out = out.contiguous()
    if index.dim() > 1:
        out = out.view(*(tuple(data.shape[:dim]) + tuple(index.shape) + tuple(data.shape[dim:][1:])))
    return out


def gather_error_pending(device=None):
    """True if a large index_select (more than 65 536 rows) on `device` met an out-of-range index since the last check.  torch
    raises eagerly; the HIP gather records it on the device instead of synchronising every call, and fills the offending
    rows with NaN so that nothing downstream can mistake them for data.  GeoTransformer.forward checks this at its first
    natural synchronisation point."""
    dev = _lib.require_gpu() if device is None else torch.device(device)
    flag = _gather_flags.get((dev.type, dev.index))
    if flag is None:
        return False
    bad = int(flag.item()) != 0
    flag.zero_()
    return bad
