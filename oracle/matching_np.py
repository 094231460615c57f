"""TEST INFRASTRUCTURE ONLY.  NumPy fp32 restatement of the reference's pure-PyTorch matching ops.

Pinned against tests/golden/matching.npz, which was produced by importing the reference's own
modules (tests/golden/gen_golden_matching.py).  Float paths: the reference runs ATen CPU/GPU
kernels whose summation order is not specified, so parity is asserted to 1e-5 relative on values
and exactly on indices wherever score gaps exceed the fp noise.

  pairwise_distance           geotransformer/modules/ops/pairwise_distance.py:4-31
  superpoint_matching         geotransformer/modules/geotransformer/superpoint_matching.py:13-50
  correspondence_matrix       geotransformer/modules/geotransformer/point_matching.py:32-66
                              (== local_global_registration.py:49-83)
  point_matching              geotransformer/modules/geotransformer/point_matching.py:68-115
  point_to_node_partition     geotransformer/modules/ops/pointcloud_partition.py:61-111
"""
import numpy as np

f32 = np.float32


def pairwise_distance(x, y, normalized=False):
    """pairwise_distance.py:19-31 (channel-last): clamp(x2 - 2 xy + y2, 0) or clamp(2 - 2 xy, 0)."""
    x, y = np.asarray(x, f32), np.asarray(y, f32)
    xy = x @ y.T
    if normalized:
        d = f32(2.0) - f32(2.0) * xy
    else:
        x2 = (x * x).sum(-1, dtype=f32)[:, None]
        y2 = (y * y).sum(-1, dtype=f32)[None, :]
        d = x2 - f32(2.0) * xy + y2
    return np.maximum(d, f32(0.0)).astype(f32)


def superpoint_matching(ref_feats, src_feats, ref_masks=None, src_masks=None, num_correspondences=256,
                        dual_normalization=True):
    """superpoint_matching.py:32-48.  Ties in the global top-k are broken by ascending flat index."""
    ref_feats, src_feats = np.asarray(ref_feats, f32), np.asarray(src_feats, f32)
    if ref_masks is None:
        ref_masks = np.ones(ref_feats.shape[0], bool)
    if src_masks is None:
        src_masks = np.ones(src_feats.shape[0], bool)
    ref_indices = np.nonzero(ref_masks)[0]
    src_indices = np.nonzero(src_masks)[0]
    rf, sf = ref_feats[ref_indices], src_feats[src_indices]
    s = np.exp(-pairwise_distance(rf, sf, normalized=True)).astype(f32)
    if dual_normalization:
        rs = s / s.sum(1, keepdims=True, dtype=f32)
        cs = s / s.sum(0, keepdims=True, dtype=f32)
        s = (rs * cs).astype(f32)
    k = min(num_correspondences, s.size)
    flat = s.reshape(-1)
    order = np.lexsort((np.arange(flat.size), -flat))[:k]  # descending score, ascending index
    scores = flat[order]
    return ref_indices[order // s.shape[1]], src_indices[order % s.shape[1]], scores, s


def _topk_mask(a, k, axis):
    """True at the k largest entries along `axis` (ties: lowest index first)."""
    n = a.shape[axis]
    idx = np.argsort(-a, axis=axis, kind="stable")
    take = np.take(idx, np.arange(min(k, n)), axis=axis)
    m = np.zeros(a.shape, bool)
    np.put_along_axis(m, take, True, axis=axis)
    return m


def correspondence_matrix(score_mat_exp, ref_knn_masks, src_knn_masks, k=3, mutual=True, confidence_threshold=0.05):
    """point_matching.py:32-66: (row top-k & > thr) AND/OR (col top-k & > thr), AND validity mask."""
    s = np.asarray(score_mat_exp, f32)
    mask = ref_knn_masks[:, :, None] & src_knn_masks[:, None, :]
    ref_corr = _topk_mask(s, k, 2) & (s > f32(confidence_threshold))
    src_corr = _topk_mask(s, k, 1) & (s > f32(confidence_threshold))
    corr = (ref_corr & src_corr) if mutual else (ref_corr | src_corr)
    return corr & mask


def point_matching(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, ref_knn_indices, src_knn_indices,
                   score_mat, global_scores, k=3, mutual=True, confidence_threshold=0.05, use_global_score=False):
    """point_matching.py:96-115 (use_dustbin=False)."""
    s = np.exp(np.asarray(score_mat, f32)).astype(f32)
    corr = correspondence_matrix(s, ref_knn_masks, src_knn_masks, k, mutual, confidence_threshold)
    if use_global_score:
        s = (s * np.asarray(global_scores, f32)[:, None, None]).astype(f32)
    b, i, j = np.nonzero(corr)
    return (ref_knn_points[b, i], src_knn_points[b, j], ref_knn_indices[b, i], src_knn_indices[b, j], s[b, i, j], corr)


def point_to_node_partition(points, nodes, point_limit):
    """pointcloud_partition.py:84-102.  Returns point_to_node (N,), node_masks (M,),
    node_knn_indices (M,K) padded with N, node_knn_masks (M,K)."""
    points, nodes = np.asarray(points, f32), np.asarray(nodes, f32)
    N, M = points.shape[0], nodes.shape[0]
    d = pairwise_distance(nodes, points)  # (M, N)
    p2n = d.argmin(0)
    node_masks = np.zeros(M, bool)
    node_masks[p2n] = True
    d = d.copy()
    own = np.zeros_like(d, bool)
    own[p2n, np.arange(N)] = True
    d[~own] = f32(1e12)
    K = point_limit
    order = np.argsort(d, axis=1, kind="stable")[:, :K]
    knn_masks = p2n[order] == np.arange(M)[:, None]
    knn_idx = np.where(knn_masks, order, N)
    return p2n, node_masks, knn_idx, knn_masks, d


# ------------------------------------------------------------------------------------------------
# "next" rows
def _logsumexp(a, axis):
    m = a.max(axis=axis, keepdims=True)
    return (np.log(np.exp(a - m).sum(axis=axis, dtype=f32, keepdims=True)) + m).squeeze(axis).astype(f32)


def sinkhorn(scores, row_masks=None, col_masks=None, alpha=1.0, num_iterations=100, inf=1e12):
    """geotransformer/modules/sinkhorn/learnable_sinkhorn.py:20-66 (forward) in fp32 NumPy."""
    s = np.asarray(scores, f32)
    B, M, N = s.shape
    rm = np.ones((B, M), bool) if row_masks is None else np.asarray(row_masks, bool)
    cm = np.ones((B, N), bool) if col_masks is None else np.asarray(col_masks, bool)
    prm = np.zeros((B, M + 1), bool)
    prm[:, :M] = ~rm
    pcm = np.zeros((B, N + 1), bool)
    pcm[:, :N] = ~cm
    P = np.full((B, M + 1, N + 1), f32(alpha), f32)
    P[:, :M, :N] = s
    P[prm[:, :, None] | pcm[:, None, :]] = f32(-inf)
    nvr, nvc = rm.sum(1).astype(f32), cm.sum(1).astype(f32)
    norm = -np.log(nvr + nvc).astype(f32)
    log_mu = np.empty((B, M + 1), f32)
    log_mu[:, :M] = norm[:, None]
    log_mu[:, M] = np.log(nvc) + norm
    log_mu[prm] = f32(-inf)
    log_nu = np.empty((B, N + 1), f32)
    log_nu[:, :N] = norm[:, None]
    log_nu[:, N] = np.log(nvr) + norm
    log_nu[pcm] = f32(-inf)
    u, v = np.zeros_like(log_mu), np.zeros_like(log_nu)
    for _ in range(num_iterations):
        u = log_mu - _logsumexp(P + v[:, None, :], 2)
        v = log_nu - _logsumexp(P + u[:, :, None], 1)
    return (P + u[:, :, None] + v[:, None, :] - norm[:, None, None]).astype(f32)


def kpconv(s_feats, q_points, s_points, neighbor_indices, kernel_points, weights, sigma, bias=None, inf=1e6):
    """geotransformer/modules/kpconv/kpconv.py:90-120 in fp32 NumPy."""
    s_feats, q_points, s_points = np.asarray(s_feats, f32), np.asarray(q_points, f32), np.asarray(s_points, f32)
    sp = np.concatenate([s_points, np.full((1, 3), inf, f32)], 0)
    nb = sp[neighbor_indices] - q_points[:, None, :]                      # (M, H, 3)
    diff = nb[:, :, None, :] - np.asarray(kernel_points, f32)[None, None]  # (M, H, K, 3)
    sq = (diff * diff).sum(-1, dtype=f32)
    w = np.maximum(f32(1.0) - np.sqrt(sq) / f32(sigma), f32(0.0)).astype(f32)  # (M, H, K)
    sf = np.concatenate([s_feats, np.zeros((1, s_feats.shape[1]), f32)], 0)
    nf = sf[neighbor_indices]                                              # (M, H, C)
    wf = np.einsum("mhk,mhc->mkc", w, nf).astype(f32)
    o = np.einsum("mkc,kco->mo", wf, np.asarray(weights, f32)).astype(f32)
    num = np.maximum((nf.sum(-1, dtype=f32) > 0).sum(-1), 1)
    o = o / num[:, None].astype(f32)
    if bias is not None:
        o = o + np.asarray(bias, f32)
    return o.astype(f32)


def maxpool(x, neighbor_indices):
    x = np.concatenate([np.asarray(x, f32), np.zeros((1, x.shape[1]), f32)], 0)
    return x[neighbor_indices].max(1)


def nearest_upsample(x, upsample_indices):
    x = np.concatenate([np.asarray(x, f32), np.zeros((1, x.shape[1]), f32)], 0)
    return x[upsample_indices[:, 0]]


# ------------------------------------------------------------------------------------------------
def weighted_procrustes(src, ref, w, eps=1e-5):
    """geotransformer/modules/registration/procrustes.py:41-66 (single problem) -> 4x4 float64."""
    w = np.where(w < 0, 0, w).astype(np.float64)
    w = w / (w.sum() + eps)
    cs, cr = (src * w[:, None]).sum(0), (ref * w[:, None]).sum(0)
    H = (src - cs).T @ (w[:, None] * (ref - cr))
    U, _, Vt = np.linalg.svd(H)
    V = Vt.T
    D = np.eye(3)
    D[2, 2] = np.sign(np.linalg.det(V @ U.T))
    R = V @ D @ U.T
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = cr - R @ cs
    return T


def local_global_registration(ref_knn_points, src_knn_points, ref_knn_masks, src_knn_masks, score_mat, k=3,
                              acceptance_radius=0.1, mutual=True, confidence_threshold=0.05,
                              correspondence_threshold=3, num_refinement_steps=5):
    """local_global_registration.py:135-235 (use_dustbin=False, use_global_score=False, no limit)."""
    s = np.exp(np.asarray(score_mat, f32)).astype(f32)
    corr = correspondence_matrix(s, ref_knn_masks, src_knn_masks, k, mutual, confidence_threshold)
    b, i, j = np.nonzero(corr)
    ref, src, sc = ref_knn_points[b, i].astype(np.float64), src_knn_points[b, j].astype(np.float64), s[b, i, j]

    def inl(T):
        return np.linalg.norm(ref - (src @ T[:3, :3].T + T[:3, 3]), axis=1) < acceptance_radius

    starts = np.concatenate([[0], np.nonzero(b[1:] != b[:-1])[0] + 1, [len(b)]])
    chunks = [(x, y) for x, y in zip(starts[:-1], starts[1:]) if y - x >= correspondence_threshold]
    if chunks:
        Ts = [weighted_procrustes(src[x:y], ref[x:y], sc[x:y]) for x, y in chunks]
        counts = [int(inl(T).sum()) for T in Ts]
        cur = sc * inl(Ts[int(np.argmax(counts))])
    else:
        cur = sc * inl(weighted_procrustes(src, ref, sc))
    T = weighted_procrustes(src, ref, cur)
    for _ in range(num_refinement_steps - 1):
        T = weighted_procrustes(src, ref, sc * inl(T))
    return ref.astype(f32), src.astype(f32), sc, T.astype(f32)


def farthest_point_sampling(points, k, start=0):
    """Exact FPS, fp32 ((dx*dx + dy*dy) + dz*dz), first maximum on ties (contract of gr_fps; fpsample unpinned)."""
    p = np.asarray(points, f32)
    d = np.full(p.shape[0], np.inf, f32)
    out = [int(start)]
    for _ in range(1, k):
        diff = p - p[out[-1]]
        dd = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
        d = np.minimum(d, dd)
        out.append(int(np.argmax(d)))
    return np.array(out, np.int64)
