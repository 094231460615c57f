"""TEST INFRASTRUCTURE ONLY -- the product path never imports this.

NumPy restatement of the RPE rows:
  geometric_structure_embedding   geotransformer/modules/geotransformer/geotransformer.py:26-73
  sinusoidal_embedding            geotransformer/modules/transformer/positional_embedding.py:21-34
  rpe_multi_head_attention        geotransformer/modules/transformer/rpe_transformer.py:34-72
Pinned against tests/golden/rpe.npz (outputs of the reference's own modules, tests/golden/gen_golden_rpe.py).
"""
import numpy as np

f32 = np.float32


def pairwise_distance(x, y):
    """geotransformer/modules/ops/pairwise_distance.py:21-31 (not normalised): clamp(x2 - 2xy + y2, 0)."""
    x, y = np.asarray(x, f32), np.asarray(y, f32)
    xy = x @ y.T
    x2 = np.sum(x * x, axis=-1, dtype=f32)[:, None]
    y2 = np.sum(y * y, axis=-1, dtype=f32)[None, :]
    return np.maximum(x2 - f32(2.0) * xy + y2, f32(0.0))


def embedding_indices(points, sigma_d, sigma_a, angle_k):
    """geotransformer.py:26-55 for one cloud (N,3) -> d_indices (N,N), a_indices (N,N,k)."""
    p = np.asarray(points, f32)
    dist = np.sqrt(pairwise_distance(p, p))
    d_idx = dist / f32(sigma_d)
    knn = np.argsort(dist, axis=1, kind="stable")[:, 1:angle_k + 1]      # topk(k+1, smallest)[1][:, 1:]
    ref = p[knn] - p[:, None, :]                                          # (N,k,3)
    anc = p[None, :, :] - p[:, None, :]                                   # (N,N,3)
    ref_e = ref[:, None, :, :]                                            # (N,1,k,3)
    anc_e = anc[:, :, None, :]                                            # (N,N,1,3)
    cross = np.cross(np.broadcast_to(ref_e, (p.shape[0], p.shape[0], angle_k, 3)),
                     np.broadcast_to(anc_e, (p.shape[0], p.shape[0], angle_k, 3))).astype(f32)
    sin_v = np.sqrt(np.sum(cross * cross, axis=-1, dtype=f32))
    cos_v = np.sum(ref_e * anc_e, axis=-1, dtype=f32)
    a_idx = np.arctan2(sin_v, cos_v).astype(f32) * f32(180.0 / (sigma_a * np.pi))
    return d_idx.astype(f32), a_idx.astype(f32)


def sinusoidal_embedding(idx, div_term):
    om = np.asarray(idx, f32)[..., None] * np.asarray(div_term, f32)
    emb = np.stack([np.sin(om), np.cos(om)], axis=-1)                     # (*, C/2, 2): (sin, cos) interleaved
    return emb.reshape(*np.shape(idx), -1).astype(f32)


def geometric_structure_embedding(points, w_d, b_d, w_a, b_a, div_term, sigma_d, sigma_a, angle_k, reduction="max"):
    d_idx, a_idx = embedding_indices(points, sigma_d, sigma_a, angle_k)
    d_emb = sinusoidal_embedding(d_idx, div_term) @ np.asarray(w_d, f32).T + np.asarray(b_d, f32)
    a_emb = sinusoidal_embedding(a_idx, div_term) @ np.asarray(w_a, f32).T + np.asarray(b_a, f32)
    a_emb = a_emb.max(axis=2) if reduction == "max" else a_emb.mean(axis=2, dtype=f32)
    return (d_emb + a_emb).astype(f32)


def rpe_multi_head_attention(sd, num_heads, xq, xk, xv, emb, key_weights=None, key_masks=None, attention_factors=None):
    """rpe_transformer.py:51-72 for one batch element: xq (N,C), xk/xv (M,C), emb (N,M,C) ->
    hidden (N,C), scores (H,N,M)."""
    lin = lambda x, n: np.asarray(x, f32) @ sd[n + ".weight"].T + sd[n + ".bias"]
    N, C = xq.shape
    M = xk.shape[0]
    ch = C // num_heads
    q = lin(xq, "proj_q").reshape(N, num_heads, ch).transpose(1, 0, 2)
    k = lin(xk, "proj_k").reshape(M, num_heads, ch).transpose(1, 0, 2)
    v = lin(xv, "proj_v").reshape(M, num_heads, ch).transpose(1, 0, 2)
    p = lin(emb, "proj_p").reshape(N, M, num_heads, ch).transpose(2, 0, 1, 3)
    s = (np.einsum("hnc,hmc->hnm", q, k) + np.einsum("hnc,hnmc->hnm", q, p)) / f32(ch ** 0.5)
    if attention_factors is not None:
        s = attention_factors[None] * s
    if key_weights is not None:
        s = s * key_weights[None, None, :]
    if key_masks is not None:
        s = np.where(key_masks[None, None, :], -np.inf, s)
    s = s - s.max(axis=-1, keepdims=True)
    e = np.exp(s)
    s = (e / e.sum(axis=-1, keepdims=True)).astype(f32)
    h = np.einsum("hnm,hmc->hnc", s, v).transpose(1, 0, 2).reshape(N, C)
    return h.astype(f32), s
