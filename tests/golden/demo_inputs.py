"""Seeded input builders shared by tests/golden/gen_golden_demo.py / gen_golden_backbone.py (which run the reference on
them, build container only) and the tests (which run the HIP path on them).  torch CPU generators and numpy default_rng
are deterministic for a given library version; the golden files carry fp64 checksums of these inputs."""
import math

import numpy as np

K015 = np.array([
    [0.0, 0.0, 0.0], [-0.49820612, 0.41826797, 0.11736718], [-0.24123565, -0.34214048, -0.5115481],
    [-0.2828808, -0.58614266, 0.11553228], [0.29054036, -0.10093209, -0.585091], [0.42820039, 0.39929883, -0.30681813],
    [-0.63586493, -0.08196441, -0.16090403], [-0.43181082, -0.14729417, 0.47830957], [-0.044666, 0.27973214, 0.59723308],
    [0.22552417, -0.34462544, 0.50794659], [0.63889212, -0.16914906, -0.01190108], [-0.22552415, 0.34462545, -0.50794659],
    [0.49054666, 0.26880703, 0.35219206], [0.25233084, -0.59706653, -0.12951142], [0.03415394, 0.65858341, 0.04513958]])


def checksum_i64(a):
    """Position-weighted 64-bit checksum of an integer array (wraps mod 2^64; order sensitive)."""
    a = np.ascontiguousarray(a).astype(np.uint64).reshape(-1)
    w = (np.arange(a.size, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345)) | np.uint64(1)
    with np.errstate(over="ignore"):
        return np.uint64((a * w).sum(dtype=np.uint64))


def point_matching_inputs():
    import torch
    g = torch.Generator().manual_seed(1234)
    P, K = 256, 128
    logits = torch.randn(P, K, K, generator=g) * 3.0
    score = (torch.log_softmax(logits, dim=2) + torch.log_softmax(logits, dim=1)) * 0.5
    rk = torch.rand(P, K, generator=g) > 0.2
    sk = torch.rand(P, K, generator=g) > 0.2
    rp, sp_ = torch.randn(P, K, 3, generator=g), torch.randn(P, K, 3, generator=g)
    rki = torch.randint(0, 30000, (P, K), generator=g)
    ski = torch.randint(0, 30000, (P, K), generator=g)
    gs = torch.rand(P, generator=g)
    return dict(score=score, ref_masks=rk, src_masks=sk, ref_points=rp, src_points=sp_, ref_idx=rki, src_idx=ski, global_scores=gs)


def sinkhorn_inputs():
    import torch
    g = torch.Generator().manual_seed(4321)
    P, K = 256, 128
    sc = torch.randn(P, K, K, generator=g) * 2.0
    rm = torch.rand(P, K, generator=g) > 0.25
    cm = torch.rand(P, K, generator=g) > 0.25
    return sc, rm, cm


def kpconv_inputs(n_points, cin=64, cout=64):
    import torch
    g = torch.Generator().manual_seed(99)
    feats = torch.nn.functional.leaky_relu(torch.randn(n_points, cin, generator=g), 0.1)
    w = torch.randn(15, cin, cout, generator=g) * 0.05
    return feats, w


def lgr_limit_inputs():
    import torch
    g = torch.Generator().manual_seed(77)
    Pn, Kn = 64, 64
    ang = 0.5
    Rm = torch.tensor([[math.cos(ang), 0.0, math.sin(ang)], [0.0, 1.0, 0.0], [-math.sin(ang), 0.0, math.cos(ang)]])
    tv = torch.tensor([-0.4, 0.25, 0.1])
    src_k = torch.rand(Pn, Kn, 3, generator=g) * 2.0
    # elementwise on purpose (no BLAS call whose summation order may depend on the host): x_ref = R x_src + t + noise
    rot = torch.stack([src_k[..., 0] * Rm[r, 0] + src_k[..., 1] * Rm[r, 1] + src_k[..., 2] * Rm[r, 2] for r in range(3)], dim=-1)
    ref_k = rot + tv + 0.01 * torch.randn(Pn, Kn, 3, generator=g)
    perm = torch.stack([torch.randperm(Kn, generator=g) for _ in range(Pn)])
    ref_k = torch.gather(ref_k, 1, perm[:, :, None].expand(-1, -1, 3))
    lg = -8.0 * torch.ones(Pn, Kn, Kn) + torch.randn(Pn, Kn, Kn, generator=g)
    lg[torch.arange(Pn)[:, None], torch.arange(Kn)[None, :], perm] = 4.0 + torch.randn(Pn, Kn, generator=g)
    bad = torch.rand(Pn, generator=g) < 0.3
    lg[bad] = torch.randn(int(bad.sum()), Kn, Kn, generator=g) * 3
    lscore = (torch.log_softmax(lg, 2) + torch.log_softmax(lg, 1)) * 0.5
    rk2, sk2 = torch.rand(Pn, Kn, generator=g) > 0.1, torch.rand(Pn, Kn, generator=g) > 0.1
    gsc = torch.rand(Pn, generator=g)
    return ref_k, src_k, rk2, sk2, lscore, gsc


def backbone_feats(n_points):
    import torch
    g = torch.Generator().manual_seed(5)
    return torch.rand(n_points, 4, generator=g)


def tie_rows(q, s, nb):
    """Rows of a neighbour tensor in which two consecutive valid neighbours have EQUAL fp32 distance (the reference's
    order inside such a run depends on its kd-tree traversal).  Distances in the reference's own
    arithmetic: ((dx*dx + dy*dy) + dz*dz) in fp32."""
    q = np.asarray(q, np.float32)
    s = np.asarray(s, np.float32)
    nb = np.asarray(nb)
    pad = s.shape[0]
    s2 = np.concatenate([s, np.full((1, 3), np.float32(1e18))], 0)
    d = q[:, None, :] - s2[nb]
    d2 = d * d
    dist = (d2[..., 0] + d2[..., 1]) + d2[..., 2]
    valid = nb != pad
    eq = (dist[:, 1:] == dist[:, :-1]) & valid[:, 1:] & valid[:, :-1]
    return np.nonzero(eq.any(axis=1))[0]


def transformer_inputs(n_ref, n_src, c_in, seed=17):
    """Superpoint-level inputs of the GeometricTransformer: two clouds in a 4 x 3 x 2.5 m room (the second one a rigid
    motion of a jittered subset-like resample), features ~ N(0, 0.25) as the coarsest backbone stage delivers them."""
    import torch
    g = torch.Generator().manual_seed(seed)
    box = torch.tensor([4.0, 3.0, 2.5])
    ref = torch.rand(1, n_ref, 3, generator=g) * box
    src = torch.rand(1, n_src, 3, generator=g) * box
    a = 0.4
    rot = torch.tensor([[np.cos(a), -np.sin(a), 0.0], [np.sin(a), np.cos(a), 0.0], [0.0, 0.0, 1.0]], dtype=torch.float32)
    src = (src.reshape(-1, 3, 1) * rot.t().reshape(1, 3, 3)).sum(1).reshape(1, n_src, 3) + torch.tensor([0.3, -0.2, 0.1])
    rf = torch.randn(1, n_ref, c_in, generator=g) * 0.5
    sf = torch.randn(1, n_src, c_in, generator=g) * 0.5
    return ref.contiguous(), src.contiguous(), rf, sf
