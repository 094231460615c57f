"""The stack-mode (batch of scene pairs) entry points of the per-pair stages must return exactly what the single-pair entry
points return pair by pair -- same kernels, same arithmetic, same order: bit-equal, floats included.
    gr_point_to_node_partition_batch   vs gr_point_to_node_partition      (model.py:99-104)
    gr_superpoint_matching_batch       vs gr_superpoint_matching          (model.py:156-160)
    gr_lgr_register_seg                vs gr_lgr_register                 (model.py:195-207)
    gr_ransac_similarity_seg           vs gr_ransac_similarity            (model.py:209-220)
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _c(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_point_to_node_partition_batch_equals_single_calls():
    from gaussreg_amd.ops import point_to_node_partition, point_to_node_partition_batch
    rng = np.random.default_rng(3)
    n_pts = [6000, 5000, 130, 20000]
    n_nodes = [150, 90, 3, 700]
    clouds = [(rng.random((n, 3)) * [4, 3, 2.5]).astype(np.float32) for n in n_pts]
    nodes = [c[rng.choice(len(c), m, replace=False)] + rng.normal(0, 0.02, (m, 3)).astype(np.float32) for c, m in zip(clouds, n_nodes)]
    nodes[1][5] = [50, 50, 50]  # a superpoint that owns no point: node mask False, an all-padding row
    got = point_to_node_partition_batch(_c(np.concatenate(clouds)), n_pts, _c(np.concatenate(nodes).astype(np.float32)), n_nodes, 128)
    po, no = np.cumsum([0] + n_pts), np.cumsum([0] + n_nodes)
    for i in range(len(n_pts)):
        want = point_to_node_partition(_c(clouds[i]), _c(nodes[i].astype(np.float32)), 128)
        assert torch.equal(got[0][po[i]:po[i + 1]], want[0]), f"cloud {i}: point_to_node"
        assert torch.equal(got[1][no[i]:no[i + 1]], want[1]), f"cloud {i}: node masks"
        assert torch.equal(got[2][no[i]:no[i + 1]], want[2]), f"cloud {i}: knn indices"
        assert torch.equal(got[3][no[i]:no[i + 1]], want[3]), f"cloud {i}: knn masks"
    assert not bool(got[1][no[1] + 5])


def test_superpoint_matching_batch_equals_single_calls():
    from gaussreg_amd.matching import SuperPointMatching
    g = torch.Generator(device="cuda").manual_seed(5)
    sizes = [700, 650, 40, 300, 10, 12, 768, 767]                       # (ref, src) of four pairs; 10 x 12 < 256 candidates
    feats = torch.nn.functional.normalize(torch.randn(sum(sizes), 256, device="cuda", generator=g), dim=1)
    masks = torch.rand(sum(sizes), device="cuda", generator=g) > 0.1
    spm = SuperPointMatching(256, True)
    ri, si, sc, counts = spm.forward_batch(feats, sizes, masks)
    off = np.cumsum([0] + sizes)
    for b in range(4):
        r0, r1, s1 = off[2 * b], off[2 * b + 1], off[2 * b + 2]
        wr, ws_, wsc = spm(feats[r0:r1], feats[r1:s1], masks[r0:r1], masks[r1:s1])
        assert counts[b] == wr.shape[0]
        assert torch.equal(ri[b, :counts[b]], wr) and torch.equal(si[b, :counts[b]], ws_), f"pair {b}"
        assert torch.equal(sc[b, :counts[b]], wsc), f"pair {b}: scores"
    assert counts[2] < 256 and counts[0] == 256
    # without masks, without dual normalisation
    spm2 = SuperPointMatching(64, False)
    ri, si, sc, counts = spm2.forward_batch(feats, sizes)
    for b in range(4):
        r0, r1, s1 = off[2 * b], off[2 * b + 1], off[2 * b + 2]
        wr, ws_, wsc = spm2(feats[r0:r1], feats[r1:s1])
        assert torch.equal(ri[b, :counts[b]], wr) and torch.equal(si[b, :counts[b]], ws_) and torch.equal(sc[b, :counts[b]], wsc)


def _patches(rng, P, K, angle, bad=()):
    Rm = np.array([[np.cos(angle), 0, np.sin(angle)], [0, 1, 0], [-np.sin(angle), 0, np.cos(angle)]])
    src = rng.random((P, K, 3)) * 3
    ref = src @ Rm.T + [0.1, 0.2, -0.3] + rng.normal(0, 0.01, (P, K, 3))
    logits = rng.normal(size=(P, K, K)) - 8.0
    logits[:, np.arange(K), np.arange(K)] = 4.0 + rng.normal(size=(P, K))
    for p in bad:
        logits[p] = -20.0                                              # no entry passes the confidence threshold
    ls = logits - np.log(np.exp(logits).sum(2, keepdims=True))
    ls = ((ls + logits - np.log(np.exp(logits).sum(1, keepdims=True))) * 0.5).astype(np.float32)
    rm, sm = rng.random((P, K)) > 0.2, rng.random((P, K)) > 0.2
    return ref.astype(np.float32), src.astype(np.float32), rm, sm, ls


def test_lgr_and_ransac_batch_equal_single_calls():
    from gaussreg_amd.matching import LocalGlobalRegistration
    from gaussreg_amd.registration import registration_with_ransac_batch, registration_with_ransac_from_correspondences
    rng = np.random.default_rng(17)
    K = 128
    per_pair = [60, 1, 256, 9]
    parts = [_patches(rng, 60, K, 0.4), _patches(rng, 1, K, 0.1, bad=(0,)),      # pair 1: no correspondence at all
             _patches(rng, 256, K, -0.7), _patches(rng, 9, K, 1.1, bad=range(1, 9))]
    ref, src, rm, sm, ls = (np.concatenate([p[i] for p in parts]) for i in range(5))
    lgr = LocalGlobalRegistration(3, 0.1)
    rc, sc, cs, T, rows = lgr.forward_batch(_c(ref), _c(src), _c(rm), _c(sm), _c(ls), None, per_pair)
    rows_h = rows.cpu().numpy()
    assert rows_h[0] == 0 and rows_h[-1] == rc.shape[0] and rows_h[1] == rows_h[2]  # pair 1 is empty
    poff = np.cumsum([0] + per_pair)
    singles = []
    for b in range(4):
        a, e = poff[b], poff[b + 1]
        w = lgr(_c(ref[a:e]), _c(src[a:e]), _c(rm[a:e]), _c(sm[a:e]), _c(ls[a:e]), None)
        singles.append(w)
        r0, r1 = rows_h[b], rows_h[b + 1]
        assert r1 - r0 == w[0].shape[0], f"pair {b}: {r1 - r0} vs {w[0].shape[0]} correspondences"
        assert torch.equal(rc[r0:r1], w[0]) and torch.equal(sc[r0:r1], w[1]) and torch.equal(cs[r0:r1], w[2]), f"pair {b}"
        assert torch.equal(T[b], w[3]), f"pair {b}: transform\n{T[b]}\n{w[3]}"
    assert torch.equal(T[1], torch.eye(4, device="cuda"))
    # RANSAC with scale on the same correspondences, seed = pair index; pair 1 (0 rows) keeps the fallback
    Tr, stats = registration_with_ransac_batch(sc, rc, rows, fallback_transforms=T, distance_threshold=0.05, ransac_n=5,
                                               num_iterations=2000, seed=0, return_stats=True)
    for b in range(4):
        w = singles[b]
        if w[0].shape[0] >= 5:
            want, wst = registration_with_ransac_from_correspondences(w[1], w[0], None, 0.05, 5, 2000, seed=b, return_stats=True)
            assert torch.equal(Tr[b], want), f"pair {b}: RANSAC transform"
            assert torch.equal(stats[b], wst)
        else:
            assert torch.equal(Tr[b], T[b]) and int(stats[b, 0]) == -1


def test_transformer_padded_batch_equals_pairs_alone():
    """GeometricTransformer over several pairs of different sizes as one padded batch (ref_lengths / src_lengths + masks):
    the rows of the real superpoints are those of each pair alone (up to the summation order of the batched GEMMs)."""
    from gaussreg_amd.transformer import GeometricTransformer
    torch.manual_seed(7)
    net = GeometricTransformer(64, 32, 64, 4, ['self', 'cross', 'self', 'cross'], 0.2, 15, 3, reduction_a='max').cuda().eval()
    g = torch.Generator(device="cuda").manual_seed(1)
    rl, sl = [150, 97, 200], [140, 201, 33]
    pts_r = [torch.rand(n, 3, device="cuda", generator=g) * 3 for n in rl]
    pts_s = [torch.rand(n, 3, device="cuda", generator=g) * 3 for n in sl]
    f_r = [torch.randn(n, 64, device="cuda", generator=g) for n in rl]
    f_s = [torch.randn(n, 64, device="cuda", generator=g) for n in sl]
    from torch.nn.utils.rnn import pad_sequence
    rp, sp = pad_sequence(pts_r, batch_first=True), pad_sequence(pts_s, batch_first=True)
    rf, sf = pad_sequence(f_r, batch_first=True), pad_sequence(f_s, batch_first=True)
    rm = torch.arange(rp.shape[1], device="cuda")[None, :] >= torch.tensor(rl, device="cuda")[:, None]
    sm = torch.arange(sp.shape[1], device="cuda")[None, :] >= torch.tensor(sl, device="cuda")[:, None]
    with torch.no_grad():
        got_r, got_s = net(rp, sp, rf, sf, rm, sm, ref_lengths=rl, src_lengths=sl)
        for b in range(3):
            want_r, want_s = net(pts_r[b][None], pts_s[b][None], f_r[b][None], f_s[b][None])
            scale = float(want_r.abs().max())
            assert float((got_r[b, :rl[b]] - want_r[0]).abs().max()) <= 2e-5 * scale, b
            assert float((got_s[b, :sl[b]] - want_s[0]).abs().max()) <= 2e-5 * scale, b
        assert torch.isfinite(got_r).all() and torch.isfinite(got_s).all()


def test_ransac_thread_per_hypothesis_scores_equal_the_sixteen_lane_kernel():
    """A stack-mode call with >= 131 072 hypotheses scores them one thread each (ransac_score_wide_kernel), a single call with
    fewer takes sixteen lanes per hypothesis: the two must agree to the bit -- the error sums break the ties between
    hypotheses of equal inlier count, and with few correspondences (40 - 1 100 here, not multiples of 16, one of them past a
    1 024-row chunk) nearly every pair is decided by such a tie.  refine=False: the winning hypothesis' own transform."""
    from gaussreg_amd.registration import registration_with_ransac_batch, registration_with_ransac_from_correspondences
    rng = np.random.default_rng(41)
    counts = [40, 57, 333, 1100, 64, 16, 5, 250]
    H = 20000                                                    # 8 x 20 000 = 160 000 hypotheses in the batch call
    srcs, refs = [], []
    for b, c in enumerate(counts):
        R = np.linalg.qr(rng.normal(size=(3, 3)))[0]
        if np.linalg.det(R) < 0:
            R[:, 0] = -R[:, 0]
        src = rng.random((c, 3)) * 2.0
        ref = 1.3 * src @ R.T + rng.normal(0, 0.5, 3) + rng.normal(0, 0.02, (c, 3))
        bad = rng.random(c) < 0.4
        ref[bad] = rng.random((int(bad.sum()), 3)) * 4.0
        srcs.append(src.astype(np.float32)); refs.append(ref.astype(np.float32))
    rows = torch.tensor(np.cumsum([0] + counts), dtype=torch.int32).cuda()
    s, r = _c(np.concatenate(srcs)), _c(np.concatenate(refs))
    for refine in (False, True):
        Tb, stb = registration_with_ransac_batch(s, r, rows, None, 0.05, 5, H, refine=refine, seed=3, return_stats=True)
        for b, c in enumerate(counts):
            want, wst = registration_with_ransac_from_correspondences(_c(srcs[b]), _c(refs[b]), None, 0.05, 5, H, refine=refine,
                                                                      seed=3 + b, return_stats=True)
            assert torch.equal(stb[b], wst), (b, c, stb[b], wst)
            assert torch.equal(Tb[b], want), (b, c)


def test_lgr_stack_mode_with_thousands_of_patches_equals_single_calls():
    """>= 2 048 patches in one stack-mode call: the hypotheses are verified one thread each over slices of the pair's
    correspondences (lgr_verify_wide_kernel: `d2 < x0` instead of `sqrtf(d2) < radius`, integer adds across the slices);
    single calls take the workgroup-per-hypothesis kernel.  Same correspondences, same estimates, bit for bit."""
    from gaussreg_amd.matching import LocalGlobalRegistration
    rng = np.random.default_rng(53)
    K = 128
    per_pair = [256, 256, 200, 256, 256, 1, 256, 256, 256, 256, 77]
    shifts = [0.4, -0.7, 1.1, 0.2, -0.3, 0.1, 0.9, -1.2, 0.6, -0.5, 0.3]
    parts = [_patches(rng, n, K, sft, bad=(0,) if n == 1 else ()) for n, sft in zip(per_pair, shifts)]
    ref, src, rm, sm, ls = (np.concatenate([p[i] for p in parts]) for i in range(5))
    assert ref.shape[0] >= 2048
    lgr = LocalGlobalRegistration(3, 0.1)
    rc, sc, cs, T, rows = lgr.forward_batch(_c(ref), _c(src), _c(rm), _c(sm), _c(ls), None, per_pair)
    rows_h = rows.cpu().numpy()
    poff = np.cumsum([0] + per_pair)
    for b in range(len(per_pair)):
        a, e = poff[b], poff[b + 1]
        w = lgr(_c(ref[a:e]), _c(src[a:e]), _c(rm[a:e]), _c(sm[a:e]), _c(ls[a:e]), None)
        r0, r1 = rows_h[b], rows_h[b + 1]
        assert r1 - r0 == w[0].shape[0], f"pair {b}"
        assert torch.equal(rc[r0:r1], w[0]) and torch.equal(sc[r0:r1], w[1]) and torch.equal(cs[r0:r1], w[2]), f"pair {b}"
        assert torch.equal(T[b], w[3]), f"pair {b}: transform\n{T[b]}\n{w[3]}"
