"""GPU parity: HIP matching operators (through the C ABI) vs golden vectors produced by the
reference's own Python modules, and vs oracle/matching_np.py on seeded inputs.
Tolerance: values 1e-5 relative (GPU/CPU summation order differs), indices exact (the fixtures have
no score ties within fp noise)."""
import numpy as np
import pytest
import torch

from helpers import load_golden

pytestmark = pytest.mark.gpu


def _c(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_pairwise_distance_vs_reference_golden():
    from gaussreg_amd.ops import pairwise_distance
    g = load_golden("matching.npz")
    d = pairwise_distance(_c(g["pd_x"]), _c(g["pd_y"])).cpu().numpy()
    np.testing.assert_allclose(d, g["pd_plain"], rtol=1e-5, atol=2e-5)
    dn = pairwise_distance(_c(g["pd_xn"]), _c(g["pd_yn"]), normalized=True).cpu().numpy()
    np.testing.assert_allclose(dn, g["pd_normalized"], rtol=1e-5, atol=2e-6)
    # batched + channel_first
    x = torch.randn(3, 17, 50, device="cuda")
    y = torch.randn(3, 17, 70, device="cuda")
    d3 = pairwise_distance(x, y, channel_first=True)
    ref = ((x.transpose(1, 2)[:, :, None, :] - y.transpose(1, 2)[:, None, :, :]) ** 2).sum(-1)
    torch.testing.assert_close(d3, ref, rtol=1e-4, atol=1e-4)


def test_superpoint_matching_vs_reference_golden():
    from gaussreg_amd.matching import SuperPointMatching
    g = load_golden("matching.npz")
    ri, si, sc = SuperPointMatching(256, True)(_c(g["spm_ref"]), _c(g["spm_src"]), _c(g["spm_ref_masks"]),
                                               _c(g["spm_src_masks"]))
    assert ri.dtype == torch.int64 and sc.dtype == torch.float32 and ri.shape == (256,)
    np.testing.assert_allclose(sc.cpu().numpy(), g["spm_scores"], rtol=1e-5)
    assert np.array_equal(ri.cpu().numpy(), g["spm_ref_idx"])
    assert np.array_equal(si.cpu().numpy(), g["spm_src_idx"])
    ri, si, sc = SuperPointMatching(64, False)(_c(g["spm_ref"]), _c(g["spm_src"]))
    np.testing.assert_allclose(sc.cpu().numpy(), g["spm_nodual_scores"], rtol=1e-5)
    assert np.array_equal(ri.cpu().numpy(), g["spm_nodual_ref_idx"])
    assert np.array_equal(si.cpu().numpy(), g["spm_nodual_src_idx"])


@pytest.mark.parametrize("nr,ns,k", [(767, 801, 256), (40, 3, 256), (1500, 1200, 256)])
def test_superpoint_matching_vs_oracle(nr, ns, k):
    from gaussreg_amd.matching import SuperPointMatching
    from oracle import matching_np as M
    rng = np.random.default_rng(nr)
    base = rng.normal(size=(max(nr, ns), 256)).astype(np.float32)
    ref = base[:nr] + 0.4 * rng.normal(size=(nr, 256)).astype(np.float32)
    src = base[rng.permutation(max(nr, ns))[:ns]] + 0.4 * rng.normal(size=(ns, 256)).astype(np.float32)
    ref /= np.linalg.norm(ref, axis=1, keepdims=True)
    src /= np.linalg.norm(src, axis=1, keepdims=True)
    rm, sm = rng.random(nr) > 0.1, rng.random(ns) > 0.1
    wri, wsi, wsc, _ = M.superpoint_matching(ref, src, rm, sm, k, True)
    ri, si, sc = SuperPointMatching(k, True)(_c(ref), _c(src), _c(rm), _c(sm))
    assert sc.shape[0] == wsc.shape[0] == min(k, int(rm.sum()) * int(sm.sum()))
    np.testing.assert_allclose(sc.cpu().numpy(), wsc, rtol=2e-5)
    # indices: exact wherever neighbouring scores differ by more than the fp noise
    gap = np.abs(np.diff(wsc)) / wsc[:-1]
    safe = np.concatenate([[True], gap > 2e-5]) & np.concatenate([gap > 2e-5, [True]])
    assert np.array_equal(ri.cpu().numpy()[safe], wri[safe]) and np.array_equal(si.cpu().numpy()[safe], wsi[safe])
    assert safe.mean() > 0.5
    # as multisets of (ref, src) pairs the two results agree except possibly at the k-th boundary
    got = set(zip(ri.cpu().numpy().tolist(), si.cpu().numpy().tolist()))
    want = set(zip(wri.tolist(), wsi.tolist()))
    assert len(got ^ want) <= 4


def test_point_matching_vs_reference_golden():
    from gaussreg_amd.matching import PointMatching
    g = load_golden("matching.npz")
    args = [_c(g[k]) for k in ("pm_ref_points", "pm_src_points", "pm_ref_masks", "pm_src_masks", "pm_ref_idx",
                               "pm_src_idx", "pm_score", "pm_global")]
    pm = PointMatching(k=3, mutual=True, confidence_threshold=0.05)
    corr = pm.compute_correspondence_matrix(torch.exp(args[6]), args[2], args[3])
    assert corr.dtype == torch.bool and np.array_equal(corr.cpu().numpy(), g["pm_corr_mat"])
    rp, sp, ri, si, sc = pm(*args)
    assert np.array_equal(ri.cpu().numpy(), g["pm_out_ref_idx"]) and np.array_equal(si.cpu().numpy(), g["pm_out_src_idx"])
    assert np.array_equal(rp.cpu().numpy(), g["pm_out_ref_points"]) and np.array_equal(sp.cpu().numpy(), g["pm_out_src_points"])
    np.testing.assert_allclose(sc.cpu().numpy(), g["pm_out_scores"], rtol=1e-5)
    pm2 = PointMatching(k=2, mutual=False, confidence_threshold=0.1, use_global_score=True)
    rp, sp, ri, si, sc = pm2(*args)
    assert np.array_equal(ri.cpu().numpy(), g["pm2_out_ref_idx"]) and np.array_equal(si.cpu().numpy(), g["pm2_out_src_idx"])
    np.testing.assert_allclose(sc.cpu().numpy(), g["pm2_out_scores"], rtol=1e-5)
    with pytest.raises(NotImplementedError):
        PointMatching(3, use_dustbin=True)


def test_point_matching_demo_shape_vs_oracle():
    """P=256 patches of 128x128 (config.py:110-125 shapes)."""
    from gaussreg_amd.matching import PointMatching
    from oracle import matching_np as M
    rng = np.random.default_rng(2)
    P, K = 256, 128
    logits = rng.normal(size=(P, K, K)).astype(np.float32) * 4
    score = (logits - np.log(np.exp(logits).sum(2, keepdims=True)) + logits - np.log(np.exp(logits).sum(1, keepdims=True))) * 0.5
    score = score.astype(np.float32)
    rm, sm = rng.random((P, K)) > 0.3, rng.random((P, K)) > 0.3
    rp, sp = rng.normal(size=(P, K, 3)).astype(np.float32), rng.normal(size=(P, K, 3)).astype(np.float32)
    ri, si = rng.integers(0, 30000, (P, K)), rng.integers(0, 30000, (P, K))
    gs = rng.random(P).astype(np.float32)
    want = M.point_matching(rp, sp, rm, sm, ri, si, score, gs)
    got = PointMatching(3)( _c(rp), _c(sp), _c(rm), _c(sm), _c(ri), _c(si), _c(score), _c(gs))
    assert got[2].shape[0] == want[2].shape[0] > 100
    assert np.array_equal(got[2].cpu().numpy(), want[2]) and np.array_equal(got[3].cpu().numpy(), want[3])
    assert np.array_equal(got[0].cpu().numpy(), want[0]) and np.array_equal(got[1].cpu().numpy(), want[1])
    np.testing.assert_allclose(got[4].cpu().numpy(), want[4], rtol=1e-5)


def test_point_to_node_vs_reference_golden():
    from gaussreg_amd.ops import point_to_node_partition
    g = load_golden("matching.npz")
    p2n, nm, idx, km = point_to_node_partition(_c(g["p2n_points"]), _c(g["p2n_nodes"]), 64)
    assert p2n.dtype == torch.int64 and nm.dtype == torch.bool and idx.dtype == torch.int64 and km.dtype == torch.bool
    assert np.array_equal(p2n.cpu().numpy(), g["p2n_point_to_node"])
    assert np.array_equal(nm.cpu().numpy(), g["p2n_node_masks"])
    assert np.array_equal(km.cpu().numpy(), g["p2n_knn_masks"])
    assert np.array_equal(idx.cpu().numpy(), g["p2n_knn_idx"])
    out = point_to_node_partition(_c(g["p2n_points"]), _c(g["p2n_nodes"]), 64, return_count=True)
    assert np.array_equal(out[1].cpu().numpy(), np.bincount(g["p2n_point_to_node"], minlength=g["p2n_nodes"].shape[0]))


def test_point_to_node_demo_shape_vs_oracle():
    """N ~ 25 k fine points, M ~ 767 nodes, K = 128."""
    from gaussreg_amd.ops import point_to_node_partition
    from oracle import matching_np as M
    rng = np.random.default_rng(4)
    N, Mn = 24745, 767
    pts = (rng.random((N, 3)) * [4.0, 3.0, 2.5]).astype(np.float32)
    nodes = pts[rng.permutation(N)[:Mn]] + rng.normal(0, 0.02, (Mn, 3)).astype(np.float32)
    wp, wm, widx, wkm, dmat = M.point_to_node_partition(pts, nodes, 128)
    p2n, nm, idx, km = [t.cpu().numpy() for t in point_to_node_partition(_c(pts), _c(nodes), 128)]
    # argmin may legitimately differ where the two best node distances are within fp noise
    full = M.pairwise_distance(nodes, pts)
    part = np.partition(full, 1, axis=0)
    clear = (part[1] - part[0]) > 1e-5
    assert clear.mean() > 0.99 and np.array_equal(p2n[clear], wp[clear])
    if np.array_equal(p2n, wp):
        assert np.array_equal(nm, wm) and np.array_equal(km, wkm)
        same_rows = (idx == widx).all(1)
        assert same_rows.mean() > 0.98  # rows differ only where two member distances tie within fp noise
        for r in np.nonzero(~same_rows)[0]:
            assert set(idx[r]) == set(widx[r])


def test_local_global_registration_vs_reference_golden():
    from geotransformer.modules.geotransformer import LocalGlobalRegistration
    g = load_golden("matching.npz")
    lgr = LocalGlobalRegistration(3, 0.1, mutual=True, confidence_threshold=0.05, correspondence_threshold=3,
                                  num_refinement_steps=5)
    r, s, sc, T = lgr(_c(g["lgr_ref_points"]), _c(g["lgr_src_points"]), _c(g["lgr_ref_masks"]), _c(g["lgr_src_masks"]),
                      _c(g["lgr_score"]), _c(g["lgr_global"]))
    assert np.array_equal(r.cpu().numpy(), g["lgr_out_ref"]) and np.array_equal(s.cpu().numpy(), g["lgr_out_src"])
    np.testing.assert_allclose(sc.cpu().numpy(), g["lgr_out_scores"], rtol=1e-5)
    assert T.shape == (4, 4)
    np.testing.assert_allclose(T.cpu().numpy(), g["lgr_out_transform"], atol=1e-5)


def test_local_global_registration_degenerate_and_demo_shape():
    from gaussreg_amd.matching import LocalGlobalRegistration
    from oracle import matching_np as M
    rng = np.random.default_rng(9)
    P, K = 256, 128
    ang = 0.4
    Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]])
    src = rng.random((P, K, 3)) * 3
    ref = src @ Rm.T + [0.1, 0.2, -0.3] + rng.normal(0, 0.01, (P, K, 3))
    logits = rng.normal(size=(P, K, K)) - 8.0
    logits[:, np.arange(K), np.arange(K)] = 4.0 + rng.normal(size=(P, K))
    ls = logits - np.log(np.exp(logits).sum(2, keepdims=True))
    ls = ((ls + logits - np.log(np.exp(logits).sum(1, keepdims=True))) * 0.5).astype(np.float32)
    rm, sm = rng.random((P, K)) > 0.2, rng.random((P, K)) > 0.2
    src, ref = src.astype(np.float32), ref.astype(np.float32)
    want = M.local_global_registration(ref, src, rm, sm, ls)
    lgr = LocalGlobalRegistration(3, 0.1)
    got = lgr(_c(ref), _c(src), _c(rm), _c(sm), _c(ls), None)
    assert got[0].shape[0] == want[0].shape[0] > 1000
    np.testing.assert_allclose(got[3].cpu().numpy(), want[3], atol=1e-5)
    # degenerate: threshold so high that no patch qualifies -> global initialisation branch
    lgr2 = LocalGlobalRegistration(3, 0.1, correspondence_threshold=10 ** 6)
    T2 = lgr2(_c(ref), _c(src), _c(rm), _c(sm), _c(ls), None)[3].cpu().numpy()
    want2 = M.local_global_registration(ref, src, rm, sm, ls, correspondence_threshold=10 ** 6)[3]
    np.testing.assert_allclose(T2, want2, atol=1e-5)


def test_superpoint_matching_more_ties_than_the_candidate_buffer():
    """Constant features (an untrained / collapsed backbone, or padded superpoints without masks): every score ties.
    The reference's topk still returns k entries; here they are the k lowest flat indices, all with the same score."""
    from gaussreg_amd.matching import SuperPointMatching
    f = torch.nn.functional.normalize(torch.ones(120, 64, device="cuda"), dim=1)
    ri, si, sc = SuperPointMatching(256, True)(f, f.clone())
    assert ri.shape == si.shape == sc.shape == (256,)
    flat = (ri * 120 + si).cpu().numpy()
    assert np.array_equal(flat, np.arange(256)), flat[:10]
    assert float(sc.max() - sc.min()) == 0.0
    # a few scores above a sea of ties: those come first, then the lowest-index ties
    g = torch.ones(90, 32, device="cuda")
    g[5, :] = 0
    g[5, 0] = 1.0
    h = g.clone()
    g, h = torch.nn.functional.normalize(g, dim=1), torch.nn.functional.normalize(h, dim=1)
    ri, si, sc = SuperPointMatching(64, False)(g, h)
    assert ri.shape == (64,) and bool((sc[:-1] >= sc[1:]).all())
    # exactly torch.topk's set under (score descending, flat index ascending): the tie value is shared by 89 x 89 entries
    # (more than the candidate buffer holds: the flat-index-ordered gather), row / column 5 score differently
    S = torch.exp(-(2.0 - 2.0 * g.double() @ h.double().T).clamp(min=0)).flatten()
    order = sorted(range(S.numel()), key=lambda e: (-round(float(S[e]), 9), e))[:64]
    assert (ri * 90 + si).tolist() == order


@pytest.mark.parametrize("n,m", [(65, 65), (1, 1), (767, 701), (64, 129)])
def test_pairwise_distance_c_entry_with_the_header_workspace_rule(n, m):
    """A C caller sizes the workspace with gr_pairwise_distance_workspace_bytes(n, m) (include/gaussreg_hip.h) and calls the
    un-batched entry un-normalised: it must be accepted and equal the batched entry bit for bit."""
    from gaussreg_amd import _lib, ops
    L = _lib.lib()
    g = torch.Generator().manual_seed(n * 1000 + m)
    x = torch.randn(n, 48, generator=g).cuda()
    y = torch.randn(m, 48, generator=g).cuda()
    nbytes = L.gr_pairwise_distance_workspace_bytes(n, m)
    assert nbytes == L.gr_pairwise_distance_batch_workspace_bytes(1, n, m)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    out = torch.empty(n, m, device="cuda")
    _lib.check(L.gr_pairwise_distance(_lib.ptr(x), _lib.ptr(y), n, m, 48, 0, _lib.ptr(out), _lib.ptr(ws), nbytes,
                                      _lib.stream_ptr(x.device)))
    assert torch.equal(out, ops.pairwise_distance(x, y))
    # one byte less is refused, not overrun
    rc = L.gr_pairwise_distance(_lib.ptr(x), _lib.ptr(y), n, m, 48, 0, _lib.ptr(out), _lib.ptr(ws), nbytes - 1,
                                _lib.stream_ptr(x.device))
    assert rc == -3  # GR_ERR_WORKSPACE (include/gaussreg_hip.h)


@pytest.mark.parametrize("nr,ns,k", [(300, 1024, 256), (300, 1025, 256), (2500, 700, 256), (130, 640, 1000), (97, 64, 40), (50, 63, 40)])
def test_superpoint_matching_fast_and_dense_path_boundaries(nr, ns, k):
    """The slab selection takes matrices of at most 1024 columns and k <= 1024 (one column per thread; the full waves must be
    able to supply k thread maxima), everything else the dense radix select; 2 500 rows = 53 slabs appending to one candidate
    buffer; 63 / 64 columns = no / one full wave; k = 1000 with 10 full waves = 100 maxima per wave asked of 64 (threshold 0,
    every score listed -> overflow -> dense path).  Against a float64 evaluation of superpoint_matching.py:32-48."""
    from gaussreg_amd.matching import SuperPointMatching
    g = torch.Generator(device="cuda").manual_seed(nr * 7 + ns)
    fr = torch.nn.functional.normalize(torch.randn(nr, 64, device="cuda", generator=g), dim=1)
    fs = torch.nn.functional.normalize(torch.randn(ns, 64, device="cuda", generator=g), dim=1)
    # a planted set of near-duplicates so that the best scores stand clear of the rest
    nm = min(nr, ns, 300)
    fs[:nm] = torch.nn.functional.normalize(fr[:nm] + 0.25 * torch.randn(nm, 64, device="cuda", generator=g), dim=1)
    ri, si, sc = SuperPointMatching(k, True)(fr, fs)
    S = torch.exp(-(2.0 - 2.0 * fr.double() @ fs.double().T).clamp(min=0))
    score = (S / S.sum(1, keepdim=True)) * (S / S.sum(0, keepdim=True))
    kk = min(k, nr * ns)
    w = torch.topk(score.flatten(), kk)
    assert ri.shape == (kk,) and bool((sc[:-1] >= sc[1:]).all())
    np.testing.assert_allclose(sc.cpu().numpy(), w.values.cpu().numpy(), rtol=2e-5)
    got, want = set((ri * ns + si).tolist()), set(w.indices.tolist())
    assert len(got ^ want) <= 4  # the k-th boundary may flip inside the fp32 noise


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_superpoint_matching_slab_selection_under_extreme_row_and_column_sums(seed):
    """The slab selection filters with reciprocal-priced approximate scores and a fixed 2e-6 relative margin.  Fuzz it where
    that margin is thinnest: un-normalised features (superpoint_matching.py:33 always uses the `2 - 2 xy` form, so d spans
    0 ... 20 and exp(-d) 1 ... 2e-9: row and column sums between ~1e-8 and a few hundred), clusters of near-identical points
    (hundreds of nearly tied scores around the k-th) and rows / columns that are far from everything.  Every emitted score
    must be the float64 score of its (row, column), the emitted set must be the float64 top-k up to entries tied within
    fp32 noise, and the dense path (k > 1024 takes the radix select) must emit the same leading entries."""
    from gaussreg_amd.matching import SuperPointMatching
    g = torch.Generator(device="cuda").manual_seed(1000 + seed)
    nr, ns, k, c = 700, 900, 256, 32
    centres = torch.randn(12, c, device="cuda", generator=g) * 0.5
    which_r = torch.randint(0, 12, (nr,), device="cuda", generator=g)
    which_s = torch.randint(0, 12, (ns,), device="cuda", generator=g)
    fr = centres[which_r] + torch.randn(nr, c, device="cuda", generator=g) * 0.02
    fs = centres[which_s] + torch.randn(ns, c, device="cuda", generator=g) * 0.02
    fr[::7] *= -1.0                                  # rows opposite to every cluster: tiny row sums
    fs[::5] *= -0.5                                  # columns likewise: tiny column sums
    fr[1::50] = fs[1:nr:50][: fr[1::50].shape[0]]    # exact duplicates across the two sets

    xy = fr.double() @ fs.double().T
    S = torch.exp(-(2.0 - 2.0 * xy).clamp(min=0))
    score = (S / S.sum(1, keepdim=True)) * (S / S.sum(0, keepdim=True))
    ri, si, sc = SuperPointMatching(k, True)(fr, fs)
    assert ri.shape == (k,) and bool((sc[:-1] >= sc[1:]).all())
    mine = score[ri, si]
    np.testing.assert_allclose(sc.cpu().numpy(), mine.cpu().numpy(), rtol=1e-4, atol=1e-37)
    kth = torch.topk(score.flatten(), k).values[-1]
    assert bool((mine >= kth * (1 - 1e-3)).all()), "an emitted entry lies clearly below the float64 k-th best"
    missing = (score >= kth * (1 + 1e-3))
    missing[ri, si] = False
    assert int(missing.sum()) == 0, "an entry clearly above the float64 k-th best was dropped by the filter"
    # the dense path (k > 1024): its leading k entries are the slab path's
    ri2, si2, sc2 = SuperPointMatching(1100, True)(fr, fs)
    np.testing.assert_allclose(sc2[:k].cpu().numpy(), sc.cpu().numpy(), rtol=1e-6, atol=1e-37)
    distinct = sc[:-1] > sc[1:] * (1 + 1e-6)     # where neighbours are not tied, the order of the entries is pinned too
    same = (ri2[:k] == ri) & (si2[:k] == si)
    strict = torch.ones(k, dtype=torch.bool, device=sc.device)
    strict[1:] &= distinct
    strict[:-1] &= distinct
    assert bool(same[strict].all())
