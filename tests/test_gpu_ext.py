"""GPU parity: HIP radius_neighbors / grid_subsampling (through the C ABI) vs the reference's golden
vectors and vs the oracle on seeded inputs.  Bit-exact for indices and barycentres."""
import numpy as np
import pytest
import torch

from helpers import load_golden, c1_points, assert_neighbors_equal_up_to_ties, sqdist_f32

pytestmark = pytest.mark.gpu


def _t(a, dev="cuda"):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _ext():
    from gaussreg_amd import ext
    return ext


def test_c1_radius_matches_reference_golden():
    g = load_golden("ext_c1.npz")
    pts = _t(c1_points())
    lens = torch.tensor([20000])
    nb = _ext().radius_neighbors(pts, pts, lens, lens, float(g["radius"]))
    assert nb.is_cuda and nb.dtype == torch.int64 and nb.is_contiguous()
    assert tuple(nb.shape) == (20000, 39)
    assert np.array_equal(nb.cpu().numpy(), g["neighbors"].astype(np.int64))


def test_c1_grid_matches_reference_golden_bit_exact():
    g = load_golden("ext_c1.npz")
    pts = _t(c1_points())
    sp, sl = _ext().grid_subsampling(pts, torch.tensor([20000]), float(g["voxel"]))
    assert sl.tolist() == [7366] and tuple(sp.shape) == (7366, 3)
    assert np.array_equal(sp.cpu().numpy().view(np.uint32), g["s_points"].view(np.uint32))
    # device-only order: same rows, sorted
    sc, slc = _ext().grid_subsampling(pts, torch.tensor([20000]), float(g["voxel"]), order="cell")
    a = np.sort(sc.cpu().numpy().view(np.uint32).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).ravel())
    b = np.sort(g["s_points"].view(np.uint32).view([("x", "u4"), ("y", "u4"), ("z", "u4")]).ravel())
    assert np.array_equal(a, b) and slc.tolist() == [7366]


def test_cpu_tensors_in_cpu_tensors_out():
    g = load_golden("ext_multibatch.npz")
    q, s = torch.from_numpy(g["q"]), torch.from_numpy(g["s"])
    ql, sl = torch.from_numpy(g["q_lengths"]), torch.from_numpy(g["s_lengths"])
    nb = _ext().radius_neighbors(q, s, ql, sl, float(g["radius"]))
    assert not nb.is_cuda
    assert np.array_equal(nb.numpy(), g["neighbors"].astype(np.int64))
    sp, spl = _ext().grid_subsampling(s, sl, float(g["voxel"]))
    assert not sp.is_cuda
    assert np.array_equal(spl.numpy(), g["sub_lengths"])
    assert np.array_equal(sp.numpy().view(np.uint32), g["s_points"].view(np.uint32))


def test_no_neighbor_and_self_only():
    g = load_golden("ext_noneighbor.npz")
    p, far, l = _t(g["p"]), _t(g["far"]), torch.from_numpy(g["lengths"])
    nb = _ext().radius_neighbors(p, p, l, l, float(g["r_self"]))
    assert np.array_equal(nb.cpu().numpy(), g["nb_self"].astype(np.int64))
    nb0 = _ext().radius_neighbors(far, p, l, l, float(g["r_none"]))
    assert tuple(nb0.shape) == (500, 0)


def test_ties_equal_up_to_tie_groups():
    g = load_golden("ext_ties.npz")
    p, l = _t(g["p"]), torch.from_numpy(g["lengths"])
    nb = _ext().radius_neighbors(p, p, l, l, float(g["radius"])).cpu().numpy()
    assert_neighbors_equal_up_to_ties(nb, g["neighbors"], g["p"], g["p"], g["lengths"], g["lengths"])
    # and exactly the oracle's (d, index) tie order
    from oracle import capi
    assert np.array_equal(nb, capi.radius_neighbors(g["p"], g["p"], g["lengths"], g["lengths"], float(g["radius"])))
    sp, sl = _ext().grid_subsampling(p, l, float(g["voxel"]))
    assert np.array_equal(sp.cpu().numpy().view(np.uint32), g["s_points"].view(np.uint32))


def test_pyramid_matches_reference_golden():
    g = load_golden("ext_pyramid.npz")
    ext = _ext()
    v, r = float(g["voxel0"]), float(g["radius0"])
    P, L = [_t(g["points0"])], [torch.from_numpy(g["lengths0"])]
    for i in range(5):
        if i > 0:
            sp, sl = ext.grid_subsampling(P[-1], L[-1], v)
            assert np.array_equal(sl.numpy(), g[f"lengths{i}"])
            assert np.array_equal(sp.cpu().numpy().view(np.uint32), g[f"points{i}"].view(np.uint32)), i
            P.append(sp)
            L.append(sl)
        v *= 2
    for i in range(5):
        pn = [p.cpu().numpy() for p in P]
        ln = [l.numpy() for l in L]
        nb = ext.radius_neighbors(P[i], P[i], L[i], L[i], r).cpu().numpy()
        assert_neighbors_equal_up_to_ties(nb, g[f"neighbors{i}"], pn[i], pn[i], ln[i], ln[i])
        if i < 4:
            sub = ext.radius_neighbors(P[i + 1], P[i], L[i + 1], L[i], r).cpu().numpy()
            assert_neighbors_equal_up_to_ties(sub, g[f"subsampling{i}"], pn[i + 1], pn[i], ln[i + 1], ln[i])
            up = ext.radius_neighbors(P[i], P[i + 1], L[i], L[i + 1], r * 2).cpu().numpy()
            assert_neighbors_equal_up_to_ties(up, g[f"upsampling{i}"], pn[i], pn[i + 1], ln[i], ln[i + 1])
        r *= 2


@pytest.mark.parametrize("seed,nq,ns,batch,radius", [(0, 5000, 7000, 3, 0.08), (1, 3000, 3000, 1, 0.2),
                                                     (2, 4000, 1000, 4, 0.5), (3, 777, 12345, 2, 0.03)])
def test_random_vs_oracle(seed, nq, ns, batch, radius):
    from oracle import capi
    rng = np.random.default_rng(seed)
    s = (rng.random((ns, 3)) * [2.0, 1.0, 0.5]).astype(np.float32)
    q = (rng.random((nq, 3)) * [2.4, 1.0, 0.5] - [0.2, 0, 0]).astype(np.float32)

    def split(n):
        cuts = np.sort(rng.integers(0, n + 1, batch - 1))
        return np.diff(np.concatenate([[0], cuts, [n]])).astype(np.int64)

    ql, sl = split(nq), split(ns)
    want = capi.radius_neighbors(q, s, ql, sl, radius)
    got = _ext().radius_neighbors(_t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl), radius)
    assert np.array_equal(got.cpu().numpy(), want)
    wp, wl = capi.grid_subsampling(s, sl, radius / 2)
    gp, gl = _ext().grid_subsampling(_t(s), torch.from_numpy(sl), radius / 2)
    assert np.array_equal(gl.numpy(), wl)
    assert np.array_equal(gp.cpu().numpy().view(np.uint32), wp.view(np.uint32))


def test_empty_batch_elements_and_limits():
    from oracle import capi
    rng = np.random.default_rng(5)
    s = rng.random((1000, 3)).astype(np.float32)
    q = rng.random((600, 3)).astype(np.float32)
    ql = np.array([0, 600, 0], np.int64)
    sl = np.array([300, 700, 0], np.int64)
    want = capi.radius_neighbors(q, s, ql, sl, 0.15)
    got = _ext().radius_neighbors(_t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl), 0.15).cpu().numpy()
    assert np.array_equal(got, want)
    lim = _ext().radius_neighbors_limited(_t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl), 0.15, 5)
    assert lim.is_contiguous() and np.array_equal(lim.cpu().numpy(), want[:, :5])


def test_error_texts_match_reference():
    ext = _ext()
    p = torch.rand(10, 3, device="cuda")
    l = torch.tensor([10])
    with pytest.raises(RuntimeError, match="q_points must be a float tensor"):
        ext.radius_neighbors(p.double(), p, l, l, 0.1)
    with pytest.raises(RuntimeError, match="s_lengths must be an long tensor"):
        ext.radius_neighbors(p, p, l, l.int(), 0.1)
    with pytest.raises(RuntimeError, match="points must be contiguous"):
        ext.grid_subsampling(torch.rand(3, 10, device="cuda").t(), l, 0.1)


def test_200k_properties():
    """BASELINE 200 k config: width / mean count as measured on the reference,
    plus size-independent properties: self first, strictly inside the radius, sorted, complete."""
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(200000, 3, generator=g) * 10 ** (1 / 3)).float()
    d = pts.cuda()
    lens = torch.tensor([200000])
    r = 0.0625
    nb = _ext().radius_neighbors(d, d, lens, lens, r)
    assert tuple(nb.shape) == (200000, 45)
    valid = nb < 200000
    assert abs(valid.sum(1).float().mean().item() - 20.8) < 0.1
    assert torch.equal(nb[:, 0], torch.arange(200000, device="cuda"))  # d = 0 sorts first
    # distances along each row: < r^2 where valid, non-decreasing
    idx = nb.clamp(max=199999)
    P = d
    diff = P[:, None, :] - P[idx]
    d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    r2 = torch.tensor(r, dtype=torch.float32).pow(2).item()
    assert bool((d2[valid] < r2).all())
    d2m = torch.where(valid, d2, torch.full_like(d2, float("inf")))
    assert bool((d2m[:, 1:] >= d2m[:, :-1]).all())
    # completeness on a sample of queries against brute force
    sel = torch.arange(0, 200000, 997, device="cuda")
    diff = P[sel][:, None, :] - P[None, :, :]
    bd2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
    assert torch.equal((bd2 < r2).sum(1), valid[sel].sum(1))
    # grid subsample: count as on the reference, rows of every voxel average back into the voxel
    sp, sl = _ext().grid_subsampling(d, lens, 0.05)
    assert sl.tolist() == [74011]
    # the whole (200 000, 45) tensor and the 74 011 subsampled rows against the CPU oracle, bit for bit
    # (radius_neighbors_cpu.cpp:3-91, grid_subsampling_cpu.cpp:3-48); the tie-free input makes the order unique
    from oracle import capi
    lens_np = np.array([200000], np.int64)
    want = capi.radius_neighbors(pts.numpy(), pts.numpy(), lens_np, lens_np, r)
    assert want.shape == (200000, 45)
    assert np.array_equal(nb.cpu().numpy(), want)
    wp, wl = capi.grid_subsampling(pts.numpy(), lens_np, 0.05)
    assert wl.tolist() == [74011]
    assert np.array_equal(sp.cpu().numpy().view(np.uint32), wp.view(np.uint32))
    # the radius_search(limit=40) wrapper, both search modes, against the truncated oracle rows
    from gaussreg_amd import _lib, ext as gext
    for mode in (0, 1, 2, 3, 4, 5):
        old = _lib.lib().gr_radius_search_mode(mode)
        try:
            lim = gext.radius_neighbors_limited(d, d, lens, lens, r, 40)
        finally:
            _lib.lib().gr_radius_search_mode(old)
        assert lim.is_contiguous() and np.array_equal(lim.cpu().numpy(), want[:, :40]), mode


def test_support_grid_reuse_gives_identical_results():
    """Searches that share a SupportGrid (same supports + radius, other queries) must equal independent searches,
    and a changed support / radius must not be served from the cache."""
    ext = _ext()
    rng = np.random.default_rng(9)
    s = _t((rng.random((9000, 3)) * [2.0, 1.5, 1.0]).astype(np.float32))
    q1 = _t((rng.random((4000, 3)) * [2.2, 1.5, 1.0] - [0.1, 0, 0]).astype(np.float32))
    q2 = _t((rng.random((12000, 3)) * [2.0, 1.5, 1.0]).astype(np.float32))
    sl = torch.tensor([5000, 4000])
    grid = ext.SupportGrid(max_queries=12000)
    pairs = [(q2, torch.tensor([7000, 5000])), (s, sl), (q1, torch.tensor([1500, 2500])), (s, sl)]
    for q, ql in pairs:   # first call builds (q != s), then self-search and another query set reuse it
        got = ext.radius_neighbors(q, s, ql, sl, 0.11, grid=grid)
        want = ext.radius_neighbors(q, s, ql, sl, 0.11)
        assert torch.equal(got, want)
        lim = ext.radius_neighbors_limited(q, s, ql, sl, 0.11, 7, grid=grid)
        assert torch.equal(lim, want[:, :7])
    # other radius / other supports with the same grid object: rebuilt, not reused
    assert torch.equal(ext.radius_neighbors(q1, s, pairs[2][1], sl, 0.2, grid=grid), ext.radius_neighbors(q1, s, pairs[2][1], sl, 0.2))
    s2 = s.clone() * 0.5
    assert torch.equal(ext.radius_neighbors(q1, s2, pairs[2][1], sl, 0.2, grid=grid), ext.radius_neighbors(q1, s2, pairs[2][1], sl, 0.2))


def test_small_shape_fuzz_vs_oracle():
    """Many tiny, ragged configurations (empty clouds, single points, duplicates, radius from tiny to covering
    everything, self-search and q != s, with and without a shared SupportGrid) against the CPU oracle."""
    from oracle import capi
    ext = _ext()
    rng = np.random.default_rng(2024)
    for it in range(60):
        batch = int(rng.integers(1, 5))
        sl = rng.integers(0, 40, batch).astype(np.int64)
        ql = rng.integers(0, 40, batch).astype(np.int64)
        if it % 5 == 0:
            sl[rng.integers(0, batch)] = 0
        s = rng.random((int(sl.sum()), 3)).astype(np.float32)
        q = rng.random((int(ql.sum()), 3)).astype(np.float32)
        if it % 3 == 0 and len(s) > 4:
            s[1] = s[0]; s[3] = s[2]                      # duplicates -> equal distances
        radius = float(rng.choice([1e-4, 0.05, 0.2, 0.7, 3.0]))
        ts, tq = _t(s), _t(q)
        grid = ext.SupportGrid(max(len(q), len(s), 1))
        for (qq, qqt, qql) in ((s, ts, sl), (q, tq, ql), (s, ts, sl)):
            want = capi.radius_neighbors(qq, s, qql, sl, radius)
            got = ext.radius_neighbors(qqt, ts, torch.from_numpy(qql), torch.from_numpy(sl), radius).cpu().numpy()
            got_c = ext.radius_neighbors(qqt, ts, torch.from_numpy(qql), torch.from_numpy(sl), radius, grid=grid).cpu().numpy()
            assert got.shape == want.shape and got_c.shape == want.shape, (it, got.shape, want.shape)
            if it % 3 == 0:   # duplicates: rows equal as sets of (distance-sorted) neighbours up to tie order
                assert np.array_equal(np.sort(got, 1), np.sort(want, 1)) and np.array_equal(np.sort(got_c, 1), np.sort(want, 1))
            else:
                assert np.array_equal(got, want) and np.array_equal(got_c, want), it
        if len(s):
            wp, wl = capi.grid_subsampling(s, sl, max(radius / 2, 0.01))
            gp, gl = ext.grid_subsampling(ts, torch.from_numpy(sl), max(radius / 2, 0.01))
            assert np.array_equal(gl.numpy(), wl) and np.array_equal(gp.cpu().numpy().view(np.uint32), wp.view(np.uint32))


def test_large_clouds_take_the_staged_radix_sort():
    """>= 2^19 keys: csrc/sort.hip switches to 8-bit digits and LDS-staged runs.  grid_subsampling of 3 x 250 k points against
    the oracle (bit-exact incl. the reference row order), and FPS of the stacked clouds against FPS of every cloud alone
    (whose Morton pre-sort takes the small-size path)."""
    from oracle import capi
    from gaussreg_amd.registration import farthest_point_sampling
    rng = np.random.default_rng(17)
    lens = np.array([250000, 250000, 250000], np.int64)
    pts = (rng.random((int(lens.sum()), 3)) * [5, 4, 3]).astype(np.float32)
    t = _t(pts)
    for order in ("reference", "cell"):
        gp, gl = _ext().grid_subsampling(t, torch.from_numpy(lens), 0.05, order=order)
        wp, wl = capi.grid_subsampling(pts, lens, 0.05)
        assert np.array_equal(gl.numpy(), wl)
        got = gp.cpu().numpy()
        if order == "reference":
            assert np.array_equal(got.view(np.uint32), wp.view(np.uint32))
        else:  # same multiset of rows per cloud
            o = 0
            for m in wl:
                a = np.sort(got[o:o + m].view([("x", "f4"), ("y", "f4"), ("z", "f4")]), axis=0)
                b = np.sort(wp[o:o + m].view([("x", "f4"), ("y", "f4"), ("z", "f4")]), axis=0)
                assert np.array_equal(a, b)
                o += m
    both = farthest_point_sampling(t, lens.tolist(), [2000] * 3, start_indices=[0, 1, 2])
    o = 0
    for b, n in enumerate(lens.tolist()):
        alone = farthest_point_sampling(t[o:o + n].contiguous(), [n], [2000], start_indices=[b])
        assert torch.equal(both[b], alone[0])
        o += n


def test_millions_of_keys_take_the_long_chunk_scan():
    """> 2 M keys = more than 1 024 chunks per digit: csrc/sort.hip scans the chunk table with one workgroup per digit
    (rs_scan_long_kernel) and common.hip's scans run their three-launch form.  11 x 220 k points against the oracle, bit-exact
    including the reference row order."""
    from oracle import capi
    rng = np.random.default_rng(23)
    lens = np.array([220000] * 11, np.int64)
    pts = (rng.random((int(lens.sum()), 3)) * [3, 2.5, 2]).astype(np.float32)
    gp, gl = _ext().grid_subsampling(_t(pts), torch.from_numpy(lens), 0.05)
    wp, wl = capi.grid_subsampling(pts, lens, 0.05)
    assert np.array_equal(gl.numpy(), wl)
    assert np.array_equal(gp.cpu().numpy().view(np.uint32), wp.view(np.uint32))


def test_many_small_clouds_per_call():
    """100 clouds per call: past the 80-cloud limit of the mailbox read-backs (csrc/grid_subsample.hip copies and synchronises
    instead), ragged sizes including empty clouds; grid_subsampling in the reference order and radius_neighbors against the oracle."""
    from oracle import capi
    rng = np.random.default_rng(31)
    lens = rng.integers(0, 900, size=100).astype(np.int64)
    lens[[3, 57, 99]] = 0
    pts = (rng.random((int(lens.sum()), 3)) * [1.5, 1.0, 0.8]).astype(np.float32)
    gp, gl = _ext().grid_subsampling(_t(pts), torch.from_numpy(lens), 0.08)
    wp, wl = capi.grid_subsampling(pts, lens, 0.08)
    assert np.array_equal(gl.numpy(), wl)
    assert np.array_equal(gp.cpu().numpy().view(np.uint32), wp.view(np.uint32))
    nb = _ext().radius_neighbors(gp, gp, gl, gl, 0.2)
    want = capi.radius_neighbors(wp, wp, wl, wl, 0.2)
    assert nb.shape == want.shape and np.array_equal(nb.cpu().numpy(), want)
