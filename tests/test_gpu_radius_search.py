"""GPU parity of the single-pass radius_search (gr_radius_search: width = neighbor_limit known before the launch) against
the oracle's full-width result truncated the way the reference truncates it (modules/ops/radius_search.py:25-26), and
against this library's own two-pass path.  Bit-exact (indices)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.fixture(params=[0, 1, 2, 3, 4, 5],
                ids=["count+fill", "single-pass", "thread-per-query", "adaptive", "thread-per-query-64", "thread-per-query-64-preselect"],
                autouse=True)
def search_mode(request):
    """Every test of this file runs in both modes of gr_radius_search (include/gaussreg_hip.h)."""
    from gaussreg_amd import _lib
    old = _lib.lib().gr_radius_search_mode(request.param)
    yield request.param
    _lib.lib().gr_radius_search_mode(old)


def _info(q, s, ql, sl, radius, limit):
    """Call the C entry point directly to read h_info[4] (did the single pass produce the rows?)."""
    from gaussreg_amd import _lib
    L = _lib.lib()
    nq, ns, nb = q.shape[0], s.shape[0], len(ql)
    info = (ctypes.c_int64 * 6)()
    out = torch.empty((nq, limit), dtype=torch.int64, device=q.device)
    ws = torch.empty(L.gr_radius_workspace_bytes(nq, ns, nb), dtype=torch.uint8, device=q.device)
    _lib.check(L.gr_radius_search(_lib.ptr(q), _lib.ptr(s), _lib.host_i64(ql), _lib.host_i64(sl), nq, ns, nb, float(radius),
                                  limit, _lib.ptr(out), _lib.ptr(ws), ws.numel(), info, None, 0, _lib.stream_ptr(q.device)))
    torch.cuda.synchronize()
    return list(info), out


@pytest.mark.parametrize("seed,nq,ns,batch,radius", [(0, 5000, 7000, 3, 0.08), (1, 3000, 3000, 1, 0.2),
                                                     (2, 4000, 1000, 4, 0.5), (3, 777, 12345, 2, 0.03)])
@pytest.mark.parametrize("limit", [1, 7, 30, 64, 89, 400])
def test_limited_vs_oracle(seed, nq, ns, batch, radius, limit):
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(seed)
    s = (rng.random((ns, 3)) * [2.0, 1.0, 0.5]).astype(np.float32)
    q = (rng.random((nq, 3)) * [2.4, 1.0, 0.5] - [0.2, 0, 0]).astype(np.float32)

    def split(n):
        cuts = np.sort(rng.integers(0, n + 1, batch - 1))
        return np.diff(np.concatenate([[0], cuts, [n]])).astype(np.int64)

    ql, sl = split(nq), split(ns)
    want = capi.radius_neighbors(q, s, ql, sl, radius)[:, :limit]
    tq, ts, tql, tsl = _t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl)
    got = ext.radius_neighbors_limited(tq, ts, tql, tsl, radius, limit)
    assert got.is_contiguous() and got.shape == want.shape
    assert np.array_equal(got.cpu().numpy(), want)
    two = ext.radius_neighbors_limited(tq, ts, tql, tsl, radius, limit, two_pass=True)
    assert torch.equal(two, got)


def test_self_search_c1_golden_truncated():
    from gaussreg_amd import ext
    from helpers import c1_points, load_golden
    g = load_golden("ext_c1.npz")
    pts = torch.from_numpy(c1_points()).cuda()
    lens = torch.tensor([pts.shape[0]])
    for limit in (16, 38, 39, 40, 89):
        nb = ext.radius_neighbors_limited(pts, pts, lens, lens, float(g["radius"]), limit)
        assert np.array_equal(nb.cpu().numpy(), g["neighbors"][:, :limit])


def test_dense_blocks_work_in_groups_and_giant_queries_fall_back(search_mode):
    """~600 hits per query: a block's 128 queries need several passes over its key area (groups); then a radius that
    makes every support a neighbour of every query: one query alone overflows the key area and the call repeats itself
    on the two-pass path.  Same rows either way."""
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(11)
    s = rng.random((6000, 3)).astype(np.float32)
    q = rng.random((900, 3)).astype(np.float32)
    ql, sl = np.array([500, 400], np.int64), np.array([4000, 2000], np.int64)
    tq, ts = _t(q), _t(s)
    for radius, limit, single in ((0.35, 50, True), (0.35, 333, True), (2.0, 64, False), (2.0, 7, False)):
        want = capi.radius_neighbors(q, s, ql, sl, radius)
        info, out = _info(tq, ts, ql.tolist(), sl.tolist(), radius, limit)
        assert info[0] == want.shape[1]
        if search_mode < 2:  # (modes 2 and 3 hand dense calls like these back to count + fill)
            assert (info[4] == 1) == (single and search_mode == 1), (radius, limit, info)
        w = min(limit, want.shape[1])
        assert np.array_equal(out[:, :w].cpu().numpy(), want[:, :w])
        if w < limit:
            assert bool((out[:, w:] == s.shape[0]).all())
        got = ext.radius_neighbors_limited(tq, ts, torch.from_numpy(ql), torch.from_numpy(sl), radius, limit)
        assert np.array_equal(got.cpu().numpy(), want[:, :limit])


def test_limit_wider_than_max_count_and_no_neighbour():
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(3)
    s = rng.random((2000, 3)).astype(np.float32)
    q = rng.random((1500, 3)).astype(np.float32)
    ql, sl = np.array([1500], np.int64), np.array([2000], np.int64)
    want = capi.radius_neighbors(q, s, ql, sl, 0.1)
    got = ext.radius_neighbors_limited(_t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl), 0.1, 200)
    assert got.shape == want.shape and got.is_contiguous() and np.array_equal(got.cpu().numpy(), want)
    far = _t(q + 10.0)
    none = ext.radius_neighbors_limited(far, _t(s), torch.from_numpy(ql), torch.from_numpy(sl), 0.1, 30)
    assert none.shape == (1500, 0)
    empty = ext.radius_neighbors_limited(_t(q[:0]), _t(s), torch.tensor([0]), torch.from_numpy(sl), 0.1, 30)
    assert empty.shape == (0, 0)


def test_ragged_batches_odd_limits_and_grid_reuse():
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(8)
    s = rng.random((3000, 3)).astype(np.float32)
    sl = np.array([0, 1300, 1, 1699, 0], np.int64)
    ts, tsl = _t(s), torch.from_numpy(sl)
    grid = ext.SupportGrid(2000)
    for k, limit in enumerate((5, 33, 41)):
        q = rng.random((1000 + 300 * k, 3)).astype(np.float32)
        ql = np.array([10, 500 + 300 * k, 0, 489, 1], np.int64)
        want = capi.radius_neighbors(q, s, ql, sl, 0.12)[:, :limit]
        got = ext.radius_neighbors_limited(_t(q), ts, torch.from_numpy(ql), tsl, 0.12, limit, grid=grid)
        assert np.array_equal(got.cpu().numpy(), want)
    # self-search through the cached grid
    want = capi.radius_neighbors(s, s, sl, sl, 0.12)[:, :17]
    got = ext.radius_neighbors_limited(ts, ts, tsl, tsl, 0.12, 17, grid=grid)
    assert np.array_equal(got.cpu().numpy(), want)


def test_200k_limited_equals_two_pass_and_properties(search_mode):
    from gaussreg_amd import ext, synthetic
    pts, lens = synthetic.cloud_200k(2, seed=3)
    d = pts.cuda()
    a = ext.radius_neighbors_limited(d, d, lens, lens, 0.0625, 40)
    a2 = ext.radius_neighbors_limited(d, d, lens, lens, 0.0625, 40)
    assert torch.equal(a, a2)
    b = ext.radius_neighbors_limited(d, d, lens, lens, 0.0625, 40, two_pass=True)
    assert a.shape == (400000, 40) and torch.equal(a, b)
    assert bool((a[:, 0] == torch.arange(400000, device=a.device)).all())  # self first (d = 0)


@pytest.mark.parametrize("limit", [6, 13, 40])
def test_equal_distances_follow_the_index_order(limit):
    """Duplicated points and an integer lattice: many exactly equal distances per row.  This library's rule for them
    (ascending support index, the oracle's rule) must hold on the single-pass path too -- its 32-bit ranking sees them as
    collisions and redoes such rows on (distance, index)."""
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(21)
    base = rng.random((700, 3)).astype(np.float32)
    dup = np.concatenate([base, base[:400], base[100:300]])                      # duplicates, shuffled below
    lattice = (np.stack(np.meshgrid(*[np.arange(9)] * 3, indexing="ij"), -1).reshape(-1, 3) * 0.125).astype(np.float32)
    s = np.concatenate([dup[rng.permutation(dup.shape[0])], lattice[rng.permutation(lattice.shape[0])]])
    sl = np.array([dup.shape[0], lattice.shape[0]], np.int64)
    for radius in (0.13, 0.26, 0.4):  # (0.4: ~120 hits at a dozen distinct distances -- whole bins of the pre-selection tie)
        want = capi.radius_neighbors(s, s, sl, sl, radius)[:, :limit]
        ts, tsl = _t(s), torch.from_numpy(sl)
        got = ext.radius_neighbors_limited(ts, ts, tsl, tsl, radius, limit)
        assert np.array_equal(got.cpu().numpy(), want)
        q = np.concatenate([base[:300] + np.float32(1e-3), lattice[:200]])
        ql = np.array([300, 200], np.int64)
        want = capi.radius_neighbors(q, s, ql, sl, radius)[:, :limit]
        got = ext.radius_neighbors_limited(_t(q), ts, torch.from_numpy(ql), tsl, radius, limit)
        assert np.array_equal(got.cpu().numpy(), want)


def test_call_site_width_hint_narrow_rows_and_the_repeat_at_full_limit():
    """`radius_search` remembers the largest count of a (radius, limit) call site and searches the next call with narrower
    rows (ext._WIDTH_HINT).  Sparse cloud first (hint 8-ish against a limit of 60), then a cloud ten times as dense at the
    same call site: its counts do not fit the hinted rows, the call repeats itself at the full limit.  Same tensor as the
    oracle's every time, contiguous, and the same as a process that has no hint."""
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(5)
    radius, limit = 0.0731, 60
    ext._WIDTH_HINT.pop((float(radius), limit), None)
    for n, want_hint_before in ((4000, False), (4000, True), (40000, True), (40000, True), (4000, True)):
        pts = rng.random((n, 3)).astype(np.float32)
        lens = np.array([n // 2, n - n // 2], np.int64)
        want = capi.radius_neighbors(pts, pts, lens, lens, radius)[:, :limit]
        assert ((float(radius), limit) in ext._WIDTH_HINT) == want_hint_before
        t = _t(pts)
        got = ext.radius_neighbors_limited(t, t, torch.from_numpy(lens), torch.from_numpy(lens), radius, limit)
        assert got.is_contiguous() and got.shape == want.shape and np.array_equal(got.cpu().numpy(), want)
        assert ext._WIDTH_HINT[(float(radius), limit)] == capi.radius_neighbors(pts, pts, lens, lens, radius).shape[1]


def test_a_cell_with_more_points_than_a_12_bit_range_and_nan_queries():
    """5 000 supports inside one cell: a thread's candidate range there is longer than the 12-bit offsets of the
    thread-per-query kernel's hit codes (give-up code 2: the call repeats itself on count + fill), every cluster query has
    thousands of neighbours; queries with NaN / infinite coordinates have none.  Same rows as the oracle in every mode."""
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(17)
    cluster = (0.5 + rng.random((5000, 3)) * 1e-3).astype(np.float32)
    spread = rng.random((3000, 3)).astype(np.float32)
    s = np.concatenate([cluster, spread])[rng.permutation(8000)]
    q = np.concatenate([cluster[:40], spread[:300], rng.random((200, 3)).astype(np.float32)])
    q[5] = np.nan
    q[50, 1] = np.inf
    q[400, 2] = -np.inf
    ql, sl = np.array([q.shape[0]], np.int64), np.array([8000], np.int64)
    want = capi.radius_neighbors(q, s, ql, sl, 0.05)
    assert want.shape[1] >= 5000
    for limit in (16, 70):
        got = ext.radius_neighbors_limited(_t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl), 0.05, limit)
        assert np.array_equal(got.cpu().numpy(), want[:, :limit])
        assert bool((got[5] == 8000).all()) and bool((got[50] == 8000).all()) and bool((got[400] == 8000).all())


@pytest.mark.parametrize("limit", [1, 20, 49, 56, 60])
def test_coarsest_level_shape_many_more_hits_than_the_row_keeps(search_mode, limit):
    """The last level of the data pyramid: clouds of ~770 points, a radius that reaches a fifth of the cloud (~150 hits), rows
    of 49.  The default mode moves such a call site to the pre-selecting kernel (mode 5 starts there): same rows as the
    truncated oracle, the reported width is the true largest count, and the site memory ends up on a kernel that finishes."""
    from gaussreg_amd import ext
    from oracle import capi
    rng = np.random.default_rng(17)
    nb = 6
    clouds, queries = [], []
    for b in range(nb):  # points on the walls and the floor of a room: surfaces, as in a scan
        n = 767 + 5 * b
        u = rng.random((n, 3)).astype(np.float32) * np.array([2.6, 2.2, 1.8], np.float32)
        face = rng.integers(0, 3, n)
        u[np.arange(n), face] = 0.0
        clouds.append(u + np.float32(b))
        queries.append((clouds[-1][rng.integers(0, n, 3000)] + rng.normal(0, 0.05, (3000, 3))).astype(np.float32))
    s, q = np.concatenate(clouds), np.concatenate(queries)
    sl, ql = np.array([c.shape[0] for c in clouds], np.int64), np.array([3000] * nb, np.int64)
    full = capi.radius_neighbors(q, s, ql, sl, 1.0)
    counts = (full < s.shape[0]).sum(1)
    assert counts.mean() > 100 and full.shape[1] > 200
    tq, ts, tql, tsl = _t(q), _t(s), torch.from_numpy(ql), torch.from_numpy(sl)
    for _ in range(3):  # (mode 3: the first calls walk the site up its kernels)
        info, out = _info(tq, ts, ql.tolist(), sl.tolist(), 1.0, limit)
        assert info[0] == full.shape[1]
        assert np.array_equal(out.cpu().numpy(), full[:, :limit])
    if search_mode == 5 and limit <= 56:
        assert info[4] == 1  # finished by the thread-per-query kernel itself
    got = ext.radius_neighbors_limited(ts, ts, tsl, tsl, 1.0, limit)
    assert np.array_equal(got.cpu().numpy(), capi.radius_neighbors(s, s, sl, sl, 1.0)[:, :limit])


def test_preselection_with_a_last_bin_of_hundreds_of_equal_distances(search_mode):
    """240 copies of one support (equal distances, one histogram bin) behind 30 nearer ones, rows of 40: the bin that holds the
    row's end cannot be cut, the wave's exact path gets 270 keys -- more than its key area -- and the call repeats itself on
    count + fill.  A second cloud with 100 copies stays inside the key area.  Rows = the oracle's (ties by ascending index)."""
    from oracle import capi
    rng = np.random.default_rng(23)

    def cloud(copies):
        near = (rng.random((30, 3)) * 0.2).astype(np.float32)
        far = np.tile(np.array([[0.45, 0.1, 0.05]], np.float32), (copies, 1))
        rest = (rng.random((400, 3)) * 0.9).astype(np.float32)
        pts = np.concatenate([near, far, rest])
        return pts[rng.permutation(pts.shape[0])]

    s = np.concatenate([cloud(240), cloud(100) + np.float32(3.0)])
    sl = np.array([670, 530], np.int64)
    q = np.concatenate([(rng.random((200, 3)) * 0.1).astype(np.float32), (rng.random((200, 3)) * 0.1).astype(np.float32) + np.float32(3.0)])
    ql = np.array([200, 200], np.int64)
    want = capi.radius_neighbors(q, s, ql, sl, 0.6)
    assert want.shape[1] > 270
    for limit in (40, 56):
        info, out = _info(_t(q), _t(s), ql.tolist(), sl.tolist(), 0.6, limit)
        assert info[0] == want.shape[1]
        assert np.array_equal(out.cpu().numpy(), want[:, :limit])


def test_column_slice_on_request_instead_of_a_dense_copy():
    """contiguous=False hands back the column slice of the searched rows where the result is narrower than the rows the
    kernel wrote (the reference's own truncated result is such a slice, radius_search.py:26): same values and shape, no copy;
    the default stays dense."""
    from gaussreg_amd import ext, ops
    from oracle import capi
    rng = np.random.default_rng(29)
    s = rng.random((4000, 3)).astype(np.float32)
    lens = np.array([4000], np.int64)
    want = capi.radius_neighbors(s, s, lens, lens, 0.07)
    assert want.shape[1] < 60
    ts, tl = _t(s), torch.from_numpy(lens)
    ext._WIDTH_HINT.pop((0.07, 89), None)
    for call in range(2):  # (first call: rows of the full limit; second: rows of the remembered width + margin)
        dense = ops.radius_search(ts, ts, tl, tl, 0.07, 89)
        sliced = ops.radius_search(ts, ts, tl, tl, 0.07, 89, contiguous=False)
        assert dense.is_contiguous() and dense.shape == want.shape and np.array_equal(dense.cpu().numpy(), want)
        assert sliced.shape == want.shape and not sliced.is_contiguous() and sliced.stride(1) == 1
        assert torch.equal(sliced, dense)
