"""GPU: grid_subsampling's bucket sort (grid_subsample.hip -> depth_sort.hip: one pass over the top nine bits of every cloud's own
voxel-key range, every bucket finished inside LDS) against the general three-pass radix sort it replaces for batches that
qualify -- the same rows, bit for bit, in both row orders; and against the oracle where the oracle finishes in seconds.
gr_grid_subsample_debug_bucket_sort(0) pins the general sort."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(points, lengths, voxel, order):
    """(rows, lengths) with the bucket sort allowed and with the general sort pinned."""
    from gaussreg_amd import _lib, ext
    L = _lib.lib()
    pts = torch.from_numpy(np.ascontiguousarray(points)).cuda()
    lens = torch.tensor(list(lengths), dtype=torch.int64)
    old = L.gr_grid_subsample_debug_bucket_sort(1)
    try:
        a, al = ext.grid_subsampling(pts, lens, voxel, order=order)
        L.gr_grid_subsample_debug_bucket_sort(0)
        b, bl = ext.grid_subsampling(pts, lens, voxel, order=order)
    finally:
        L.gr_grid_subsample_debug_bucket_sort(old)
    return a.cpu().numpy(), al.tolist(), b.cpu().numpy(), bl.tolist()


def _same(points, lengths, voxel):
    for order in ("reference", "cell"):
        a, al, b, bl = _both(points, lengths, voxel, order)
        assert al == bl, order
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), order
    return al


@pytest.mark.parametrize("seed,lengths,voxel", [
    (0, [200000], 0.05),
    (1, [30000, 30011], 0.025),
    (2, [5000, 0, 70000, 1, 12345, 2048, 2049, 0], 0.04),       # ragged, empty clouds, chunk edges
    (3, [100] * 40, 0.1),                                         # many small clouds
    (4, [150000, 900], 0.02),                                     # too ragged for the tables?  either way the same rows
    (5, [800000, 0, 800000, 5, 700000], 0.03),                    # > 2 M points: the cells are ranked by a second bucket sort
    (6, [30000] * 67 + [0, 7], 0.05),                             # a pyramid level of many pairs: six-bit digits, clouds dealt to
                                                                  # XCDs (69 is no multiple of 8), > 2 M points
    (7, [4500] * 130 + [60000, 3, 0] + [700] * 140, 0.2),         # 273 clouds (past the mailbox's 256), sizes of three digit widths
    (8, [25000] * 40, 0.05),                                      # 40 clouds: XCD mapping, counts through the mailbox
])
def test_bucket_sort_rows_equal_the_general_sort(seed, lengths, voxel):
    rng = np.random.default_rng(seed)
    n = int(sum(lengths))
    pts = (rng.random((n, 3)) * np.array([2.2, 1.7, 2.9]) + rng.normal(0, 5, 3)).astype(np.float32)
    out = _same(pts, lengths, voxel)
    assert sum(out) > 0 and all((l == 0) == (o == 0) for l, o in zip(lengths, out))


def test_bucket_sort_matches_the_oracle():
    from oracle import capi
    rng = np.random.default_rng(7)
    lengths = [20000, 15000, 3]
    pts = (rng.random((sum(lengths), 3)) * 1.9).astype(np.float32)
    want_p, want_l = capi.grid_subsampling(pts, np.array(lengths, np.int64), 0.05)
    a, al, b, bl = _both(pts, lengths, 0.05, "reference")
    assert al == list(want_l) and np.array_equal(a.view(np.uint32), np.asarray(want_p, np.float32).view(np.uint32))


def test_points_crowded_into_one_slab_overflow_a_bucket_and_fall_back():
    """60 000 points inside ONE voxel layer of a cloud whose bounding box is a thousand layers tall: the top nine key bits
    cannot tell them apart, the bucket (> 7 936 entries) raises the flag and the call starts over with the general sort."""
    from gaussreg_amd import _lib
    L = _lib.lib()
    before = L.gr_grid_subsample_debug_bucket_sort(2)
    rng = np.random.default_rng(5)
    n = 60002
    pts = np.empty((n, 3), np.float32)
    pts[:, 0] = rng.random(n) * 2.0
    pts[:, 1] = rng.random(n) * 2.0
    pts[:, 2] = 10.0 + rng.random(n) * 0.04
    pts[0] = (0.0, 0.0, 0.0)
    pts[1] = (1.0, 1.0, 50.0)
    out = _same(pts, [n], 0.05)
    assert out[0] > 1000
    assert L.gr_grid_subsample_debug_bucket_sort(2) == before + 2  # once per row order
    # two clouds, only the second one crowded
    other = (rng.random((40000, 3)) * 2.0).astype(np.float32)
    _same(np.concatenate([other, pts]), [40000, n], 0.05)
    assert L.gr_grid_subsample_debug_bucket_sort(2) == before + 4
    # ... and a cloud that fits does not start over
    _same(other, [40000], 0.05)
    assert L.gr_grid_subsample_debug_bucket_sort(2) == before + 4


def test_sixty_four_clouds_of_200k_cell_order_full_size():
    """BASELINE's pyramid input size in one call (64 x 200 000 points, 12.8 M keys): bucket sort == general sort."""
    g = torch.Generator().manual_seed(0)
    pts = (torch.rand(200000 * 64, 3, generator=g) * 10 ** (1 / 3)).float().numpy()
    _same(pts, [200000] * 64, 0.05)


def test_many_clouds_of_mixed_sizes_match_the_oracle():
    """The digit width of the bucket sorts follows the cloud sizes, the hash-order stages deal clouds to XCDs from 32 clouds
    on: 45 clouds between 0 and 9 000 points against the reference's own code, both steps of its row order included."""
    from oracle import capi
    rng = np.random.default_rng(19)
    lengths = [int(x) for x in rng.integers(0, 9000, size=45)]
    lengths[5] = 0
    lengths[44] = 1
    pts = (rng.random((sum(lengths), 3)) * np.array([3.0, 2.0, 1.2])).astype(np.float32)
    want_p, want_l = capi.grid_subsampling(pts, np.array(lengths, np.int64), 0.06)
    a, al, b, bl = _both(pts, lengths, 0.06, "reference")
    assert al == list(want_l) and bl == list(want_l)
    assert np.array_equal(a.view(np.uint32), np.asarray(want_p, np.float32).view(np.uint32))
    assert np.array_equal(b.view(np.uint32), np.asarray(want_p, np.float32).view(np.uint32))
