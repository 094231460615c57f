"""The device evaluation of std::unordered_map's iteration order (csrc/hash_order_device.hip) against the host replay of
the container's own linking rules (csrc/hash_order.hip, itself checked against the real container in
tests/test_hash_order.py): many clouds per call, sizes on both sides of every rehash threshold, colliding keys."""
import ctypes

import numpy as np
import pytest
import torch

from gaussreg_amd import _lib

pytestmark = pytest.mark.gpu


def host_order(keys):
    L = _lib.lib()
    k = np.ascontiguousarray(keys, np.uint64)
    perm = np.zeros(max(len(k), 1), np.int32)
    _lib.check(L.gr_host_unordered_map_order(k.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), len(k),
                                             perm.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
    return perm[: len(k)]


def device_order(clouds):
    L = _lib.lib()
    begins = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).astype(np.int64)
    n, nb = int(begins[-1]), len(clouds)
    keys = torch.from_numpy(np.concatenate(clouds).astype(np.uint64).view(np.int64)).cuda() if n else torch.zeros(0, dtype=torch.int64, device="cuda")
    perm = torch.full((max(n, 1),), -1, dtype=torch.int32, device="cuda")
    ws = torch.empty(L.gr_hash_order_device_workspace_bytes(n, nb) + 256, dtype=torch.uint8, device="cuda")
    hb = (ctypes.c_int64 * (nb + 1))(*begins.tolist())
    _lib.check(L.gr_hash_order_device(_lib.ptr(keys), hb, nb, _lib.ptr(perm), _lib.ptr(ws), ws.numel(),
                                      _lib.stream_ptr(keys.device)))
    torch.cuda.synchronize()
    return perm.cpu().numpy()[:n], begins


def distinct(rng, n, kind):
    if kind == "random":
        k = rng.integers(0, 2 ** 40, size=2 * n + 8, dtype=np.uint64)
    elif kind == "dense":      # voxel-like: many keys congruent modulo small primes
        k = rng.integers(0, 3 * n + 8, size=2 * n + 8, dtype=np.uint64)
    else:                      # multiples of a bucket count the policy will choose: heavy collisions in that table
        k = rng.integers(0, 4 * n + 8, size=2 * n + 8, dtype=np.uint64) * np.uint64(541)
    _, first = np.unique(k, return_index=True)
    return k[np.sort(first)][:n]


def test_device_order_equals_host_replay_across_rehash_thresholds():
    rng = np.random.default_rng(0)
    sizes = [0, 1, 2, 12, 13, 14, 28, 29, 30, 59, 60, 126, 127, 128, 256, 257, 258, 541, 542, 1109, 1110, 2357, 2358, 5087,
             5088, 10273, 10274, 20753, 20754, 3001, 777]
    clouds = [distinct(rng, n, ("random", "dense", "colliding")[i % 3]) for i, n in enumerate(sizes)]
    for c, n in zip(clouds, sizes):
        assert len(c) == n
    perm, begins = device_order(clouds)
    for i, c in enumerate(clouds):
        want = host_order(c) + begins[i]
        got = perm[begins[i]:begins[i + 1]]
        assert np.array_equal(got, want), (i, len(c))


@pytest.mark.parametrize("prescan", [False, True])
def test_device_order_large_clouds(prescan, monkeypatch):
    """prescan: the path clouds of more than 4 M clocks take (slab totals scanned by a launch of their own)."""
    from gaussreg_amd import _lib
    rng = np.random.default_rng(1)
    clouds = [distinct(rng, n, kind) for n, kind in ((60000, "dense"), (49505, "random"), (100000, "dense"), (33333, "colliding"),
                                                     (300000, "dense"))]
    old = _lib.lib().gr_hash_order_debug_force_prescan(1 if prescan else 0)
    try:
        perm, begins = device_order(clouds)
    finally:
        _lib.lib().gr_hash_order_debug_force_prescan(old)
    for i, c in enumerate(clouds):
        assert np.array_equal(perm[begins[i]:begins[i + 1]], host_order(c) + begins[i]), i


def test_device_order_medium_clouds_take_the_slab_chains():
    """Tables of at most 16 x 16 384 buckets: the chains are threaded per bucket slab in LDS (ho_bucket_slab_kernel) -- 32-bit
    and 64-bit keys, heavy collisions, sizes on both sides of a slab boundary; the call above with a 300 000-key cloud keeps
    the device-scope atomics of ho_bucket_kernel covered."""
    rng = np.random.default_rng(2)
    clouds = [distinct(rng, n, kind) for n, kind in ((60000, "dense"), (49505, "random"), (100000, "dense"), (33333, "colliding"),
                                                     (130000, "dense"), (16384, "dense"), (16385, "colliding"), (5, "random"),
                                                     (120000, "random"), (90000, "colliding"))]
    assert sum(len(c) for c in clouds) > 600000  # (the slabs are for calls with many elements; fewer take the atomics)
    perm, begins = device_order(clouds)
    for i, c in enumerate(clouds):
        assert np.array_equal(perm[begins[i]:begins[i + 1]], host_order(c) + begins[i]), i


def test_grid_subsample_same_rows_with_either_order_engine(monkeypatch):
    """ext.grid_subsampling(order="reference") through the device evaluation equals the golden (reference C++) rows."""
    from helpers import c1_points, load_golden
    from gaussreg_amd import ext
    g = load_golden("ext_c1.npz")
    sp, sl = ext.grid_subsampling(torch.from_numpy(c1_points()).cuda(), torch.tensor([20000]), 0.05)
    assert np.array_equal(sp.cpu().numpy().view(np.uint32), g["s_points"].view(np.uint32)) and sl.tolist() == g["s_lengths"].tolist()
