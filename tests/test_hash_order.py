"""CPU: the flat-array replay of libstdc++'s unordered_map linking rules (product, host-only helper of
GR_ORDER_REFERENCE) gives exactly the iteration order of the real container (oracle side)."""
import ctypes

import numpy as np
import pytest

from oracle import capi


def _both(keys):
    from gaussreg_amd import _lib
    keys = np.ascontiguousarray(keys, dtype=np.uint64)
    n = keys.shape[0]
    got = np.zeros(max(n, 1), np.int32)
    _lib.check(_lib.lib().gr_host_unordered_map_order(keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n,
                                                      got.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))))
    O = capi.lib()
    O.oracle_unordered_map_order.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int64,
                                             ctypes.POINTER(ctypes.c_int32)]
    want = np.zeros(max(n, 1), np.int32)
    O.oracle_unordered_map_order(keys.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n,
                                 want.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
    return got[:n], want[:n]


@pytest.mark.parametrize("n", [0, 1, 2, 12, 13, 14, 29, 30, 59, 60, 127, 541, 1000, 5087, 20000, 74011])
def test_matches_real_container_random_keys(n):
    rng = np.random.default_rng(n)
    keys = rng.choice(np.arange(0, max(4 * n, 8), dtype=np.uint64), size=n, replace=False)
    got, want = _both(keys)
    assert np.array_equal(got, want)


def test_matches_real_container_adversarial_keys():
    # all keys collide modulo small primes; huge keys; consecutive keys
    for keys in (np.arange(0, 13 * 500, 13, dtype=np.uint64), np.arange(500, dtype=np.uint64) * np.uint64(2 ** 40) + np.uint64(7),
                 np.arange(3000, dtype=np.uint64)[::-1].copy(), np.uint64(2 ** 64 - 1) - np.arange(300, dtype=np.uint64)):
        got, want = _both(keys)
        assert np.array_equal(got, want)
