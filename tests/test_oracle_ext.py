"""CPU: pin oracle/{radius_neighbors_oracle.c, grid_subsample_oracle.cpp} against golden vectors
that were produced by the reference's own C++ (tests/golden/gen_golden_ext.py), and -- where
oracle/_ref is present -- against the reference core directly on fresh random inputs."""
import numpy as np
import pytest

from oracle import capi
from helpers import load_golden, c1_points, assert_neighbors_equal_up_to_ties


def test_c1_inputs_reproducible():
    g = load_golden("ext_c1.npz")
    pts = c1_points()
    assert np.array_equal(pts[:4], g["first_points"])
    assert float(pts.astype(np.float64).sum()) == float(g["points_sum"])


def test_radius_c1_matches_reference():
    g = load_golden("ext_c1.npz")
    pts = c1_points()
    lens = np.array([20000], np.int64)
    nb = capi.radius_neighbors(pts, pts, lens, lens, float(g["radius"]))
    assert nb.shape == (20000, 39)
    assert np.array_equal(nb, g["neighbors"].astype(np.int64))


def test_grid_c1_matches_reference_bit_exact():
    g = load_golden("ext_c1.npz")
    sp, sl = capi.grid_subsampling(c1_points(), np.array([20000], np.int64), float(g["voxel"]))
    assert sp.shape == (7366, 3) and sl.tolist() == [7366]
    assert np.array_equal(sp.view(np.uint32), g["s_points"].view(np.uint32))


def test_multibatch_matches_reference():
    g = load_golden("ext_multibatch.npz")
    nb = capi.radius_neighbors(g["q"], g["s"], g["q_lengths"], g["s_lengths"], float(g["radius"]))
    assert np.array_equal(nb, g["neighbors"].astype(np.int64))
    nbb = capi.radius_neighbors(g["q"], g["s"], g["q_lengths"], g["s_lengths"], float(g["radius"]), brute=True)
    assert np.array_equal(nbb, nb)
    sp, sl = capi.grid_subsampling(g["s"], g["s_lengths"], float(g["voxel"]))
    assert np.array_equal(sl, g["sub_lengths"])
    assert np.array_equal(sp.view(np.uint32), g["s_points"].view(np.uint32))


def test_no_neighbor_and_self_only():
    g = load_golden("ext_noneighbor.npz")
    nb = capi.radius_neighbors(g["p"], g["p"], g["lengths"], g["lengths"], float(g["r_self"]))
    assert np.array_equal(nb, g["nb_self"].astype(np.int64)) and nb.shape[1] == 1
    nb0 = capi.radius_neighbors(g["far"], g["p"], g["lengths"], g["lengths"], float(g["r_none"]))
    assert list(nb0.shape) == g["nb_none_shape"].tolist() == [500, 0]


def test_ties_equal_up_to_tie_groups():
    g = load_golden("ext_ties.npz")
    nb = capi.radius_neighbors(g["p"], g["p"], g["lengths"], g["lengths"], float(g["radius"]))
    n = assert_neighbors_equal_up_to_ties(nb, g["neighbors"], g["p"], g["p"], g["lengths"], g["lengths"])
    assert n > 0  # the fixture really exercises ties
    sp, sl = capi.grid_subsampling(g["p"], g["lengths"], float(g["voxel"]))
    assert np.array_equal(sp.view(np.uint32), g["s_points"].view(np.uint32))


def test_pyramid_matches_reference():
    g = load_golden("ext_pyramid.npz")
    v, r = float(g["voxel0"]), float(g["radius0"])
    pts, lens = g["points0"], g["lengths0"]
    P, L = [pts], [lens]
    for i in range(5):
        if i > 0:
            sp, sl = capi.grid_subsampling(P[-1], L[-1], v)
            assert np.array_equal(sp.view(np.uint32), g[f"points{i}"].view(np.uint32)), i
            assert np.array_equal(sl, g[f"lengths{i}"])
            P.append(sp)
            L.append(sl)
        v *= 2
    for i in range(5):
        nb = capi.radius_neighbors(P[i], P[i], L[i], L[i], r)
        assert_neighbors_equal_up_to_ties(nb, g[f"neighbors{i}"], P[i], P[i], L[i], L[i])
        if i < 4:
            sub = capi.radius_neighbors(P[i + 1], P[i], L[i + 1], L[i], r)
            assert_neighbors_equal_up_to_ties(sub, g[f"subsampling{i}"], P[i + 1], P[i], L[i + 1], L[i])
            up = capi.radius_neighbors(P[i], P[i + 1], L[i], L[i + 1], r * 2)
            assert_neighbors_equal_up_to_ties(up, g[f"upsampling{i}"], P[i], P[i + 1], L[i], L[i + 1])
        r *= 2


@pytest.mark.skipif(not capi.have_ref(), reason="oracle/_ref not built (reference not mounted)")
@pytest.mark.parametrize("seed", [3, 4])
def test_restatement_vs_ref_random(seed):
    rng = np.random.default_rng(seed)
    s = (rng.random((5000, 3)) * [2.0, 1.0, 0.5]).astype(np.float32)
    q = (rng.random((3000, 3)) * [2.2, 1.0, 0.5] - 0.1).astype(np.float32)
    sl = np.array([3000, 2000], np.int64)
    ql = np.array([1000, 2000], np.int64)
    want = capi.ref_radius_neighbors(q, s, ql, sl, 0.09)
    got = capi.radius_neighbors(q, s, ql, sl, 0.09)
    assert_neighbors_equal_up_to_ties(got, want, q, s, ql, sl)
    sp_w, sl_w = capi.ref_grid_subsampling(s, sl, 0.04)
    sp_g, sl_g = capi.grid_subsampling(s, sl, 0.04)
    assert np.array_equal(sl_w, sl_g) and np.array_equal(sp_w.view(np.uint32), sp_g.view(np.uint32))
