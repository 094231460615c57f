"""CPU: pin oracle/matching_np.py against golden vectors produced by the reference's own Python
modules (tests/golden/gen_golden_matching.py)."""
import numpy as np

from helpers import load_golden
from oracle import matching_np as M


def test_pairwise_distance():
    g = load_golden("matching.npz")
    np.testing.assert_allclose(M.pairwise_distance(g["pd_x"], g["pd_y"]), g["pd_plain"], rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(M.pairwise_distance(g["pd_xn"], g["pd_yn"], True), g["pd_normalized"], rtol=1e-5, atol=2e-6)


def test_superpoint_matching():
    g = load_golden("matching.npz")
    ri, si, sc, _ = M.superpoint_matching(g["spm_ref"], g["spm_src"], g["spm_ref_masks"], g["spm_src_masks"], 256, True)
    np.testing.assert_allclose(sc, g["spm_scores"], rtol=1e-5)
    assert np.array_equal(ri, g["spm_ref_idx"]) and np.array_equal(si, g["spm_src_idx"])
    ri, si, sc, _ = M.superpoint_matching(g["spm_ref"], g["spm_src"], None, None, 64, False)
    np.testing.assert_allclose(sc, g["spm_nodual_scores"], rtol=1e-5)
    assert np.array_equal(ri, g["spm_nodual_ref_idx"]) and np.array_equal(si, g["spm_nodual_src_idx"])


def test_point_matching():
    g = load_golden("matching.npz")
    args = (g["pm_ref_points"], g["pm_src_points"], g["pm_ref_masks"], g["pm_src_masks"], g["pm_ref_idx"],
            g["pm_src_idx"], g["pm_score"], g["pm_global"])
    rp, sp, ri, si, sc, corr = M.point_matching(*args)
    assert np.array_equal(corr, g["pm_corr_mat"])
    assert np.array_equal(ri, g["pm_out_ref_idx"]) and np.array_equal(si, g["pm_out_src_idx"])
    assert np.array_equal(rp, g["pm_out_ref_points"]) and np.array_equal(sp, g["pm_out_src_points"])
    np.testing.assert_allclose(sc, g["pm_out_scores"], rtol=1e-6)
    rp, sp, ri, si, sc, corr = M.point_matching(*args, k=2, mutual=False, confidence_threshold=0.1, use_global_score=True)
    assert np.array_equal(ri, g["pm2_out_ref_idx"]) and np.array_equal(si, g["pm2_out_src_idx"])
    np.testing.assert_allclose(sc, g["pm2_out_scores"], rtol=1e-6)


def test_point_to_node_partition():
    g = load_golden("matching.npz")
    p2n, nm, idx, km, _ = M.point_to_node_partition(g["p2n_points"], g["p2n_nodes"], 64)
    assert np.array_equal(p2n, g["p2n_point_to_node"])
    assert np.array_equal(nm, g["p2n_node_masks"]) and not nm[-1]
    assert np.array_equal(km, g["p2n_knn_masks"])
    assert np.array_equal(idx, g["p2n_knn_idx"])


def test_local_global_registration():
    g = load_golden("matching.npz")
    r, s_, sc, T = M.local_global_registration(g["lgr_ref_points"], g["lgr_src_points"], g["lgr_ref_masks"],
                                               g["lgr_src_masks"], g["lgr_score"])
    assert np.array_equal(r, g["lgr_out_ref"]) and np.array_equal(s_, g["lgr_out_src"])
    np.testing.assert_allclose(sc, g["lgr_out_scores"], rtol=1e-6)
    np.testing.assert_allclose(T, g["lgr_out_transform"], atol=2e-5)
    np.testing.assert_allclose(T, g["lgr_true_transform"], atol=5e-3)  # and it found the planted motion
