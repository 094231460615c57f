"""CPU: pin the NumPy restatements of the "next" rows against golden vectors from the reference's modules."""
import numpy as np

from helpers import load_golden
from oracle import matching_np as M


def test_sinkhorn():
    g = load_golden("next_rows.npz")
    o = M.sinkhorn(g["sk_scores"], g["sk_row_masks"], g["sk_col_masks"], alpha=1.0)
    np.testing.assert_allclose(o, g["sk_out_alpha1"], rtol=2e-5, atol=2e-4)
    o = M.sinkhorn(g["sk_scores"], g["sk_row_masks"], g["sk_col_masks"], alpha=0.37)
    np.testing.assert_allclose(o, g["sk_out_alpha037"], rtol=2e-5, atol=2e-4)
    o = M.sinkhorn(g["sk_scores"], alpha=0.37)
    np.testing.assert_allclose(o, g["sk_out_nomask_alpha037"], rtol=2e-5, atol=2e-4)


def test_kpconv_and_pools():
    g = load_golden("next_rows.npz")
    y = M.kpconv(g["kp_s_feats"], g["kp_q_points"], g["kp_s_points"], g["kp_neighbors"], g["kp_kernel_points"],
                 g["kp_weights"], float(g["kp_sigma"]), g["kp_bias"])
    np.testing.assert_allclose(y, g["kp_out"], rtol=1e-4, atol=1e-5)
    assert np.array_equal(M.maxpool(g["kp_s_feats"], g["kp_neighbors"]), g["kp_maxpool"])
    assert np.array_equal(M.nearest_upsample(g["kp_s_feats"], g["kp_neighbors"]), g["kp_upsample"])


def test_fps_c_restatement_equals_the_numpy_one():
    """oracle/fps_oracle.c (used for the 200 000 -> 30 000 GPU test) against the NumPy definition it restates."""
    from oracle import capi
    rng = np.random.default_rng(4)
    p = (rng.random((7000, 3)) * [6, 5, 3]).astype(np.float32)
    p[100] = p[7]; p[2000] = p[7]                       # identical points: equal distances, first maximum wins
    for k, start in ((1, 0), (2, 5), (900, 0), (7000, 6999)):
        assert np.array_equal(capi.farthest_point_sampling(p, k, start), M.farthest_point_sampling(p, k, start))
