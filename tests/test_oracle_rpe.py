"""CPU: the NumPy restatement of the RPE rows vs the reference's own outputs (tests/golden/rpe.npz)."""
import numpy as np
import pytest

from helpers import load_golden
from oracle import rpe_np as R


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_gse_oracle_matches_reference(tag):
    g = load_golden("rpe.npz")
    c, k, mean = g[f"gse_{tag}_cfg"]
    pts = g[f"gse_{tag}_points"][0]
    d_idx, a_idx = R.embedding_indices(pts, 0.2, 15, int(k))
    off = ~np.eye(pts.shape[0], dtype=bool)   # the diagonal of d_indices is rounding noise of x2 - 2xy + y2
    np.testing.assert_allclose(d_idx[off], g[f"gse_{tag}_d_idx"][0][off], rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(a_idx[off], g[f"gse_{tag}_a_idx"][0][off], rtol=1e-4, atol=2e-4)
    out = R.geometric_structure_embedding(pts, g[f"gse_{tag}_w_d"], g[f"gse_{tag}_b_d"], g[f"gse_{tag}_w_a"],
                                          g[f"gse_{tag}_b_a"], g[f"gse_{tag}_div"], 0.2, 15, int(k),
                                          "mean" if mean else "max")
    np.testing.assert_allclose(out[off], g[f"gse_{tag}_out"][0][off], rtol=1e-3, atol=2e-3)


def test_rpe_attention_oracle_matches_reference():
    g = load_golden("rpe.npz")
    c, h = g["rpe_cfg"]
    sd = {k[len("rpe_sd_"):].replace("__", "."): v for k, v in g.items() if k.startswith("rpe_sd_")}
    hid, sc = R.rpe_multi_head_attention(sd, int(h), g["rpe_q"][0], g["rpe_k"][0], g["rpe_k"][0], g["rpe_emb"][0])
    np.testing.assert_allclose(hid, g["rpe_h0"][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sc, g["rpe_s0"][0], rtol=1e-4, atol=1e-6)
    hid, sc = R.rpe_multi_head_attention(sd, int(h), g["rpe_q"][0], g["rpe_k"][0], g["rpe_k"][0], g["rpe_emb"][0],
                                         g["rpe_weights"][0], g["rpe_masks"][0], g["rpe_factors"][0])
    np.testing.assert_allclose(hid, g["rpe_h1"][0], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(sc, g["rpe_s1"][0], rtol=1e-4, atol=1e-6)
