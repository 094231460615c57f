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


def test_host_mirrors_keep_reference_state_dict_keys_and_embedding_matches():
    """CPU-only: the nn.Module mirrors expose the reference's parameter / buffer names (checkpoints load unchanged) and
    the plain-torch sinusoidal embedding equals the NumPy restatement."""
    import torch
    from gaussreg_amd.embedding import GeometricStructureEmbedding, SinusoidalPositionalEmbedding
    from gaussreg_amd.rpe_attention import RPEMultiHeadAttention
    m = GeometricStructureEmbedding(64, 0.2, 15, 3)
    assert sorted(m.state_dict().keys()) == ["embedding.div_term", "proj_a.bias", "proj_a.weight", "proj_d.bias", "proj_d.weight"]
    a = RPEMultiHeadAttention(64, 4)
    assert sorted(a.state_dict().keys()) == sorted(f"proj_{n}.{p}" for n in "qkvp" for p in ("weight", "bias"))
    with pytest.raises(ValueError):
        RPEMultiHeadAttention(65, 4)
    with pytest.raises(ValueError):
        SinusoidalPositionalEmbedding(63)
    with pytest.raises(ValueError):
        GeometricStructureEmbedding(64, 0.2, 15, 3, reduction_a="sum")
    emb = SinusoidalPositionalEmbedding(32)
    idx = torch.rand(5, 7) * 20
    np.testing.assert_allclose(emb(idx).numpy(), R.sinusoidal_embedding(idx.numpy(), emb.div_term.numpy()), rtol=1e-5, atol=1e-6)
    g = load_golden("rpe.npz")
    np.testing.assert_allclose(emb.div_term.numpy(), g["gse_c_div"], rtol=1e-6)   # same buffer as the reference builds
