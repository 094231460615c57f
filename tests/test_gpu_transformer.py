"""GPU parity of the GeometricTransformer stack against goldens generated from the reference's module
(tests/golden/gen_golden_transformer.py): GaussReg's configuration on 767 / 701 superpoints, and a small configuration with
key masks whose per-layer attention scores are compared too.  Bars: 1e-5 of the tensor scale against the fp64 evaluation
and against the reference's fp32 values."""
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, load_golden

sys.path.insert(0, GOLDEN)
import demo_inputs  # noqa: E402

pytestmark = pytest.mark.gpu
BLOCKS6 = ['self', 'cross', 'self', 'cross', 'self', 'cross']


def _sums(module):
    return np.array([float(p.detach().double().sum()) for _, p in sorted(module.state_dict().items())], np.float64)


def _close(got, ref32, ref64, what, ref_factor=0.0):
    """1e-5 of the tensor scale against the fp64 evaluation and against the reference's fp32 values.  `ref_factor` > 0 adds
    that multiple of the reference's OWN fp32 error to the bar: the softmax rows of the RPE layers amplify the fp32
    rounding of the sinusoidal arguments (distance / sigma_d up to 25 rad times the frequencies), which the reference's
    fp32 evaluation carries just the same (its |f32 - f64| on those scores is 1.5e-4 of their scale)."""
    got = np.asarray(got, np.float64)
    scale = np.abs(ref64).max()
    bar = 1e-5 * scale + ref_factor * np.abs(ref32.astype(np.float64) - ref64).max()
    e64 = np.abs(got - ref64).max()
    e32 = np.abs(got - ref32.astype(np.float64)).max()
    assert e64 <= bar, f"{what}: |hip - f64| = {e64:.3e} at scale {scale:.3e} (bar {bar:.3e})"
    assert e32 <= bar, f"{what}: |hip - ref32| = {e32:.3e} at scale {scale:.3e} (bar {bar:.3e})"


@pytest.fixture(scope="module")
def gold():
    return load_golden("transformer.npz")


def test_geometric_transformer_demo_configuration(gold):
    from geotransformer.modules.geotransformer import GeometricTransformer
    rp, sp, rf, sf = demo_inputs.transformer_inputs(767, 701, 2048)
    np.testing.assert_allclose([float(t.double().sum()) for t in (rp, sp, rf, sf)], gold["demo_sums"], rtol=1e-11)
    torch.manual_seed(int(gold["seed"]))
    net = GeometricTransformer(2048, 256, 256, 4, BLOCKS6, 0.2, 15, 3, reduction_a='max')
    np.testing.assert_allclose(_sums(net), gold["demo_param_sums"], rtol=0, atol=1e-6)  # uniform_ rounds the last ulp differently across host CPUs
    net = net.cuda().eval()
    a, b = net(rp.cuda(), sp.cuda(), rf.cuda(), sf.cuda())
    assert a.shape == (1, 767, 256) and b.shape == (1, 701, 256)
    _close(a[0].cpu().numpy()[gold["demo_rows_ref"]], gold["demo_ref32"], gold["demo_ref64"], "ref feats")
    _close(b[0].cpu().numpy()[gold["demo_rows_src"]], gold["demo_src32"], gold["demo_src64"], "src feats")
    # every row takes part: column sums over all superpoints (n terms, each within 1e-5 * scale, signs mixed)
    for got, want in ((a, gold["demo_colsum_ref64"]), (b, gold["demo_colsum_src64"])):
        n, scale = got.shape[1], float(got.abs().max())
        assert np.abs(got[0].double().sum(0).cpu().numpy() - want).max() <= 1e-5 * scale * 2 * np.sqrt(n)


def test_geometric_transformer_masks_and_scores(gold):
    from geotransformer.modules.geotransformer import GeometricTransformer
    n_ref, n_src = 90, 75
    rp, sp, rf, sf = demo_inputs.transformer_inputs(n_ref, n_src, 96, seed=23)
    np.testing.assert_allclose([float(t.double().sum()) for t in (rp, sp, rf, sf)], gold["small_sums"], rtol=1e-11)
    masks = torch.from_numpy(gold["small_masks"])
    rm, sm = masks[None, :n_ref].cuda(), masks[None, n_ref:].cuda()
    torch.manual_seed(int(gold["seed"]) + 1)
    net = GeometricTransformer(96, 32, 64, 4, ['self', 'cross', 'self', 'cross'], 0.2, 15, 3, reduction_a='mean')
    np.testing.assert_allclose(_sums(net), gold["small_param_sums"], rtol=0, atol=1e-6)  # uniform_ rounds the last ulp differently across host CPUs
    net = net.cuda().eval()
    net.transformer.return_attention_scores = True
    with torch.no_grad():
        e0, e1 = net.embedding(rp.cuda()), net.embedding(sp.cuda())
        f0, f1, scores = net.transformer(net.in_proj(rf.cuda()), net.in_proj(sf.cuda()), e0, e1, masks0=rm, masks1=sm)
        a, b = net.out_proj(f0), net.out_proj(f1)
    _close(a[0].cpu().numpy(), gold["small_ref32"], gold["small_ref64"], "ref feats (masked)")
    _close(b[0].cpu().numpy(), gold["small_src32"], gold["small_src64"], "src feats (masked)")
    assert len(scores) == 4
    for i, pair in enumerate(scores):
        for j in range(2):
            got = pair[j][0].cpu().numpy()
            want32, want64 = gold[f"small_scores32_{i}_{j}"], gold[f"small_scores64_{i}_{j}"]
            assert got.shape == want64.shape
            # self blocks (even i): direction 0 attends to the ref cloud; cross blocks (odd i): direction 0 attends to src
            key_mask = gold["small_masks"][:n_ref] if (i % 2) == j else gold["small_masks"][n_ref:]
            assert got.shape[-1] == key_mask.shape[0]
            assert (got[..., key_mask] == 0).all(), "masked keys must receive exactly zero attention"
            _close(got, want32, want64, f"attention scores layer {i} direction {j}", ref_factor=2.0)
    # the two-function default path returns the same features
    net.transformer.return_attention_scores = False
    a2, b2 = net(rp.cuda(), sp.cuda(), rf.cuda(), sf.cuda(), ref_masks=rm, src_masks=sm)
    assert torch.equal(a2, a) and torch.equal(b2, b)


def test_parallel_cross_attention_reads_previous_features():
    """conditional_transformer.py:103-107: with parallel=True both directions of a cross block see the un-updated features."""
    from geotransformer.modules.transformer import RPEConditionalTransformer
    torch.manual_seed(0)
    seq = RPEConditionalTransformer(['cross'], 64, 4).cuda().eval()
    par = RPEConditionalTransformer(['cross'], 64, 4, parallel=True).cuda().eval()
    par.load_state_dict(seq.state_dict())
    f0, f1 = torch.randn(1, 40, 64, device="cuda"), torch.randn(1, 33, 64, device="cuda")
    s0, s1 = seq(f0, f1, None, None)
    p0, p1 = par(f0, f1, None, None)
    want1, _ = par.layers[0](f1, f0)
    assert torch.equal(s0, p0) and torch.equal(p1, want1) and not torch.equal(s1, p1)
