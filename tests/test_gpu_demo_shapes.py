"""GPU parity AT DEMO SHAPES against goldens generated from the reference itself (tests/golden/gen_golden_demo.py,
gen_golden_backbone.py): the 2 x 30 000-point pyramid, PointMatching / Sinkhorn at 256 x 128 x 128, KPConv at a real stage
shape, LocalGlobalRegistration with correspondence_limit, index_select, calibrate_neighbors_stack_mode, the KPConv blocks
and the whole KPConvFPN backbone.  Inputs are regenerated from the seeds the generators used (checksums asserted).

Float bars: where the reference's own fp32 result and an fp64 evaluation of the same module are both stored, the test
asserts max|hip - f64| <= max|ref32 - f64| * FACTOR + tiny -- the HIP result must sit as close to the exact answer as the
reference's own arithmetic does (GPU and CPU sum in different orders, so elementwise equality with ref32 is not the
right question) -- and, separately, 1e-5 relative to the tensor's scale against the reference's fp32 values."""
import os
import sys

import numpy as np
import pytest
import torch

from helpers import GOLDEN, assert_neighbors_equal_up_to_ties, load_golden

sys.path.insert(0, GOLDEN)
import demo_inputs  # noqa: E402
from gen_golden_ext import room_pair  # noqa: E402

pytestmark = pytest.mark.gpu
LIMITS = [89, 30, 43, 49, 49]


def _c(a):
    if isinstance(a, torch.Tensor):
        return a.cuda()
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def same_sum(t, stored):
    """fp64 sum of a regenerated input vs the generator's (the reduction order depends on the host's thread count)."""
    return abs(float(t.double().sum()) - float(stored)) <= 1e-11 * max(1.0, abs(float(stored)))


def assert_as_exact_as_reference(got, ref32, ref64, factor=8.0, what=""):
    """The north_star bar, against the EXACT answer: |hip - f64| <= 1e-5 * scale (and the same against the reference's fp32
    values); plus: the HIP rounding error stays within `factor` of the reference's own fp32 rounding error (the GPU kernels
    accumulate longer fp32 chains -- e.g. 15 x Cin products per KPConv output -- than ATen's blocked CPU sums)."""
    got = np.asarray(got, np.float64)
    e_hip = np.abs(got - ref64).max()
    e_ref = np.abs(ref32.astype(np.float64) - ref64).max()
    scale = np.abs(ref64).max()
    assert e_hip <= 1e-5 * scale, f"{what}: |hip - f64| = {e_hip:.3e}, scale {scale:.3g}"
    assert np.abs(got - ref32).max() <= 1e-5 * scale, f"{what}: max |hip - ref32| = {np.abs(got - ref32).max():.3e}, scale {scale:.3g}"
    assert e_hip <= factor * e_ref + 1e-7 * scale, f"{what}: |hip - f64| = {e_hip:.3e} vs |ref32 - f64| = {e_ref:.3e} (scale {scale:.3g})"


@pytest.fixture(scope="module")
def demo():
    return load_golden("demo_shapes.npz")


@pytest.fixture(scope="module")
def pyramid(demo):
    from geotransformer.utils.data import precompute_data_stack_mode
    ref, src = room_pair(30000, 0)
    points = np.concatenate([ref, src]).astype(np.float32)
    assert float(points.astype(np.float64).sum()) == float(demo["pyr_points_sum"])
    return points, precompute_data_stack_mode(_c(points), torch.tensor([30000, 30000]), 5, 0.025, 0.0625, LIMITS)


def test_pyramid_2x30000_matches_reference_core(demo, pyramid):
    """Points, lengths and row order bit-exact at every level; neighbour tensors identical to the reference core's except
    inside runs of EQUAL fp32 distance (a barycentre is equidistant from the two points of its voxel, and the reference's
    order there depends on its kd-tree traversal): sampled rows are compared tie-tolerantly and a
    row-order-invariant checksum covers ALL rows; the self-searches are also compared with the exact checksum."""
    _, d = pyramid
    for i in range(5):
        assert d["lengths"][i].tolist() == demo[f"pyr_len_{i}"].tolist()
        p = d["points"][i].cpu().numpy()
        assert demo_inputs.checksum_i64(p.view(np.uint32)) == demo[f"pyr_pts_bits_{i}"], f"level {i} points (bits, order)"
        assert np.array_equal(p[demo[f"pyr_rows_{i}"]], demo[f"pyr_pts_rows_{i}"])
        for name, key in (("nb", "neighbors"), ("sub", "subsampling"), ("up", "upsampling")):
            if i < len(d[key]):
                a = d[key][i].cpu().numpy()
                assert list(a.shape) == demo[f"pyr_{name}_shape_{i}"].tolist(), (name, i)
                assert a.flags["C_CONTIGUOUS"]
                rows = demo[f"pyr_{name}_rows_{i}"]
                qi, si = {"nb": (i, i), "sub": (i + 1, i), "up": (i, i + 1)}[name]
                q, s = d["points"][qi].cpu().numpy(), d["points"][si].cpu().numpy()
                assert_neighbors_equal_up_to_ties(a[rows], demo[f"pyr_{name}_vals_{i}"].astype(np.int64), q[rows], s,
                                                  np.array([len(rows)]), np.array([s.shape[0]]))
                assert demo_inputs.checksum_i64(np.sort(a, axis=1)) == demo[f"pyr_{name}_setsum_{i}"], \
                    f"{name} level {i}: row-set checksum over all rows"
                # exact order on EVERY row that has no equal-distance run (same rows as in the reference's output)
                tr = demo_inputs.tie_rows(q, s, a)
                assert np.array_equal(tr, demo[f"pyr_{name}_tierows_{i}"]), f"{name} level {i}: rows with equal-distance runs"
                keep = np.ones(a.shape[0], bool)
                keep[tr] = False
                assert demo_inputs.checksum_i64(a[keep]) == demo[f"pyr_{name}_notie_sum_{i}"], f"{name} level {i}: exact checksum"


def test_point_matching_256x128x128(demo):
    from geotransformer.modules.geotransformer import PointMatching
    x = demo_inputs.point_matching_inputs()
    assert same_sum(x["score"], demo["pm_score_sum"])
    pm = PointMatching(k=3, mutual=True, confidence_threshold=0.05, use_dustbin=False, use_global_score=False)
    want = np.unpackbits(demo["pm_corr_bits"])[: 256 * 128 * 128].reshape(256, 128, 128).astype(bool)
    corr = pm.compute_correspondence_matrix(torch.exp(x["score"]).cuda(), _c(x["ref_masks"]), _c(x["src_masks"])).cpu().numpy()
    assert np.array_equal(corr, want)
    a, b, c, d, e = pm(_c(x["ref_points"]), _c(x["src_points"]), _c(x["ref_masks"]), _c(x["src_masks"]), _c(x["ref_idx"]),
                       _c(x["src_idx"]), _c(x["score"]), _c(x["global_scores"]))
    assert np.array_equal(c.cpu().numpy(), demo["pm_out_ref_idx"]) and np.array_equal(d.cpu().numpy(), demo["pm_out_src_idx"])
    assert np.array_equal(a.cpu().numpy(), demo["pm_out_ref_points"]) and np.array_equal(b.cpu().numpy(), demo["pm_out_src_points"])
    np.testing.assert_allclose(e.cpu().numpy(), demo["pm_out_scores"], rtol=1e-5, atol=0)


def test_sinkhorn_256x128x128(demo):
    from geotransformer.modules.sinkhorn import LearnableLogOptimalTransport
    sc, rm, cm = demo_inputs.sinkhorn_inputs()
    assert same_sum(sc, demo["sk_scores_sum"])
    ot = LearnableLogOptimalTransport(100).cuda()
    with torch.no_grad():
        ot.alpha.fill_(float(demo["sk_alpha"]))
    o = ot(sc.cuda(), rm.cuda(), cm.cuda())
    assert o.shape == (256, 129, 129)
    got = o[torch.from_numpy(demo["sk_pick"]).cuda()].cpu().numpy()
    r32, r64 = demo["sk_out32"], demo["sk_out64"]
    valid = r64 > -1e6  # masked rows / columns hold -inf stand-ins (-1e12 + small terms): compared separately
    assert np.array_equal(got > -1e6, valid)
    g64 = got.astype(np.float64)
    e_hip, e_ref = np.abs(g64 - r64)[valid].max(), np.abs(r32.astype(np.float64) - r64)[valid].max()
    scale = np.abs(r64[valid]).max()
    assert e_hip <= 1e-5 * scale and e_hip <= 8.0 * e_ref + 1e-7 * scale, (e_hip, e_ref, scale)
    assert np.abs(g64 - r32)[valid].max() <= 1e-5 * scale
    np.testing.assert_allclose(got[~valid], r32[~valid], rtol=1e-6)


def test_kpconv_stage_shape(demo, pyramid):
    from geotransformer.modules.kpconv import KPConv
    p1, nb1 = pyramid[1]["points"][1], pyramid[1]["neighbors"][1]
    feats, w = demo_inputs.kpconv_inputs(p1.shape[0])
    assert same_sum(feats, demo["kp_feats_sum"]) and same_sum(w, demo["kp_w_sum"])
    conv = KPConv(64, 64, 15, float(demo["kp_radius"]), float(demo["kp_sigma"]), bias=False,
                  kernel_points=demo["kp_kernel_points"]).cuda()
    with torch.no_grad():
        conv.weights.copy_(w.cuda())
    y = conv(feats.cuda(), p1, p1, nb1)
    assert y.shape == (p1.shape[0], 64)
    rows = torch.from_numpy(demo["kp_rows"]).cuda()
    assert_as_exact_as_reference(y[rows].cpu().numpy(), demo["kp_out32"], demo["kp_out64"], what="KPConv rows")
    colsum = y.double().sum(0).cpu().numpy()
    assert np.abs(colsum - demo["kp_colsum64"]).max() <= 1e-5 * np.abs(demo["kp_colsum64"]).max() + 1e-3


def test_lgr_correspondence_limit(demo):
    from geotransformer.modules.geotransformer import LocalGlobalRegistration
    ref_k, src_k, rk, sk, lscore, gsc = demo_inputs.lgr_limit_inputs()
    assert same_sum(lscore, demo["lgrl_score_sum"])
    lgr = LocalGlobalRegistration(3, 0.1, mutual=True, confidence_threshold=0.05, use_dustbin=False, use_global_score=True,
                                  correspondence_threshold=3, correspondence_limit=int(demo["lgrl_limit"]), num_refinement_steps=5)
    a, b, c, T = lgr(ref_k.cuda(), src_k.cuda(), rk.cuda(), sk.cuda(), lscore.cuda(), gsc.cuda())
    assert np.array_equal(a.cpu().numpy(), demo["lgrl_out_ref"]) and np.array_equal(b.cpu().numpy(), demo["lgrl_out_src"])
    np.testing.assert_allclose(c.cpu().numpy(), demo["lgrl_out_scores"], rtol=1e-5, atol=0)
    np.testing.assert_allclose(T.cpu().numpy(), demo["lgrl_out_transform"], rtol=0, atol=2e-5)


def test_index_select_and_calibrate(demo):
    from geotransformer.modules.ops import index_select
    from geotransformer.utils.data import calibrate_neighbors_stack_mode, registration_collate_fn_stack_mode
    data, idx = _c(demo["is_data"]), _c(demo["is_idx"].astype(np.int64))
    o0 = index_select(data, idx, dim=0)
    assert o0.shape == (40, 9, 7) and np.array_equal(o0.cpu().numpy(), demo["is_out0"])
    o1 = index_select(data.t().contiguous(), idx, dim=1)
    assert o1.shape == (7, 40, 9) and np.array_equal(o1.cpu().numpy(), demo["is_out1"])
    assert torch.equal(index_select(data, idx[0], 0), data[idx[0]])
    with pytest.raises(IndexError):
        index_select(data, torch.tensor([3, 500], device="cuda"), 0)

    class TinySet:
        def __len__(self):
            return 3

        def __getitem__(self, i):
            r_, s_ = room_pair(4000, 100 + i)
            return {"ref_points": r_, "src_points": s_, "ref_feats": np.ones((r_.shape[0], 1), np.float32),
                    "src_feats": np.ones((s_.shape[0], 1), np.float32)}

    limits = calibrate_neighbors_stack_mode(TinySet(), registration_collate_fn_stack_mode, 4, 0.025, 0.0625, keep_ratio=0.8,
                                            sample_threshold=2000)
    assert np.asarray(limits).tolist() == demo["calib_limits"].tolist()


# ---------------------------------------------------------------------------------------- KPConv blocks / KPConvFPN
@pytest.fixture(scope="module")
def bb():
    return load_golden("backbone.npz")


@pytest.fixture(scope="module")
def bb_inputs(bb):
    from geotransformer.utils.data import precompute_data_stack_mode
    ref, src = room_pair(6000, 11)
    points = np.concatenate([ref, src]).astype(np.float32)
    assert float(points.astype(np.float64).sum()) == float(bb["points_sum"])
    d = precompute_data_stack_mode(_c(points), torch.tensor([6000, 6000]), 5, 0.025, 0.0625, bb["limits"].tolist())
    assert [p.shape[0] for p in d["points"]] == bb["level_sizes"].tolist()
    feats = demo_inputs.backbone_feats(points.shape[0])
    assert same_sum(feats, bb["feats_sum"])
    return d, feats.cuda()


def _set_kernel_points(module):
    from gaussreg_amd.kpconv import KPConv
    for m in module.modules():
        if isinstance(m, KPConv):
            m.kernel_points.copy_(torch.from_numpy((demo_inputs.K015 * m.radius).astype(np.float32)))


def _param_sums(module):
    return np.array([float(p.detach().double().sum()) for _, p in sorted(module.state_dict().items())
                     if not _.endswith("kernel_points")], np.float64)


def _ref_param_sums(stored, module):
    keys = [k for k, _ in sorted(module.state_dict().items())]
    keep = [i for i, k in enumerate(keys) if not k.endswith("kernel_points")]
    return stored[keep]


def test_kpconv_blocks_vs_reference(bb, bb_inputs):
    from geotransformer.modules.kpconv import ConvBlock, ResidualBlock
    d, feats = bb_inputs
    torch.manual_seed(int(bb["seed"]))
    cb = ConvBlock(4, 64, 15, 0.0625, 0.05, 32)
    rb = ResidualBlock(64, 128, 15, 0.0625, 0.05, 32)
    rbs = ResidualBlock(128, 128, 15, 0.0625, 0.05, 32, strided=True)
    # the reference's state-dict keys, and the same seeded parameter values
    assert set(rb.state_dict().keys()) >= {"unary1.mlp.weight", "unary1.norm.norm.weight", "KPConv.weights", "KPConv.kernel_points",
                                           "norm_conv.norm.bias", "unary2.mlp.bias", "unary_shortcut.mlp.weight"}
    n1, n2 = len(cb.state_dict()), len(rb.state_dict())
    stored = bb["blk_param_sums"]
    for m, part in ((cb, stored[:n1]), (rb, stored[n1:n1 + n2]), (rbs, stored[n1 + n2:])):
        np.testing.assert_allclose(_param_sums(m), _ref_param_sums(part, m), rtol=0, atol=1e-9)
        _set_kernel_points(m)
    cb, rb, rbs = cb.cuda().eval(), rb.cuda().eval(), rbs.cuda().eval()
    with torch.no_grad():
        y1 = cb(feats, d["points"][0], d["points"][0], d["neighbors"][0])
        y2 = rb(y1, d["points"][0], d["points"][0], d["neighbors"][0])
        y3 = rbs(y2, d["points"][1], d["points"][0], d["subsampling"][0])
    r0, r1 = torch.from_numpy(bb["blk_rows0"]).cuda(), torch.from_numpy(bb["blk_rows1"]).cuda()
    assert_as_exact_as_reference(y1[r0].cpu().numpy(), bb["blk_conv32"], bb["blk_conv64"], what="ConvBlock")
    assert_as_exact_as_reference(y2[r0].cpu().numpy(), bb["blk_res32"], bb["blk_res64"], what="ResidualBlock")
    assert_as_exact_as_reference(y3[r1].cpu().numpy(), bb["blk_strided32"], bb["blk_strided64"], what="strided ResidualBlock")


def test_kpconv_fpn_backbone_vs_reference(bb, bb_inputs):
    from gaussreg_amd.kpconv_blocks import KPConvFPN
    d, feats = bb_inputs
    torch.manual_seed(int(bb["seed"]))
    net = KPConvFPN(4, 256, 64, 15, 0.0625, 0.05, 32)
    assert sum(p.numel() for p in net.parameters()) == int(bb["fpn_param_count"])
    np.testing.assert_allclose(_param_sums(net), _ref_param_sums(bb["fpn_param_sums"], net), rtol=0, atol=1e-9)
    _set_kernel_points(net)
    net = net.cuda().eval()
    with torch.no_grad():
        fl = net(feats, d)
    assert len(fl) == 4
    for i, f in enumerate(fl):
        assert list(f.shape) == bb[f"fpn_shape_{i}"].tolist()
        rows = torch.from_numpy(bb[f"fpn_rows_{i}"]).cuda()
        assert_as_exact_as_reference(f[rows].cpu().numpy(), bb[f"fpn_out32_{i}"], bb[f"fpn_out64_{i}"],
                                     what=f"KPConvFPN output {i}")
