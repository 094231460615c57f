"""configs[4] pair path (gaussreg_amd/pair_pipeline.py): synthetic room pairs register to their ground truth, the stream /
thread fan-out of the per-pair stage changes nothing, and the stage profiler accounts for every stage."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pairs():
    from gaussreg_amd import pair_pipeline
    dev = torch.device("cuda", 0)
    return [pair_pipeline.synthetic_room_pair(100 + b, 60000, dev) for b in range(5)]


def test_pairs_register_and_streams_do_not_change_results(pairs):
    from gaussreg_amd import pair_pipeline
    dev = torch.device("cuda", 0)
    seq = pair_pipeline.PairRegistrar(dev, pair_streams=1).register_pairs(pairs)
    assert seq.shape == (5, pair_pipeline.RESULT_LEN)
    assert (seq[:, 16] < 5.0).all() and (seq[:, 17] < 0.1).all(), seq[:, 16:18]      # RRE deg, RTE m
    assert (seq[:, 18] >= 3).all()                                                    # correspondences
    for b, (_, _, T_gt) in enumerate(pairs):
        T = seq[b, :16].reshape(4, 4)
        assert torch.allclose(T[3], torch.tensor([0.0, 0.0, 0.0, 1.0], device=dev))
        assert torch.linalg.norm(T[:3, 3] - T_gt[:3, 3]) < 0.1
    for S in (2, 4):
        reg = pair_pipeline.PairRegistrar(dev, pair_streams=S)
        assert torch.equal(reg.register_pairs(pairs), seq)
        assert torch.equal(reg.register_pairs(pairs), seq)      # the pool and its streams are reused
    # the default: every per-pair stage through the stack-mode entry points (two read-backs per batch, no host threads)
    batched = pair_pipeline.PairRegistrar(dev)
    assert batched.pair_streams == 0
    got = batched.register_pairs(pairs)
    # transform, RRE, RTE, number of correspondences: bit for bit; the inlier ratio against the ground truth is a metric of
    # this pipeline (not of the reference) and is summed differently in the batched form
    assert torch.equal(got[:, :19], seq[:, :19]), (got - seq).abs().max(0).values
    assert torch.allclose(got[:, 19], seq[:, 19], atol=1e-3)
    many = batched.register_many(pairs, 2)         # FPS over all clouds first, then blocks of two pairs
    blocks = torch.cat([batched.register_pairs(pairs[i:i + 2]) for i in range(0, 5, 2)])
    assert torch.equal(many[:, :19], blocks[:, :19])
    sub = batched.register_pairs(pairs[1:4])       # another batch composition: same correspondences; RANSAC draws with the
    assert torch.equal(sub[:, 18], seq[1:4, 18])   # pair's position in the batch as seed, so its estimate moves a little
    assert torch.allclose(sub[:, :16], seq[1:4, :16], atol=2e-2)


def test_stage_profile_covers_the_pipeline(pairs):
    from gaussreg_amd import pair_pipeline
    reg = pair_pipeline.PairRegistrar(torch.device("cuda", 0), profile=True)
    reg.register_pairs(pairs[:2])
    assert set(reg.section_ms) == {"fps", "pyramid", "point_to_node", "standin_descriptors", "superpoint_matching",
                                   "patch_features", "patch_scores", "sinkhorn", "local_global_registration", "ransac",
                                   "metrics"}
    assert all(v > 0 for v in reg.section_ms.values())


def test_segmented_group_norm_equals_per_segment_calls():
    """gr_group_norm_seg (several pairs through the backbone at once) against one plain GroupNorm call per segment."""
    from gaussreg_amd.kpconv_blocks import GroupNorm, norm_segments
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    for C, G in ((64, 32), (256, 32), (2048, 32)):
        gn = GroupNorm(G, C).to(dev)
        with torch.no_grad():
            gn.norm.weight.copy_(torch.rand(C, generator=g, device=dev) + 0.5)
            gn.norm.bias.copy_(torch.rand(C, generator=g, device=dev) - 0.5)
            lens = [1000, 1, 7777, 300, 12001]
            x = torch.randn((sum(lens), C), generator=g, device=dev) * 3 + 1
            offs = [0]
            for n in lens:
                offs.append(offs[-1] + n)
            for slope in (None, 0.1):
                want = torch.cat([gn(x[a:b].contiguous(), slope).reshape(b - a, C) for a, b in zip(offs[:-1], offs[1:])])
                with norm_segments({x.shape[0]: (torch.tensor(offs, dtype=torch.int64, device=dev), max(lens))}):
                    got = gn(x, slope)
                assert torch.allclose(got, want, rtol=1e-5, atol=1e-6), (C, slope, (got - want).abs().max())


def test_network_features_batched_backbone_equals_the_pair_alone(pairs):
    """features="model": KPConvFPN runs once over all clouds of the batch with per-pair GroupNorm statistics -- the fine and
    coarse features of a pair must be what the backbone gives that pair alone; and the pipeline runs end to end."""
    from gaussreg_amd import pair_pipeline
    from gaussreg_amd.data import precompute_data_stack_mode
    from gaussreg_amd.kpconv_blocks import norm_segments
    dev = torch.device("cuda", 0)
    reg = pair_pipeline.PairRegistrar(dev, features="model", pair_streams=2)
    out = reg.register_pairs(pairs[:3])
    assert out.shape == (3, pair_pipeline.RESULT_LEN) and torch.isfinite(out[:, :16]).all()
    regb = pair_pipeline.PairRegistrar(dev, features="model")          # stack-mode per-pair stages, same weights (seeded)
    outb = regb.register_pairs(pairs[:3])
    # (random weights: the estimates are noise and amplify any difference in summation order, so only the contract is checked)
    assert outb.shape == out.shape and torch.isfinite(outb[:, :16]).all()
    # batched vs alone, on the FPS-sampled clouds of two pairs
    sampled = reg._sample(pairs[:2], 24)

    def pyramid(clouds):
        pts = torch.cat(clouds, 0).contiguous()
        lens = torch.tensor([c.shape[0] for c in clouds], dtype=torch.int64)
        return precompute_data_stack_mode(pts, lens, pair_pipeline.NUM_STAGES, pair_pipeline.INIT_VOXEL,
                                          pair_pipeline.INIT_RADIUS, pair_pipeline.NEIGHBOR_LIMITS)

    with torch.no_grad():
        both = pyramid(sampled)
        table = {}
        for lv in range(pair_pipeline.NUM_STAGES):
            ll = both["lengths"][lv].tolist()
            offs = [0, ll[0] + ll[1], sum(ll)]
            table[offs[-1]] = (torch.tensor(offs, dtype=torch.int64, device=dev), max(offs[1], offs[2] - offs[1]))
        with norm_segments(table):
            fb = reg.net.backbone(torch.ones((both["points"][0].shape[0], 4), device=dev), both)
        alone = pyramid(sampled[2:4])                                    # the second pair on its own
        fa = reg.net.backbone(torch.ones((alone["points"][0].shape[0], 4), device=dev), alone)
    for lv_feats_b, lv_feats_a, lv in ((fb[0], fa[0], 1), (fb[-1], fa[-1], 4)):
        ll = both["lengths"][lv].tolist()
        mine = lv_feats_b[ll[0] + ll[1]:]
        assert mine.shape == lv_feats_a.shape
        scale = lv_feats_a.abs().max()
        assert (mine - lv_feats_a).abs().max() <= 2e-5 * scale, ((mine - lv_feats_a).abs().max(), scale)
    reg.close()
