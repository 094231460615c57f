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


def test_stage_profile_covers_the_pipeline(pairs):
    from gaussreg_amd import pair_pipeline
    reg = pair_pipeline.PairRegistrar(torch.device("cuda", 0), profile=True)
    reg.register_pairs(pairs[:2])
    assert set(reg.section_ms) == {"fps", "pyramid", "point_to_node", "coarse_features", "superpoint_matching",
                                   "patch_features", "sinkhorn", "local_global_registration", "ransac", "metrics"}
    assert all(v > 0 for v in reg.section_ms.values())
