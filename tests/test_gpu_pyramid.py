"""GPU parity for row a7: the device-resident pyramid builder (gaussreg_amd.data, mirror of
geotransformer/utils/data.py:13-77) against the reference's golden 5-level pyramid, including the
neighbor_limit truncation done inside the fill kernel."""
import numpy as np
import pytest
import torch

from helpers import load_golden, assert_neighbors_equal_up_to_ties

pytestmark = pytest.mark.gpu


def test_precompute_data_stack_mode_matches_reference_pyramid():
    from geotransformer.utils.data import precompute_data_stack_mode  # drop-in alias package
    g = load_golden("ext_pyramid.npz")
    limits = [20, 15, 25, 30, 30]
    pts = torch.from_numpy(g["points0"]).cuda()
    out = precompute_data_stack_mode(pts, torch.from_numpy(g["lengths0"]), 5, float(g["voxel0"]), float(g["radius0"]),
                                     limits)
    assert set(out) == {"points", "lengths", "neighbors", "subsampling", "upsampling"}
    P = [p.cpu().numpy() for p in out["points"]]
    L = [l.cpu().numpy() for l in out["lengths"]]
    for i in range(5):
        assert np.array_equal(L[i], g[f"lengths{i}"])
        assert np.array_equal(P[i].view(np.uint32), g[f"points{i}"].view(np.uint32)), f"level {i} points"
        nb = out["neighbors"][i]
        assert nb.is_contiguous() and nb.is_cuda
        want = g[f"neighbors{i}"][:, :limits[i]]
        assert_neighbors_equal_up_to_ties(nb.cpu().numpy(), want, P[i], P[i], L[i], L[i])
        if i < 4:
            sub = out["subsampling"][i].cpu().numpy()
            assert_neighbors_equal_up_to_ties(sub, g[f"subsampling{i}"][:, :limits[i]], P[i + 1], P[i], L[i + 1], L[i])
            up = out["upsampling"][i].cpu().numpy()
            assert_neighbors_equal_up_to_ties(up, g[f"upsampling{i}"][:, :limits[i + 1]], P[i], P[i + 1], L[i], L[i + 1])


def test_registration_collate_builds_on_device():
    from gaussreg_amd.data import registration_collate_fn_stack_mode
    rng = np.random.default_rng(0)
    d = {"ref_points": rng.random((4000, 3)).astype(np.float32) * 3, "src_points": rng.random((3500, 3)).astype(np.float32) * 3,
         "ref_feats": np.ones((4000, 4), np.float32), "src_feats": np.ones((3500, 4), np.float32),
         "transform": np.eye(4, dtype=np.float32)}
    out = registration_collate_fn_stack_mode([d], 5, 0.05, 0.125, [30, 30, 30, 30, 30], device="cuda")
    assert out["batch_size"] == 1 and out["features"].shape == (7500, 4) and out["transform"].shape == (4, 4)
    assert len(out["points"]) == 5 and out["points"][0].is_cuda
    assert out["lengths"][0].tolist() == [4000, 3500]
    for i in range(5):
        n = out["points"][i].shape[0]
        assert out["neighbors"][i].shape[0] == n and int(out["neighbors"][i].max()) <= n
        assert torch.equal(out["neighbors"][i][:, 0], torch.arange(n, device="cuda"))  # self first


def test_pipelined_pyramid_is_identical():
    """The subsampling chain on a second host thread + stream (gaussreg_amd/data.py) must not change a single value."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from gen_golden_ext import room_pair
    from gaussreg_amd.data import precompute_data_stack_mode
    clouds = []
    for b in range(4):
        r_, s_ = room_pair(12000, 50 + b)
        clouds += [r_, s_]
    pts = torch.from_numpy(np.concatenate(clouds)).cuda()
    lens = torch.tensor([12000] * 8)
    limits = [89, 30, 43, 49, 49]
    a = precompute_data_stack_mode(pts, lens, 5, 0.025, 0.0625, limits, pipeline=False)
    for _ in range(3):
        b = precompute_data_stack_mode(pts, lens, 5, 0.025, 0.0625, limits, pipeline=True)
        for key in ("points", "lengths", "neighbors", "subsampling", "upsampling"):
            assert len(a[key]) == len(b[key])
            for x, y in zip(a[key], b[key]):
                assert x.shape == y.shape and torch.equal(x.cpu(), y.cpu()), key
