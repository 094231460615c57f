"""gaussreg_amd.gs_points (the data preparation of the reference's demo.py) against a NumPy restatement of
demo.py:30-75 / :82-127 / :171-174 on synthetic GS records, and the end-to-end example script in --synthetic mode."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _records(n, seed):
    rng = np.random.default_rng(seed)
    rec = np.zeros((n, 62), np.float32)
    rec[:, 0:3] = rng.normal(0, 1.5, (n, 3))
    rec[:, 6:54] = rng.normal(0, 0.3, (n, 48))
    rec[:, 54] = rng.normal(1.0, 1.5, n)
    rec[:, 55:58] = rng.normal(-4, 0.3, (n, 3))
    rec[:, 58:62] = rng.normal(0, 1, (n, 4))
    return rec


def _np_extract(rec):
    """demo.py:30-75 without the FPS step, in float64 NumPy."""
    from gaussreg_amd.gs_points import _C0, _C1, _C2, _C3
    r = rec.astype(np.float64)
    opacity = 1 / (1 + np.exp(-r[:, 54]))
    x, y, z = r[:, 0], r[:, 1], r[:, 2]
    m = opacity > 0.7
    for c in (x, y, z):
        m &= (c < np.percentile(c, 95)) & (c > np.percentile(c, 5))
    idx = np.where(m)[0]
    pts = r[idx, 0:3]
    sh = np.concatenate([r[idx, 6:9].reshape(-1, 3, 1), r[idx, 9:54].reshape(-1, 3, 15)], 2)
    view = pts.mean(0) + np.array([0, 2 * np.linalg.norm(pts.max(0) - pts.min(0)), 0])
    d = pts - view
    d = d / (np.linalg.norm(d, axis=1, keepdims=True) + 1e-6)
    X, Y, Z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    xx, yy, zz, xy, yz, xz = X * X, Y * Y, Z * Z, X * Y, Y * Z, X * Z
    B = np.concatenate([np.full_like(X, _C0), -_C1 * Y, _C1 * Z, -_C1 * X, _C2[0] * xy, _C2[1] * yz, _C2[2] * (2 * zz - xx - yy),
                        _C2[3] * xz, _C2[4] * (xx - yy), _C3[0] * Y * (3 * xx - yy), _C3[1] * xy * Z, _C3[2] * Y * (4 * zz - xx - yy),
                        _C3[3] * Z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * X * (4 * zz - xx - yy), _C3[5] * Z * (xx - yy),
                        _C3[6] * X * (xx - 3 * yy)], 1)
    col = np.clip((sh * B[:, None, :]).sum(-1) + 0.5, 0, 1) * 255
    return idx, pts, np.concatenate([opacity[idx, None], col], 1)


def test_extract_points_matches_restatement():
    from gaussreg_amd import gs_points
    rec = _records(50000, 3)
    idx, pts, feats = _np_extract(rec)
    p, f, i = gs_points.extract_points(torch.from_numpy(rec).cuda())
    assert np.array_equal(i.cpu().numpy(), idx)
    assert np.array_equal(p.cpu().numpy(), rec[idx, 0:3])
    assert np.abs(f.cpu().numpy() - feats).max() <= 2e-3        # colours in 0..255 computed in fp32
    p2, f2, i2 = gs_points.extract_points(torch.from_numpy(rec).cuda(), point_limit=5000)
    assert p2.shape == (5000, 3) and f2.shape == (5000, 4) and len(set(i2.tolist())) == 5000
    assert set(i2.tolist()) <= set(idx.tolist()) and int(i2[0]) == int(idx[0])   # exact FPS from the first kept Gaussian


def test_normalise_and_denormalise_are_consistent():
    from gaussreg_amd import gs_points
    g = torch.Generator(device="cuda").manual_seed(0)
    ref = torch.rand(4000, 3, device="cuda", generator=g) * torch.tensor([9.0, 4.0, 3.0], device="cuda") + 5.0   # volume 108 -> 50
    src = torch.rand(3000, 3, device="cuda", generator=g) * torch.tensor([1.5, 1.0, 2.0], device="cuda") - 2.0   # volume 3 -> 30
    d = gs_points.normalize_pair(ref, torch.zeros(4000, 4, device="cuda"), src, torch.zeros(3000, 4, device="cuda"))
    vol = lambda p: float(torch.prod(p.max(0).values - p.min(0).values))
    assert abs(vol(d["ref_points"]) - 50) < 0.5 and abs(vol(d["src_points"]) - 30) < 0.5
    # a transform that maps normalised src onto normalised ref exactly ...
    a = 0.3
    R = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
    T = torch.eye(4)
    T[:3, :3], T[:3, 3] = 1.2 * R, torch.tensor([0.1, 0.2, -0.3])
    # ... becomes, between the original clouds:
    Tw = gs_points.denormalize_transform(T, d["ref_center"], d["src_center"], d["ref_adjust_scale"], d["src_adjust_scale"])
    x_src = src[:50].cpu().numpy().astype(np.float64)
    xn = (x_src - d["src_center"].cpu().numpy()) * d["src_adjust_scale"]            # normalised src
    yn = xn @ T[:3, :3].numpy().astype(np.float64).T + T[:3, 3].numpy()            # in the normalised ref frame
    y = yn / d["ref_adjust_scale"] + d["ref_center"].cpu().numpy()                   # original ref frame
    assert np.abs(x_src @ Tw[:3, :3].T + Tw[:3, 3] - y).max() < 1e-5


def test_example_script_runs_end_to_end(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "register_scenes.py"), "--synthetic", "--num_sample", "8000",
                          "--output_path", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    T = np.load(tmp_path / "estimated_transform.npz")["estimated_transform"]
    assert T.shape == (4, 4) and np.isfinite(T).all() and np.allclose(T[3], [0, 0, 0, 1])
    from gaussreg_amd import gs_io
    fused = gs_io.read_gs_ply(str(tmp_path / "fused.ply"))
    assert fused.shape[1] == 62 and fused.shape[0] > 0
