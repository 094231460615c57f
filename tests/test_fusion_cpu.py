"""CPU: GS .ply wire format round trip, and the NumPy restatement of gs_fusion.gaussian_fuse vs the golden
vector produced by the reference's own gaussian_fuse."""
import numpy as np

from helpers import load_golden


def test_ply_round_trip(tmp_path):
    from gaussreg_amd.gs_io import PROPERTIES, read_gs_ply, write_gs_ply
    assert len(PROPERTIES) == 62 and PROPERTIES[6] == "f_dc_0" and PROPERTIES[54] == "opacity" and PROPERTIES[58] == "rot_0"
    rec = np.random.default_rng(0).normal(size=(123, 62)).astype(np.float32)
    p = tmp_path / "x.ply"
    write_gs_ply(str(p), rec)
    head = open(p, "rb").read(200).decode("ascii", "replace")
    assert head.startswith("ply\nformat binary_little_endian 1.0\nelement vertex 123\nproperty float x\n")
    assert (p.stat().st_size - open(p, "rb").read().index(b"end_header\n") - len(b"end_header\n")) == 123 * 248
    assert np.array_equal(read_gs_ply(str(p)), rec)


def test_oracle_fusion_matches_reference():
    from oracle import fusion_np
    g = load_golden("gs_fusion.npz")
    got = fusion_np.gaussian_fuse(g["rec1"], g["rec2"], g["transform"])
    assert got.shape == g["fused"].shape
    np.testing.assert_allclose(got, g["fused"], rtol=2e-5, atol=2e-6)
    # cloud-1 rows are copied verbatim
    n1 = int((np.abs(got[:, 0:3] - g["fused"][:, 0:3]).max(1) == 0).sum())
    assert n1 > 100


def test_split_records_conventions():
    import torch
    from gaussreg_amd.gs_io import split_records
    rec = np.zeros((2, 62), np.float32)
    rec[:, 6:9] = [[1, 2, 3], [4, 5, 6]]
    rec[:, 9:54] = np.arange(45)[None]
    rec[:, 54] = 0.0
    rec[:, 55:58] = np.log(0.5)
    rec[:, 58:62] = [2, 0, 0, 0]
    d = split_records(rec)
    assert d["shs"].shape == (2, 16, 3)
    assert d["shs"][0, 0].tolist() == [1, 2, 3]
    assert d["shs"][0, 1].tolist() == [0, 15, 30]      # coefficient 1 of channels r,g,b = f_rest_{c*15}
    assert torch.allclose(d["opacities"], torch.full((2, 1), 0.5)) and torch.allclose(d["scales"], torch.full((2, 3), 0.5))
    assert d["rotations"][0].tolist() == [1, 0, 0, 0]
