"""From a Gaussian-Splatting scene to the network's input, and back: the data preparation of GaussReg's demo on the GPU.

Mirrors experiments/geotransformer.gaussian_splatting.indoor/demo.py:
  extract_points         demo.py:30-75 `_read_ply_by_opacity`: sigmoid(opacity) > 0.7, inside the 5th..95th percentile box,
                         farthest point sampling down to `point_limit`, features = [opacity, RGB in 0..255] where RGB is the
                         degree-3 SH colour seen from one fixed far-away point (demo.py:62-72, graphics_utils.py:34-89)
  normalize_pair         demo.py:82-127 `load_data`: centre each cloud, rescale clouds whose bounding-box volume is outside
                         [10, 50] (to 30 resp. 50)
  denormalize_transform  demo.py:171-174: the estimate in the normalised frames -> the similarity between the original scenes
Everything runs on the device the records live on; FPS is gaussreg_amd's (exact FPS from index 0; the reference calls
fpsample.bucket_fps_kdline_sampling, whose start point is random: parity unpinned).
"""
import numpy as np
import torch

from .registration import farthest_point_sampling

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def eval_sh3(sh, dirs):
    """Degree-3 real SH colour: sh (N, 3, 16) coefficients (DC first), dirs (N, 3) unit vectors -> (N, 3).
    The polynomial basis of graphics_utils.py:51-81 (the 3DGS convention)."""
    x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    basis = [torch.full_like(x, _C0), -_C1 * y, _C1 * z, -_C1 * x,
             _C2[0] * xy, _C2[1] * yz, _C2[2] * (2.0 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy),
             _C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy),
             _C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy),
             _C3[6] * x * (xx - 3 * yy)]
    return (sh * torch.cat(basis, 1)[:, None, :]).sum(-1)


def _percentile(v, q):
    """np.percentile(v, q) (linear interpolation) for a 1-D tensor of any length (torch.quantile stops at 16 M)."""
    s, _ = torch.sort(v.double())
    pos = (s.numel() - 1) * (q / 100.0)
    lo = int(np.floor(pos))
    hi = min(lo + 1, s.numel() - 1)
    return (s[lo] + (s[hi] - s[lo]) * (pos - lo)).to(v.dtype)


@torch.no_grad()
def extract_points(records, point_limit=None, opacity_threshold=0.7):
    """records: (N, 62) GS vertex records (gs_io.read_gs_ply order) on the GPU -> points (M, 3), features (M, 4)
    = [opacity, R, G, B] with the colours in 0..255, and the indices (M,) of the kept Gaussians."""
    r = torch.as_tensor(records, dtype=torch.float32)
    if not r.is_cuda:
        raise RuntimeError("extract_points: the records must live on the GPU (gaussreg_amd has no CPU path)")
    opacity = torch.sigmoid(r[:, 54])
    xyz = r[:, 0:3]
    keep = opacity > opacity_threshold
    for a in range(3):
        c = xyz[:, a]
        keep &= (c < _percentile(c, 95)) & (c > _percentile(c, 5))
    index = torch.nonzero(keep).squeeze(1)
    if point_limit is not None and index.numel() > point_limit:
        picked = farthest_point_sampling(xyz[index].contiguous(), [int(index.numel())], [int(point_limit)])[0]
        index = index[picked]
    points = xyz[index]
    n = points.shape[0]
    sh = torch.cat([r[index, 6:9].reshape(n, 3, 1), r[index, 9:54].reshape(n, 3, 15)], 2)   # (M, 3, 16), channel-major
    # one fixed view point far along +y of the cloud (demo.py:62-66)
    center = points.mean(0)
    max_length = torch.linalg.norm(points.max(0).values - points.min(0).values)
    view = center + torch.stack([torch.zeros_like(max_length), 2 * max_length, torch.zeros_like(max_length)])
    d = points - view
    d = d / (torch.linalg.norm(d, dim=1, keepdim=True) + 1e-6)
    colors = torch.clamp(eval_sh3(sh, d) + 0.5, 0.0, 1.0) * 255
    feats = torch.cat([opacity[index, None], colors], 1)
    return points.contiguous(), feats.contiguous(), index


def _normalise_one(points):
    ext = points.max(0).values - points.min(0).values
    volume = float(ext[0] * ext[1] * ext[2])
    center = (points.max(0).values + points.min(0).values) / 2
    scale = 1.0
    if volume > 50:
        scale = (50 / volume) ** (1 / 3)
    elif volume < 10:
        scale = (30 / volume) ** (1 / 3)
    return (points - center) * scale, center, scale


@torch.no_grad()
def normalize_pair(ref_points, ref_feats, src_points, src_feats):
    """-> the per-pair dict registration_collate_fn_stack_mode takes (utils/data.py:139-189), tensors on the GPU."""
    rp, rc, rs = _normalise_one(ref_points)
    sp, sc, ss = _normalise_one(src_points)
    return {"ref_points": rp.float().contiguous(), "src_points": sp.float().contiguous(), "ref_feats": ref_feats.float(),
            "src_feats": src_feats.float(), "ref_adjust_scale": rs, "src_adjust_scale": ss, "ref_center": rc, "src_center": sc,
            "transform": torch.eye(4, device=ref_points.device)}


def denormalize_transform(estimated_transform, ref_center, src_center, ref_adjust_scale, src_adjust_scale):
    """The network's estimate maps normalised src to normalised ref; this is the map between the original scenes."""
    T = np.asarray(torch.as_tensor(estimated_transform).detach().cpu(), np.float64)
    rc = np.asarray(torch.as_tensor(ref_center).detach().cpu(), np.float64)
    sc = np.asarray(torch.as_tensor(src_center).detach().cpu(), np.float64)
    out = np.zeros((4, 4))
    out[:3, :3] = T[:3, :3] / ref_adjust_scale * src_adjust_scale
    out[:3, 3] = T[:3, 3] / ref_adjust_scale + rc - out[:3, :3] @ sc
    out[3, 3] = 1.0
    return out
