"""GS `.ply` wire format + the fusion step of gs_fusion.py, on the GPU.

Wire format (gs_fusion.py:172-193, the 3DGS convention): binary little-endian PLY, one `vertex` element
with 62 float32 properties in the order x y z nx ny nz f_dc_0..2 f_rest_0..44 opacity scale_0..2 rot_0..3;
f_rest is channel-major (P,3,15) (gs_fusion.py:203-215); opacity is a logit, scales are logs, rot is a
quaternion with the real part first.  `plyfile` is not a dependency here: the reader /
writer below handle exactly this layout (any property order in the header is accepted on read).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

PROPERTIES = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)] + [f"f_rest_{i}" for i in range(45)] +
              ["opacity"] + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])  # gs_fusion.py:172-184


def read_gs_ply(path):
    """-> (N, 62) float32 array in canonical property order."""
    with open(path, "rb") as fh:
        header = []
        while True:
            line = fh.readline()
            if not line:
                raise ValueError("truncated PLY header")
            header.append(line.decode("ascii", "replace").strip())
            if header[-1] == "end_header":
                break
        if "format binary_little_endian 1.0" not in header:
            raise ValueError("only binary_little_endian PLY is supported")
        n, names, in_vertex = 0, [], False
        for h in header:
            t = h.split()
            if t[:2] == ["element", "vertex"]:
                n, in_vertex = int(t[2]), True
            elif t[:1] == ["element"]:
                in_vertex = False
            elif t[:1] == ["property"] and in_vertex:
                if t[1] not in ("float", "float32"):
                    raise ValueError(f"vertex property {t[2]} is {t[1]}, expected float32")
                names.append(t[2])
        raw = np.fromfile(fh, dtype="<f4", count=n * len(names)).reshape(n, len(names))
    idx = [names.index(p) for p in PROPERTIES]
    return np.ascontiguousarray(raw[:, idx])


def write_gs_ply(path, records):
    rec = np.ascontiguousarray(np.asarray(records, dtype="<f4"))
    assert rec.ndim == 2 and rec.shape[1] == len(PROPERTIES)
    head = ["ply", "format binary_little_endian 1.0", f"element vertex {rec.shape[0]}"]
    head += [f"property float {p}" for p in PROPERTIES] + ["end_header"]
    with open(path, "wb") as fh:
        fh.write(("\n".join(head) + "\n").encode("ascii"))
        rec.tofile(fh)


def split_records(rec):
    """(N,62) -> dict with the tensors the rasterizer takes (activations applied as demo.py:33-34,
    gs_fusion.py:83,242 imply: sigmoid(opacity), exp(scale), normalised quaternion; shs (N,16,3))."""
    r = torch.as_tensor(rec)
    n = r.shape[0]
    dc = r[:, 6:9].reshape(n, 1, 3)
    rest = r[:, 9:54].reshape(n, 3, 15).transpose(1, 2)
    rot = r[:, 58:62]
    return dict(means3D=r[:, 0:3].contiguous(), shs=torch.cat([dc, rest], 1).contiguous(),
                opacities=torch.sigmoid(r[:, 54:55]).contiguous(), scales=torch.exp(r[:, 55:58]).contiguous(),
                rotations=(rot / rot.norm(dim=1, keepdim=True)).contiguous())


# ---- SH band transforms: new_coeffs = old_coeffs @ T_l with Y_l(R d) = Y_l(d) @ T_l ... gs_fusion.py:53-68
_C1 = 0.4886025119029199
_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]


def _sh_basis(d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
    b1 = np.stack([-_C1 * y, _C1 * z, -_C1 * x], 1)
    b2 = np.stack([_C2[0] * xy, _C2[1] * yz, _C2[2] * (2 * zz - xx - yy), _C2[3] * xz, _C2[4] * (xx - yy)], 1)
    b3 = np.stack([_C3[0] * y * (3 * xx - yy), _C3[1] * xy * z, _C3[2] * y * (4 * zz - xx - yy),
                   _C3[3] * z * (2 * zz - 3 * xx - 3 * yy), _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy),
                   _C3[6] * x * (xx - 3 * yy)], 1)
    return b1, b2, b3


def sh_band_transforms(rotation):
    """The matrices gs_fusion.py:53-68 fits per call (pinv of the basis at random directions): they depend
    only on the rotation, so a fixed well-spread direction set gives the same matrices (to fp64 rounding)."""
    rng = np.random.default_rng(12345)
    d = rng.normal(size=(64, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    a = _sh_basis(d)
    b = _sh_basis(d @ np.asarray(rotation, np.float64).T)
    return tuple(np.linalg.pinv(ai) @ bi for ai, bi in zip(a, b))


def gaussian_fuse_records(rec1, rec2, estimated_transform):
    """gs_fusion.py:231-262 on (N,62) record arrays / tensors -> fused (M,62) float32 CUDA tensor."""
    dev = _lib.require_gpu()
    L = _lib.lib()
    r1 = torch.as_tensor(rec1, dtype=torch.float32).to(dev).contiguous()
    r2 = torch.as_tensor(rec2, dtype=torch.float32).to(dev).contiguous()
    # gr_gs_fuse fetches records as 16-byte vectors: a view that starts at an odd row (248-byte records) is copied
    if r1.data_ptr() % 16:
        r1 = r1.clone()
    if r2.data_ptr() % 16:
        r2 = r2.clone()
    T = np.asarray(estimated_transform, np.float64)
    rotation = T[:3, :3]
    translation = np.ascontiguousarray(T[:3, 3])
    scale = float((rotation @ rotation.T)[0, 0] ** 0.5)          # :239
    rotation = np.ascontiguousarray(rotation / scale)            # :240
    t1, t2, t3 = [np.ascontiguousarray(t, np.float32) for t in sh_band_transforms(rotation)]
    n1, n2 = r1.shape[0], r2.shape[0]
    out = torch.empty((n1 + n2, 62), dtype=torch.float32, device=dev)
    n_out = ctypes.c_int64(0)
    dp = ctypes.POINTER(ctypes.c_double)
    fp = ctypes.POINTER(ctypes.c_float)
    with torch.cuda.device(dev):
        ws = _lib.workspace(dev, L.gr_gs_fuse_workspace_bytes(n1, n2))
        _lib.check(L.gr_gs_fuse(_lib.ptr(r1), n1, _lib.ptr(r2), n2, rotation.ctypes.data_as(dp),
                                translation.ctypes.data_as(dp), scale, t1.ctypes.data_as(fp), t2.ctypes.data_as(fp),
                                t3.ctypes.data_as(fp), _lib.ptr(out), ctypes.byref(n_out), _lib.ptr(ws), ws.numel(),
                                _lib.stream_ptr(dev)))
    return out[: n_out.value]


def gaussian_fuse(input_path_1, input_path_2, transform_path, output_path):
    """Same signature as gs_fusion.py:231."""
    est = np.load(transform_path)["estimated_transform"]
    fused = gaussian_fuse_records(read_gs_ply(input_path_1), read_gs_ply(input_path_2), est)
    os.makedirs(os.path.dirname(os.path.abspath(output_path)), exist_ok=True)
    write_gs_ply(output_path, fused.cpu().numpy())
