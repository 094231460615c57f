"""Synthetic workloads named by BASELINE.json / SURVEY.md section 8(d) (inputs only; no datasets offline)."""
import math

import numpy as np


def gaussians_c2(P, seed=0, sh_degree=3):
    """C2 recipe: P Gaussians uniform in a 4 x 3 x 2.5 m box centred 3 m in front of a camera at the
    origin looking down +z; log-scales N(log 0.01, 0.3^2); rotations = normalised N(0,1)^4;
    opacity = sigmoid(N(0, 1.5^2)); SH (P,16,3) ~ N(0, 0.2^2) with DC ~ N(0.5, 0.5^2)."""
    rng = np.random.default_rng(seed)
    means = (rng.random((P, 3)) - 0.5) * np.array([4.0, 3.0, 2.5]) + np.array([0.0, 0.0, 3.0])
    scales = np.exp(rng.normal(math.log(0.01), 0.3, (P, 3)))
    rot = rng.normal(0.0, 1.0, (P, 4))
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    opac = 1.0 / (1.0 + np.exp(-rng.normal(0.0, 1.5, (P, 1))))
    K = (sh_degree + 1) ** 2
    shs = rng.normal(0.0, 0.2, (P, K, 3))
    shs[:, 0, :] = rng.normal(0.5, 0.5, (P, 3))
    f = np.float32
    return dict(means3D=means.astype(f), scales=scales.astype(f), rotations=rot.astype(f),
                opacities=opac.astype(f), shs=shs.astype(f))


def camera(W, H, fovx_deg=60.0, R_c2w=None, C=None, znear=0.01, zfar=100.0):
    """Pinhole camera in the 3DGS convention: returns the dict of fields of
    GaussianRasterizationSettings that describe the camera (numpy fp32, matrices TRANSPOSED)."""
    R = np.eye(3) if R_c2w is None else np.asarray(R_c2w, np.float64)
    C = np.zeros(3) if C is None else np.asarray(C, np.float64)
    tanx = math.tan(math.radians(fovx_deg) / 2)
    tany = tanx * H / W
    Rt = np.eye(4)
    Rt[:3, :3] = R.T
    Rt[:3, 3] = -R.T @ C
    Pm = np.zeros((4, 4))
    Pm[0, 0] = 1 / tanx
    Pm[1, 1] = 1 / tany
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -zfar * znear / (zfar - znear)
    Pm[3, 2] = 1.0
    return dict(image_height=H, image_width=W, tanfovx=tanx, tanfovy=tany,
                viewmatrix=Rt.T.astype(np.float32), projmatrix=(Pm @ Rt).T.astype(np.float32),
                campos=C.astype(np.float32))


def rot_yx(yaw, pitch):
    cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    return Ry @ Rx


def camera_ring(V, W, H, seed=0, fovx_deg=60.0):
    """V cameras jittered around the C2 viewpoint (small yaw/pitch/translation)."""
    rng = np.random.default_rng(1000 + seed)
    cams = []
    for v in range(V):
        if v == 0:
            cams.append(camera(W, H, fovx_deg))
            continue
        yaw, pitch = rng.uniform(-0.25, 0.25), rng.uniform(-0.15, 0.15)
        C = rng.uniform(-0.4, 0.4, 3) * np.array([1.0, 0.6, 0.5])
        cams.append(camera(W, H, fovx_deg, rot_yx(yaw, pitch), C))
    return cams


def cloud_200k(batch=1, seed=0):
    """north_star 200 k-point scaling config: torch.rand(200000,3) * 10**(1/3) per cloud."""
    import torch
    g = torch.Generator().manual_seed(seed)
    pts = (torch.rand(200000 * batch, 3, generator=g) * 10 ** (1 / 3)).float()
    return pts, torch.tensor([200000] * batch, dtype=torch.int64)
