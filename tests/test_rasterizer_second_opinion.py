"""The rasterizer has no reference implementation in the GaussReg tree, so its bit-exact fp32 oracle
(oracle/rasterizer_oracle.c) is checked here against an INDEPENDENT float64 NumPy renderer written from the published
algorithm (oracle/rasterizer_np64.py: libm exp, no fmaf, no shared code).  Agreement bounds how far the chosen fp32
operation sequence sits from the real-number algorithm:
  * radii (integer) identical;
  * mean |error| below 2e-6, at least 99.5 % of the pixels within 1e-6 abs + 1e-5 rel (99.95 % at 100 k / 640 x 480);
  * the few pixels outside are single threshold decisions (alpha < 1/255, T < 1e-4) taken differently in fp32 and fp64:
    bounded by the weight one just-visible Gaussian can carry (alpha ~ 1/255 times colour <= ~1.5).
CPU test: the C oracle at 20 k Gaussians / 320 x 240.  GPU test: the HIP image at 100 k Gaussians / 640 x 480."""
import numpy as np
import pytest

from gaussreg_amd import synthetic
from oracle import capi, rasterizer_np64


def _scene(P, W, H):
    g = synthetic.gaussians_c2(P, 3)
    cam = synthetic.camera_ring(2, W, H, 5)[1]
    kw = dict(shs=g["shs"], scales=g["scales"], rotations=g["rotations"], viewmatrix=cam["viewmatrix"],
              projmatrix=cam["projmatrix"], campos=cam["campos"], bg=np.array([0.1, 0.2, 0.3], np.float32), W=W, H=H,
              tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3)
    return g, cam, kw


def _check(img32, rad32, img64, rad64, npix):
    assert np.array_equal(np.asarray(rad32, np.int64), rad64), "radii differ from the float64 renderer"
    d = np.abs(img32.astype(np.float64) - img64)
    assert d.mean() < 2e-6, d.mean()
    outside = (d > 1e-6 + 1e-5 * np.abs(img64)).any(axis=0).sum()
    assert outside <= 5e-3 * npix, f"{outside} of {npix} pixels outside 1e-6 + 1e-5 rel"
    assert d.max() < 8e-3, d.max()  # one just-visible Gaussian (alpha ~ 1/255) at full transmittance


def test_fp32_oracle_vs_independent_float64_renderer():
    g, cam, kw = _scene(20000, 320, 240)
    img32, rad32, _ = capi.rasterize_forward(g["means3D"], g["opacities"], **kw)
    img64, rad64, st = rasterizer_np64.render(g["means3D"], g["opacities"], **kw)
    assert st["visible"] > 5000 and st["blended_pairs"] > 100000
    _check(img32, rad32, img64, rad64, 320 * 240)


def test_float64_renderer_precomputed_inputs():
    """colours / 3-D covariances passed in instead of SH / (scale, rotation): same image from both renderers."""
    g, cam, kw = _scene(4000, 160, 128)
    rng = np.random.default_rng(0)
    cols = rng.random((4000, 3)).astype(np.float32)
    kw2 = {k: v for k, v in kw.items() if k not in ("shs", "sh_degree")}
    img32, rad32, _ = capi.rasterize_forward(g["means3D"], g["opacities"], colors_precomp=cols, **kw2)
    img64, rad64, _ = rasterizer_np64.render(g["means3D"], g["opacities"], colors_precomp=cols, **kw2)
    _check(img32, rad32, img64, rad64, 160 * 128)


@pytest.mark.gpu
def test_hip_image_vs_independent_float64_renderer_100k():
    import torch
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    P, W, H = 100000, 640, 480
    g, cam, kw = _scene(P, W, H)
    rs = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], torch.tensor([0.1, 0.2, 0.3]), 1.0,
                                       torch.from_numpy(cam["viewmatrix"]), torch.from_numpy(cam["projmatrix"]), 3,
                                       torch.from_numpy(cam["campos"]), False, False)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    img, radii = GaussianRasterizer(rs)(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"],
                                        rotations=t["rotations"])
    img64, rad64, st = rasterizer_np64.render(g["means3D"], g["opacities"], **kw)
    assert st["visible"] > 40000
    _check(img.cpu().numpy(), radii.cpu().numpy(), img64, rad64, W * H)
