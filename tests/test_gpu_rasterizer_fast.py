"""The opt-in fast-exponential blend (GR_RASTER_FAST_EXP / fast_exp=True: v_exp_f32 instead of the oracle's deterministic
polynomial) against (1) the independent float64 renderer oracle/rasterizer_np64.py -- north_star's bar for rendered RGB is
1e-5 relative, written here as |err| <= 1e-6 + 1e-5 |ref| on >= 99.9 % of the pixels, the rest being single threshold
decisions (alpha < 1/255, T < 1e-4) taken differently, exactly as for the bit-exact mode -- and (2) the bit-exact HIP image.
Also: the one-call boundary path (gr_raster_forward) and the camera cache of GaussianRasterizer."""
import numpy as np
import pytest
import torch

from gaussreg_amd import synthetic

pytestmark = pytest.mark.gpu


def _scene(P, W, H):
    g = synthetic.gaussians_c2(P, 3)
    cam = synthetic.camera_ring(2, W, H, 5)[1]
    return g, cam


def _settings(cam, W, H, bg=(0.1, 0.2, 0.3)):
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], torch.tensor(bg), 1.0,
                                         torch.from_numpy(cam["viewmatrix"].copy()), torch.from_numpy(cam["projmatrix"].copy()), 3,
                                         torch.from_numpy(cam["campos"].copy()), False, False)


def test_fast_exp_vs_float64_renderer_and_vs_exact_100k():
    from gaussreg_amd.rasterizer import GaussianRasterizer
    from oracle import rasterizer_np64
    P, W, H = 100000, 640, 480
    g, cam = _scene(P, W, H)
    rs = _settings(cam, W, H)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    args = (t["means3D"], None, t["opacities"])
    kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
    exact, radii_e = GaussianRasterizer(rs, fast_exp=False)(*args, **kw)
    fast, radii_f = GaussianRasterizer(rs, fast_exp=True)(*args, **kw)
    assert torch.equal(radii_e, radii_f)
    img64, rad64, st = rasterizer_np64.render(g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"],
                                              rotations=g["rotations"], viewmatrix=cam["viewmatrix"],
                                              projmatrix=cam["projmatrix"], campos=cam["campos"],
                                              bg=np.array([0.1, 0.2, 0.3], np.float32), W=W, H=H, tanfovx=cam["tanfovx"],
                                              tanfovy=cam["tanfovy"], sh_degree=3)
    assert np.array_equal(radii_f.cpu().numpy().astype(np.int64), rad64)
    f = fast.cpu().numpy().astype(np.float64)
    e = exact.cpu().numpy().astype(np.float64)
    npix = W * H
    # (1) against the real-number algorithm
    d = np.abs(f - img64)
    assert d.mean() < 2e-6, d.mean()
    outside = (d > 1e-6 + 1e-5 * np.abs(img64)).any(axis=0).sum()
    assert outside <= 1e-3 * npix, f"{outside} of {npix} pixels outside 1e-6 + 1e-5 rel of the float64 image"
    assert d.max() < 8e-3, d.max()  # one just-visible Gaussian (alpha ~ 1/255) at full transmittance
    # (2) against the bit-exact mode: the same bar; the fast image is not expected to be bit-equal
    d2 = np.abs(f - e)
    outside2 = (d2 > 1e-6 + 1e-5 * np.abs(e)).any(axis=0).sum()
    assert outside2 <= 1e-3 * npix, f"{outside2} of {npix} pixels outside 1e-6 + 1e-5 rel of the exact image"
    assert d2.mean() < 1e-6 and d2.max() < 8e-3, (d2.mean(), d2.max())


def test_fast_exp_batched_views_match_single_view_calls():
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer, rasterize_views
    P, W, H = 30000, 320, 240
    g = synthetic.gaussians_c2(P, 1)
    cams = synthetic.camera_ring(3, W, H, 2)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    sets = [GaussianRasterizationSettings(H, W, c["tanfovx"], c["tanfovy"], torch.zeros(3), 1.0,
                                          torch.from_numpy(c["viewmatrix"]), torch.from_numpy(c["projmatrix"]), 3,
                                          torch.from_numpy(c["campos"]), False, False) for c in cams]
    img, radii, nr = rasterize_views(sets, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                     rotations=t["rotations"], fast_exp=True)
    for v, s in enumerate(sets):
        one, r1 = GaussianRasterizer(s, fast_exp=True)(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"],
                                                       rotations=t["rotations"])
        assert torch.equal(one, img[v]) and torch.equal(r1, radii[v])  # the same kernel either way: bit-equal


def test_one_call_path_grows_its_binning_buffer_and_stays_bit_exact():
    """GaussianRasterizer.forward enters the library once per frame with a binning buffer sized from the previous frame of
    the same shape; a frame that needs more (here: the same Gaussians three times larger) is rendered correctly too."""
    from gaussreg_amd.rasterizer import GaussianRasterizer
    from oracle import capi
    P, W, H = 20000, 256, 192
    g, cam = _scene(P, W, H)
    rs = _settings(cam, W, H, bg=(0.0, 0.0, 0.0))
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    rast = GaussianRasterizer(rs)
    for scale_up in (0.0, 0.0, 1.1, 0.0):   # log-scales: + 1.1 = three times larger footprints -> many more instances
        sc = g["scales"] + np.float32(scale_up)
        img, radii = rast(t["means3D"], None, t["opacities"], shs=t["shs"], scales=torch.from_numpy(sc).cuda(),
                          rotations=t["rotations"])
        wimg, wr, _ = capi.rasterize_forward(g["means3D"], g["opacities"], shs=g["shs"], scales=sc, rotations=g["rotations"],
                                             viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
                                             bg=np.zeros(3, np.float32), W=W, H=H, tanfovx=cam["tanfovx"],
                                             tanfovy=cam["tanfovy"], sh_degree=3)
        assert np.array_equal(radii.cpu().numpy(), wr)
        assert np.array_equal(img.cpu().numpy().view(np.uint32), wimg.view(np.uint32))


def test_camera_updated_in_place_is_rendered():
    """The marshalled camera is cached per settings object; an in-place update of its tensors (pose optimisation) must
    invalidate the cache (the reference reads the tensors on every forward)."""
    from gaussreg_amd.rasterizer import GaussianRasterizer
    P, W, H = 5000, 128, 96
    g = synthetic.gaussians_c2(P, 4)
    c0, c1 = synthetic.camera_ring(2, W, H, 9)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    rs = _settings(c0, W, H)
    rast = GaussianRasterizer(rs)
    call = lambda r: r(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])  # noqa: E731
    a0, _ = call(rast)
    rs.viewmatrix.copy_(torch.from_numpy(c1["viewmatrix"]))
    rs.projmatrix.copy_(torch.from_numpy(c1["projmatrix"]))
    rs.campos.copy_(torch.from_numpy(c1["campos"]))
    a1, _ = call(rast)
    b1, _ = call(GaussianRasterizer(_settings(c1, W, H)))
    assert torch.equal(a1, b1) and not torch.equal(a0, a1)


def test_sorted_output_self_check_passes_on_every_frame():
    """GR_RASTER_VERIFY=1: the device-side check of the depth order and of every per-tile list runs on every frame (by default
    only on the first three of a process) -- with lane-ordered LDS atomics and with explicit ballot ranking -- and the
    image stays bit-equal to the oracle's."""
    import os
    import subprocess
    import sys
    code = r'''
import numpy as np, torch
from gaussreg_amd import synthetic, _lib
from gaussreg_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
from oracle import capi
P, W, H = 60000, 320, 240
g = synthetic.gaussians_c2(P, 2)
cam = synthetic.camera(W, H)
rs = GaussianRasterizationSettings(H, W, cam["tanfovx"], cam["tanfovy"], torch.zeros(3), 1.0, torch.from_numpy(cam["viewmatrix"]),
                                   torch.from_numpy(cam["projmatrix"]), 3, torch.from_numpy(cam["campos"]), False, False)
t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
r = GaussianRasterizer(rs)
for _ in range(5):
    img, radii = r(t["means3D"], None, t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
wimg, wr, _ = capi.rasterize_forward(g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"], rotations=g["rotations"],
                                     viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"],
                                     bg=np.zeros(3, np.float32), W=W, H=H, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3)
assert np.array_equal(img.cpu().numpy().view(np.uint32), wimg.view(np.uint32))
print("STATE", _lib.lib().gr_raster_lds_atomics_lane_ordered())
'''
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    for extra in ({}, {"GR_RASTER_BALLOT_RANKING": "1"}):
        env = dict(os.environ, GR_RASTER_VERIFY="1", PYTHONPATH=root, **extra)
        res = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=root)
        assert res.returncode == 0, res.stderr[-2000:]
        assert "STATE" in res.stdout
