"""GPU: the four-launch depth sort of a few views per call (depth_sort.hip: top-digit pass relative to the view's key range +
in-LDS bucket sort) and the scatter that scans its own segments behind it (tile_scatter_kernel SELF_SEG).

White box: the order a deferred gr_raster_forward leaves in the geometry buffer (gr_raster_debug_geom_layout) must be the
stable argsort of the depth fields of the same frame -- bit-exact integer work, every visible Gaussian of every view.
Black box: the images of such frames equal the ones of the three-pass sort with its separate count / scan launches (pinned
through gr_raster_debug_bucket_cooldown), bit for bit -- one camera and many per call, including a scene that overflows a
bucket and falls back."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("P,W,H,V,frames", [(3000, 64, 48, 1, 8), (50000, 320, 240, 2, 6), (300000, 640, 480, 4, 5),
                                            (1000000, 640, 480, 1, 6), (7, 64, 48, 3, 5)])
def test_deferred_frames_leave_the_stable_depth_order(P, W, H, V, frames):
    import depth_order_check
    assert depth_order_check.run(P, W, H, V, frames, verbose=False) == 0


def _settings(cam, bg=(0.0, 0.0, 0.0)):
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
        bg=torch.tensor(bg, dtype=torch.float32, device="cuda"), scale_modifier=1.0,
        viewmatrix=torch.from_numpy(cam["viewmatrix"]).cuda(), projmatrix=torch.from_numpy(cam["projmatrix"]).cuda(),
        sh_degree=3, campos=torch.from_numpy(cam["campos"]).cuda(), prefiltered=False, debug=False)


def _render_each(sets, t, n_frames):
    from gaussreg_amd.rasterizer import GaussianRasterizer
    out = []
    for f in range(n_frames):
        r = GaussianRasterizer(sets[f % len(sets)])
        img, radii = r(means3D=t["means3D"], means2D=None, opacities=t["opacities"], shs=t["shs"], scales=t["scales"],
                       rotations=t["rotations"])
        out.append((img.cpu().numpy(), radii.cpu().numpy()))
    return out


def _plain_reference(sets, t):
    """More than four cameras in one call (the host reads the counts between the two halves, count and scan launches of their
    own) with the three-pass sort pinned; the thread's waiting period is cleared afterwards."""
    from gaussreg_amd import _lib
    from gaussreg_amd.rasterizer import rasterize_views
    L = _lib.lib()
    many = list(sets) + [sets[0]] * max(0, 5 - len(sets))
    L.gr_raster_debug_bucket_cooldown(1 << 20)
    try:
        color, radii, _ = rasterize_views(many, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                          rotations=t["rotations"])
    finally:
        L.gr_raster_debug_bucket_cooldown(0)
    return color.cpu().numpy(), radii.cpu().numpy()


def test_three_cameras_per_call_plain_and_deferred_equal_the_three_pass_sort():
    """The first call of a shape has no list-size hint and takes the plain path (the host reads the counts between the two
    halves; the bucket sort is used there too), the later ones are deferred."""
    from gaussreg_amd import _lib, synthetic
    from gaussreg_amd.rasterizer import rasterize_views
    L = _lib.lib()
    P, W, H, V = 300017, 320, 240, 3
    g = synthetic.gaussians_c2(P, seed=6, sh_degree=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    sets = [_settings(c) for c in synthetic.camera_ring(V, W, H, seed=2)]
    want_c, want_r = _plain_reference(sets, t)
    for _ in range(5):  # (the first frames of a process are the self-checked ones: both kinds pass through here)
        color, radii, _ = rasterize_views(sets, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                          rotations=t["rotations"])
        assert np.array_equal(color.cpu().numpy().view(np.uint32), want_c[:V].view(np.uint32))
        assert np.array_equal(radii.cpu().numpy(), want_r[:V])
    assert L.gr_raster_debug_bucket_cooldown(-1) == 0


def test_one_camera_frames_equal_the_plain_path_bit_for_bit():
    from gaussreg_amd import _lib, synthetic
    L = _lib.lib()
    L.gr_raster_debug_bucket_cooldown(0)
    P, W, H = 200000, 320, 240
    g = synthetic.gaussians_c2(P, seed=5, sh_degree=3)
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    sets = [_settings(c) for c in synthetic.camera_ring(4, W, H, seed=1)]
    want_c, want_r = _plain_reference(sets, t)
    for f, (img, radii) in enumerate(_render_each(sets, t, 12)):
        assert np.array_equal(img.view(np.uint32), want_c[f % 4].view(np.uint32)), f"frame {f}"
        assert np.array_equal(radii, want_r[f % 4])
    assert L.gr_raster_debug_bucket_cooldown(-1) == 0  # no bucket of this scene overflowed: the frames took the bucket sort


def test_a_crowded_bucket_falls_back_and_the_frame_is_still_exact():
    """40 000 Gaussians inside 4 mm of depth plus one very near and one very far: the key range is wide, so all of them land
    in ONE bucket of the top-digit pass (> 7 936 entries) -- the bucket launch raises the flag, the call repeats the frame with
    the three-pass sort and stays with it for a while."""
    from gaussreg_amd import _lib, synthetic
    L = _lib.lib()
    L.gr_raster_debug_bucket_cooldown(0)
    P, W, H = 40002, 256, 192
    rng = np.random.default_rng(11)
    g = synthetic.gaussians_c2(P, seed=2, sh_degree=3)
    g["means3D"][:, 0] = (rng.random(P) - 0.5) * 3.0
    g["means3D"][:, 1] = (rng.random(P) - 0.5) * 2.0
    g["means3D"][:, 2] = 3.0 + rng.random(P) * 0.004
    g["means3D"][0] = (0.0, 0.0, 0.3)
    g["means3D"][1] = (0.5, 0.5, 60.0)
    g["means3D"] = g["means3D"].astype(np.float32)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in g.items()}
    sets = [_settings(synthetic.camera(W, H), bg=(0.2, 0.1, 0.0))]
    want_c, want_r = _plain_reference(sets, t)
    frames = _render_each(sets, t, 6)
    for f, (img, radii) in enumerate(frames):
        assert np.array_equal(img.view(np.uint32), want_c[0].view(np.uint32)), f"frame {f}"
        assert np.array_equal(radii, want_r[0])
    assert (want_r[0] > 0).sum() > 30000
    left = L.gr_raster_debug_bucket_cooldown(0)
    assert 0 < left <= 256, left  # the overflow was seen and the thread is in its three-pass period
    # the same through the plain path (first call of a three-camera shape: the host reads the flag next to the counts and
    # orders the views again in place)
    from gaussreg_amd.rasterizer import rasterize_views
    color, radii5, _ = rasterize_views(sets * 3, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                       rotations=t["rotations"])
    for v in range(3):
        assert np.array_equal(color[v].cpu().numpy().view(np.uint32), want_c[0].view(np.uint32)), f"view {v}"
        assert np.array_equal(radii5[v].cpu().numpy(), want_r[0])
    assert L.gr_raster_debug_bucket_cooldown(0) == 256
    # ... and a scene that fits takes the bucket sort again once the period is cleared
    g2 = synthetic.gaussians_c2(20000, seed=3, sh_degree=3)
    t2 = {k: torch.from_numpy(v).cuda() for k, v in g2.items()}
    want2_c, _ = _plain_reference(sets, t2)
    for img, _ in _render_each(sets, t2, 3):
        assert np.array_equal(img.view(np.uint32), want2_c[0].view(np.uint32))
    assert L.gr_raster_debug_bucket_cooldown(-1) == 0


def test_equal_depths_take_the_copy_branch():
    """Every Gaussian at exactly the same camera depth: the key range is empty, one bucket holds everything and is already in
    id order (no LDS sort, no overflow however many there are)."""
    from gaussreg_amd import _lib, synthetic
    L = _lib.lib()
    L.gr_raster_debug_bucket_cooldown(0)
    P, W, H = 30000, 192, 144
    rng = np.random.default_rng(4)
    g = synthetic.gaussians_c2(P, seed=8, sh_degree=3)
    g["means3D"][:, 0] = (rng.random(P) - 0.5) * 3.0
    g["means3D"][:, 1] = (rng.random(P) - 0.5) * 2.0
    g["means3D"][:, 2] = 2.5
    g["means3D"] = g["means3D"].astype(np.float32)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).cuda() for k, v in g.items()}
    sets = [_settings(synthetic.camera(W, H))]
    want_c, want_r = _plain_reference(sets, t)
    for img, radii in _render_each(sets, t, 4):
        assert np.array_equal(img.view(np.uint32), want_c[0].view(np.uint32))
        assert np.array_equal(radii, want_r[0])
    assert L.gr_raster_debug_bucket_cooldown(-1) == 0


def test_fuzz_small_odd_shapes_plain_and_deferred_frames():
    """Random small scenes and image sizes, 1 - 4 cameras per call, three calls each (the first takes the plain path, the others
    are deferred: bucket sort + the scatter that scans its own segments), against the three-pass sort with its own count / scan
    launches -- images and radii bit for bit."""
    from gaussreg_amd import _lib, synthetic
    from gaussreg_amd.rasterizer import rasterize_views
    L = _lib.lib()
    rng = np.random.default_rng(2024)
    for case in range(14):
        P = int(rng.integers(1, 6000))
        W, H = int(rng.integers(17, 300)), int(rng.integers(9, 200))
        V = int(rng.integers(1, 5))
        g = synthetic.gaussians_c2(P, seed=100 + case, sh_degree=3)
        if case % 3 == 0:  # bigger splats: rectangles of many tiles, the marker path
            g["scales"] = (g["scales"] + np.float32(1.5)).astype(np.float32)
        t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
        sets = [_settings(c, bg=(0.1, 0.2, 0.3)) for c in synthetic.camera_ring(V, W, H, seed=case)]
        kw = dict(shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        if case % 4 == 1:  # the other preprocess variants: colours given, no SH
            kw = dict(colors_precomp=torch.rand(P, 3, generator=torch.Generator().manual_seed(case)).cuda(), scales=t["scales"],
                      rotations=t["rotations"])
        elif case % 4 == 2:  # ... 4 SH coefficients (degree 1): not the 16-coefficient fast path
            sets = [s._replace(sh_degree=1) for s in sets]
            kw = dict(shs=t["shs"][:, :4].contiguous(), scales=t["scales"], rotations=t["rotations"])
        L.gr_raster_debug_bucket_cooldown(1 << 20)
        try:
            want_c, want_r, _ = rasterize_views(sets, t["means3D"], t["opacities"], **kw)
            want_c, want_r = want_c.cpu().numpy(), want_r.cpu().numpy()
        finally:
            L.gr_raster_debug_bucket_cooldown(0)
        for call in range(3):
            c, r, _ = rasterize_views(sets, t["means3D"], t["opacities"], **kw)
            assert np.array_equal(c.cpu().numpy().view(np.uint32), want_c.view(np.uint32)), (case, call, P, W, H, V)
            assert np.array_equal(r.cpu().numpy(), want_r), (case, call)
        assert L.gr_raster_debug_bucket_cooldown(-1) == 0, case


def test_large_image_few_cameras_deferred():
    """1920 x 1080 (8 160 tiles: the scatter keeps four waves per chunk and, with rectangles of 5 - 8 tiles a side, expands
    them through its per-wave instance list), two cameras per call, deferred frames with the self-scanning scatter."""
    from gaussreg_amd import _lib, synthetic
    from gaussreg_amd.rasterizer import rasterize_views
    L = _lib.lib()
    P, W, H, V = 60000, 1920, 1080, 2
    g = synthetic.gaussians_c2(P, seed=21, sh_degree=3)
    g["scales"] = (g["scales"] * np.float32(2.5)).astype(np.float32)  # splats of a few tiles a side
    t = {k: torch.from_numpy(v).cuda() for k, v in g.items()}
    sets = [_settings(c) for c in synthetic.camera_ring(V, W, H, seed=4)]
    L.gr_raster_debug_bucket_cooldown(1 << 20)
    try:
        want_c, want_r, _ = rasterize_views(sets, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"],
                                            rotations=t["rotations"])
        want_c, want_r = want_c.cpu().numpy(), want_r.cpu().numpy()
    finally:
        L.gr_raster_debug_bucket_cooldown(0)
    for call in range(3):
        c, r, _ = rasterize_views(sets, t["means3D"], t["opacities"], shs=t["shs"], scales=t["scales"], rotations=t["rotations"])
        assert np.array_equal(c.cpu().numpy().view(np.uint32), want_c.view(np.uint32)), call
        assert np.array_equal(r.cpu().numpy(), want_r), call
