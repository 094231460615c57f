"""GPU parity: HIP rasterizer forward (through the C ABI) vs oracle/rasterizer_oracle.c, BIT-EXACT
on colour and radii (same IEEE operation sequence on both sides; see the oracle's header)."""
import numpy as np
import pytest
import torch

from helpers import raster_scene, oracle_render

pytestmark = pytest.mark.gpu


def _settings(cam, sh_degree=3, bg=(0.0, 0.0, 0.0), scale_modifier=1.0):
    from gaussreg_amd.rasterizer import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=cam["image_height"], image_width=cam["image_width"], tanfovx=cam["tanfovx"],
        tanfovy=cam["tanfovy"], bg=torch.tensor(bg, dtype=torch.float32, device="cuda"),
        scale_modifier=scale_modifier, viewmatrix=torch.from_numpy(cam["viewmatrix"]).cuda(),
        projmatrix=torch.from_numpy(cam["projmatrix"]).cuda(), sh_degree=sh_degree,
        campos=torch.from_numpy(cam["campos"]).cuda(), prefiltered=False, debug=False)


def _cu(g):
    return {k: torch.from_numpy(v).cuda() for k, v in g.items()}


def _assert_bit_equal(a, b, what):
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
        bad = np.argwhere(a.view(np.uint32) != b.view(np.uint32))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ; first at {bad[0]}: "
                             f"{a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}; max abs diff {np.abs(a - b).max()}")


@pytest.mark.parametrize("P,W,H,deg", [(20000, 160, 120, 3), (5000, 100, 70, 2), (3000, 64, 48, 1), (3000, 64, 48, 0)])
def test_forward_bit_exact_vs_oracle(P, W, H, deg):
    from gaussreg_amd.rasterizer import GaussianRasterizer
    g, cams = raster_scene(P, W, H, seed=P)
    cam = cams[0]
    bg = (0.1, 0.3, 0.7)
    want_img, want_radii, want_R = oracle_render(g, cam, sh_degree=deg, bg=bg)
    d = _cu(g)
    r = GaussianRasterizer(_settings(cam, deg, bg))
    img, radii = r(means3D=d["means3D"], means2D=None, opacities=d["opacities"], shs=d["shs"],
                   scales=d["scales"], rotations=d["rotations"])
    assert img.shape == (3, H, W) and img.dtype == torch.float32 and radii.dtype == torch.int32
    assert np.array_equal(radii.cpu().numpy(), want_radii)
    _assert_bit_equal(img.cpu().numpy(), want_img, "colour")


def test_colors_precomp_cov_precomp_scale_modifier():
    from gaussreg_amd.rasterizer import GaussianRasterizer, rasterize_views
    P, W, H = 6000, 128, 96
    g, cams = raster_scene(P, W, H, seed=3)
    rng = np.random.default_rng(0)
    colors = rng.random((P, 3)).astype(np.float32)
    # a valid symmetric 3D covariance per Gaussian (upper triangle, 6 floats)
    A = rng.normal(0, 0.02, (P, 3, 3))
    S = A @ np.transpose(A, (0, 2, 1)) + 1e-5 * np.eye(3)
    cov6 = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    d = _cu(g)
    # precomputed colours + scale modifier
    want, wr, _ = oracle_render(g, cams[0], colors_precomp=colors, scale_modifier=1.7)
    img, radii = GaussianRasterizer(_settings(cams[0], 3, scale_modifier=1.7))(
        d["means3D"], None, d["opacities"], colors_precomp=torch.from_numpy(colors).cuda(), scales=d["scales"],
        rotations=d["rotations"])
    assert np.array_equal(radii.cpu().numpy(), wr)
    _assert_bit_equal(img.cpu().numpy(), want, "colour (colors_precomp, scale_modifier)")
    # precomputed covariance
    want, wr, _ = oracle_render(g, cams[0], cov3D_precomp=cov6)
    img, radii = GaussianRasterizer(_settings(cams[0], 3))(
        d["means3D"], None, d["opacities"], shs=d["shs"], cov3D_precomp=torch.from_numpy(cov6).cuda())
    assert np.array_equal(radii.cpu().numpy(), wr)
    _assert_bit_equal(img.cpu().numpy(), want, "colour (cov3D_precomp)")


def test_multi_view_batch_equals_single_views():
    from gaussreg_amd.rasterizer import rasterize_views
    P, W, H, V = 8000, 160, 112, 5
    g, cams = raster_scene(P, W, H, seed=11, V=V)
    d = _cu(g)
    imgs, radii, nr = rasterize_views([_settings(c) for c in cams], d["means3D"], d["opacities"], shs=d["shs"],
                                      scales=d["scales"], rotations=d["rotations"])
    assert imgs.shape == (V, 3, H, W) and radii.shape == (V, P)
    for v in range(V):
        want, wr, wR = oracle_render(g, cams[v])
        assert 0 < nr[v] <= wR   # instances emitted: the reference count minus (tile, Gaussian) pairs that blend nothing
        assert np.array_equal(radii[v].cpu().numpy(), wr)
        _assert_bit_equal(imgs[v].cpu().numpy(), want, f"view {v}")


def test_argument_errors_and_mark_visible():
    from gaussreg_amd.rasterizer import GaussianRasterizer
    from oracle import capi
    g, cams = raster_scene(500, 64, 48, seed=1)
    d = _cu(g)
    r = GaussianRasterizer(_settings(cams[0]))
    with pytest.raises(Exception, match="excatly one of either SHs or precomputed colors"):
        r(d["means3D"], None, d["opacities"], scales=d["scales"], rotations=d["rotations"])
    with pytest.raises(Exception, match="exactly one of either scale/rotation pair"):
        r(d["means3D"], None, d["opacities"], shs=d["shs"])
    far = d["means3D"] - torch.tensor([0, 0, 3.0], device="cuda")
    vis = r.markVisible(far)
    assert np.array_equal(vis.cpu().numpy(), capi.mark_visible(far.cpu().numpy(), cams[0]["viewmatrix"]))


def test_empty_scene_renders_background():
    from gaussreg_amd.rasterizer import GaussianRasterizer
    g, cams = raster_scene(100, 64, 48, seed=1)
    d = _cu(g)
    behind = d["means3D"] * torch.tensor([1, 1, -1.0], device="cuda")
    img, radii = GaussianRasterizer(_settings(cams[0], bg=(0.25, 0.5, 0.75)))(
        behind, None, d["opacities"], shs=d["shs"], scales=d["scales"], rotations=d["rotations"])
    assert int(radii.abs().sum()) == 0
    assert torch.equal(img[:, 0, 0].cpu(), torch.tensor([0.25, 0.5, 0.75]))
    assert bool((img == img[:, :1, :1]).all())


@pytest.mark.parametrize("P,W,H,V", [(1_000_000, 640, 480, 2), (3_000_000, 1920, 1080, 1)])
def test_full_size_configs_bit_exact(P, W, H, V):
    """BASELINE configs[1] (1 M Gaussians, 640x480) and configs[3] (3 M Gaussians, 1920x1080) at FULL size:
    image and radii bit-identical to the oracle (the oracle needs a few seconds per view on one core)."""
    from gaussreg_amd.rasterizer import rasterize_views
    g, cams = raster_scene(P, W, H, seed=0, V=V)
    d = _cu(g)
    imgs, radii, nr = rasterize_views([_settings(c) for c in cams], d["means3D"], d["opacities"], shs=d["shs"],
                                      scales=d["scales"], rotations=d["rotations"])
    for v in range(V):
        want, wr, wR = oracle_render(g, cams[v])
        assert 0 < nr[v] <= wR   # instances emitted: the reference count minus (tile, Gaussian) pairs that blend nothing
        assert np.array_equal(radii[v].cpu().numpy(), wr)
        _assert_bit_equal(imgs[v].cpu().numpy(), want, f"{P} Gaussians {W}x{H} view {v}")


def test_large_image_many_views_tile_bands_bit_exact():
    """1920 x 1080 with several cameras per call: the tile scatter runs in bands of tile rows (8 160 tiles: the cursor rows of
    all of them do not fit a workgroup's LDS), incl. huge Gaussians that cross bands; every view bit-identical to the oracle."""
    from gaussreg_amd.rasterizer import rasterize_views
    P, W, H, V = 150_000, 1920, 1080, 5
    g, cams = raster_scene(P, W, H, seed=31, V=V)
    g["scales"][:40] = np.float32([0.4, 0.3, 0.2])   # a few screen-sized ones across many tile rows
    d = _cu(g)
    imgs, radii, nr = rasterize_views([_settings(c) for c in cams], d["means3D"], d["opacities"], shs=d["shs"],
                                      scales=d["scales"], rotations=d["rotations"])
    for v in (0, 2, 4):
        want, wr, wR = oracle_render(g, cams[v])
        assert 0 < nr[v] <= wR
        assert np.array_equal(radii[v].cpu().numpy(), wr)
        _assert_bit_equal(imgs[v].cpu().numpy(), want, f"1080p view {v}")


def test_huge_gaussians_take_the_rect_marker_path():
    """A rectangle wider than 63 tiles does not fit the bits packed into the depth-sort key: those Gaussians
    go through the gather fallback.  1232 x 48 image, a few screen-filling Gaussians among small ones."""
    from gaussreg_amd.rasterizer import GaussianRasterizer
    W, H, P = 1232, 48, 400
    g, cams = raster_scene(P, W, H, seed=21)
    g["scales"][:6] = np.float32([2.0, 0.05, 0.05])      # metres: hundreds of pixels wide
    g["means3D"][:6, :2] *= 0.2
    want, wr, wR = oracle_render(g, cams[0])
    assert wr[:6].max() > 64 * 16 // 2
    d = _cu(g)
    img, radii = GaussianRasterizer(_settings(cams[0]))(d["means3D"], None, d["opacities"], shs=d["shs"],
                                                        scales=d["scales"], rotations=d["rotations"])
    assert np.array_equal(radii.cpu().numpy(), wr)
    _assert_bit_equal(img.cpu().numpy(), want, "wide image with screen-filling Gaussians")


def test_zero_gaussians_and_odd_sizes():
    from gaussreg_amd.rasterizer import GaussianRasterizer, rasterize_views
    g, cams = raster_scene(50, 37, 23, seed=2)
    d = _cu(g)
    img, radii, nr = rasterize_views([_settings(cams[0], bg=(0.2, 0.4, 0.6))], d["means3D"][:0], d["opacities"][:0],
                                     shs=d["shs"][:0], scales=d["scales"][:0], rotations=d["rotations"][:0])
    assert img.shape == (1, 3, 23, 37) and radii.shape == (1, 0) and nr == [0]
    assert torch.equal(img[0, :, 5, 7].cpu(), torch.tensor([0.2, 0.4, 0.6]))
    want, wr, _ = oracle_render(g, cams[0])
    img, radii = GaussianRasterizer(_settings(cams[0]))(d["means3D"], None, d["opacities"], shs=d["shs"],
                                                        scales=d["scales"], rotations=d["rotations"])
    assert np.array_equal(radii.cpu().numpy(), wr)
    _assert_bit_equal(img.cpu().numpy(), want, "37x23 image")


def test_far_depth_takes_full_key_sort():
    """Depths >= 8192 do not fit the 27-bit rebased depth key: the call re-sorts with full-width keys.  The whole
    scene (positions, sizes, camera offsets) is scaled by 4000 -- same picture, depths of ~12 km."""
    from gaussreg_amd.rasterizer import rasterize_views
    P, W, H, V = 6000, 128, 96, 3
    g, cams = raster_scene(P, W, H, seed=5, V=V)
    k = np.float32(4000.0)
    g["means3D"] = (g["means3D"] * k).astype(np.float32)
    g["scales"] = (g["scales"] * k).astype(np.float32)
    cams = [dict(c) for c in cams]
    for c in cams:   # translate the camera centre with the scene: view = [R^T | -R^T C]
        vm = c["viewmatrix"].copy()          # stored transposed: last ROW holds the translation
        vm[3, :3] *= k
        proj = c["projmatrix"].T.astype(np.float64) @ np.linalg.inv(c["viewmatrix"].T.astype(np.float64))
        c["viewmatrix"] = vm
        c["projmatrix"] = (proj @ vm.T.astype(np.float64)).T.astype(np.float32)
        c["campos"] = (c["campos"] * k).astype(np.float32)
    d = _cu(g)
    imgs, radii, nr = rasterize_views([_settings(c) for c in cams], d["means3D"], d["opacities"], shs=d["shs"],
                                      scales=d["scales"], rotations=d["rotations"])
    for v in range(V):
        want, wr, wR = oracle_render(g, cams[v])
        assert wr.max() > 0 and want.std() > 0
        assert np.array_equal(radii[v].cpu().numpy(), wr)
        _assert_bit_equal(imgs[v].cpu().numpy(), want, f"far view {v}")


def test_fuzz_extreme_gaussians_bit_exact():
    """Random small scenes seeded with the extremes the cull rules have to survive: opacity 0 / 1 / barely above
    1/255, needle-thin and huge Gaussians, points behind / on the near plane, image sizes that are not tile
    multiples, 1-5 views -- images and radii must stay bit-identical to the oracle."""
    from gaussreg_amd.rasterizer import rasterize_views
    rng = np.random.default_rng(77)
    for it in range(12):
        P = int(rng.integers(200, 3000))
        W, H, V = int(rng.integers(17, 140)), int(rng.integers(17, 110)), int(rng.integers(1, 6))
        g, cams = raster_scene(P, W, H, seed=100 + it, V=V)
        k = P // 10
        g["opacities"][:k] = rng.choice(np.float32([0.0, 1.0, 1.0 / 255.0, 0.00393, 0.004, 0.0045, 0.5]), (k, 1))
        g["scales"][k:2 * k] = np.exp(rng.normal(np.log(0.05), 1.5, (k, 3))).astype(np.float32)       # from dust to walls
        g["scales"][2 * k:3 * k, 0] *= np.float32(40.0)                                               # needles
        g["means3D"][3 * k:4 * k, 2] = rng.choice(np.float32([-1.0, 0.0, 0.19, 0.2, 0.2001, 0.25]), k)  # near plane
        d = _cu(g)
        bg = tuple(float(x) for x in rng.random(3).astype(np.float32))
        imgs, radii, nr = rasterize_views([_settings(c, bg=bg) for c in cams], d["means3D"], d["opacities"], shs=d["shs"],
                                          scales=d["scales"], rotations=d["rotations"])
        for v in range(V):
            want, wr, wR = oracle_render(g, cams[v], bg=bg)
            assert np.array_equal(radii[v].cpu().numpy(), wr), (it, v)
            assert nr[v] <= wR
            _assert_bit_equal(imgs[v].cpu().numpy(), want, f"fuzz scene {it} view {v} ({P} Gaussians, {W}x{H})")


def test_ragged_sizes_bit_exact():
    """Gaussian counts around the wave / workgroup / chunk boundaries (1, 63 .. 65, 255 .. 257, 2047 .. 2049) with 1-9 views:
    the preprocess writes records wave-cooperatively, the depth sort packs ids into its words, the binning preloads
    whole 512-Gaussian wave slices -- none of it may depend on full waves, workgroups or chunks."""
    from gaussreg_amd.rasterizer import rasterize_views
    rng = np.random.default_rng(4242)
    for it, P in enumerate([1, 3, 63, 64, 65, 255, 257, 2047, 2049]):
        W, H, V = int(rng.integers(16, 120)), int(rng.integers(16, 90)), int(rng.integers(1, 10))
        g, cams = raster_scene(P, W, H, seed=1000 + it, V=V)
        d = _cu(g)
        imgs, radii, nr = rasterize_views([_settings(c) for c in cams], d["means3D"], d["opacities"], shs=d["shs"],
                                          scales=d["scales"], rotations=d["rotations"])
        for v in range(V):
            want, wr, _ = oracle_render(g, cams[v])
            assert np.array_equal(radii[v].cpu().numpy(), wr), (P, v)
            _assert_bit_equal(imgs[v].cpu().numpy(), want, f"{P} Gaussians, view {v} of {V}, {W}x{H}")
