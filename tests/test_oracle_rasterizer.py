"""CPU: analytic known-answer tests that pin oracle/rasterizer_oracle.c (the rasterizer has no
reference implementation in the GaussReg tree -- SURVEY.md section 0 F3 -- so these closed forms ARE the pin)."""
import math

import numpy as np
import pytest

from oracle import capi
from gaussreg_amd import synthetic

W, H = 64, 48


def _cam():
    return synthetic.camera(W, H, 60.0)


def _render(means, scales, opac, rgb, cam=None, bg=(0, 0, 0), rot=None, **kw):
    cam = cam or _cam()
    P = len(means)
    rot = np.tile(np.array([[1, 0, 0, 0]], np.float32), (P, 1)) if rot is None else rot
    return capi.rasterize_forward(np.asarray(means, np.float32), np.asarray(opac, np.float32),
                                  colors_precomp=np.asarray(rgb, np.float32), scales=np.asarray(scales, np.float32),
                                  rotations=rot, viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"],
                                  campos=cam["campos"], bg=np.asarray(bg, np.float32), W=W, H=H,
                                  tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], **kw)


def test_exp_det_is_exp_to_a_few_ulp():
    """The blend's deterministic exponential (13 basic operations): within 2.5e-7 relative of exp where alpha can reach 1/255
    (x >= -5.55), 4.5e-7 down to -20, monotone, exactly 1 at 0, clamped at -86."""
    x = -np.abs(np.random.default_rng(0).normal(0, 3, 4000)).astype(np.float32)
    x = np.concatenate([x, np.float32([0.0, -5.5413, -20.0, -85.9, -86.0])])
    got = capi.exp_det(x).astype(np.float64)
    want = np.exp(x.astype(np.float64))
    rel = np.abs(got - want) / want
    assert rel[x >= -6.0].max() < 2.5e-7 and rel[x >= -20.0].max() < 4.5e-7 and rel.max() < 2e-6
    assert capi.exp_det([0.0])[0] == 1.0
    assert capi.exp_det([-100.0])[0] == capi.exp_det([-86.0])[0] and capi.exp_det([-1e30])[0] == capi.exp_det([-86.0])[0]
    xs = np.sort(np.float32(-np.random.default_rng(1).random(3000) * 8))
    g = capi.exp_det(xs)
    assert bool((np.diff(g.astype(np.float64)) >= -1e-7 * g[1:]).all())  # monotone up to the last bits


def test_single_isotropic_gaussian_closed_form():
    """Gaussian of sigma s on the optical axis at depth z: 2D covariance = (fx*s/z)^2 + 0.3 on both
    axes, alpha(p) = min(0.99, o * exp(-|p - c|^2 / (2 var))), colour = rgb*alpha, centre pixel
    coordinate ((0+1)*W-1)/2."""
    s, z, o = 0.05, 2.0, 0.8
    col = np.array([[0.2, 0.5, 0.9]], np.float32)
    img, radii, R = _render([[0, 0, z]], [[s, s, s]], [[o]], col)
    cam = _cam()
    fx = W / (2 * cam["tanfovx"])
    fy = H / (2 * cam["tanfovy"])
    assert abs(fx - fy) < 1e-3
    var = (fx * s / z) ** 2 + 0.3
    cx, cy = (W - 1) / 2, (H - 1) / 2
    assert radii[0] == math.ceil(3 * math.sqrt(var))
    ys, xs = np.mgrid[0:H, 0:W]
    d2 = (xs - cx) ** 2 + (ys - cy) ** 2
    alpha = np.minimum(0.99, o * np.exp(-d2 / (2 * var)))
    alpha[alpha < 1 / 255] = 0
    # pixels outside the tiles the 3-sigma radius touches are never visited
    r = radii[0]
    tx0, tx1 = int((cx - r) / 16), int((cx + r + 15) / 16)
    ty0, ty1 = int((cy - r) / 16), int((cy + r + 15) / 16)
    mask = np.zeros((H, W), bool)
    mask[ty0 * 16:ty1 * 16, tx0 * 16:tx1 * 16] = True
    alpha[~mask] = 0
    want = col[0][:, None, None] * alpha[None]
    assert R == (tx1 - tx0) * (ty1 - ty0)
    np.testing.assert_allclose(img, want, rtol=2e-5, atol=1e-6)


def test_front_to_back_compositing_and_background():
    o1, o2 = 0.6, 0.9
    c1, c2 = np.float32([1, 0, 0]), np.float32([0, 1, 0])
    bg = (0.1, 0.2, 0.3)
    # big flat splats so alpha at the centre pixel is ~ the opacity
    img, _, _ = _render([[0, 0, 1.0], [0, 0, 2.0]], [[0.5] * 3, [1.0] * 3], [[o1], [o2]], [c1, c2], bg=bg)
    cy, cx = H // 2, W // 2
    a1 = min(0.99, o1 * math.exp(-0.5 * (0.5 ** 2 * 2) / ((W / (2 * _cam()["tanfovx"]) * 0.5 / 1.0) ** 2 + 0.3)))
    a2 = min(0.99, o2 * math.exp(-0.5 * (0.5 ** 2 * 2) / ((W / (2 * _cam()["tanfovx"]) * 1.0 / 2.0) ** 2 + 0.3)))
    T = (1 - a1) * (1 - a2)
    want = c1 * a1 + c2 * a2 * (1 - a1) + np.float32(bg) * T
    np.testing.assert_allclose(img[:, cy, cx], want, rtol=1e-4)
    # swapping the depths swaps the compositing order
    img2, _, _ = _render([[0, 0, 2.0], [0, 0, 1.0]], [[1.0] * 3, [0.5] * 3], [[o2], [o1]], [c2, c1], bg=bg)
    np.testing.assert_allclose(img2[:, cy, cx], want, rtol=1e-4)


def test_near_plane_cull_and_alpha_rules():
    col = np.float32([[1, 1, 1]])
    # z <= 0.2 is culled (radius 0, background only)
    img, radii, R = _render([[0, 0, 0.2]], [[0.01] * 3], [[1.0]], col, bg=(0.5, 0.5, 0.5))
    assert radii[0] == 0 and R == 0 and np.all(img == 0.5)
    img, radii, R = _render([[0, 0, 0.21]], [[0.01] * 3], [[1.0]], col)
    assert radii[0] > 0
    # opacity 1: alpha is clamped to 0.99 at the centre
    img, _, _ = _render([[0, 0, 1.0]], [[0.5] * 3], [[1.0]], col)
    assert abs(img[0, H // 2, W // 2] - 0.99) < 2e-3 and img.max() <= 0.99 + 1e-6
    # opacity below 1/255 contributes nothing
    img, _, _ = _render([[0, 0, 1.0]], [[0.5] * 3], [[1.0 / 256]], col)
    assert np.all(img == 0)
    # T < 1e-4 stops: many opaque layers then a bright far one that must not show
    n = 6
    means = [[0, 0, 1.0 + 0.1 * i] for i in range(n)] + [[0, 0, 3.0]]
    cols = [[0, 0, 1]] * n + [[1, 0, 0]]
    img, _, _ = _render(means, [[0.5] * 3] * (n + 1), [[1.0]] * (n + 1), np.float32(cols))
    assert img[0, H // 2, W // 2] == 0.0  # 0.01^3 < 1e-4 reached before the red layer


def test_mark_visible_matches_cull():
    cam = _cam()
    m = np.float32([[0, 0, 0.1], [0, 0, 0.2], [0, 0, 0.2001], [0, 0, 5], [0, 0, -1]])
    assert capi.mark_visible(m, cam["viewmatrix"]).tolist() == [False, False, True, True, False]


def test_sh_basis_matches_reference_eval_sh():
    """Golden vector from the reference's own eval_sh (geotransformer/utils/graphics_utils.py:34-77),
    generated by tests/golden/gen_golden_sh.py: colour = max(eval_sh + 0.5, 0)."""
    from helpers import load_golden
    g = load_golden("sh_eval.npz")
    P = g["means"].shape[0]
    cam = synthetic.camera(W, H, 60.0, C=g["campos"])
    for deg in range(4):
        pre = capi.raster_preprocess(g["means"], np.ones((P, 1), np.float32), shs=g["shs"],
                                     scales=np.full((P, 3), 0.05, np.float32),
                                     rotations=np.tile(np.float32([[1, 0, 0, 0]]), (P, 1)), viewmatrix=cam["viewmatrix"],
                                     projmatrix=cam["projmatrix"], campos=cam["campos"], W=W, H=H,
                                     tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=deg)
        vis = pre["radii"] > 0
        assert vis.sum() > P // 4
        want = np.maximum(g[f"eval_deg{deg}"] + 0.5, 0.0)
        np.testing.assert_allclose(pre["rgb"][vis], want[vis], rtol=1e-5, atol=1e-6)


def test_scene_statistics_are_sane():
    g = synthetic.gaussians_c2(5000, 0)
    cam = synthetic.camera(160, 120)
    img, radii, R = capi.rasterize_forward(g["means3D"], g["opacities"], shs=g["shs"], scales=g["scales"],
                                           rotations=g["rotations"], viewmatrix=cam["viewmatrix"],
                                           projmatrix=cam["projmatrix"], campos=cam["campos"], bg=np.zeros(3, np.float32),
                                           W=160, H=120, tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"], sh_degree=3)
    assert (radii > 0).mean() > 0.3 and R >= (radii > 0).sum()
    assert np.isfinite(img).all() and img.max() > 0.05
