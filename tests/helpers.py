"""Shared test helpers (CPU + GPU tests)."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name)))


def c1_points():
    """BASELINE config C1 inputs: torch.manual_seed(0); torch.rand(20000, 3)."""
    import torch
    g = torch.Generator().manual_seed(0)
    return torch.rand(20000, 3, generator=g).numpy()


def sqdist_f32(q, s):
    """The reference metric (nanoflann.hpp:432-440) in numpy fp32: ((dx*dx + dy*dy) + dz*dz)."""
    d = q.astype(np.float32) - s.astype(np.float32)
    d2 = d * d
    return (d2[..., 0] + d2[..., 1]) + d2[..., 2]


def assert_neighbors_equal_up_to_ties(got, want, q, s, q_lengths, s_lengths):
    """Rows must be identical except inside runs of EQUAL fp32 distance, where the reference's
    order depends on its kd-tree traversal: those runs compare as sets."""
    got = np.asarray(got).astype(np.int64)
    want = np.asarray(want).astype(np.int64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if np.array_equal(got, want):
        return 0
    pad = s.shape[0]
    bad_rows = np.nonzero((got != want).any(axis=1))[0]
    q_start = np.concatenate([[0], np.cumsum(q_lengths)])
    n_tie_rows = 0
    for r in bad_rows:
        g, w = got[r], want[r]
        assert ((g == pad) == (w == pad)).all(), f"row {r}: different neighbour count"
        k = int((w != pad).sum())
        dg = sqdist_f32(q[r][None, :], s[g[:k]])
        dw = sqdist_f32(q[r][None, :], s[w[:k]])
        assert np.array_equal(dg, dw), f"row {r}: distance sequences differ"
        # group by distance run and compare as sets
        i = 0
        while i < k:
            j = i
            while j + 1 < k and dw[j + 1] == dw[i]:
                j += 1
            assert set(g[i:j + 1]) == set(w[i:j + 1]), f"row {r}: tie group differs"
            i = j + 1
        n_tie_rows += 1
    return n_tie_rows


# ---------------------------------------------------------------- rasterizer scenes
def raster_scene(P, W, H, seed=0, sh_degree=3, V=1):
    from gaussreg_amd import synthetic
    g = synthetic.gaussians_c2(P, seed, sh_degree=3)
    cams = synthetic.camera_ring(V, W, H, seed)
    return g, cams


def oracle_render(g, cam, *, sh_degree=3, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, colors_precomp=None,
                  cov3D_precomp=None):
    from oracle import capi
    kw = dict(viewmatrix=cam["viewmatrix"], projmatrix=cam["projmatrix"], campos=cam["campos"], bg=np.asarray(bg, np.float32),
              W=cam["image_width"], H=cam["image_height"], tanfovx=cam["tanfovx"], tanfovy=cam["tanfovy"],
              sh_degree=sh_degree, scale_modifier=scale_modifier)
    if colors_precomp is None:
        kw["shs"] = g["shs"]
    else:
        kw["colors_precomp"] = colors_precomp
    if cov3D_precomp is None:
        kw["scales"], kw["rotations"] = g["scales"], g["rotations"]
    else:
        kw["cov3D_precomp"] = cov3D_precomp
    return capi.rasterize_forward(g["means3D"], g["opacities"], **kw)


def assert_rel_scale(got, want, rel=1e-5, what="", mask=None):
    """The north_star float bar: max |got - want| <= rel * max |want| (error relative to the tensor's scale; an
    elementwise rtol is meaningless for entries that cancel to ~0)."""
    got = np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, (got.shape, want.shape)
    if mask is not None:
        got, want = got[mask], want[mask]
    scale = float(np.abs(want).max()) if want.size else 0.0
    err = float(np.abs(got - want).max()) if want.size else 0.0
    assert err <= rel * scale + 1e-30, f"{what}: max |got - want| = {err:.3e} > {rel:g} x scale {scale:.3g} (= {err / max(scale, 1e-300):.2e} rel)"
    return err / max(scale, 1e-300)
