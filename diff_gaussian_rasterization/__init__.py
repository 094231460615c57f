"""``diff_gaussian_rasterization`` drop-in (forward only), backed by the gfx950 rasterizer."""
from gaussreg_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    ViewBatch,
    rasterize_gaussians,
    rasterize_views,
)
