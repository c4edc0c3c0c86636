"""Drop-in module name: the reference does
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
(gaussian_renderer/__init__.py:14).  With this repo root on sys.path that import resolves here."""
from gaussian_transformer_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians"]
