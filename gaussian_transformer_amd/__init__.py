"""MI355X-native differentiable Gaussian rasterizer (the hot path of stu214634/gaussian-transformer).

Public surface = the reference's rasterizer API (gaussian_renderer/__init__.py:14):
    from gaussian_transformer_amd import GaussianRasterizationSettings, GaussianRasterizer
also importable under the reference's module name `diff_gaussian_rasterization` (repo root).
The compute path is hand-written HIP for gfx950 behind the C ABI of include/gsr.h; there is no
CPU fallback.
"""
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: F401

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
