"""Drop-in module name for MonoGS: ``from diff_gaussian_rasterization import
GaussianRasterizationSettings, GaussianRasterizer``
(/root/reference/gaussian_splatting/gaussian_renderer/__init__.py:13-16)."""
from monogs_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    rasterize_gaussians,
)
