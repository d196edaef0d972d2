"""`diff_gaussian_rasterization` (reference README.md:43) served by lvdgs."""
from lvdgs.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer,  # noqa: F401
                              rasterize_gaussians)
