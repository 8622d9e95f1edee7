"""Drop-in for RTG-SLAM's un-vendored `diff_gaussian_rasterization_depth` package
(/root/reference/SLAM/render.py:8-13 imports exactly these two names) - the MI355X build."""
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer"]
