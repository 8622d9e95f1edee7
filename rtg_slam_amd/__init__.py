"""rtg_slam_amd - MI355X-native (gfx950) hot path of RTG-SLAM: the differentiable
Gaussian-splatting rasterizer and the ICP frame-to-model tracker, as hand-written HIP
kernels behind a C-ABI (include/*.h) with a thin Python host mirroring the reference's
call signatures (SLAM/render.py, SLAM/icp.py)."""
__version__ = "0.1.0"
