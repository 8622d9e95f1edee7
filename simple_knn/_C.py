"""`simple_knn._C`: distCUDA2(points[N,3]) -> (mean squared distance to the 3 nearest other points [N], indices [N,3]),
the signature RTG-SLAM's fork uses (gaussian_pointcloud.py:376: `_, knn_indices = distCUDA2(total_xyz.float().cuda())`)."""
from rtg_slam_amd.slam_ops import distCUDA2

__all__ = ["distCUDA2"]
