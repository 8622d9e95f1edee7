"""`cuda_utils._C`: accumulate_gaussian_error(H, W, P, color_error, depth_error, normal_error, color_index, depth_index,
color_thres, depth_thres, normal_thres, True) -> 4 tensors of length P (mapper.py:541-565)."""
from rtg_slam_amd.slam_ops import accumulate_gaussian_error

__all__ = ["accumulate_gaussian_error"]
