"""Drop-in for RTG-SLAM's un-vendored `simple_knn` package (/root/reference/SLAM/gaussian_pointcloud.py:3 does
`from simple_knn._C import distCUDA2`) - the MI355X build (rtg_slam_amd.slam_ops.distCUDA2 over include/rtgs_slam.h)."""
