"""Drop-in for RTG-SLAM's un-vendored `cuda_utils` package (/root/reference/SLAM/multiprocess/mapper.py:15 does
`from cuda_utils._C import accumulate_gaussian_error`) - the MI355X build."""
