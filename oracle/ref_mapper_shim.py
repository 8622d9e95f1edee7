"""Run the reference's OWN map lifecycle - SLAM/multiprocess/mapper.py::Mapping with SLAM/gaussian_pointcloud.py,
SLAM/render.py, SLAM/utils.py, utils/general_utils.py, utils/loss_utils.py - on the CPU, from where it lies under
/root/reference, so that rtg_slam_amd.mapping.Mapping can be pinned to it (tests/golden/mapping_ref.npz,
oracle/gen_mapping_golden.py).

TEST INFRASTRUCTURE ONLY, build container only (the GPU box has no /root/reference).  Nothing is copied: the reference's
modules are imported in place; three of them carry literal `device="cuda"` arguments, for those the source TEXT is read,
the literal replaced by "cpu" in memory and the result executed as the module.  What the reference gets from native code
is supplied by this repository's CPU oracles:

    diff_gaussian_rasterization_depth.{GaussianRasterizationSettings, GaussianRasterizer}  <- oracle/raster_oracle.py (autograd)
    simple_knn._C.distCUDA2                                                                  <- slam_ops_oracle.dist2_knn
    pytorch3d.ops.knn_points                                                                 <- slam_ops_oracle.knn_query
    cuda_utils._C.accumulate_gaussian_error                                                  <- slam_ops_oracle.accumulate_gaussian_error

so a difference between the two lifecycles is a difference in HOST LOGIC (masks, thresholds, row moves, schedules,
learning-rate groups, the loss and its optimiser), which is what this pin is for - the kernels have their own parity tests."""
from __future__ import annotations

import importlib
import math
import os
import sys
import types
from collections import namedtuple

import torch

from oracle import raster_oracle as ro
from oracle import ref_shim
from oracle import slam_ops_oracle as so

REF = ref_shim.REF
_PATCH = [('device="cuda"', 'device="cpu"'), ("device='cuda'", "device='cpu'"), ('torch.device("cuda")', 'torch.device("cpu")')]


def _load_patched(modname: str, relpath: str):
    path = os.path.join(REF, relpath)
    src = open(path).read()
    for a, b in _PATCH:
        src = src.replace(a, b)
    mod = types.ModuleType(modname)
    mod.__file__ = path
    sys.modules[modname] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


# ---------------------------------------------------------------- native modules of the reference, from the oracles
_Settings = namedtuple("GaussianRasterizationSettings", [
    "image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree",
    "campos", "opaque_threshold", "depth_threshold", "normal_threshold", "color_sigma", "prefiltered", "debug", "cx", "cy",
    "T_threshold"])


class _Rasterizer(torch.nn.Module):
    """GaussianRasterizer as SLAM/render.py:110-120 calls it; the seven maps come from the autograd oracle."""

    def __init__(self, raster_settings):
        super().__init__()
        self.rs = raster_settings

    def forward(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
                normal_w=None, tile_mask=None):
        s = self.rs
        fx, fy = s.image_width / (2 * s.tanfovx), s.image_height / (2 * s.tanfovy)
        st = ro.make_settings(s.image_height, s.image_width, fx, fy, s.cx, s.cy, viewmatrix=s.viewmatrix, campos=s.campos,
                              sh_degree=s.sh_degree, opaque_threshold=s.opaque_threshold, depth_threshold=s.depth_threshold,
                              normal_threshold_deg=math.degrees(math.acos(float(s.normal_threshold))), color_sigma=s.color_sigma)
        if means3D.shape[0] == 0:
            H, W = s.image_height, s.image_width
            z = torch.zeros
            return (z(3, H, W), z(1, H, W), -torch.ones(1, H, W, dtype=torch.int32), -torch.ones(1, H, W, dtype=torch.int32),
                    z(1, H, W), z(1, H, W), torch.ones(1, H, W))
        out = ro.rasterize(st, means3D, opacities, shs, scales, rotations, normal_w, tile_mask)
        # A custom autograd Function hands EVERY input a gradient tensor (zeros where nothing was rendered, e.g. under an empty
        # tile mask); plain autograd leaves `.grad` None there and mapper.py:455 dereferences it.  Tie the inputs in with an
        # exact zero so that the stand-in behaves like the Function it stands for (values unchanged: x + 0.0).
        tie = sum((t * 0.0).sum() for t in (means3D, opacities, shs, scales, rotations) if t is not None and t.requires_grad)
        if torch.is_tensor(tie):
            out = tuple(o + tie if o.is_floating_point() else o for o in out)
        return out


def _dist_cuda2(points):
    mean_d2, idx, _ = so.dist2_knn(points.float())
    return mean_d2, idx


def _knn_points(p1, p2, K=3, **kw):
    """pytorch3d.ops.knn_points(p1[1,N,3], p2[1,M,3], K) -> (dists[1,N,K] squared, idx[1,N,K], None), nearest first."""
    d2, idx = so.knn_query(p2[0].float(), p1[0].float(), -1, None)
    assert K == 3
    return d2[None], idx[None].long(), None


def install():
    """Stub / oracle-back everything the reference's mapper imports; returns the reference's mapper module.
    PROCESS-WIDE and not undone (sys.modules entries for the four native modules, torch.Tensor.cuda): call it from a process
    of its own (oracle/gen_mapping_golden.py), never from inside the test suite."""
    ref_shim.install_stubs()
    sys.modules["pytorch3d.ops"].knn_points = _knn_points
    for name in ("simple_knn", "simple_knn._C", "cuda_utils", "cuda_utils._C", "diff_gaussian_rasterization_depth"):
        sys.modules[name] = types.ModuleType(name)
    sys.modules["simple_knn"].__path__ = []
    sys.modules["cuda_utils"].__path__ = []
    sys.modules["simple_knn._C"].distCUDA2 = _dist_cuda2
    sys.modules["cuda_utils._C"].accumulate_gaussian_error = so.accumulate_gaussian_error
    sys.modules["diff_gaussian_rasterization_depth"].GaussianRasterizationSettings = _Settings
    sys.modules["diff_gaussian_rasterization_depth"].GaussianRasterizer = _Rasterizer
    # the real utils.general_utils (build_rotation, inverse_sigmoid, devF ...) instead of ref_shim's stub
    pkg = sys.modules["utils"]
    pkg.__path__ = [os.path.join(REF, "utils")]
    _load_patched("utils.general_utils", "utils/general_utils.py")
    slam_pkg = types.ModuleType("SLAM")
    slam_pkg.__path__ = [os.path.join(REF, "SLAM")]
    sys.modules["SLAM"] = slam_pkg
    mp_pkg = types.ModuleType("SLAM.multiprocess")
    mp_pkg.__path__ = [os.path.join(REF, "SLAM", "multiprocess")]
    sys.modules["SLAM.multiprocess"] = mp_pkg
    _load_patched("SLAM.utils", "SLAM/utils.py")
    _load_patched("SLAM.gaussian_pointcloud", "SLAM/gaussian_pointcloud.py")
    if not hasattr(torch.Tensor, "_rtgs_cuda_patched"):
        torch.Tensor.cuda = lambda self, *a, **k: self            # `.cuda()` on a CPU-only box: stay where you are
        torch.Tensor._rtgs_cuda_patched = True
    return importlib.import_module("SLAM.multiprocess.mapper")
