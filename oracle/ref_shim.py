"""Import modules of the reference tree (/root/reference) on CPU, from where they lie, with their heavy imports stubbed.

TEST INFRASTRUCTURE ONLY, and only usable in the build container: the GPU box has no /root/reference (tests that need
this module skip there; what they pin travels as committed fixtures under tests/golden/).

The reference's modules import cv2 / open3d / plyfile / pytorch3d / skimage at module scope and
utils/general_utils.py allocates CUDA tensors at import (general_utils.py:22-24); none of that is on the paths we pin,
so those modules are stubbed and devF / devI / devB become CPU identities.  Nothing is copied."""
from __future__ import annotations

import importlib
import os
import sys
import types

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(REF)


def install_stubs():
    if not available():
        raise RuntimeError("reference tree not present (build container only)")
    for name in ["cv2", "open3d", "plyfile", "pytorch3d", "pytorch3d.loss", "pytorch3d.ops", "skimage", "skimage.color",
                 "skimage.filters", "GPUtil", "tensorboardX", "torch.utils.tensorboard"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["cv2"].COLORMAP_JET = 2
    sys.modules["plyfile"].PlyData = object
    sys.modules["plyfile"].PlyElement = object
    sys.modules["pytorch3d.loss"].chamfer_distance = None
    sys.modules["pytorch3d.ops"].knn_points = None
    sys.modules["skimage"].filters = sys.modules["skimage.filters"]
    sys.modules["skimage.color"].rgb2gray = None
    sys.modules["torch.utils.tensorboard"].SummaryWriter = object
    if "scene.cameras" not in sys.modules:
        cams = types.ModuleType("scene.cameras")
        cams.Camera = object
        scene = types.ModuleType("scene")
        scene.__path__ = []
        sys.modules.setdefault("scene", scene)
        sys.modules["scene.cameras"] = cams
    if "utils.general_utils" not in sys.modules or not getattr(sys.modules["utils.general_utils"], "_rtgs_stub", False):
        gu = types.ModuleType("utils.general_utils")
        gu._rtgs_stub = True
        gu.devF = lambda t: t.float()
        gu.devI = lambda t: t.int()
        gu.devB = lambda t: t.bool()
        gu.quaternion_from_axis_angle = None
        gu.build_covariance_from_scaling_rotation = None
        gu.build_rotation = None
        gu.inverse_sigmoid = None
        pkg = sys.modules.get("utils")
        if pkg is None or not hasattr(pkg, "__path__"):
            pkg = types.ModuleType("utils")
            pkg.__path__ = [os.path.join(REF, "utils")]
            sys.modules["utils"] = pkg
        sys.modules["utils.general_utils"] = gu
    if REF not in sys.path:
        sys.path.append(REF)          # appended: this repository's drop-in packages win over same-named reference dirs


def load(name: str):
    """e.g. load("SLAM.utils"), load("utils.loss_utils")"""
    install_stubs()
    return importlib.import_module(name)
