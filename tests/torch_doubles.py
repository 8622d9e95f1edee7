"""TEST DOUBLES - torch restatements of product kernels, used only by tests (the gloo tests inject them into
ShardedMapOptimizer in place of the HIP kernels; the GPU tests check the HIP kernels against them):

    activate8 / activate      raw8 -> opacity / scales / rotations / normal      gaussian_pointcloud.py:16-25, 538-550
    ssim                      utils/loss_utils.py:27-100
    slam_losses               the image terms of Mapping.loss_update             mapper.py:402-448

They lived in rtg_slam_amd/map_optim.py until round 3 (test doubles do not belong in the product module)."""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from rtg_slam_amd.map_optim import normal_loss_term


def rotmat_cols(q: torch.Tensor):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    c0 = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y + r * z), 2 * (x * z - r * y)], -1)
    c1 = torch.stack([2 * (x * y - r * z), 1 - 2 * (x * x + z * z), 2 * (y * z + r * x)], -1)
    c2 = torch.stack([2 * (x * z + r * y), 2 * (y * z - r * x), 1 - 2 * (x * x + y * y)], -1)
    return torch.stack([c0, c1, c2], dim=1)          # [N, 3 (column), 3]


def activate8(raw8: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Torch restatement (differentiable) of the raw8 activations: reference semantics, used by the
    gloo tests and as the checker of the HIP kernels."""
    N = raw8.shape[0]
    scales = torch.exp(raw8[:, 1:4])
    rot = torch.nn.functional.normalize(raw8[:, 4:8])
    cols = rotmat_cols(rot)
    k = scales.argmin(dim=1)
    n = cols[torch.arange(N, device=raw8.device), k]
    normal = n / (n.norm(dim=-1, keepdim=True) + 1e-8)
    return dict(opacity=torch.sigmoid(raw8[:, 0:1]), scales=scales, rotations=rot, normal=normal)


def activate(packed: torch.Tensor) -> Dict[str, torch.Tensor]:
    """Packed raw [N,59] -> the `gaussian_data` dict Renderer.render consumes (torch, differentiable)."""
    N = packed.shape[0]
    out = activate8(packed[:, 51:59])
    out["xyz"] = packed[:, 0:3]
    out["shs"] = packed[:, 3:51].reshape(N, 16, 3)
    return out


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """utils/loss_utils.py:58-100 (11x11 Gaussian window, sigma 1.5, zero padding, mean of the map), torch ops."""
    Cn = img1.shape[-3]
    g = torch.tensor([math.exp(-((x - window_size // 2) ** 2) / float(2 * 1.5 ** 2)) for x in range(window_size)])
    g = (g / g.sum()).to(img1)
    win = (g[:, None] @ g[None, :])[None, None].expand(Cn, 1, window_size, window_size).contiguous()
    a, b = img1.reshape(1, Cn, *img1.shape[-2:]), img2.reshape(1, Cn, *img2.shape[-2:])
    conv = lambda t: torch.nn.functional.conv2d(t, win, padding=window_size // 2, groups=Cn)
    mu1, mu2 = conv(a), conv(b)
    s11, s22, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))).mean()


def slam_losses(render, gt_color: torch.Tensor, gt_depth: torch.Tensor, color_weight: float = 0.8,
                depth_weight: float = 1.0, ssim_weight: float = 0.2, add_depth_thres: float = 0.1,
                render_mask: Optional[torch.Tensor] = None, normal_weight: float = 0.0,
                normal_w: Optional[torch.Tensor] = None, gt_normal: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Sync-free torch restatement of the image terms of Mapping.loss_update (mapper.py:402-448); `render` = the
    rasterizer's tuple (color[3,H,W], depth[1,H,W], ..., depth_index[1,H,W] at [3]).
      render_mask None -> every pixel AND the SSIM term 1 - ssim(render, gt) is live (:411-417); otherwise bool [H,W]
      colour: mean |C - C_gt| over the mask (:421); depth: mean |D - D_gt| over depth_index != -1 & D_gt > 0 &
      (D - D_gt) < add_depth_thres & mask (:423-431; an empty set gives 0 here, nan in the reference).
    Weights: configs/base.yaml:78-81.  The HIP kernel `slam_losses_hip` computes the same thing."""
    color, depth, didx = render[0], render[1], render[3]
    if render_mask is None:
        m = torch.ones_like(depth[0])
        ssim_loss = 1 - ssim(color, gt_color)
    else:
        m = render_mask.to(depth.dtype)
        ssim_loss = 0.0
    color_loss = ((color - gt_color).abs() * m).sum() / (3 * m.sum().clamp_min(1.0))
    err = depth[0] - gt_depth[0]
    vm = ((didx[0] != -1) & (gt_depth[0] > 0) & (err < add_depth_thres)).to(depth.dtype) * m
    depth_loss = (err.abs() * vm).sum() / vm.sum().clamp_min(1.0)
    total = depth_weight * depth_loss + color_weight * color_loss + ssim_weight * ssim_loss
    if normal_weight > 0:
        total = total + normal_weight * normal_loss_term(render, normal_w, gt_normal, render_mask)
    return total
