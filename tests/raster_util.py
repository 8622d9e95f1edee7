"""Shared helpers of the rasterizer parity tests."""
import math

import torch

from oracle import raster_oracle as ro
from rtg_slam_amd import synth

FIELDS = ("xyz", "opacity", "shs", "scales", "rotations", "normal")


def make_scene(N, cam, seed=1, pose_seed=None, **kw):
    c2w = synth.look_at_pose(pose_seed) if pose_seed is not None else None
    g = synth.random_gaussians(N, cam, seed=seed, c2w=c2w, **kw)
    view = torch.eye(4) if c2w is None else torch.linalg.inv(c2w).float().t().contiguous()
    s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, viewmatrix=view)
    return g, s


def oracle_run(s, g, tile_mask=None, grads=None, dtype=torch.float32):
    """Oracle forward (+ backward if `grads`=(g_color, g_depth)).  Returns (outs, grad dict, aux)."""
    leaves = {k: g[k].detach().to(dtype).clone().requires_grad_(grads is not None) for k in FIELDS}
    s2 = s._replace(bg=s.bg.to(dtype), viewmatrix=s.viewmatrix.to(dtype), campos=s.campos.to(dtype))
    outs, aux = ro.rasterize(s2, leaves["xyz"], leaves["opacity"], leaves["shs"], leaves["scales"],
                             leaves["rotations"], leaves["normal"], tile_mask, return_aux=True)
    gd = None
    if grads is not None:
        loss = (outs[0] * grads[0].to(dtype)).sum() + (outs[1] * grads[1].to(dtype)).sum()
        loss.backward()
        gd = {k: (leaves[k].grad if leaves[k].grad is not None else torch.zeros_like(leaves[k])) for k in FIELDS}
    return tuple(o.detach() for o in outs), gd, aux


def hip_settings(s, dev):
    from diff_gaussian_rasterization_depth import GaussianRasterizationSettings
    return GaussianRasterizationSettings(
        image_height=s.image_height, image_width=s.image_width, tanfovx=s.tanfovx, tanfovy=s.tanfovy,
        bg=s.bg.to(dev), scale_modifier=s.scale_modifier, viewmatrix=s.viewmatrix.to(dev),
        projmatrix=s.projmatrix.to(dev), sh_degree=s.sh_degree, campos=s.campos.to(dev),
        opaque_threshold=s.opaque_threshold, depth_threshold=s.depth_threshold,
        normal_threshold=s.normal_threshold, color_sigma=s.color_sigma, prefiltered=False, debug=False,
        cx=s.cx, cy=s.cy, T_threshold=s.T_threshold)


def hip_run(s, g, tile_mask=None, grads=None, dev="cuda:0"):
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    leaves = {k: g[k].detach().to(dev).clone().requires_grad_(grads is not None) for k in FIELDS}
    rast = GaussianRasterizer(raster_settings=hip_settings(s, dev))
    outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None,
                normal_w=leaves["normal"], tile_mask=None if tile_mask is None else tile_mask.to(dev))
    gd = None
    if grads is not None:
        loss = (outs[0] * grads[0].to(dev)).sum() + (outs[1] * grads[1].to(dev)).sum()
        loss.backward()
        gd = {k: leaves[k].grad.detach().cpu() for k in FIELDS}
    return tuple(o.detach().cpu() for o in outs), gd


def frac_bad(a, b, atol):
    return float(((a - b).abs() > atol).float().mean())
