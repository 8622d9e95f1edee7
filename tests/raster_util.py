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


def explain_outliers(out_h, out_o, T_thr=1e-4, tol=1e-4, limit=24):
    """For every pixel whose colour / depth / weights / T differ by more than `tol` between the HIP maps and the oracle's:
    WHICH discontinuous decision of the blend flipped (SURVEY.md Appendix B "per-pixel forward").  Read off the seven
    output maps alone:
      depth gate         depth_index differs (the first opaque contributor passed |cos|, z_hit > 0 or |z_hit - z| < thr on one
                         side only, or a different contributor crossed opaque_threshold first)
      1/255 skip         the two final T differ by the factor (1 - alpha) of ONE entry with alpha ~ 1/255 (ratio in
                         [0.990, 0.9985]): that entry was blended by one side only; the colour moves by alpha T c at its depth
      T-threshold stop   final T differ otherwise and the smaller one is below 4 T_threshold: one side blended a last
                         contributor the other stopped before (T' < T_threshold)
      other flip         final T differ otherwise (power ~ 0 skip, clamp)
      arg-max tie        only color_index / color_weight differ: two contributors with (almost) equal alpha T
      rounding           none of the above: accumulated float error over a long list
    Returns {kind: count}; prints one line per pixel (at most `limit`)."""
    H, W = out_o[0].shape[-2:]
    over = torch.zeros(H, W, dtype=torch.bool)
    for k in (0, 1, 4, 5, 6):
        over |= ((out_h[k].float().cpu() - out_o[k].float().cpu()).abs() > tol).reshape(-1, H, W).any(0)
    kinds = {}
    ys, xs = torch.nonzero(over, as_tuple=True)
    for n, (y, x) in enumerate(zip(ys.tolist(), xs.tolist())):
        Th, To = float(out_h[6][0, y, x]), float(out_o[6][0, y, x])
        dih, dio = int(out_h[3][0, y, x]), int(out_o[3][0, y, x])
        cih, cio = int(out_h[2][0, y, x]), int(out_o[2][0, y, x])
        dc = float((out_h[0][:, y, x].cpu() - out_o[0][:, y, x].cpu()).abs().max())
        dd = abs(float(out_h[1][0, y, x]) - float(out_o[1][0, y, x]))
        ratio = min(Th, To) / max(Th, To, 1e-30)
        if dih != dio:
            kind = "depth gate"
        elif 0.990 <= ratio <= 0.9985:
            kind = "1/255 skip"
        elif ratio < 0.999 and min(Th, To) < 4 * T_thr:
            kind = "T-threshold stop"
        elif ratio < 0.999:
            kind = "other flip"
        elif cih != cio:
            kind = "arg-max tie"
        else:
            kind = "rounding"
        kinds[kind] = kinds.get(kind, 0) + 1
        if n < limit:
            print(f"  outlier pixel ({y},{x}): {kind}; T hip {Th:.3e} oracle {To:.3e}; |d colour| {dc:.2e} |d depth| {dd:.2e}; "
                  f"depth owner {dih} / {dio}; colour arg-max {cih} / {cio}")
    return kinds
