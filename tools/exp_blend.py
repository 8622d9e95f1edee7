"""Experiment: how does blend_fwd / blend_bwd time scale with the number of active tiles?"""
import sys, os, math, ctypes as C, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import synth, _lib
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
lib = _lib.load()
cam = synth.REPLICA
dev = "cuda:0"
g = {k: v.to(dev) for k, v in synth.random_gaussians(1_200_000, cam, seed=2024).items()}
rs = GaussianRasterizationSettings(cam.H, cam.W, cam.W/(2*cam.fx), cam.H/(2*cam.fy), torch.zeros(3, device=dev), 1.0,
     torch.eye(4, device=dev), torch.eye(4, device=dev), 3, torch.zeros(3, device=dev), 0.6, 1.0, 0.5, 3.0, False, False, cam.cx, cam.cy, 1e-4)
rast = GaussianRasterizer(raster_settings=rs)
gy, gx = (cam.H+15)//16, (cam.W+15)//16
lib.rtgs_raster_set_profiling(1)
gc = torch.randn(3, cam.H, cam.W, device=dev); gd = torch.randn(1, cam.H, cam.W, device=dev)
for name, frac in (("all", 1.0), ("half", 0.5), ("quarter", 0.25), ("1/16", 1/16)):
    mask = torch.zeros(gy*gx, dtype=torch.int32, device=dev)
    k = int(gy*gx*frac)
    idx = torch.linspace(0, gy*gx-1, k).long().to(dev)
    mask[idx] = 1
    mask = mask.view(gy, gx)
    acc = [0.0]*8
    for it in range(6):
        leaves = {k2: v.clone().requires_grad_(True) for k2, v in g.items()}
        out = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None, scales=leaves["scales"],
                   rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"], tile_mask=mask)
        ((out[0]*gc).sum() + (out[1]*gd).sum()).backward()
        torch.cuda.synchronize()
        ms = (C.c_float*12)(); lib.rtgs_raster_last_timings(ms)
        if it >= 2:
            for q in range(8): acc[q] += max(0.0, ms[q])/4
    st = (C.c_int64*8)(); lib.rtgs_raster_last_stats(st)
    print(f"{name:8s} tiles {k:5d} R {st[0]:9d} | pre {acc[0]:.3f} count {acc[1]:.3f} scatter {acc[2]:.3f} sort {acc[3]:.3f} blend_fwd {acc[5]:.3f} blend_bwd {acc[6]:.3f} pre_bwd {acc[7]:.3f}")
