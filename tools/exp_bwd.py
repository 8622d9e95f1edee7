"""Experiment: blend_bwd stage time of the bench scene for the library given by RTGS_LIB_PATH."""
import sys, os, ctypes as C, torch
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
lib.rtgs_raster_set_profiling(1)
gc = torch.randn(3, cam.H, cam.W, device=dev); gd = torch.randn(1, cam.H, cam.W, device=dev)
acc = [0.0]*10
for it in range(8):
    leaves = {k2: v.clone().requires_grad_(True) for k2, v in g.items()}
    out = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None, scales=leaves["scales"],
               rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"], tile_mask=None)
    ((out[0]*gc).sum() + (out[1]*gd).sum()).backward()
    torch.cuda.synchronize()
    ms = (C.c_float*12)(); lib.rtgs_raster_last_timings(ms)
    if it >= 3:
        for q in range(10): acc[q] += max(0.0, ms[q])/5
print(f"{os.environ.get('RTGS_LIB_PATH','default'):40s} bwd_env={os.environ.get('RTGS_BLEND_BWD','-')} | slice_bin {acc[8]:.3f} slice_blend {acc[9]:.3f} blend_bwd(+memset) {acc[6]:.3f} pre_bwd {acc[7]:.3f}")
