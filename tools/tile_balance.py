"""Per-tile consumed list lengths of a scene (device counters of blend_fwd) and what the tile launch order costs:
greedy list scheduling of the tiles onto S workgroup slots in index order vs longest-first.
    python tools/tile_balance.py [headline|surface]"""
import ctypes as C
import heapq
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import _lib, synth
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings
from diff_gaussian_rasterization_depth import GaussianRasterizer

which = sys.argv[1] if len(sys.argv) > 1 else "surface"
N = int(os.environ.get("RTGS_N", 1_200_000))
lib = _lib.load()
cam = synth.REPLICA
dev = torch.device("cuda", 0)
g = synth.random_gaussians(N, cam, seed=2024) if which == "headline" else synth.surface_gaussians(N, cam, seed=7)
g = {k: v.to(dev) for k, v in g.items()}
rs = GaussianRasterizationSettings(
    image_height=cam.H, image_width=cam.W, tanfovx=cam.W / (2 * cam.fx), tanfovy=cam.H / (2 * cam.fy),
    bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=torch.eye(4, device=dev), projmatrix=torch.eye(4, device=dev),
    sh_degree=3, campos=torch.zeros(3, device=dev), opaque_threshold=0.6, depth_threshold=1.0,
    normal_threshold=math.cos(math.radians(60.0)), color_sigma=3.0, prefiltered=False, debug=False, cx=cam.cx, cy=cam.cy,
    T_threshold=1e-4)
gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
counters = torch.zeros(2 * gx * gy, dtype=torch.int64, device=dev)
rast = GaussianRasterizer(raster_settings=rs)
for _ in range(2):
    counters.zero_()
    lib.rtgs_raster_set_counters(C.c_void_p(counters.data_ptr()))
    with torch.no_grad():
        rast(means3D=g["xyz"], opacities=g["opacity"], shs=g["shs"], colors_precomp=None, scales=g["scales"],
             rotations=g["rotations"], cov3D_precomp=None, normal_w=g["normal"], tile_mask=None)
    lib.rtgs_raster_set_counters(None)
    torch.cuda.synchronize()
c = counters.view(-1, 2).cpu()
length, evals = c[:, 0].double(), c[:, 1].double()
print(which, "tiles", len(length), "consumed entries: mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f" % (
    length.mean(), length.median(), length.quantile(0.9), length.quantile(0.99), length.max()))
cost = (evals / 64.0).tolist()          # (entry, wave) pairs evaluated per tile ~ the tile's run time
for slots in (1024, 1536):
    for name, order in (("index order", list(range(len(cost)))), ("longest first", sorted(range(len(cost)), key=lambda t: -cost[t]))):
        h = [0.0] * slots
        heapq.heapify(h)
        for t in order:
            heapq.heappush(h, heapq.heappop(h) + cost[t])
        print(f"  {slots} slots, {name}: makespan {max(h):.0f}  (ideal {sum(cost) / slots:.0f}, longest tile {max(cost):.0f})")
