"""Map-optimisation iterations only (no ICP, no CPU baseline) for rocprofv3 kernel traces:
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o x -- python tools/prof_raster.py [headline|surface] [iters]
Prints the per-stage HIP-event timings too."""
import ctypes as C
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import _lib, synth, map_optim as mo
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings

which = sys.argv[1] if len(sys.argv) > 1 else "headline"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N = int(os.environ.get("RTGS_N", 1_200_000))
lib = _lib.load()
if os.environ.get("RTGS_BWD_DEBUG_TOOL"):       # timing decompositions (tools/r06_decomp.sh): parts of the backward walk off, results wrong
    lib.rtgs_raster_set_bwd_debug(int(os.environ["RTGS_BWD_DEBUG_TOOL"]))
cam = synth.REPLICA
dev = torch.device("cuda", 0)
g = synth.random_gaussians(N, cam, seed=2024) if which == "headline" else synth.surface_gaussians(N, cam, seed=7)
opt = mo.ShardedMapOptimizer(mo.pack_from_activated({k: v.to(dev) for k, v in g.items()}), lr_col=mo.default_lr_columns() * 1e-4)
rs = GaussianRasterizationSettings(
    image_height=cam.H, image_width=cam.W, tanfovx=cam.W / (2 * cam.fx), tanfovy=cam.H / (2 * cam.fy),
    bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=torch.eye(4, device=dev), projmatrix=torch.eye(4, device=dev),
    sh_degree=3, campos=torch.zeros(3, device=dev), opaque_threshold=0.6, depth_threshold=1.0,
    normal_threshold=math.cos(math.radians(60.0)), color_sigma=3.0, prefiltered=False, debug=False, cx=cam.cx, cy=cam.cy,
    T_threshold=1e-4)
gt_color = torch.rand(3, cam.H, cam.W, generator=torch.Generator().manual_seed(7)).to(dev)
gt_depth = synth.box_room_depth(cam, torch.eye(4, dtype=torch.float64), bump=0.0).to(dev).reshape(1, cam.H, cam.W)
rm = torch.ones(cam.H, cam.W, dtype=torch.uint8, device=dev)
opt.begin_local_optimization()
for _ in range(20):
    opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
torch.cuda.synchronize()
lib.rtgs_raster_set_profiling(1)
acc = [0.0] * 11
import time
t0 = time.perf_counter()
for _ in range(iters):
    opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
    ms = (C.c_float * 12)()
    lib.rtgs_raster_last_timings(ms)
    for k in range(11):
        acc[k] += max(0.0, ms[k]) / iters
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters * 1e3
names = ["preprocess_fwd", "bin_count", "bin_scatter", "bin_tilesort", "ranges", "blend_fwd", "blend_bwd", "preprocess_bwd", "slice_bin", "slice_blend", "grad_reduce"]
sl = (C.c_int64 * 4)()
lib.rtgs_raster_last_slice_stats(sl)
print("near slice: used %d instances %d tiles finished %d left %d" % tuple(sl))
print(which, f"iter {dt:.3f} ms (with per-iteration sync) |", " ".join(f"{n}={v * 1e3:.0f}us" for n, v in zip(names, acc)))
lib.rtgs_raster_set_profiling(0)
torch.cuda.synchronize()
blocks = []
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(100):
        opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
    torch.cuda.synchronize()
    blocks.append((time.perf_counter() - t0) * 10)
sp = (C.c_int64 * 3)()
lib.rtgs_raster_speculation_stats_ctx(None, sp)
print(which, "un-profiled iteration ms, 5 x 100:", " ".join(f"{b:.4f}" for b in blocks), "| speculation (ok, failed, not eligible):", list(sp))
