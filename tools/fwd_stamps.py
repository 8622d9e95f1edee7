"""Where a launch of blend_fwd spends its time: per-wave stamps (rtgs_raster_set_fwd_stamps, include/rtgs_debug.h) of ONE
map-optimisation iteration on the 1.2 M / 1200x680 scenes.
    python tools/fwd_stamps.py [headline|surface]
Prints the launch's span, tile lifetimes, resident waves over time, and the share of a wave's cycles before its first walk
step (tile range -> list ids -> record gather -> block test -> barrier), inside the walk loops and behind them."""
import math
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import _lib, synth, map_optim as mo
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings

which = sys.argv[1] if len(sys.argv) > 1 else "headline"
N = int(os.environ.get("RTGS_N", 1_200_000))
lib = _lib.load()
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
for _ in range(30):
    opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
torch.cuda.synchronize()
gx, gy = (cam.W + 15) // 16, (cam.H + 15) // 16
tiles = gx * gy
runs = []
for rep in range(3):
    st = torch.zeros(tiles * 4 * 8, dtype=torch.int64, device=dev)
    lib.rtgs_raster_set_fwd_stamps(st.data_ptr())
    try:
        for _ in range(3):
            opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
        torch.cuda.synchronize()
    finally:
        lib.rtgs_raster_set_fwd_stamps(None)
    runs.append(st.cpu().numpy().reshape(tiles, 4, 8).astype(np.int64))
s = runs[-1]
live = s[:, :, 0] > 0
t0 = s[:, :, 0][live].min()
start = (s[:, :, 0] - t0) * 0.01
end = (s[:, :, 1] - t0) * 0.01
span = float(end[live].max())
act = live.any(1)
print(f"{which}: {int(act.sum())} of {tiles} tiles stamped; launch span {span:.1f} us")
life = np.where(live, end - start, 0.0)
tile_life = life.max(1)
print("tile lifetime us: p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(np.percentile(tile_life[act], [10, 50, 90, 99, 100])))
print("wave lifetime us: mean %.1f;  sum of wave lifetimes / (span x 1024 SIMDs x 6 slots) = %.2f" %
      (life[live].mean(), life[live].sum() / (span * 1024 * 6)))
tt = np.linspace(0, span, 21)[1:-1]
print("resident waves at 5 % steps of the span (6 144 slots):", " ".join(str(int(((start <= t) & (end > t) & live).sum())) for t in tt))
tot = s[:, :, 5][live].astype(np.float64)
mhz = tot.sum() / life[live].sum()
rng = s[:, :, 2][live].astype(np.float64); first = s[:, :, 3][live].astype(np.float64); walk = s[:, :, 4][live].astype(np.float64)
steps = s[:, :, 6][live].astype(np.float64); batches = s[:, :, 7][live].astype(np.float64)
print(f"shader clock seen by the waves: {mhz:.0f} MHz")
print(f"share of the waves' cycles: tile range arrives {rng.sum() / tot.sum():.3f}; ids + gather + block test + first barrier "
      f"{(first - rng).sum() / tot.sum():.3f} (mean {first.mean() / mhz:.2f} us until the first step); walk loops {walk.sum() / tot.sum():.3f}; "
      f"rest (later batches' staging, epilogue) {1.0 - (first.sum() + walk.sum()) / tot.sum():.3f}")
print(f"per wave: walk steps mean {steps.mean():.1f} (max {steps.max():.0f}), cycles per step {walk.sum() / max(1.0, steps.sum()):.0f}; batches mean {batches.mean():.2f} (max {batches.max():.0f})")
import heapq
for label, seq in (("launch order", list(tile_life[act])), ("longest first", sorted(tile_life[act], reverse=True))):
    slots = [0.0] * 1536
    heapq.heapify(slots)
    for L in seq:
        heapq.heappush(slots, heapq.heappop(slots) + L)
    print(f"list scheduling of the measured tile lifetimes on 1 536 slots, {label}: {max(slots):.1f} us")
print(f"mean load {tile_life[act].sum() / 1536:.1f} us")
order = np.argsort(-tile_life)[:8]
wst = np.where(live, s[:, :, 6], 0)
print("longest tiles: " + "; ".join(f"tile {t}: {tile_life[t]:.1f} us, starts {start[t][live[t]].min():.1f}, steps {int(wst[t].max())}, batches {int(s[t, :, 7].max())}" for t in order))
spans = [float(((r[:, :, 1][r[:, :, 0] > 0]).max() - (r[:, :, 0][r[:, :, 0] > 0]).min()) * 0.01) for r in runs]
print("span of the three stamped launches:", " ".join(f"{x:.1f}" for x in spans))
