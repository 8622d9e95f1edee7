"""Where a launch of blend_bwd_entry spends its time: per-wave stamps (rtgs_raster_set_bwd_stamps, include/rtgs_debug.h) of ONE
map-optimisation iteration on the 1.2 M / 1200x680 scenes.
    python tools/bwd_stamps.py [headline|surface]
Prints: the launch's span, the distribution of tile (workgroup) lifetimes against it, how many waves are resident over time,
the share of a wave's cycles inside the group loop, cycles per quad step, and the critical tiles."""
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
    st = torch.zeros(tiles * 4 * 14, dtype=torch.int64, device=dev)
    lib.rtgs_raster_set_bwd_stamps(st.data_ptr())
    for _ in range(3):                                   # the stamped kernel's third launch: clocks and caches as in a run
        opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
    torch.cuda.synchronize()
    lib.rtgs_raster_set_bwd_stamps(None)
    runs.append(st.cpu().numpy().reshape(tiles, 4, 14).astype(np.int64))
s = runs[-1]
live = s[:, :, 0] > 0
t0 = s[:, :, 0][live].min()
start = (s[:, :, 0] - t0) * 0.01                         # us (100 MHz wall clock)
end = (s[:, :, 1] - t0) * 0.01
span = float(end[live].max())
print(f"{which}: {int(live.any(1).sum())} of {tiles} tiles walked; launch span (first wave in -> last wave out) {span:.1f} us")
life = np.where(live, end - start, 0.0)
tile_life = life.max(1)
act = live.any(1)
q = np.percentile(tile_life[act], [10, 50, 90, 99, 100])
print("tile lifetime us: p10 %.1f  p50 %.1f  p90 %.1f  p99 %.1f  max %.1f" % tuple(q))
print("wave lifetime us: mean %.1f;  sum of wave lifetimes / (span x 1024 SIMDs x 5 slots) = %.2f" %
      (life[live].mean(), life[live].sum() / (span * 1024 * 5)))
tt = np.linspace(0, span, 21)[1:-1]
res = [(int(((start <= t) & (end > t) & live).sum())) for t in tt]
print("resident waves at 5 % steps of the span (5 120 slots):", " ".join(str(r) for r in res))
cyc_total = s[:, :, 3][live].astype(np.float64)
cyc_walk = s[:, :, 2][live].astype(np.float64)
groups = s[:, :, 4][live].astype(np.float64)
steps = s[:, :, 5][live].astype(np.float64)
mhz = cyc_total.sum() / life[live].sum()
print(f"shader clock seen by the waves: {mhz:.0f} MHz;  cycles inside the group loop / all cycles = {cyc_walk.sum() / cyc_total.sum():.3f}")
print(f"per wave: groups mean {groups.mean():.1f} (max {groups.max():.0f}), quad steps entered mean {steps.mean():.1f} (max {steps.max():.0f});  "
      f"cycles per quad step inside the loop {cyc_walk.sum() / max(1.0, steps.sum()):.0f};  steps per group {steps.sum() / max(1.0, groups.sum()):.1f}")
print(f"whole launch: {int(steps.sum())} quad steps = {steps.sum() / 1024:.0f} per SIMD;  at the loop's cycles per step, one SIMD running its share "
      f"back to back needs {steps.sum() / 1024 * cyc_walk.sum() / max(1.0, steps.sum()) / mhz:.1f} us")
seg = s[:, :, 6:12][live].astype(np.float64).sum(0)
names = ["prologue (every first load landed)", "accumulator zeroing + depth partials", "staging (records -> LDS)",
         "compaction + barrier before the walk", "group loop", "barrier behind the loop", "per-entry tail (moments, slot store)"]
vals = [seg[0], seg[1], seg[2], seg[3], cyc_walk.sum(), seg[4], seg[5]]
print("share of the waves' cycles: " + "; ".join(f"{n} {v / cyc_total.sum():.3f}" for n, v in zip(names, vals)) +
      f"; unaccounted {1.0 - sum(vals) / cyc_total.sum():.3f}")
p1w = s[:, :, 12][live]
p_issue = (p1w & 0xffffffff).astype(np.float64).sum(); p1 = (p1w >> 32).astype(np.float64).sum(); p2 = s[:, :, 13][live].astype(np.float64).sum()
print(f"prologue, mean us per wave: all loads issued {p_issue / live.sum() / mhz:.2f}; first load back {p1 / live.sum() / mhz:.2f}; all back {p2 / live.sum() / mhz:.2f}")
print(f"inside the prologue: per-tile words arrive {p1 / cyc_total.sum():.3f}; the rest of the first loads {(p2 - p1) / cyc_total.sum():.3f} "
      f"(mean prologue {seg[0] / live.sum() / mhz:.2f} us)")
# waves of a tile: how unequal (the tile waits for its slowest quadrant at every batch barrier)
wsteps = np.where(live, s[:, :, 5], 0).astype(np.float64)
imb = wsteps.max(1)[act].sum() * 4 / max(1.0, wsteps[act].sum())
print(f"quadrant imbalance: sum over tiles of 4 x (slowest wave's steps) / all steps = {imb:.2f}")
# the critical tiles
order = np.argsort(-tile_life)[:8]
print("longest tiles: " + "; ".join(f"tile {t}: {tile_life[t]:.1f} us, starts {start[t][live[t]].min():.1f}, steps {int(wsteps[t].max())}" for t in order))
late = np.argsort(-end.max(1))[:8]
print("last to finish: " + "; ".join(f"tile {t}: in {start[t][live[t]].min():.1f} out {end[t].max():.1f}" for t in late))
# what would perfect packing of these tile lifetimes onto 1 280 workgroup slots give?
import heapq
slots = [0.0] * 1280
heapq.heapify(slots)
for L in tile_life[act]:                                  # in launch order
    heapq.heappush(slots, heapq.heappop(slots) + L)
print(f"list scheduling of the measured tile lifetimes on 1 280 slots, launch order: {max(slots):.1f} us; longest first: ", end="")
slots = [0.0] * 1280
heapq.heapify(slots)
for L in sorted(tile_life[act], reverse=True):
    heapq.heappush(slots, heapq.heappop(slots) + L)
print(f"{max(slots):.1f} us; mean load {tile_life[act].sum() / 1280:.1f} us")
spans = [float(((r[:, :, 1][r[:, :, 0] > 0]).max() - (r[:, :, 0][r[:, :, 0] > 0]).min()) * 0.01) for r in runs]
print("span of the three stamped launches:", " ".join(f"{x:.1f}" for x in spans))
