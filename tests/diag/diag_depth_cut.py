"""Diagnostic (CPU, oracle pre-processing): how deep does each tile of the bench scene read its list before every
pixel is saturated, and what would a two-pass "near slice first" forward save?  Not part of the product."""
import sys, time
import torch
sys.path.insert(0, ".")
from oracle import raster_oracle as ro
from rtg_slam_amd import synth

torch.set_num_threads(8)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
cam = synth.REPLICA
g = synth.random_gaussians(N, cam, seed=2024)
s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy)
t0 = time.time()
pre = ro.preprocess(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"])
gx, gy = pre["gx"], pre["gy"]
mask = torch.ones(gy, gx, dtype=torch.int32)
gid, tile, ranges = ro.bin_tiles(pre, mask)
print("instances (3-sigma rects)", gid.numel(), "prep+bin s", round(time.time() - t0, 1))
depth = pre["depth"].float()
zi = depth[gid]
T16 = 16
ly, lx = torch.meshgrid(torch.arange(T16), torch.arange(T16), indexing="ij")
zstop = torch.full((gx * gy,), float("inf"))
cons = torch.zeros(gx * gy, dtype=torch.int64)
t0 = time.time()
for t in range(gx * gy):
    lo, hi = int(ranges[t, 0]), int(ranges[t, 1])
    ty, tx = divmod(t, gx)
    px = (tx * T16 + lx).reshape(-1).float(); py = (ty * T16 + ly).reshape(-1).float()
    inside = (px < cam.W) & (py < cam.H)
    T = torch.ones(256); done = ~inside
    pos = lo
    while pos < hi and not bool(done.all()):
        ids = gid[pos:min(pos + 32, hi)]
        dx = pre["u"][ids][:, None] - px[None]; dy = pre["v"][ids][:, None] - py[None]
        con = pre["conic"][ids]
        power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
        alpha = torch.clamp(pre["opacity"][ids][:, None] * torch.exp(power), max=0.99)
        ok = (power <= 0) & (alpha >= 1 / 255.)
        a = torch.where(ok, alpha, torch.zeros_like(alpha))
        Tn = T[None] * torch.cumprod(1 - a, 0)
        # per pixel: first row where T_after < thr
        below = Tn < s.T_threshold
        first = torch.where(below.any(0), below.float().argmax(0), torch.full((256,), ids.numel()))
        # how many rows until ALL (not yet done) pixels are done
        need = torch.where(done, torch.zeros(256, dtype=torch.long), first + 1)
        allrows = int(need.max())
        if allrows <= ids.numel() and bool((below.any(0) | done).all()):
            pos += allrows
            done[:] = True
            break
        T = Tn[-1]; done = done | below.any(0)
        pos += ids.numel()
    cons[t] = pos - lo
    if bool(done.all()) and pos > lo:
        zstop[t] = float(zi[pos - 1])
print("blend sim s", round(time.time() - t0, 1), "consumed", int(cons.sum()), "tiles saturated", int(torch.isfinite(zstop).sum()), "of", gx * gy)
n_t = (ranges[:, 1] - ranges[:, 0])
qs = torch.tensor([0.01, 0.02, 0.03, 0.05, 0.08, 0.12, 0.2, 0.3])
zs = torch.quantile(depth[pre["valid"]][::7], qs)
for q, z in zip(qs.tolist(), zs.tolist()):
    inst1 = int((zi <= z).sum())
    fin = zstop <= z
    inst2 = int(n_t[~fin].sum())
    print(f"gaussian depth quantile {q:5.2f} z<={z:6.3f}: pass-1 instances {inst1/1e6:6.2f} M, tiles finished {int(fin.sum()):5d}/{gx*gy}, "
          f"pass-2 instances {inst2/1e6:6.2f} M, total {(inst1+inst2)/1e6:6.2f} M vs {gid.numel()/1e6:.2f} M")
zq = torch.quantile(zstop[torch.isfinite(zstop)], torch.tensor([0.0, 0.1, 0.5, 0.9, 0.99, 1.0]))
print("z_stop quantiles over tiles", [round(v, 3) for v in zq.tolist()], "depth range", float(depth[pre['valid']].min()), float(depth[pre['valid']].max()))
