"""BASELINE.json configs[0] in miniature, on the CPU only: "first 50 frames, GT poses, PyTorch-CPU render path (no GPU,
plumbing reference)".  A synthetic Replica-shaped stream (the box room of rtg_slam_amd/synth.py, smooth trajectory) is
rendered frame by frame AT ITS GROUND-TRUTH POSES through the CPU oracle (oracle/raster_oracle.py - the repository's
PyTorch-CPU render path; the reference has none of its own, its rasterizer is CUDA-only) from a fixed single-layer map
painted with the room's colour function, and compared with the frames: PSNR, depth L1, seconds per frame.

    python -m tests.test_config0_cpu            # the 50 frames at quarter Replica resolution (a few minutes of CPU)

The pytest form runs 4 frames at 96x128.  Nothing here touches the HIP library."""
import math
import sys
import time

import torch

from oracle import raster_oracle as ro
from rtg_slam_amd import synth


def run(cam, n_frames, n_gaussians, seed=5, log=None):
    g = synth.surface_gaussians(n_gaussians, cam, seed=7)
    poses = synth.trajectory(n_frames, seed=seed)               # c2w per frame, <= 2 cm / 1 degree apart
    stats = []
    for fid, c2w in enumerate(poses):
        depth = synth.box_room_depth(cam, c2w, bump=0.0)
        color = synth.box_room_color(cam, c2w, depth)
        view = torch.linalg.inv(c2w).float().t().contiguous()
        s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, viewmatrix=view, campos=c2w[:3, 3].float())
        t0 = time.perf_counter()
        with torch.no_grad():
            out = ro.rasterize(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"])
        dt = time.perf_counter() - t0
        covered = out[6][0] < 0.5
        mse = float(((out[0] - color) ** 2)[:, covered].mean())
        hit = covered & (out[1][0] > 0)
        d_l1 = float((out[1][0] - depth[..., 0]).abs()[hit].mean())
        stats.append(dict(frame=fid, psnr=10 * math.log10(1.0 / max(mse, 1e-12)), depth_l1_m=d_l1,
                          covered=float(covered.float().mean()), depth_hits=float(hit.float().mean()), seconds=dt))
        if log:
            log(stats[-1])
    return stats


def test_cpu_render_path_over_a_gt_pose_stream():
    cam = synth.CameraSpec(96, 128, 64.0, 64.0, 63.5, 47.5)
    stats = run(cam, 4, 120_000)
    for st in stats:
        assert st["covered"] > 0.97, st
        assert st["psnr"] > 24.0, st
        assert st["depth_hits"] > 0.7 and st["depth_l1_m"] < 0.005, st      # grazing walls fail the 60-degree normal gate


if __name__ == "__main__":
    c = synth.REPLICA
    down = 4
    cam = synth.CameraSpec(c.H // down, c.W // down, c.fx / down, c.fy / down, (c.cx + 0.5) / down - 0.5, (c.cy + 0.5) / down - 0.5)
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    out = run(cam, frames, 400_000, log=lambda s: print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in s.items()}))
    print("mean PSNR %.2f dB, mean depth L1 %.4f m, %.2f s per frame" % (sum(s["psnr"] for s in out) / len(out),
          sum(s["depth_l1_m"] for s in out) / len(out), sum(s["seconds"] for s in out) / len(out)))
