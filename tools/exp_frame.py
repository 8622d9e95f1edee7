"""Experiment: how the tracker stage is enqueued next to the map step (frame time of bench.py's unit).
    python tools/exp_frame.py"""
import math, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import synth, icp as hicp, map_optim as mo
from rtg_slam_amd.pipeline import TrackMapPipeline
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings
dev = torch.device("cuda", 0)
cam = synth.REPLICA
N = 1_200_000
g = synth.random_gaussians(N, cam, seed=2024)
opt = mo.ShardedMapOptimizer(mo.pack_from_activated({k: v.to(dev) for k, v in g.items()}), lr_col=mo.default_lr_columns() * 1e-4)
rs = GaussianRasterizationSettings(cam.H, cam.W, cam.W / (2 * cam.fx), cam.H / (2 * cam.fy), torch.zeros(3, device=dev), 1.0,
                                   torch.eye(4, device=dev), torch.eye(4, device=dev), 3, torch.zeros(3, device=dev), 0.6, 1.0,
                                   math.cos(math.radians(60.0)), 3.0, False, False, cam.cx, cam.cy, 1e-4)
gt_color = torch.rand(3, cam.H, cam.W, generator=torch.Generator().manual_seed(7)).to(dev)
poses = synth.trajectory(2, seed=9); base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
d0 = synth.box_room_depth(cam, base @ poses[0]).to(dev); d1 = synth.box_room_depth(cam, base @ poses[1]).to(dev)
gt_depth = d1.reshape(1, cam.H, cam.W)
K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32, device=dev)
vp0, np0 = hicp.build_pyramids(d0, K, 3)
cos_thr = math.cos(math.radians(20.0))
rm = torch.ones(cam.H, cam.W, dtype=torch.uint8, device=dev)
opt.begin_local_optimization()
def track():
    vp1, np1 = hicp.build_pyramids(d1, K, 3)
    return hicp.icp_track(vp1, np1, vp0, np0, K, [0.25, 0.5, 1.0], [5, 5, 5], 0.1, cos_thr, 1e-4)
def mapstep():
    return opt.step_slam(rs, gt_color, gt_depth, None, render_mask=rm)
pipe = TrackMapPipeline(dev)
ts = torch.cuda.Stream(device=dev, priority=-1)
def frame_thread():
    pipe.track(track); mapstep(); return pipe.result()
def frame_main_icp_first():
    main = torch.cuda.current_stream(dev); ts.wait_stream(main)
    with torch.cuda.stream(ts): out = track()
    mapstep(); main.wait_stream(ts); return out
def frame_main_map_first():
    main = torch.cuda.current_stream(dev); ts.wait_stream(main)
    mapstep()
    with torch.cuda.stream(ts): out = track()
    main.wait_stream(ts); return out
def frame_serial():
    mapstep(); return track()
def only_map(): return mapstep()
def only_icp(): return track()
def only_icp_pipe():
    pipe.track(track); return pipe.result()
def host_time(fn, n=50):
    t0 = time.perf_counter()
    for _ in range(n): fn()
    h = time.perf_counter() - t0
    torch.cuda.synchronize(dev)
    return 1e3 * h / n
for name, fn in (("helper thread (bench)", frame_thread), ("main: icp first", frame_main_icp_first), ("main: map first", frame_main_map_first),
                 ("serial one stream", frame_serial), ("map only", only_map), ("icp only", only_icp), ("icp only via pipeline", only_icp_pipe)):
    for _ in range(60): fn()
    torch.cuda.synchronize(dev)
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(40): fn()
        torch.cuda.synchronize(dev)
        res.append(1e3 * (time.perf_counter() - t0) / 40)
    print(f"{name:26s} ms/frame {min(res):.4f} .. {max(res):.4f}   host enqueue ms {host_time(fn):.4f}")
