"""Host-side cost of one bench frame (diagnostic): wall time of each Python-level segment, measured without extra
device syncs, plus a cProfile of 100 frames.  Run on the GPU box: python tools/host_profile.py"""
import cProfile, pstats, sys, time, math, io
sys.path.insert(0, ".")
import torch
import bench as B
from rtg_slam_amd import synth, map_optim as mo, icp as hicp
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

dev = torch.device("cuda:0")
cam = synth.REPLICA
N = 1_200_000
g = synth.random_gaussians(N, cam, seed=2024)
packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
opt = mo.ShardedMapOptimizer(packed, lr_col=mo.default_lr_columns() * 1e-4)
view = torch.eye(4, device=dev)
rs = GaussianRasterizationSettings(
    image_height=cam.H, image_width=cam.W, tanfovx=cam.W / (2 * cam.fx), tanfovy=cam.H / (2 * cam.fy),
    bg=torch.zeros(3, device=dev), scale_modifier=1.0, viewmatrix=view, projmatrix=view, sh_degree=3,
    campos=torch.zeros(3, device=dev), opaque_threshold=0.6, depth_threshold=1.0,
    normal_threshold=math.cos(math.radians(60.0)), color_sigma=3.0, prefiltered=False, debug=False,
    cx=cam.cx, cy=cam.cy, T_threshold=1e-4)
rast = GaussianRasterizer(raster_settings=rs)
tile_mask = torch.ones((cam.H + 15) // 16, (cam.W + 15) // 16, dtype=torch.int32, device=dev)
gt_color = torch.rand(3, cam.H, cam.W, device=dev)
poses = synth.trajectory(2, seed=9)
base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
d0 = synth.box_room_depth(cam, base @ poses[0]).to(dev)
d1 = synth.box_room_depth(cam, base @ poses[1]).to(dev)
gt_depth = d1.reshape(1, cam.H, cam.W)
K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32, device=dev)
vp0, np0 = hicp.build_pyramids(d0, K, 3)
cos_thr = math.cos(math.radians(20.0))
T = {}
def tick(name, t0):
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0

def loss_fn(gd):
    t0 = time.perf_counter()
    out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None, scales=gd["scales"],
               rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"], tile_mask=tile_mask,
               grad_rows=gd.get("grad_rows"))
    tick("raster forward (incl. its host sync)", t0)
    t0 = time.perf_counter()
    l = mo.slam_losses_hip(out, gt_color, gt_depth)
    tick("loss forward", t0)
    return l

icp_stream = torch.cuda.Stream(device=dev)
def frame():
    t0 = time.perf_counter()
    main = torch.cuda.current_stream(dev)
    icp_stream.wait_stream(main)
    with torch.cuda.stream(icp_stream):
        vp1, np1 = hicp.build_pyramids(d1, K, 3)
        out = hicp.icp_track(vp1, np1, vp0, np0, K, [0.25, 0.5, 1.0], [5, 5, 5], 0.1, cos_thr, 1e-4)
    tick("icp enqueue", t0)
    t0 = time.perf_counter()
    opt.step(loss_fn)
    tick("opt.step total", t0)
    main.wait_stream(icp_stream)

for _ in range(200):
    frame()
torch.cuda.synchronize()
T.clear()
n = 200
t0 = time.perf_counter()
for _ in range(n):
    frame()
torch.cuda.synchronize()
wall = time.perf_counter() - t0
print(f"wall per frame {1e6 * wall / n:.0f} us")
for k, v in T.items():
    print(f"  {k:45s} {1e6 * v / n:8.1f} us")
pr = cProfile.Profile()
pr.enable()
for _ in range(100):
    frame()
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
