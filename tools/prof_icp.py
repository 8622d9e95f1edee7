"""ICP tracker alone: pyramids + one 3 x 5 Gauss-Newton track per frame, timed with HIP events.
    [RTGS_ICP_PERSISTENT=0] python tools/prof_icp.py [replica|tum] [iters]"""
import math
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import synth, icp as hicp

cam = synth.REPLICA if (len(sys.argv) < 2 or sys.argv[1] == "replica") else synth.TUM_FR1
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
dev = torch.device("cuda", 0)
poses = synth.trajectory(2, seed=9)
base = synth.look_at_pose(seed=3, max_angle_deg=5, max_trans=0.3)
d0 = synth.box_room_depth(cam, base @ poses[0]).to(dev)
d1 = synth.box_room_depth(cam, base @ poses[1]).to(dev)
K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32, device=dev)
vp0, np0 = hicp.build_pyramids(d0, K, 3)
cos_thr = math.cos(math.radians(20.0))
def one():
    vp1, np1 = hicp.build_pyramids(d1, K, 3)
    return vp1, np1, hicp.icp_track(vp1, np1, vp0, np0, K, [0.25, 0.5, 1.0], [5, 5, 5], 0.1, cos_thr, 1e-4)
for _ in range(10):
    one()
torch.cuda.synchronize()
tp = tt = 0.0
for _ in range(iters):
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    vp1, np1 = hicp.build_pyramids(d1, K, 3)
    e1.record()
    out = hicp.icp_track(vp1, np1, vp0, np0, K, [0.25, 0.5, 1.0], [5, 5, 5], 0.1, cos_thr, 1e-4)
    e2.record()
    torch.cuda.synchronize()
    tp += e0.elapsed_time(e1) / iters
    tt += e1.elapsed_time(e2) / iters
print(f"persistent={os.environ.get('RTGS_ICP_PERSISTENT', '0 (default)')} {cam.H}x{cam.W}: pyramids {tp * 1e3:.0f} us, track {tt * 1e3:.0f} us, stats {out[16:].tolist()}")

if os.environ.get("RTGS_ICP_DEBUG_TIMING"):
    import numpy as np
    sc = hicp._get_scratch(dev)
    nbytes = sc.numel()
    dbg = sc[nbytes - 96 * 8:].view(torch.int64).cpu().numpy()[:75].reshape(15, 5).astype(np.float64) * 0.01   # 100 MHz -> us
    print("per iteration (us): compute | barrier | sum_partials | gn_update")
    for e in range(15):
        t = dbg[e]
        print(f"  it {e:2d}: {t[1] - t[0]:6.1f} | {t[2] - t[1]:6.1f} | {t[3] - t[2]:6.1f} | {t[4] - t[3]:6.1f}   (since start {t[0] - dbg[0][0]:7.1f})")
