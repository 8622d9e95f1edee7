"""One cold-process forward of the 2000-Gaussian 64x96 scene against cached oracle outputs (run many times from a shell
loop: hunts first-launch-only faults).  python tests/diag/cold_forward.py <cache.pt>"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtg_slam_amd import synth
from tests import raster_util as ru
SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
cache = sys.argv[1]
g, s = ru.make_scene(2000, SMALL, seed=3, pose_seed=11)
if not os.path.exists(cache):
    out_o, _, _ = ru.oracle_run(s, g)
    torch.save([o.clone() for o in out_o], cache)
out_o = torch.load(cache)
# the two smaller scenes first, as the test file does
for N, seed, pose, cam in ((300, 1, None, SMALL), (500, 2, 7, synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0))):
    g2, s2 = ru.make_scene(N, cam, seed=seed, pose_seed=pose)
    ru.hip_run(s2, g2)
out_h, _ = ru.hip_run(s, g)
bad = max(ru.frac_bad(out_h[k], out_o[k], 1e-4) for k in (0, 1, 4, 5, 6))
badi = max(float((out_h[k] != out_o[k]).float().mean()) for k in (2, 3))
print("bad", round(bad, 5), round(badi, 5))
