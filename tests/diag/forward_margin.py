import sys, torch
sys.path.insert(0,'.')
from rtg_slam_amd import synth
from tests import raster_util as ru
SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
for rep in range(3):
    g, s = ru.make_scene(2000, SMALL, seed=3, pose_seed=11)
    out_o, _, aux = ru.oracle_run(s, g)
    out_h, _ = ru.hip_run(s, g)
    names = ["color", "depth", "color_index", "depth_index", "color_weight", "depth_weight", "T"]
    res={}
    for k in (0,1,4,5,6): res[names[k]] = ru.frac_bad(out_h[k], out_o[k], 1e-4)
    for k in (2,3): res[names[k]] = float((out_h[k] != out_o[k]).float().mean())
    print(rep, {k: round(v,5) for k,v in res.items()})
