"""Host-side profile of the SLAM sequence (cProfile around bench.sequence_leg): where the interpreter's time goes.
    python tools/seq_hostprof.py [frames]   -> top functions by own time and by cumulative time, per frame"""
import cProfile, pstats, sys, os, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from rtg_slam_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 600
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
bench.sequence_leg(synth.REPLICA, dev, 60)                      # warm: builds, plans, allocator
pr = cProfile.Profile()
pr.enable()
out = bench.sequence_leg(synth.REPLICA, dev, n)
pr.disable()
print({k: out[k] for k in ("frames", "fps", "mapping_ms_mean_optimised_frames", "mapping_ms_mean_other_frames")})
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(45)
    print(s.getvalue())
