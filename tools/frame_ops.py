"""Which tensor operations an ordinary (not optimised) SLAM frame issues from Python, and from where: a TorchDispatchMode
around Mapping.mapping + get_render_output + update_last_status of frames 31-35 of a sequence, grouped by the innermost
rtg_slam_amd source line.     python tools/frame_ops.py"""
import collections, os, sys, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from torch.utils._python_dispatch import TorchDispatchMode
from rtg_slam_amd import mapping as mp, slam, synth

dev = torch.device("cuda:0")
cam = synth.REPLICA
args = mp.replica_args(seed=1)
poses = synth.room_tour(60, seed=21)
hist = collections.Counter()
frames = [0]


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        where = "?"
        for fs in reversed(traceback.extract_stack(limit=14)):
            if "rtg_slam_amd" in fs.filename:
                where = "%s:%d %s" % (os.path.basename(fs.filename), fs.lineno, fs.name)
                break
        hist[(where, str(func))] += 1
        return func(*args, **(kwargs or {}))


def stream():
    for c2w in poses:
        d = synth.box_room_depth(cam, c2w, device=dev)
        c = synth.box_room_color(cam, c2w, d)
        yield d.reshape(cam.H, cam.W), c, c2w.numpy()


orig = mp.Mapping.mapping
mode = Log()
state = {"on": False}


def wrapped(self, frame, frame_map, frame_id, *a, **k):
    if 30 <= frame_id < 35 and frame_id % 6 != 5:
        frames[0] += 1
        state["on"] = True
        mode.__enter__()
    return orig(self, frame, frame_map, frame_id, *a, **k)


def on_frame(fid, *a):
    if state["on"]:
        mode.__exit__(None, None, None)
        state["on"] = False


mp.Mapping.mapping = wrapped
slam.run_sequence(cam, stream(), args, dev, capacity=800_000, on_frame=on_frame)
n = frames[0]
print("frames logged:", n, " tensor ops per frame:", sum(hist.values()) / n)
by_site = collections.Counter()
for (w, f), c in hist.items():
    by_site[w] += c
for w, c in by_site.most_common(60):
    ops = ", ".join("%s x%.1f" % (f.replace("aten.", ""), k / n) for (ww, f), k in sorted(hist.items(), key=lambda t: -t[1]) if ww == w)
    print("%6.1f  %-46s %s" % (c / n, w, ops[:200]))
