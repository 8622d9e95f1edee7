"""Where the SLAM sequence's device memory goes:  python tools/seq_memory.py [frames]
Prints live / peak device memory every 50 frames, what a garbage collection frees, and the largest live tensors at the end."""
import gc, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import mapping as mp, slam, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda", 0)
cam = synth.REPLICA
args = mp.replica_args(seed=1)
poses = synth.room_tour(n, seed=21)
def stream():
    for c2w in poses:
        d = synth.box_room_depth(cam, c2w, device=dev)
        c = synth.box_room_color(cam, c2w, d)
        torch.cuda.synchronize(dev)
        yield d.reshape(cam.H, cam.W), c, c2w.numpy()
MB = 2 ** 20
def on_frame(fid, frame, fm, mapper, tracker):
    if fid % 50 == 49:
        a = torch.cuda.memory_allocated(dev) / MB
        gc.collect()
        b = torch.cuda.memory_allocated(dev) / MB
        print(f"frame {fid + 1}: live {a:.0f} MB, after gc.collect {b:.0f} MB, peak {torch.cuda.max_memory_allocated(dev) / MB:.0f} MB, keyframes {mapper.get_keyframe_num}, N {mapper.opt.N}", flush=True)
mapper, tracker, rep = slam.run_sequence(cam, stream(), args, dev, capacity=800_000, on_frame=on_frame)
print("fps", rep["fps"], "peak MB", torch.cuda.max_memory_allocated(dev) / MB)
sizes = {}
for o in gc.get_objects():
    try:
        if torch.is_tensor(o) and o.is_cuda:
            st = o.untyped_storage()
            sizes[st.data_ptr()] = max(sizes.get(st.data_ptr(), 0), st.nbytes())
    except Exception:
        pass
tot = sum(sizes.values()) / MB
big = sorted(sizes.values(), reverse=True)[:12]
print(f"live storages {len(sizes)}, {tot:.0f} MB; largest (MB):", [round(b / MB, 1) for b in big])
km = mapper.keymap_list[-1]
print("a keyframe's map:", {k: (tuple(v.shape), str(v.dtype).replace('torch.', '')) for k, v in km.items() if torch.is_tensor(v)})
