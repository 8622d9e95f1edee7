"""Timing of rtgs_knn3_query at the sizes the SLAM sequence calls it with (HIP events, median of 20)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from rtg_slam_amd import synth, slam_ops as ops
dev = torch.device("cuda", 0)
cam = synth.REPLICA
def t(fn, n=20):
    for _ in range(3): fn()
    ms = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]
for Nr in (6000, 40800, 100000, 170000):
    ref = synth.surface_gaussians(Nr, cam, seed=3)["xyz"].to(dev)
    for Nq in (3500, 40800):
        g = torch.Generator().manual_seed(1)
        q = ref[torch.randperm(Nr, generator=g)[:min(Nq, Nr)].to(dev)] + 0.01 * torch.randn(min(Nq, Nr), 3, generator=g).to(dev)
        box = torch.cat([q.min(0).values - 0.05, q.max(0).values + 0.05])
        total = torch.cat([q, ref])
        a = t(lambda: ops.knn_query(ref, q, -1, box))
        b = t(lambda: ops.knn_query(total, q, 0, box))
        c = t(lambda: ops.distCUDA2(total))
        print(f"refs {Nr:7d} queries {q.shape[0]:6d}: filter-form {a*1e3:7.0f} us   update-form (refs = queries + map) {b*1e3:7.0f} us   distCUDA2(all) {c*1e3:7.0f} us")
