"""Determinism / uninitialised-memory stress of the forward (and backward): the same scene rendered many times while the
caching allocator's free blocks are poisoned with random bits between calls; every output must equal the first call's.
    python tools/stress_forward.py [iters]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import synth
from tests import raster_util as ru

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = "cuda:0"
SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
cases = [(SMALL, 2000, 3, 11, False), (SMALL, 300, 1, None, True), (synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0), 500, 2, 7, True)]
bad = 0
for cam, N, seed, pose, with_bwd in cases:
    g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose)
    gen = torch.Generator().manual_seed(seed)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen)) if with_bwd else None
    ref = None
    for it in range(iters):
        junk = [torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev) for n in (1 << 12, 1 << 16, 1 << 20, 3 << 20)]
        junk.append(torch.full((1 << 18,), float("nan"), device=dev))
        del junk                                   # back to the caching allocator, contents intact
        out, gd = ru.hip_run(s, g, grads=grads, dev=dev)
        if ref is None:
            ref = (out, gd)
            continue
        for k, (a, b) in enumerate(zip(out, ref[0])):
            if not torch.equal(a, b):
                bad += 1
                print(f"case N={N}: iteration {it}: output {k} differs in {int((a != b).sum())} places, max |d| = {float((a.float() - b.float()).abs().max()):.3e}")
        if gd is not None:
            for k in ru.FIELDS:
                sc = float(ref[1][k].abs().max()) + 1e-12
                err = float((gd[k] - ref[1][k]).abs().max()) / sc
                if not (err < 1e-4):
                    bad += 1
                    print(f"case N={N}: iteration {it}: grad {k} rel err {err:.3e}")
print("stress done, mismatches:", bad)
