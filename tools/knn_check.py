"""rtgs_knn3_query at SLAM sizes against brute force, timed:  python tools/knn_check.py [Nr] [Nq]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import slam_ops as so, synth
Nr = int(sys.argv[1]) if len(sys.argv) > 1 else 290_000
Nq = int(sys.argv[2]) if len(sys.argv) > 2 else 500
dev = torch.device("cuda", 0)
g = synth.surface_gaussians(Nr, synth.REPLICA, seed=7)["xyz"].to(dev)
gen = torch.Generator().manual_seed(3)
for case in ("cluster", "spread"):
    if case == "cluster":      # new points of one frame: a patch of the wall
        c = g[12345]
        sel = ((g - c).norm(dim=1) < 0.4).nonzero().reshape(-1)
        q = g[sel[torch.randperm(sel.numel(), generator=gen)[:Nq].to(dev)]] + 0.003 * torch.randn(min(Nq, sel.numel()), 3, generator=gen).to(dev)
    else:
        q = g[torch.randperm(Nr, generator=gen)[:Nq].to(dev)] + 0.003 * torch.randn(Nq, 3, generator=gen).to(dev)
    for with_self in (False, True):
        ref = torch.cat([q, g]) if with_self else g
        lo, hi = q.min(0)[0] - 0.05, q.max(0)[0] + 0.05
        box = torch.cat([lo, hi])
        d2, idx = so.knn_query(ref, q, 0 if with_self else -1, box)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            d2, idx = so.knn_query(ref, q, 0 if with_self else -1, box)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 100
        inb = ((ref > lo) & (ref < hi)).all(dim=1)
        D = torch.cdist(q.double(), ref.double()) ** 2
        D[:, ~inb] = float("inf")
        if with_self:
            D[torch.arange(q.shape[0]), torch.arange(q.shape[0])] = float("inf")
        bd, bi = D.topk(3, dim=1, largest=False)
        ok_d = torch.allclose(d2.double(), bd, rtol=1e-5, atol=1e-9)
        same = (idx.long() == bi).float().mean().item()
        print(f"{case} self={with_self} Nr={ref.shape[0]} Nq={q.shape[0]}: {ms:.3f} ms per call (build + query), distances equal {ok_d}, indices equal {same:.4f}")

print("--- more shapes (a SLAM frame's two calls) ---")
def brute(ref, q, self_off, lo, hi):
    inb = ((ref > lo) & (ref < hi)).all(dim=1)
    D = torch.cdist(q.double(), ref.double()) ** 2
    D[:, ~inb] = float("inf")
    if self_off >= 0:
        D[torch.arange(q.shape[0]), self_off + torch.arange(q.shape[0])] = float("inf")
    k = min(3, ref.shape[0])
    bd, bi = D.topk(k, dim=1, largest=False)
    return bd, bi
for (nr, nq, self_) in ((40800, 40800, True), (3000, 2000, False), (22, 600, False), (290000, 6, True), (5, 3000, False), (100000, 4000, True), (700, 64, False), (2, 40, False)):
    pts = synth.surface_gaussians(max(nr, nq) + 10, synth.REPLICA, seed=11)["xyz"].to(dev)
    perm = torch.randperm(pts.shape[0], generator=gen).to(dev)
    q = pts[perm[:nq]] + (0.002 * torch.randn(nq, 3, generator=gen)).to(dev)
    if self_:
        rest = pts[perm[:max(nr - nq, 0)]]
        ref = torch.cat([q, rest])
    else:
        ref = pts[perm[-nr:]]
    lo, hi = q.min(0)[0] - 0.05, q.max(0)[0] + 0.05
    d2, idx = so.knn_query(ref, q, 0 if self_ else -1, torch.cat([lo, hi]))
    torch.cuda.synchronize()
    bd, bi = brute(ref, q, 0 if self_ else -1, lo, hi)
    k = bd.shape[1]
    fin = torch.isfinite(bd)
    got = d2[:, :k].double()
    ok = bool(((got - bd).abs()[fin] <= 1e-5 * bd[fin] + 1e-12).all()) and bool((idx[:, :k][~fin] == -1).all())
    print(f"Nr={ref.shape[0]} Nq={nq} self={self_}: distances equal {ok}; missing {(~fin).sum().item()}")
