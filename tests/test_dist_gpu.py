"""Two ranks sharing ONE GPU over gloo: the sharded optimisation step with the real HIP rasterizer,
fused activations and fused Adam (RCCL refuses two ranks per device, so the collective backend here
is gloo on device tensors - the rank/shard logic is identical)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(dev):
    import math
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from tests import torch_doubles as td
    from rtg_slam_amd import synth
    from rtg_slam_amd import map_optim as mo
    from rtg_slam_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
    g = synth.random_gaussians(2001, cam, seed=3)
    packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
    fns = []
    gen = torch.Generator().manual_seed(5)
    for r in range(2):
        c2w = synth.look_at_pose(seed=50 + r, max_angle_deg=3.0, max_trans=0.05)
        view = torch.linalg.inv(c2w).float().t().contiguous().to(dev)
        rs = GaussianRasterizationSettings(cam.H, cam.W, cam.W / (2 * cam.fx), cam.H / (2 * cam.fy),
                                           torch.zeros(3, device=dev), 1.0, view, view, 3, c2w[:3, 3].float().to(dev), 0.6,
                                           1.0, math.cos(math.radians(60)), 3.0, False, False, cam.cx, cam.cy, 1e-4)
        rast = GaussianRasterizer(raster_settings=rs)
        gt_c = torch.rand(3, cam.H, cam.W, generator=gen).to(dev)
        gt_d = (1.0 + torch.rand(1, cam.H, cam.W, generator=gen)).to(dev)

        def fn(gd, rast=rast, gt_c=gt_c, gt_d=gt_d):
            out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None, scales=gd["scales"],
                       rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"], tile_mask=None)
            return td.slam_losses(out, gt_c, gt_d)
        fn.spec = (rs, gt_c, gt_d)
        fns.append(fn)
    return packed, fns


def _init(rank, world):
    """RTGS_TEST_BACKEND=nccl: one rank per GPU over RCCL (needs >= `world` GPUs); default: gloo, both ranks on GPU 0."""
    backend = os.environ.get("RTGS_TEST_BACKEND", "gloo")
    if backend == "nccl":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        return torch.device("cuda", rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return torch.device("cuda", 0)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = _init(rank, world)
    from rtg_slam_amd import map_optim as mo
    packed, fns = _setup(dev)
    opt = mo.ShardedMapOptimizer(packed, n_frozen=NF())          # HIP activations + HIP Adam, gloo (or RCCL) collectives
    for _ in range(2):
        opt.step(fns[rank])
    ret[rank] = opt.params.cpu()
    dist.barrier()
    dist.destroy_process_group()


def NF():
    """Frozen prefix of the map in the workers and the single-process reference (set by the parametrised tests)."""
    return int(os.environ.get("RTGS_TEST_NFROZEN", "0"))


@pytest.mark.parametrize("nf", [0, 701])
def test_two_ranks_one_gpu_match_single_process(nf, monkeypatch):
    """nf > 0: rows [0, nf) are frozen (rendered, not parameters): the row shards partition the trainable rows only."""
    monkeypatch.setenv("RTGS_TEST_NFROZEN", str(nf))
    sys.path.insert(0, ROOT)
    from rtg_slam_amd import map_optim as mo
    port = 29700 + (os.getpid() % 200) + (nf % 7)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert torch.equal(ret[0], ret[1])
    dev = torch.device("cuda", 0)
    packed, fns = _setup(dev)
    ref = mo.ShardedMapOptimizer(packed, n_frozen=nf)          # world 1: both views summed in one process
    for _ in range(2):
        ref.step(lambda gd: fns[0](gd) + fns[1](gd))
    d = float((ret[0] - ref.params.cpu()).abs().max())
    assert d < 2e-5, d
    assert torch.equal(ret[0][:nf], packed[:nf].cpu())
    assert float((ret[0] - packed.cpu()).abs().max()) > 1e-4


def _worker_slam(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = _init(rank, world)
    from rtg_slam_amd import map_optim as mo
    packed, fns = _setup(dev)
    opt = mo.ShardedMapOptimizer(packed, n_frozen=NF())
    opt._row_capacity, opt._shrink_every = 1 << 16, 1              # far too large: must shrink (same steps on every rank)
    rs, gt_c, gt_d = fns[rank].spec
    losses = []
    for _ in range(3):
        losses.append(float(opt.step_slam(rs, gt_c, gt_d)))        # replicated map, sparse row exchange
    opt.flush()
    ret[rank] = (opt.params.cpu(), losses)
    ret[f"cap{rank}"] = opt._row_capacity
    try:
        opt.step(fns[rank])
        ret[f"mixed{rank}"] = "no error"
    except RuntimeError as e:
        ret[f"mixed{rank}"] = str(e)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nf", [0, 701])
def test_two_ranks_sparse_slam_step_matches_single_process(nf, monkeypatch):
    """step_slam with two ranks: only the gradient rows that exist are exchanged, every rank takes the same Adam
    step.  Replicas bit-identical; result equal to one process optimising the sum of both views.  nf > 0: a frozen
    prefix - its rows never enter the exchange and stay bitwise untouched."""
    monkeypatch.setenv("RTGS_TEST_NFROZEN", str(nf))
    sys.path.insert(0, ROOT)
    from rtg_slam_amd import map_optim as mo
    port = 29300 + (os.getpid() % 200) + (nf % 7)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_slam, args=(2, port, ret), nprocs=2, join=True)
    (p0, l0), (p1, l1) = ret[0], ret[1]
    assert torch.equal(p0, p1)
    assert ret["cap0"] == ret["cap1"] < (1 << 16)             # a mostly empty exchange buffer halves itself, in step
    assert "different Adam state" in ret["mixed0"]            # step() after step_slam() on > 1 rank is refused
    dev = torch.device("cuda", 0)
    packed, fns = _setup(dev)
    ref = mo.ShardedMapOptimizer(packed, n_frozen=nf)
    for _ in range(3):
        ref.step(lambda gd: fns[0](gd) + fns[1](gd))
    rp = ref.params.cpu()
    assert torch.equal(p0[:nf], packed[:nf].cpu())
    bad = float(((p0 - rp).abs() > 2e-5).float().mean())
    assert bad < 2e-3, bad                                     # Adam's +-lr first steps: a gradient ~0 may flip
    moved = (p0 - packed.cpu()).abs().max(dim=1).values > 0
    assert 0 < int(moved.sum()) < packed.shape[0]
    assert torch.equal(moved, (rp - packed.cpu()).abs().max(dim=1).values > 0)


def _worker_slam_grow(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = _init(rank, world)
    from rtg_slam_amd import map_optim as mo
    packed, fns = _setup(dev)
    opt = mo.ShardedMapOptimizer(packed[:1500].clone(), n_frozen=300, capacity=1600)
    rs, gt_c, gt_d = fns[rank].spec
    for _ in range(2):
        opt.step_slam(rs, gt_c, gt_d)
    opt.append_rows(packed[1500:].clone())             # 501 rows: beyond the capacity - every per-row array is re-allocated
    assert opt.N == 2001 and opt.capacity >= 2001
    for _ in range(2):
        opt.step_slam(rs, gt_c, gt_d)
    opt.flush()
    ret[rank] = opt.params.cpu()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_sparse_slam_step_with_a_map_that_outgrows_its_capacity():
    """The replicated form's full-size Adam state follows the map through a re-allocation (rows keep their moments, as
    on one rank): two ranks stepping, appending beyond the capacity and stepping again equal one process doing the same
    with the sum of both views."""
    sys.path.insert(0, ROOT)
    from rtg_slam_amd import map_optim as mo
    port = 29520 + (os.getpid() % 100)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_slam_grow, args=(2, port, ret), nprocs=2, join=True)
    assert torch.equal(ret[0], ret[1])
    dev = torch.device("cuda", 0)
    packed, fns = _setup(dev)
    ref = mo.ShardedMapOptimizer(packed[:1500].clone(), n_frozen=300, capacity=1600)
    both = lambda gd: fns[0](gd) + fns[1](gd)
    for _ in range(2):
        ref.step(both)
    ref.append_rows(packed[1500:].clone())
    for _ in range(2):
        ref.step(both)
    rp = ref.params.cpu()
    assert torch.equal(ret[0][:300], packed[:300].cpu())
    bad = float(((ret[0] - rp).abs() > 2e-5).float().mean())
    assert bad < 2e-3, bad


def _gt_normal(H, W, dev):
    n = torch.randn(H, W, 3, generator=torch.Generator().manual_seed(12))
    n = n / n.norm(dim=-1, keepdim=True)
    n[:7] = 0                                                           # rows without a frame normal: not in the mean
    return n.to(dev)


def _worker_band(rank, world, port, ret, nw=0.0):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = _init(rank, world)
    from rtg_slam_amd import map_optim as mo
    packed, fns = _setup(dev)
    rs, gt_c, gt_d = fns[0].spec                                    # ONE view, split into tile bands across the ranks
    H, W = gt_c.shape[1:]
    rm = (torch.rand(H, W, generator=torch.Generator().manual_seed(9)) < 0.8).to(dev)
    tm = torch.ones((H + 15) // 16, (W + 15) // 16, dtype=torch.int32, device=dev)
    tm[0, 0] = 0
    # a DIFFERENT bool render mask every other step (the reference optimises a random frame of its window per
    # iteration, mapper.py:176-183): each becomes a fresh uint8 temporary inside step_slam, which the caching allocator
    # may hand the same address - the band's loss mask must follow the content, not the pointer
    rm2 = (torch.rand(H, W, generator=torch.Generator().manual_seed(10)) < 0.5).to(dev)
    opt = mo.ShardedMapOptimizer(packed)
    opt._row_capacity = 32                                          # far too small: the first exchange must overflow
    opt.begin_local_optimization()
    losses = []
    for i in range(4):
        losses.append(float(opt.step_slam(rs, gt_c, gt_d, tm, render_mask=(rm2 if i % 2 else rm), tile_band=True,
                                          normal_weight=nw, gt_normal=_gt_normal(H, W, dev) if nw > 0 else None)))
    band = opt.band_tile_mask(tm).cpu()
    ret[rank] = (opt.params.cpu(), losses, band, opt.overflow_redos, opt._row_capacity)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nw", [0.0, 0.3])
def test_tile_band_split_of_one_view_and_overflow_redo(nw):
    """ONE view, each rank renders and differentiates its band of tiles, loss normalisers all-reduced,
    gradient rows summed by the sparse exchange - equal to a single process stepping on the whole view.  The exchange
    starts with a capacity that is too small: the overflow is detected on the device, nothing is applied, and the host
    repeats it with a larger capacity one call later (no synchronisation in the steady state)."""
    sys.path.insert(0, ROOT)
    from rtg_slam_amd import map_optim as mo
    port = 29500 + (os.getpid() % 200)
    ctx = mp.get_context("spawn")
    mgr = ctx.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_band, args=(2, port + (37 if nw > 0 else 0), ret, nw), nprocs=2, join=True)
    (p0, l0, b0, redo0, cap0), (p1, l1, b1, redo1, cap1) = ret[0], ret[1]
    assert torch.equal(p0, p1) and l0 == l1                          # replicas bit-identical, same (global) loss
    assert redo0 >= 1 and redo0 == redo1 and cap0 == cap1 > 32
    assert int((b0 & b1).sum()) == 0 and abs(int(b0.sum()) - int(b1.sum())) <= 1      # disjoint, balanced bands
    dev = torch.device("cuda", 0)
    packed, fns = _setup(dev)
    rs, gt_c, gt_d = fns[0].spec
    H, W = gt_c.shape[1:]
    rm = (torch.rand(H, W, generator=torch.Generator().manual_seed(9)) < 0.8).to(dev)
    tm = torch.ones((H + 15) // 16, (W + 15) // 16, dtype=torch.int32, device=dev)
    tm[0, 0] = 0
    assert int((b0 | b1).sum()) == int(tm.sum())
    rm2 = (torch.rand(H, W, generator=torch.Generator().manual_seed(10)) < 0.5).to(dev)
    ref = mo.ShardedMapOptimizer(packed)
    ref.begin_local_optimization()
    # nw > 0: the normal term's mean is over the pixels of the WHOLE view (mapper.py:433-442) - its two sums are all-reduced
    # across the bands like the image terms' normalisers (it used the rank-local count)
    lr = [float(ref.step_slam(rs, gt_c, gt_d, tm, render_mask=(rm2 if i % 2 else rm), normal_weight=nw,
                              gt_normal=_gt_normal(H, W, dev) if nw > 0 else None)) for i in range(4)]
    for a, b in zip(l0, lr):
        assert abs(a - b) <= 1e-5 * max(1.0, abs(b))
    rp = ref.params.cpu()
    assert float(((p0 - rp).abs() > 2e-5).float().mean()) < 2e-3
    moved = (p0 - packed.cpu()).abs().max(dim=1).values > 0
    assert torch.equal(moved, (rp - packed.cpu()).abs().max(dim=1).values > 0)


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="real RCCL needs two GPUs (the development lease has one)")
def test_the_three_multi_gpu_forms_on_real_rccl(monkeypatch):
    """The same three comparisons with backend "nccl" (= RCCL over xGMI), one rank per GPU: dense-sharded step,
    sparse row exchange, tile-band split with overflow redo.  Skipped on a 1-GPU box; the driver's multi-GPU tier
    runs it."""
    monkeypatch.setenv("RTGS_TEST_BACKEND", "nccl")
    test_two_ranks_one_gpu_match_single_process()
    test_two_ranks_sparse_slam_step_matches_single_process()
    test_tile_band_split_of_one_view_and_overflow_redo(0.0)
    test_tile_band_split_of_one_view_and_overflow_redo(0.3)
