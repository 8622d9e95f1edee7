"""world_size-2 gloo test of the sharded map-optimisation step (rtg_slam_amd/map_optim.py): every
rank renders its own view (CPU oracle injected as the render function), gradients are summed across
ranks, Adam runs on the rank's row shard, updated rows are gathered.  Checked against a single
process that sums both views' gradients and steps all rows."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests import torch_doubles as td  # noqa: E402


def _scene():
    from rtg_slam_amd import synth
    from rtg_slam_amd import map_optim as mo
    from oracle import raster_oracle as ro
    cam = synth.CameraSpec(32, 48, 40.0, 40.0, 23.5, 15.5)
    g = synth.random_gaussians(101, cam, seed=3)        # odd count: exercises the shard padding
    packed = mo.pack_from_activated(g)
    views = []
    for r in range(2):
        c2w = synth.look_at_pose(seed=50 + r, max_angle_deg=3.0, max_trans=0.05)
        view = torch.linalg.inv(c2w).float().t().contiguous()
        views.append(ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, viewmatrix=view))
    gen = torch.Generator().manual_seed(5)
    gts = [(torch.rand(3, cam.H, cam.W, generator=gen), 1.0 + torch.rand(1, cam.H, cam.W, generator=gen)) for _ in range(2)]
    return packed, views, gts


def _loss_fn(settings, gt):
    from oracle import raster_oracle as ro
    from rtg_slam_amd import map_optim as mo

    def fn(gd):
        out = ro.rasterize(settings, gd["xyz"], gd["opacity"], gd["shs"], gd["scales"], gd["rotations"], gd["normal"])
        return td.slam_losses(out, gt[0], gt[1])
    return fn


class _Done:
    def wait(self):
        return True


def _emulate_rccl_collectives():
    """reduce_scatter_tensor / all_gather_into_tensor (async) on top of gloo primitives, so the RCCL branch of
    ShardedMapOptimizer.step (queue all reduce-scatters, Adam under the big one, async all-gathers) runs on CPU."""
    def reduce_scatter_tensor(out, inp, op=None, group=None, async_op=False):
        tmp = inp.clone()
        dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=group)
        per = out.shape[0]
        r = dist.get_rank(group)
        out.copy_(tmp[r * per:(r + 1) * per])
        return _Done()

    def all_gather_into_tensor(out, inp, group=None, async_op=False):
        parts = [torch.empty_like(inp) for _ in range(dist.get_world_size(group))]
        dist.all_gather(parts, inp.clone(), group=group)
        out.copy_(torch.cat(parts, dim=0))
        return _Done()
    dist.reduce_scatter_tensor = reduce_scatter_tensor
    dist.all_gather_into_tensor = all_gather_into_tensor


def _worker(rank, world, port, ret, rccl_branch=False, n_frozen=0):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    packed, views, gts = _scene()
    opt = mo.ShardedMapOptimizer(packed, adam_fn=adam_reference, activate_fn=td.activate8, n_frozen=n_frozen)
    if rccl_branch:
        _emulate_rccl_collectives()
        opt.backend = "nccl"
    assert opt.world == 2 and opt.per == (101 - n_frozen + 1) // 2 and opt.Npad == n_frozen + 2 * opt.per
    for _ in range(2):
        opt.step(_loss_fn(views[rank], gts[rank]))
    ret[rank] = opt.params.clone()
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("rccl_branch,n_frozen", [(False, 0), (True, 0), (False, 30), (True, 31)])
def test_sharded_step_matches_single_process(rccl_branch, n_frozen):
    """n_frozen > 0: the first rows are rendered but are no parameters (the stable part of an RTG-SLAM map,
    mapper.py:1026-1108) - the shards partition the trainable rows only, the frozen rows stay bitwise untouched."""
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    port = 29600 + (os.getpid() % 300) + (7 if rccl_branch else 0) + n_frozen
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, port, ret, rccl_branch, n_frozen), nprocs=2, join=True)
    p0, p1 = ret[0], ret[1]
    assert torch.equal(p0, p1), "all ranks hold the same gathered parameters"
    # single-process reference: sum of both views' gradients, Adam on all rows
    packed, views, gts = _scene()
    p = packed.clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr = mo.default_lr_columns()
    for step in (1, 2):
        leaf = p.detach().clone().requires_grad_(True)
        loss = _loss_fn(views[0], gts[0])(td.activate(leaf)) + _loss_fn(views[1], gts[1])(td.activate(leaf))
        (g,) = torch.autograd.grad(loss, leaf)
        adam_reference(p[n_frozen:], g[n_frozen:], m[n_frozen:], v[n_frozen:], lr, step, 1e-15)
    assert torch.equal(p0[:n_frozen], packed[:n_frozen])
    assert float((p0 - p).abs().max()) < 1e-5
    assert float((p0 - packed).abs().max()) > 1e-4, "the step moved the parameters"


def _grow_worker(rank, world, port, ret, capacity=120):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    packed, views, gts = _scene()
    opt = mo.ShardedMapOptimizer(packed[:90].clone(), adam_fn=adam_reference, activate_fn=td.activate8, n_frozen=20, capacity=capacity)
    opt.step(_loss_fn(views[rank], gts[rank]))
    opt.append_rows(packed[90:].clone())             # 11 new trainable rows on every rank: the shards re-partition
    assert opt.N == 101 and opt.per == (101 - 20 + 1) // 2
    opt.step(_loss_fn(views[rank], gts[rank]))
    mask = torch.zeros(101, dtype=torch.bool)
    mask[25:40] = True
    opt.freeze_rows(mask)                            # 15 trainable rows join the frozen prefix
    assert opt.n_frozen == 35
    opt.step(_loss_fn(views[rank], gts[rank]))
    ret[rank] = opt.params.clone()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("capacity", [120, 92])
def test_a_map_that_grows_and_freezes_between_sharded_steps_matches_one_process(capacity):
    """append_rows / freeze_rows on two ranks (every rank applies the same change): the row shards, and with them the rows
    a rank's Adam state belongs to, change - the result must be what ONE process gets from the same sequence with the
    sum of both views' losses - whether the append fits the capacity or forces a re-allocation."""
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    port = 29950 + (os.getpid() % 40) + (capacity % 7)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_grow_worker, args=(2, port, ret, capacity), nprocs=2, join=True)
    assert torch.equal(ret[0], ret[1])
    packed, views, gts = _scene()

    def both(gd):
        return _loss_fn(views[0], gts[0])(gd) + _loss_fn(views[1], gts[1])(gd)
    one = mo.ShardedMapOptimizer(packed[:90].clone(), adam_fn=adam_reference, activate_fn=td.activate8, n_frozen=20, capacity=120)
    one.step(both)
    one.append_rows(packed[90:].clone())
    one.step(both)
    mask = torch.zeros(101, dtype=torch.bool)
    mask[25:40] = True
    one.freeze_rows(mask)
    one.step(both)
    assert float((ret[0] - one.params).abs().max()) < 1e-5


def test_shard_rows_partition():
    from rtg_slam_amd import map_optim as mo
    for N in (1, 7, 8, 1_200_000, 5_000_001):
        for w in (1, 2, 4, 8):
            per, npad = mo.shard_rows(N, w)
            assert per * w == npad and npad >= N and npad - N < w


def test_adam_leaves_never_touched_rows_bitwise_unchanged():
    """The invariant the row-skipping HIP Adam relies on (rtgs_fused_adam_rows): a row whose gradient is zero and whose
    moments never left zero is a fixed point of the dense update, whatever the step count and eps."""
    from tests.dist_util import adam_reference
    from rtg_slam_amd import map_optim as mo
    gen = torch.Generator().manual_seed(0)
    p = torch.randn(64, 59, generator=gen)
    g = torch.randn(64, 59, generator=gen)
    live = torch.rand(64, generator=gen) < 0.3
    g[~live] = 0.0
    g[3] = -0.0                                              # negative zero is a zero gradient too
    live[3] = False
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p0 = p.clone()
    lr = mo.default_lr_columns() + 1e-3
    for step in range(1, 6):
        adam_reference(p, g, m, v, lr, step, 1e-15)
    assert torch.equal(p[~live], p0[~live])
    assert not m[~live].any() and not v[~live].any()
    assert (p[live] != p0[live]).any()


def test_tile_band_partition_of_the_switched_on_tiles():
    """ShardedMapOptimizer.band_tile_mask (pure torch): for any world size the ranks' bands are disjoint, cover exactly
    the switched-on tiles, are balanced to within one tile, and the REGIONS partition every tile, on or off."""
    import types
    from rtg_slam_amd import map_optim as mo
    g = torch.Generator().manual_seed(3)
    for world in (1, 2, 3, 8):
        for frac in (1.0, 0.3, 0.02):
            tm = (torch.rand(43, 75, generator=g) < frac).int()
            bands, regions = [], []
            for r in range(world):
                me = types.SimpleNamespace(world=world, rank=r)
                b, reg = mo.ShardedMapOptimizer.band_tile_mask(me, tm, with_region=True)
                bands.append(b); regions.append(reg)
            tot = torch.stack(bands).sum(0)
            assert torch.equal(tot, tm)
            assert torch.equal(torch.stack(regions).int().sum(0), torch.ones_like(tm))
            counts = [int(b.sum()) for b in bands]
            assert max(counts) - min(counts) <= 1, (world, frac, counts)


def _global_worker(rank, world, port, ret, rccl_branch, nf):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    packed, views, gts = _scene()
    opt = mo.ShardedMapOptimizer(packed, adam_fn=adam_reference, activate_fn=td.activate8, n_frozen=nf)
    if rccl_branch:
        _emulate_rccl_collectives()
        opt.backend = "nccl"
    opt.step(_loss_fn(views[rank], gts[rank]))                       # a local step: the unstable suffix moves
    mid = opt.params.clone()
    opt.begin_global_optimization(mo.global_lr_scale(final=False))
    assert opt.per == (nf + 1) // 2 and opt.my_rows().start == rank * opt.per
    for _ in range(2):
        opt.step(_loss_fn(views[rank], gts[rank]))                   # the loss renders the stable rows only
    opt.end_global_optimization()
    ret[rank] = (mid, opt.params.clone())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("rccl_branch,nf", [(False, 60), (True, 61)])
def test_global_optimization_on_two_ranks_matches_one_process(rccl_branch, nf):
    """Mapping.global_optimization (mapper.py:594-707) with the stable rows sharded over two ranks: equals one process that
    renders the stable prefix from both views, sums the gradients and steps every stable row with the rescaled learning
    rates; the unstable suffix - which includes the rows the shard padding reaches into - is bit-unchanged."""
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    port = 29950 + (os.getpid() % 300) + (11 if rccl_branch else 0)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_global_worker, args=(2, port, ret, rccl_branch, nf), nprocs=2, join=True)
    (mid0, p0), (mid1, p1) = ret[0], ret[1]
    assert torch.equal(mid0, mid1) and torch.equal(p0, p1)
    assert torch.equal(p0[nf:], mid0[nf:]), "unstable rows are neither rendered nor stepped"
    packed, views, gts = _scene()
    p = mid0[:nf].clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    lr = mo.default_lr_columns() * mo.global_lr_scale(final=False)
    for step in (1, 2):
        leaf = p.detach().clone().requires_grad_(True)
        loss = _loss_fn(views[0], gts[0])(td.activate(leaf)) + _loss_fn(views[1], gts[1])(td.activate(leaf))
        (g,) = torch.autograd.grad(loss, leaf)
        adam_reference(p, g, m, v, lr, step, 1e-15)
    assert float((p0[:nf] - p).abs().max()) < 1e-5
    assert torch.equal(p0[:nf, :3], mid0[:nf, :3]) and float((p0[:nf] - mid0[:nf]).abs().max()) > 1e-5
