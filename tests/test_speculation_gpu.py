"""The one-call map step without a host wait in the middle of the forward (RTGS_FWD_SPECULATE, include/rtgs_raster.h):
the host sizes the binning buffer and picks the sort classes from the last verified call and verifies afterwards.

* with speculation on and off the same sequence of steps gives the same losses and parameters (the guarded kernels are
  the plain kernels; gradient-slot order is the only run-to-run difference, as without speculation);
* when the guess does NOT hold - the tile mask jumps from a few tiles to all of them, the view changes, the near slice
  stops being declined - nothing persistent changes in the failed pass and the redo gives the plain result;
* the three pass structures the host can guess: plain single pass, slice declined + visible list, slice finishing
  every tile."""
import pytest
import torch

from rtg_slam_amd import map_optim as mo
from rtg_slam_amd import rasterizer as rz
from rtg_slam_amd import synth
from tests import raster_util as ru

pytestmark = pytest.mark.gpu

MID = synth.CameraSpec(272, 400, 300.0, 300.0, 199.5, 135.5)      # 17 x 25 = 425 tiles: the near slice is considered


def _run(g, cam, masks, poses, speculate, steps, lr=1e-4):
    dev = "cuda:0"
    ctx = rz.current_context()
    ctx.set_speculation(speculate)
    try:
        before = ctx.speculation_stats()
        packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
        opt = mo.ShardedMapOptimizer(packed, lr_col=mo.default_lr_columns() * lr)
        gen = torch.Generator().manual_seed(3)
        gt_c = torch.rand(3, cam.H, cam.W, generator=gen).to(dev)
        gt_d = (1.0 + torch.rand(1, cam.H, cam.W, generator=gen)).to(dev)
        losses = []
        opt.begin_local_optimization()
        for k in range(steps):
            _, s = ru.make_scene(8, cam, seed=1, pose_seed=poses[k % len(poses)])
            rs = ru.hip_settings(s, dev)
            m = masks[k % len(masks)]
            losses.append(float(opt.step_slam(rs, gt_c, gt_d, None if m is None else m.to(dev))))
        after = ctx.speculation_stats()
        stats = {k: after[k] - before[k] for k in after}
        return losses, opt.params.detach().cpu(), stats
    finally:
        ctx.set_speculation(True)


def _masks(cam, fracs):
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    out = []
    for i, f in enumerate(fracs):
        out.append(None if f >= 1.0 else (torch.rand(gy, gx, generator=torch.Generator().manual_seed(10 + i)) < f).int())
    return out


def _same(a, b):
    la, pa, _ = a
    lb, pb, _ = b
    for x, y in zip(la, lb):
        assert abs(x - y) <= 1e-5 * max(1.0, abs(x)), (la, lb)
    assert ru.frac_bad(pa, pb, 2e-6) < 1e-3


@pytest.mark.parametrize("kind", ["plain", "declined", "slice"])
def test_speculative_steps_equal_plain_steps(kind):
    cam = MID
    if kind == "plain":            # a small map: no near slice is considered
        g, _ = ru.make_scene(20_000, cam, seed=4)
    elif kind == "declined":       # a single-layer map: the kernels decline the slice, single pass over the visible list
        g = synth.surface_gaussians(150_000, cam, seed=5)
    else:                          # bench.py's headline scene: a depth-complex volume, the slice finishes every tile
        cam = synth.REPLICA
        g, _ = ru.make_scene(1_200_000, cam, seed=2024)
    masks, poses = _masks(cam, [1.0]), ([None] if kind == "slice" else [1, 1, 2, 2, 1])
    on = _run(g, cam, masks, poses, True, 8)
    off = _run(g, cam, masks, poses, False, 8)
    _same(on, off)
    assert off[2]["speculative"] == 0
    if kind == "slice":
        slice_stats = rz.current_context().last_slice_stats()
        assert slice_stats["used"] == 1 and slice_stats["tiles_left_to_pass2"] == 0, slice_stats   # else: not a guessable structure
    assert on[2]["speculative"] >= 5, on[2]        # the first call(s) build the history, the rest speculate


@pytest.mark.parametrize("onepass", [False, True])
def test_a_failed_guess_changes_nothing_and_is_redone(onepass):
    cam = MID
    g = synth.surface_gaussians(150_000, cam, seed=5)
    # three steps on a few tiles build the history (the first two cannot speculate: the slice's decision has to be
    # learnt first), then the instance total jumps by far more than the 12 % margin; the view changes too
    masks, poses = _masks(cam, [0.15, 0.15, 0.15, 1.0, 0.15, 1.0]), [1, 3, 5]
    ctx = rz.current_context()
    ctx.set_onepass(onepass)
    try:
        on = _run(g, cam, masks, poses, True, 12)
        off = _run(g, cam, masks, poses, False, 12)
    finally:
        ctx.set_onepass(True)
    _same(on, off)
    if not onepass:     # count + scan + scatter sizes ONE buffer by the guessed total: the jump to the full mask overruns it
        assert on[2]["failed"] >= 1, on[2]  # (afterwards the decaying maximum covers the loop)
    # one-pass placement has no bound on the total - every tile owns a segment - so the jump is not a wrong guess there;
    # what is one, is a tile list that outgrows its segment: next test


def test_a_list_that_outgrows_its_segment_fails_the_guess_and_is_redone():
    """One-pass placement (bin_place_kernel): the segment of a tile is the sort class of the last verified longest list.
    2 500 faint specks sit on ONE tile that the first steps mask out; when the mask opens, that tile's list no longer
    fits, the device word is raised by the placement itself, nothing persistent changes and the step is redone plainly."""
    cam = MID
    g = synth.surface_gaussians(150_000, cam, seed=5)
    n = 2500
    gen = torch.Generator().manual_seed(11)
    g["xyz"][:n] = torch.cat([(torch.rand(n, 2, generator=gen) - 0.5) * 0.002, 0.5 + 0.01 * torch.rand(n, 1, generator=gen)], 1)
    g["scales"][:n] = 0.002
    g["opacity"][:n] = 0.02
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    closed = torch.ones(gy, gx, dtype=torch.int32)
    closed[int(cam.cy) // 16, int(cam.cx) // 16] = 0           # the tile under the principal point (identity view)
    masks, poses = [closed, closed, closed, None, closed, None], [None]
    on = _run(g, cam, masks, poses, True, 12)
    off = _run(g, cam, masks, poses, False, 12)
    _same(on, off)
    assert on[2]["failed"] >= 1, on[2]
    assert on[2]["speculative"] >= 4, on[2]


def test_pass_structure_change_is_caught():
    """Steps on a surface map (slice declined) and then the SAME optimiser state size on a volume map (slice taken):
    the first speculative call after the switch assumes 'declined' and must fail cleanly."""
    cam = MID
    dev = "cuda:0"
    ctx = rz.current_context()
    gs = synth.surface_gaussians(150_000, cam, seed=5)
    gv, _ = ru.make_scene(150_000, cam, seed=6)
    res = {}
    for spec in (True, False):
        ctx.set_speculation(spec)
        try:
            before = ctx.speculation_stats()
            out = []
            for g in (gs, gv, gs):
                packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
                opt = mo.ShardedMapOptimizer(packed, lr_col=mo.default_lr_columns() * 1e-4)
                gen = torch.Generator().manual_seed(3)
                gt_c = torch.rand(3, cam.H, cam.W, generator=gen).to(dev)
                gt_d = (1.0 + torch.rand(1, cam.H, cam.W, generator=gen)).to(dev)
                _, s = ru.make_scene(8, cam, seed=1, pose_seed=1)
                rs = ru.hip_settings(s, dev)
                opt.begin_local_optimization()
                for _ in range(3):
                    out.append(float(opt.step_slam(rs, gt_c, gt_d, None)))
                out.append(opt.params.detach().cpu())
            after = ctx.speculation_stats()
            res[spec] = (out, {k: after[k] - before[k] for k in after})
        finally:
            ctx.set_speculation(True)
    for a, b in zip(res[True][0], res[False][0]):
        if isinstance(a, float):
            assert abs(a - b) <= 1e-5 * max(1.0, abs(a))
        else:
            assert ru.frac_bad(a, b, 2e-6) < 1e-3
    assert res[True][1]["failed"] >= 1, res[True][1]


def test_alternating_views_do_not_keep_failing():
    """An optimisation that alternates between two views whose instance totals differ by more than the 12 % margin (a random
    frame of the window per iteration, mapper.py:176-183): the capacities follow a slowly decaying maximum of the totals,
    so the guess fails once, when the larger view is first met, and not every time the loop returns to it."""
    cam = MID
    g = synth.surface_gaussians(150_000, cam, seed=5)
    masks, poses = _masks(cam, [0.6, 1.0]), [1]
    on = _run(g, cam, masks, poses, True, 16)
    off = _run(g, cam, masks, poses, False, 16)
    _same(on, off)
    assert on[2]["speculative"] >= 12 and on[2]["failed"] <= 2, on[2]


def _plain_render(g, rows, cam, pose_seed, dev="cuda:0"):
    """A forward without a backward (what SLAM/render.py does under torch.no_grad) of the first `rows` Gaussians."""
    from rtg_slam_amd.rasterizer import GaussianRasterizer
    _, s = ru.make_scene(8, cam, seed=1, pose_seed=pose_seed)
    rs = ru.hip_settings(s, dev)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    with torch.no_grad():
        out = GaussianRasterizer(rs)(means3D=g["xyz"][:rows], opacities=g["opacity"][:rows], shs=g["shs"][:rows], scales=g["scales"][:rows],
                                     rotations=g["rotations"][:rows], normal_w=g["normal"][:rows],
                                     tile_mask=torch.ones(gy, gx, dtype=torch.int32, device=dev))
    return [t.clone() for t in out]


def test_plain_renders_place_in_one_pass_and_equal_the_classic_path():
    """Forwards WITHOUT a backward on a large surface map (round 6, include/rtgs_raster.h: set_plain_onepass): after the
    forward that learns the near slice is declined, every further one - whatever its Gaussian count - places its instances
    in one pass and checks the assumed sort class itself; images, index maps and weights equal the classic path's bit for
    bit; a list that outgrows the assumed class (2 500 specks on one tile appear) is redone inside the call."""
    dev = "cuda:0"
    cam = MID
    g = {k: v.to(dev) for k, v in synth.surface_gaussians(150_000, cam, seed=5).items()}
    n = 2500
    gen = torch.Generator().manual_seed(11)
    specks = dict(xyz=torch.cat([(torch.rand(n, 2, generator=gen) - 0.5) * 0.002, 0.5 + 0.01 * torch.rand(n, 1, generator=gen)], 1).to(dev))
    # the specks are the LAST rows: row counts below 147 500 leave them out
    g["xyz"][-n:] = specks["xyz"]
    g["scales"][-n:] = 0.002
    g["opacity"][-n:] = 0.02
    ctx = rz.current_context()
    # large maps (the near slice is considered and declined) and small ones (below the large-map size: one pass over every
    # Gaussian) alternate, as the renders of a SLAM frame do (the whole map, the stable rows, the unstable rows): each kind
    # keeps its own history
    plan = [(147_000, None), (3_000, None), (146_000, None), (5_000, None), (147_400, 3), (4_000, 3), (120_000, None), (3_000, None),
            (147_500, None), (150_000, None), (60_000, None), (150_000, None), (147_000, 3)]
    results = {}
    try:
        for onepass in (False, True):
            ctx.set_plain_onepass(onepass)
            before = ctx.plain_stats()
            results[onepass] = [_plain_render(g, rows, cam, pose) for rows, pose in plan]
            torch.cuda.synchronize()
            after = ctx.plain_stats()
            results[(onepass, "stats")] = {k: after[k] - before[k] for k in after}
    finally:
        ctx.set_plain_onepass(True)
    for a, b in zip(results[False], results[True]):
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    assert results[(False, "stats")] == dict(onepass=0, redone=0)
    st = results[(True, "stats")]
    assert st["onepass"] >= 9 and st["redone"] >= 1, st          # the first large two and the first small one learn; the specks' first appearance is redone
