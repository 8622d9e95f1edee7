"""The trainable range and the growing map of ShardedMapOptimizer (rtg_slam_amd/map_optim.py).

RTG-SLAM renders cat(unstable, stable) but optimises only the unstable Gaussians (mapper.py:143-156 parametrizes
self.pointcloud; :1026-1108 concatenates the stable rows without requires_grad) and the map changes every frame
(gaussians_add -> GaussianPointCloud.cat, gaussian_pointcloud.py:286-303; delete / remove :195-235; gaussians_fix,
mapper.py:253-271).  Here: frozen rows are rendered, never differentiated (exact zeros, no row state), never stepped
(bitwise untouched); the trainable rows move exactly as if the frozen ones were constants of an autograd graph; appending,
removing and freezing rows keeps parameters and renders consistent and costs no re-allocation inside the capacity."""
import pytest
import torch

from rtg_slam_amd import synth
from tests import raster_util as ru

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CAM = synth.CameraSpec(128, 192, 160.0, 160.0, 95.5, 63.5)


def _setup(N=12000, seed=4):
    from rtg_slam_amd import map_optim as mo
    g, s = ru.make_scene(N, CAM, seed=seed, pose_seed=2)
    packed = mo.pack_from_activated({k: v.to(DEV) for k, v in g.items()})
    gen = torch.Generator().manual_seed(seed)
    gt_c = torch.rand(3, CAM.H, CAM.W, generator=gen).to(DEV)
    gt_d = (1.0 + torch.rand(1, CAM.H, CAM.W, generator=gen)).to(DEV)
    return mo, packed, ru.hip_settings(s, DEV), gt_c, gt_d


@pytest.mark.parametrize("nf", [0, 7000, 11990])
def test_frozen_rows_untouched_and_trainable_rows_match_the_autograd_step(nf):
    """step_slam (one C call: range backward + tail over the trainable rows) against step(loss_fn) (autograd through the
    arena with the same range) on the same optimiser state, five iterations with the attach regulariser and a confidence
    array; and against an optimiser that trains everything, after ONE step (same forward, so the trainable rows receive
    the same gradients whether or not the others are differentiated)."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    mo, packed, rs, gt_c, gt_d = _setup()
    N = packed.shape[0]
    oa = mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf)
    ob = mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf)
    full = mo.ShardedMapOptimizer(packed.clone())
    rast = GaussianRasterizer(raster_settings=rs)
    rm = torch.ones(CAM.H, CAM.W, dtype=torch.uint8, device=DEV)

    def loss_fn(gd):
        out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None, scales=gd["scales"],
                   rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"], tile_mask=None,
                   grad_rows=gd.get("grad_rows"))
        return mo.slam_losses_hip(out, gt_c, gt_d, render_mask=rm)
    conf = torch.zeros(N - nf, device=DEV)
    full.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    for it in range(5):
        la = float(oa.step(loss_fn))
        lb = float(ob.step_slam(rs, gt_c, gt_d, None, render_mask=rm, confidence=conf))
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(la)), it
        if it == 0:
            pf, pb = full.params, ob.params
            # same gradients for the trainable rows (to the rounding of the slot-summation order, which the first Adam
            # step - a move of lr * sign(g) - turns into 2 lr where a gradient component is ~0: bound the fraction)
            assert ru.frac_bad(pf[nf:].cpu(), pb[nf:].cpu(), 1e-6) < 1e-3
            if nf:
                assert not torch.equal(pf[:nf], packed[:nf])               # ... while `full` did move the others
    pa, pb = oa.params, ob.params
    assert torch.equal(pb[:nf], packed[:nf]) and torch.equal(pa[:nf], packed[:nf])      # frozen rows: bitwise untouched
    assert ru.frac_bad(pa.cpu(), pb.cpu(), 1e-5) < 2e-3
    moved = (pb[nf:] - packed[nf:]).abs().max(dim=1).values > 0
    assert int(moved.sum()) > 0
    a = ob.grad_rows
    assert int(a.row_state[:nf].sum()) == 0                                # never differentiated: no state, exact zeros
    for t in (a.d_means, a.d_shs, a.d_opac, a.d_scales, a.d_rots, a.d_normal, a.d_raw8):
        assert float(t[:nf].abs().sum()) == 0.0
    # confidence (mapper.py:454-456) counts the trainable rows whose f_dc gradient was non-zero, element 0 = row nf
    assert float(conf.sum()) > 0 and float(conf.max()) <= 5.0
    # _features_dc.grad == 0 semantics (mapper.py:455): a trainable row that reached no pixel has exactly zero gradient
    assert float(a.d_shs[nf:][a.row_state[nf:] != 1].abs().sum()) == 0.0


def test_frozen_rows_with_the_attach_regulariser_count_only_trainable_rows():
    mo, packed, rs, gt_c, gt_d = _setup()
    nf = 5000
    N = packed.shape[0]
    part = mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf)
    only = mo.ShardedMapOptimizer(packed[nf:].clone())                     # the trainable rows as a map of their own
    part.begin_local_optimization(); only.begin_local_optimization()
    assert float(part.attach_init["info"][0]) == float(only.attach_init["info"][0]) > 0
    rm = torch.ones(CAM.H, CAM.W, dtype=torch.uint8, device=DEV)
    for _ in range(3):
        part.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    assert torch.equal(part.params[:nf], packed[:nf])
    assert float(part.attach_loss()) > 0.0


def test_append_remove_freeze_keep_the_map_consistent():
    """Rows are appended behind the map (inside the capacity: no re-allocation, the parameter tensors keep their
    addresses), removed and frozen; after every change the zero-copy gaussian_data() renders exactly what a fresh
    optimiser built from `.params` renders, and optimisation continues."""
    from rtg_slam_amd.render import Renderer          # noqa: F401  (import check only)
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    mo, packed, rs, gt_c, gt_d = _setup(N=9000)
    rast = GaussianRasterizer(raster_settings=rs)

    def render(gd):
        with torch.no_grad():
            return rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None, scales=gd["scales"],
                        rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"], tile_mask=None)

    def check(opt):
        ref = mo.ShardedMapOptimizer(opt.params.clone(), n_frozen=opt.n_frozen)
        a, b = render(opt.gaussian_data()), render(ref.gaussian_data())
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        return a

    opt = mo.ShardedMapOptimizer(packed[:6000].clone(), capacity=12000)
    addr = opt.state["shs"]["p"].data_ptr()
    rm = torch.ones(CAM.H, CAM.W, dtype=torch.uint8, device=DEV)
    opt.begin_local_optimization()
    for _ in range(3):
        opt.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    gd = opt.gaussian_data()
    assert gd["xyz"].data_ptr() == opt.state["xyz"]["p"].data_ptr() and gd["opacity"].data_ptr() == opt.act["opacity"].data_ptr()
    check(opt)
    before = opt.params.clone()
    opt.append_rows(packed[6000:9000])
    assert opt.N == 9000 and opt.state["shs"]["p"].data_ptr() == addr      # inside the capacity: nothing moved
    assert torch.equal(opt.params[:6000], before) and torch.equal(opt.params[6000:], packed[6000:9000])
    check(opt)
    opt.begin_local_optimization()
    for _ in range(3):
        opt.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    check(opt)
    # freeze every third row, then remove a fifth of all rows
    p0 = opt.params.clone()
    fmask = torch.zeros(9000, dtype=torch.bool, device=DEV); fmask[::3] = True
    opt.freeze_rows(fmask)
    assert opt.n_frozen == 3000 and torch.equal(opt.params[:3000], p0[fmask]) and torch.equal(opt.params[3000:], p0[~fmask])
    check(opt)
    p1 = opt.params.clone()
    rmask = torch.zeros(9000, dtype=torch.bool, device=DEV); rmask[::5] = True
    opt.remove_rows(rmask)
    assert opt.N == 9000 - int(rmask.sum()) and opt.n_frozen == 3000 - int(rmask[:3000].sum())
    assert torch.equal(opt.params, p1[~rmask])
    check(opt)
    frozen = opt.params[:opt.n_frozen].clone()
    opt.begin_local_optimization()
    for _ in range(3):
        opt.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    assert torch.equal(opt.params[:opt.n_frozen], frozen)
    check(opt)
    # beyond the capacity: re-allocation, contents preserved
    opt.append_rows(packed[:8000])
    assert opt.capacity >= opt.N == 9000 - int(rmask.sum()) + 8000
    check(opt)
    opt.begin_local_optimization()
    opt.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    check(opt)


def test_history_merge_equals_the_oracle_on_the_trainable_rows():
    """rtgs_history_merge (mapper.py:212-251) through ShardedMapOptimizer.history_merge against the oracle that is pinned to
    the reference's own method (tests/test_oracle_slam_ops.py): a few iterations move the trainable rows, the merge pulls
    them back towards the snapshot by their confidence; frozen rows and opacities stay untouched."""
    from oracle import slam_ops_oracle as so
    mo, packed, rs, gt_c, gt_d = _setup()
    nf, N = 4000, packed.shape[0]
    opt = mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf, lr_col=mo.default_lr_columns() * 5.0)
    gen = torch.Generator().manual_seed(1)
    conf = torch.randint(0, 30, (N - nf,), generator=gen).float().to(DEV)
    conf_then = conf.clone()
    opt.begin_local_optimization(confidence=conf)
    rm = torch.ones(CAM.H, CAM.W, dtype=torch.uint8, device=DEV)
    for _ in range(4):
        opt.step_slam(rs, gt_c, gt_d, None, render_mask=rm, confidence=conf)
    assert float((conf - conf_then).sum()) > 0
    before = opt.params.clone()
    b = before[nf:].cpu()
    t = packed[nf:].cpu()
    xyz, shs, raw8 = so.history_merge(b[:, 0:3], b[:, 3:51], b[:, 51:59], t[:, 0:3], t[:, 3:51], t[:, 51:59],
                                      conf_then.cpu().reshape(-1, 1), conf.cpu().reshape(-1, 1), 0.5)
    opt.history_merge(conf, 0.5)
    after = opt.params
    assert torch.equal(after[:nf], packed[:nf])
    a = after[nf:].cpu()
    assert torch.allclose(a[:, 0:3], xyz, atol=1e-6) and torch.allclose(a[:, 3:51], shs, atol=1e-6)
    assert torch.allclose(a[:, 51:59], raw8, atol=2e-6)
    assert float((after - before).abs().max()) > 1e-4
    gd = opt.gaussian_data()                                   # the activated views follow the merged raw8
    fresh = mo.activate8_hip(opt.state["raw8"]["p"][:N])
    assert torch.equal(gd["rotations"], fresh["rotations"]) and torch.equal(gd["scales"], fresh["scales"])


@pytest.mark.parametrize("kind,nf", [("volume", 0), ("surface", 0), ("surface", 9000)])
def test_fused_tail_equals_the_three_kernel_tail(kind, nf):
    """rtgs_slam_map_step with its fused per-Gaussian tail (map_fused.hip: slot sums, chain rule, activation backward,
    attach gradient, Adam, re-activation in ONE kernel, gradient rows never written) against the same step through
    grad_reduce + preprocess_bwd + rtgs_map_tail_rows (tail_mode 1): same device functions, so the parameters, the Adam
    moments, the activated arrays and the confidence agree to the rounding of the slot-summation order; the fused form
    leaves the arena's gradient rows zero."""
    from rtg_slam_amd import map_optim as mo
    N = 30000
    g, s = ru.make_scene(N, CAM, seed=11, pose_seed=2)
    if kind == "surface":
        g = synth.surface_gaussians(N, CAM, seed=5)
        _, s = ru.make_scene(1, CAM, seed=1)
    packed = mo.pack_from_activated({k: v.to(DEV) for k, v in g.items()})
    rs = ru.hip_settings(s, DEV)
    gen = torch.Generator().manual_seed(3)
    gt_c = torch.rand(3, CAM.H, CAM.W, generator=gen).to(DEV)
    gt_d = (1.0 + torch.rand(1, CAM.H, CAM.W, generator=gen)).to(DEV)
    rm = (torch.rand(CAM.H, CAM.W, generator=gen) < 0.9).to(torch.uint8).to(DEV)
    opts = [mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf), mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf)]
    opts[1].tail_mode = 1
    confs = [torch.zeros(N - nf, device=DEV), torch.zeros(N - nf, device=DEV)]
    for o, c in zip(opts, confs):
        o.begin_local_optimization()
        for _ in range(6):
            o.step_slam(rs, gt_c, gt_d, None, render_mask=rm, confidence=c)
    pa, pb = opts[0].params.cpu(), opts[1].params.cpu()
    assert torch.equal(pa[:nf], packed[:nf].cpu())
    assert float((pa - packed.cpu()).abs().max()) > 1e-4
    assert ru.frac_bad(pa, pb, 1e-5) < 2e-3, float((pa - pb).abs().max())
    assert torch.equal(confs[0] > 0, confs[1] > 0) and float((confs[0] - confs[1]).abs().max()) <= 1.0
    for k in ("opacity", "scales", "rotations", "normal"):      # the fused tail re-activates what it steps
        fresh = mo.activate8_hip(opts[0].state["raw8"]["p"][:N])
        assert torch.equal(opts[0].act[k][:N], fresh[k].reshape(opts[0].act[k][:N].shape)), k
    for name in ("xyz", "shs", "raw8"):
        assert ru.frac_bad(opts[0].state[name]["m"][:N - nf].cpu(), opts[1].state[name]["m"][:N - nf].cpu(), 1e-5) < 2e-3, name
    a = opts[0].grad_rows
    assert float(a.d_shs.abs().sum()) == 0.0 and int(a.row_state.sum()) == 0            # gradient rows never materialised
    assert int(opts[1].grad_rows.row_state.sum()) > 0
    lc = opts[0].live_counts.cpu()
    assert 0 < int(lc[0]) <= int(lc[1]) <= 6 * (N - nf)


def test_step_slam_on_a_fully_frozen_map_changes_nothing():
    """an EMPTY trainable range (n_frozen == N) is a legal state and must not read as "every row": the one-call
    step renders, evaluates the loss and steps nothing - parameters, Adam state and gradient rows stay as they were."""
    mo, packed, rs, gt_c, gt_d = _setup(N=6000)
    N = packed.shape[0]
    opt = mo.ShardedMapOptimizer(packed.clone(), n_frozen=N)
    rm = torch.ones(CAM.H, CAM.W, dtype=torch.uint8, device=DEV)
    opt.begin_local_optimization()
    conf = torch.zeros(0, device=DEV)
    for _ in range(2):
        loss = opt.step_slam(rs, gt_c, gt_d, None, render_mask=rm, confidence=conf)
    assert float(loss) > 0 and torch.isfinite(loss)
    assert torch.equal(opt.params, packed)
    a = opt.grad_rows
    assert int(a.row_state.sum()) == 0 and float(a.d_means.abs().sum()) == 0.0 and float(a.d_shs.abs().sum()) == 0.0
    for n in ("xyz", "shs", "raw8"):
        assert float(opt.state[n]["m"].abs().sum()) == 0.0 and int(opt.state[n]["ever"].sum()) == 0
    ref = mo.ShardedMapOptimizer(packed.clone())                             # ... and it rendered what a normal forward renders
    ref.begin_local_optimization()
    l2 = ref.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    assert abs(float(l2) - float(loss)) <= 1e-6 * max(1.0, abs(float(l2)))


@pytest.mark.parametrize("tail_mode", [0, 1])
def test_global_optimization_equals_the_autograd_step_on_the_stable_rows(tail_mode):
    """Mapping.global_optimization (mapper.py:594-707) as a mode of the map object: between begin_ and
    end_global_optimization the one-call step renders the STABLE prefix only and trains all of it with the rescaled
    learning-rate columns (:605-616).  Against (a) `step(loss_fn)` - autograd through the rasterizer - in the same mode, and
    (b) a map that IS the stable prefix with the scaled rates; the unstable suffix stays bit for bit, and the local mode
    works again afterwards."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    mo, packed, rs, gt_c, gt_d = _setup()
    N, nf = packed.shape[0], 8000
    scale = mo.global_lr_scale(final=False)
    oa = mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf)
    ob = mo.ShardedMapOptimizer(packed.clone(), n_frozen=nf)
    ob.tail_mode = tail_mode
    only = mo.ShardedMapOptimizer(packed[:nf].clone(), lr_col=mo.default_lr_columns() * scale)
    only.tail_mode = tail_mode
    rast = GaussianRasterizer(raster_settings=rs)
    rm = torch.ones(CAM.H, CAM.W, dtype=torch.uint8, device=DEV)
    seen = []

    def loss_fn(gd):
        seen.append(int(gd["xyz"].shape[0]))
        out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None, scales=gd["scales"],
                   rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"], tile_mask=None,
                   grad_rows=gd.get("grad_rows"))
        return mo.slam_losses_hip(out, gt_c, gt_d, render_mask=rm)
    for o in (oa, ob):                                                       # a local optimisation first: the suffix moves
        o.begin_local_optimization()
        for _ in range(2):
            o.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    mid, mid_a = ob.params.clone(), oa.params.clone()          # (two runs of the same steps agree to summation-order rounding)
    assert torch.equal(mid[:nf], packed[:nf]) and torch.equal(mid_a[:nf], packed[:nf])
    assert ru.frac_bad(mid_a.cpu(), mid.cpu(), 1e-5) < 2e-3
    conf = torch.zeros(nf, device=DEV)
    oc = mo.ShardedMapOptimizer(mid_a.clone(), n_frozen=nf)                  # step_slam WITHOUT the attach term: the partner of
    oc.tail_mode = tail_mode                                                 # oa's autograd step, whose loss_fn has none either
    oa.begin_global_optimization(scale)
    ob.begin_global_optimization(scale)
    oc.begin_global_optimization(scale)
    oa.attach_init = oc.attach_init = None
    only.begin_local_optimization()
    assert float(ob.attach_init["info"][0]) == float(only.attach_init["info"][0]) > 0  # attach snapshot covers the stable rows
    for it in range(4):
        la = float(oa.step(loss_fn))
        lb = float(ob.step_slam(rs, gt_c, gt_d, None, render_mask=rm, confidence=conf))
        lc = float(only.step_slam(rs, gt_c, gt_d, None, render_mask=rm))
        ld = float(oc.step_slam(rs, gt_c, gt_d, None, render_mask=rm))
        assert abs(lb - lc) <= 1e-6 * max(1.0, abs(lc)), it                  # the same map is rendered: the prefix alone
        assert abs(la - ld) <= 1e-4 * max(1.0, abs(la)), it
    assert seen == [nf] * 4
    pa, pb, pc, pd = oa.params, ob.params, only.params, oc.params
    assert torch.equal(pb[nf:], mid[nf:]) and torch.equal(pa[nf:], mid_a[nf:])         # unstable suffix: bit for bit
    assert ru.frac_bad(pb[:nf].cpu(), pc.cpu(), 1e-6) < 1e-3                 # = the stable rows as a map of their own
    assert torch.equal(pb[:nf, 0:3], mid[:nf, 0:3])                          # position lr 0 (mapper.py:607)
    assert float((pb[:nf, 3:] - mid[:nf, 3:]).abs().max()) > 0
    assert ru.frac_bad(pa[:nf].cpu(), pd[:nf].cpu(), 1e-5) < 2e-3            # autograd through the rasterizer = the one-call step
    assert torch.equal(pd[nf:], mid_a[nf:])
    assert float(conf.sum()) > 0 and float(conf.max()) <= 4.0                # confidence of the STABLE rows (loss_update, unstable=False)
    ob.end_global_optimization()
    ob.begin_local_optimization()
    for _ in range(2):
        ob.step_slam(rs, gt_c, gt_d, None, render_mask=rm)
    pl = ob.params
    assert torch.equal(pl[:nf], pb[:nf]) and float((pl[nf:] - pb[nf:]).abs().max()) > 0
