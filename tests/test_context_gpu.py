"""Library state lives in contexts (include/rtgs_raster.h: rtgs_ctx), not in process globals:
two threads render different scenes concurrently with different near-slice settings and get the results and the
per-call statistics of their own calls; a backward does not depend on the near-slice budget in force when it runs."""
import threading

import pytest
import torch

from rtg_slam_amd import synth
from tests import raster_util as ru

pytestmark = pytest.mark.gpu

SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
ODD = synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0)


def _run(s, g, grads, context=None, dev="cuda:0"):
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    leaves = {k: g[k].detach().to(dev).clone().requires_grad_(True) for k in ru.FIELDS}
    rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
    outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"],
                tile_mask=None, context=context)
    ((outs[0] * grads[0].to(dev)).sum() + (outs[1] * grads[1].to(dev)).sum()).backward()
    return tuple(o.detach().cpu() for o in outs), {k: leaves[k].grad.detach().cpu() for k in ru.FIELDS}


def test_two_threads_render_concurrently_with_their_own_contexts():
    from rtg_slam_amd.rasterizer import RasterContext, current_context
    cases = [(SMALL, 4000, 21, 2, (1, 24)), (ODD, 6000, 22, 3, (0, 0))]      # (camera, N, seed, pose, (slice mode, budget))
    scenes, serial = [], []
    for cam, N, seed, pose, (mode, budget) in cases:
        g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose, r_range=(0.02, 0.12))
        gen = torch.Generator().manual_seed(seed)
        grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
        scenes.append((s, g, grads))
        ctx = RasterContext.create()
        ctx.set_near_slice(mode, budget)
        out, gd = _run(s, g, grads, context=ctx)
        serial.append((out, gd, ctx.last_stats()[0], ctx.last_slice_stats()["used"]))
    results = [None, None]
    errors = []

    def worker(i):
        try:
            torch.cuda.set_device(0)
            ctx = current_context()                       # not the main thread: a private context
            assert ctx.handle is not None
            mode, budget = cases[i][4]
            ctx.set_near_slice(mode, budget)
            s, g, grads = scenes[i]
            with torch.cuda.stream(torch.cuda.Stream()):
                rec = []
                for _ in range(12):
                    out, gd = _run(s, g, grads)
                    rec.append((out, gd, ctx.last_stats()[0], ctx.last_slice_stats()["used"]))
                torch.cuda.current_stream().synchronize()
            results[i] = rec
        except Exception as e:                            # surface in the main thread
            errors.append(e)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    for i in range(2):
        out0, gd0, R0, used0 = serial[i]
        assert used0 == (1 if cases[i][4][0] == 1 else 0)
        for out, gd, R, used in results[i]:
            for a, b in zip(out, out0):
                assert torch.equal(a, b)                  # the forward is deterministic: bit-identical to the serial run
            assert R == R0 and used == used0               # statistics of THIS thread's call
            for k in ru.FIELDS:
                sc = float(gd0[k].abs().max()) + 1e-12
                assert float((gd[k] - gd0[k]).abs().max()) / sc < 1e-4, k
    assert current_context().handle is None               # the main thread keeps the default context


def test_backward_does_not_depend_on_the_budget_in_force():
    """the geometry buffer's layout used to be derived from the process-global budget at backward time.
    Now a forward under budget 48 followed by a budget change gives the same gradients."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from rtg_slam_amd.rasterizer import RasterContext
    dev = "cuda:0"
    cam = ODD
    g, s = ru.make_scene(6000, cam, seed=21, pose_seed=2, r_range=(0.02, 0.12))
    gen = torch.Generator().manual_seed(5)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    ctx = RasterContext.create()
    ctx.set_near_slice(1, 48)
    _, gd_ref = _run(s, g, grads, context=ctx)
    assert ctx.last_slice_stats()["used"] == 1 and ctx.last_slice_stats()["tiles_finished"] > 0
    leaves = {k: g[k].detach().to(dev).clone().requires_grad_(True) for k in ru.FIELDS}
    rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
    outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"],
                tile_mask=None, context=ctx)
    ctx.set_near_slice(1, 400)                             # a different budget is in force when the backward runs
    ((outs[0] * grads[0].to(dev)).sum() + (outs[1] * grads[1].to(dev)).sum()).backward()
    for k in ru.FIELDS:
        sc = float(gd_ref[k].abs().max()) + 1e-12
        assert float((leaves[k].grad.cpu() - gd_ref[k]).abs().max()) / sc < 1e-4, k


def test_automatic_mode_decides_per_call_not_from_history():
    """Automatic near-slice mode on a depth-complex 200 k map runs the slice on EVERY call (no cooldown state), and
    on a single-layer surface map of the same size never does - whatever was rendered before (the host only changes HOW
    it learns the decision: after a declined call it asks before launching the slice's kernels); outputs equal the
    single-pass forward in every case, gradients too."""
    from rtg_slam_amd.rasterizer import RasterContext
    cam = synth.CONFIG2
    ctx = RasterContext.create()
    ctx.set_near_slice(2, 0)
    off = RasterContext.create()
    off.set_near_slice(0, 0)
    gen = torch.Generator().manual_seed(1)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    g, s = ru.make_scene(200_000, cam, seed=2024)
    gs = synth.surface_gaussians(200_000, cam, seed=7)
    ref_c, gd_c = _run(s, g, grads, context=off)
    ref_s, gd_s = _run(s, gs, grads, context=off)

    def check(scene, ref, gd_ref, taken):
        out, gd = _run(s, scene, grads, context=ctx)
        st = ctx.last_slice_stats()
        if taken:
            assert st["used"] == 1 and st["tiles_finished"] > st["tiles_left_to_pass2"], st
        else:
            assert st["used"] == 1 and st["instances"] == 0 and st["tiles_finished"] == 0, st   # the kernels declined
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        for k in ru.FIELDS:
            sc = float(gd_ref[k].abs().max()) + 1e-12
            assert float((gd[k] - gd_ref[k]).abs().max()) / sc < 1e-4, k

    # taken blind x2, declined blind, declined asked x2, taken asked, taken blind, declined blind
    for scene, ref, gd_ref, taken in [(g, ref_c, gd_c, True), (g, ref_c, gd_c, True), (gs, ref_s, gd_s, False),
                                      (gs, ref_s, gd_s, False), (gs, ref_s, gd_s, False), (g, ref_c, gd_c, True),
                                      (g, ref_c, gd_c, True), (gs, ref_s, gd_s, False)]:
        check(scene, ref, gd_ref, taken)
