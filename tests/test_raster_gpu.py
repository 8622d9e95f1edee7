"""Parity of the HIP rasterizer (through the drop-in package and the C ABI) with the oracle.
Tolerances (BASELINE.json north_star): rendered RGB / depth 1e-4 abs, gradients 1e-3 relative to
each tensor's max.  Threshold decisions (alpha < 1/255, T' < T_threshold, depth gates, ceil of
the radius) are discontinuous, so a pixel whose decisive value sits within float rounding of a
threshold may legitimately differ; the tests bound the FRACTION of such pixels instead of
pretending it is zero."""
import pytest
import torch

from tests import torch_doubles as td

from rtg_slam_amd import synth
from tests import raster_util as ru

pytestmark = pytest.mark.gpu

SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
ODD = synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0)       # not multiples of 16, cx != (W-1)/2


def check_forward(out_h, out_o, max_bad=2e-3):
    names = ["color", "depth", "color_index", "depth_index", "color_weight", "depth_weight", "T"]
    for k in (0, 1, 4, 5, 6):
        bad = ru.frac_bad(out_h[k], out_o[k], 1e-4)
        assert bad <= max_bad, (names[k], bad)
    for k in (2, 3):
        assert out_h[k].dtype == torch.int32
        bad = float((out_h[k] != out_o[k]).float().mean())
        assert bad <= max_bad, (names[k], bad)


@pytest.mark.parametrize("cam,N,seed,pose", [(SMALL, 300, 1, None), (ODD, 500, 2, 7), (SMALL, 2000, 3, 11)])
def test_forward_matches_oracle(cam, N, seed, pose):
    g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose)
    out_o, _, aux = ru.oracle_run(s, g)
    out_h, _ = ru.hip_run(s, g)
    check_forward(out_h, out_o)
    assert aux["num_rendered"] > 0


@pytest.mark.parametrize("cam,N,seed,pose", [(SMALL, 300, 1, None), (ODD, 500, 2, 7)])
def test_backward_matches_oracle(cam, N, seed, pose):
    g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose)
    gen = torch.Generator().manual_seed(seed)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    _, gd_o, _ = ru.oracle_run(s, g, grads=grads)
    _, gd_h = ru.hip_run(s, g, grads=grads)
    for k in ru.FIELDS:
        ref = gd_o[k]
        scale = float(ref.abs().max()) + 1e-12
        err = float((gd_h[k] - ref).abs().max()) / scale
        assert err < 1e-3, (k, err, scale)


def test_tile_mask_and_sentinels():
    cam = SMALL
    g, s = ru.make_scene(400, cam, seed=4)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = torch.zeros(gy, gx, dtype=torch.int32)
    mask[1:3, 2:5] = 1
    gen = torch.Generator().manual_seed(0)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    out_o, gd_o, _ = ru.oracle_run(s, g, tile_mask=mask, grads=grads)
    out_h, gd_h = ru.hip_run(s, g, tile_mask=mask, grads=grads)
    check_forward(out_h, out_o)
    off = torch.ones(cam.H, cam.W, dtype=torch.bool)
    off[16:48, 32:80] = False
    assert torch.all(out_h[0][:, off] == 0) and torch.all(out_h[1][0][off] == 0)
    assert torch.all(out_h[2][0][off] == -1) and torch.all(out_h[3][0][off] == -1)
    assert torch.all(out_h[6][0][off] == 1.0)                      # mapper.py:501,504
    # Gaussians that reach no rendered pixel get exactly-zero gradients (mapper.py:455)
    untouched = gd_o["shs"].abs().sum(dim=(1, 2)) == 0
    assert untouched.any()
    for k in ru.FIELDS:
        assert torch.all(gd_h[k][untouched] == 0), k
    for k in ru.FIELDS:
        scale = float(gd_o[k].abs().max()) + 1e-12
        assert float((gd_h[k] - gd_o[k]).abs().max()) / scale < 1e-3, k


def test_empty_and_all_masked():
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    cam = SMALL
    g, s = ru.make_scene(50, cam, seed=5)
    dev = "cuda:0"
    rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
    e = torch.empty(0, device=dev)
    outs = rast(means3D=e, opacities=e, shs=e, colors_precomp=None, scales=e, rotations=e, cov3D_precomp=None,
                normal_w=e, tile_mask=None)
    assert outs[0].shape == (3, cam.H, cam.W) and float(outs[0].abs().max()) == 0
    assert torch.all(outs[6] == 1) and torch.all(outs[2] == -1) and torch.all(outs[3] == -1)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    out_h, _ = ru.hip_run(s, g, tile_mask=torch.zeros(gy, gx, dtype=torch.int32))
    assert torch.all(out_h[6] == 1) and torch.all(out_h[1] == 0) and torch.all(out_h[3] == -1)
    # every Gaussian culled (behind the camera): zero instances, blank maps, exactly-zero gradients
    g2 = {k: v.clone() for k, v in g.items()}
    g2["xyz"][:, 2] = -g2["xyz"][:, 2].abs() - 1.0
    gen = torch.Generator().manual_seed(1)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    out_h, gd_h = ru.hip_run(s, g2, grads=grads)
    assert torch.all(out_h[6] == 1) and torch.all(out_h[2] == -1) and float(out_h[0].abs().max()) == 0
    for k in ru.FIELDS:
        assert torch.all(gd_h[k] == 0), k


def test_behind_camera_and_stacked_opaque():
    """Two stacked alpha=0.99 discs seen exactly through their centres: after the first
    T = 1-0.99f = 0.00999999; the second would give T' ~ 9.99998e-5 < 1e-4 in float32 and is NOT
    blended (exact arithmetic says 1e-4, i.e. blended) - SURVEY.md Appendix B (iv)."""
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 48.0, 32.0)
    g, s = ru.make_scene(3, cam, seed=6)
    g["xyz"] = torch.tensor([[0.0, 0.0, 1.0], [0.0, 0.0, 2.0], [0.0, 0.0, -1.0]])
    g["scales"] = torch.tensor([[0.2, 0.2, 0.02]] * 3)
    g["rotations"] = torch.tensor([[1.0, 0, 0, 0]] * 3)
    g["opacity"] = torch.full((3, 1), 0.99)
    g["normal"] = torch.tensor([[0.0, 0.0, -1.0]] * 3)
    out_o, _, _ = ru.oracle_run(s, g)
    out_h, _ = ru.hip_run(s, g)
    check_forward(out_h, out_o, max_bad=0.0)
    cy, cx = 32, 48
    assert int(out_h[2][0, cy, cx]) == 0 and int(out_h[3][0, cy, cx]) == 0
    assert abs(float(out_h[1][0, cy, cx]) - 1.0) < 1e-6
    t_expect = float(torch.tensor(1.0) - torch.tensor(0.99))
    assert float(out_h[6][0, cy, cx]) == t_expect == float(out_o[6][0, cy, cx])   # second disc not blended
    assert float(out_h[4][0, cy, cx]) == float(torch.tensor(0.99))
    assert not bool((out_h[2] == 2).any())                            # behind the camera: culled


def test_config2_200k_forward_properties():
    """BASELINE.json configs[1] shape (200k Gaussians, 640x480): size-independent properties."""
    cam = synth.CONFIG2
    g, s = ru.make_scene(200_000, cam, seed=2024)
    out_h, _ = ru.hip_run(s, g)
    color, depth, cidx, didx, cw, dw, T = out_h
    assert torch.isfinite(color).all() and torch.isfinite(depth).all()
    assert float(T.min()) >= 0 and float(T.max()) <= 1
    assert torch.all((cidx >= -1) & (cidx < 200_000)) and torch.all((didx >= -1) & (didx < 200_000))
    assert torch.all((depth[didx >= 0] > 0)) and torch.all(depth[didx < 0] == 0)
    assert torch.all(cw[cidx < 0] == 0) and torch.all(T[cidx < 0] == 1)
    # determinism of the forward (stable sort, no atomics on the forward path)
    out_2, _ = ru.hip_run(s, g)
    for a, b in zip(out_h, out_2):
        assert torch.equal(a, b)


def test_fallback_sort_path_is_equivalent():
    """The LDS tile-sort binning (default) and the global radix-sort fallback give the same maps:
    the default path only drops (Gaussian, tile) entries no pixel of the tile can see."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam = ODD
    g, s = ru.make_scene(3000, cam, seed=8, pose_seed=5)
    gen = torch.Generator().manual_seed(3)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    out_a, gd_a = ru.hip_run(s, g, grads=grads)
    lib.rtgs_raster_force_sort_path(1)
    try:
        out_b, gd_b = ru.hip_run(s, g, grads=grads)
    finally:
        lib.rtgs_raster_force_sort_path(0)
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)
    for k in ru.FIELDS:
        sc = float(gd_a[k].abs().max()) + 1e-12
        assert float((gd_a[k] - gd_b[k]).abs().max()) / sc < 1e-4, k
    out_o, gd_o, _ = ru.oracle_run(s, g, grads=grads)
    check_forward(out_b, out_o)


def test_fused_activation_and_adam_match_torch():
    """rtgs_map_activate8_{forward,backward} vs the torch restatement `map_optim.activate8` (autograd),
    rtgs_fused_adam (scalar and 16-B vector paths) vs torch.optim.Adam arithmetic with per-column rates."""
    from rtg_slam_amd import map_optim as mo
    from tests.dist_util import adam_reference
    dev = "cuda:0"
    g = synth.random_gaussians(5000, SMALL, seed=11)
    packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
    packed[:, 55:59] *= 1.7                      # un-normalised raw quaternions
    raw8 = packed[:, 51:59].contiguous()
    a = raw8.clone().requires_grad_(True)
    b = raw8.clone().requires_grad_(True)
    ra, rb = td.activate8(a), mo.activate8_hip(b)
    gen = torch.Generator().manual_seed(1)
    la = lb = 0
    for k in ("opacity", "scales", "rotations", "normal"):
        assert float((ra[k] - rb[k]).detach().abs().max()) < 2e-6, k
        w = torch.randn(ra[k].shape, generator=gen).to(dev)
        la = la + (ra[k] * w).sum()
        lb = lb + (rb[k] * w).sum()
    la.backward(); lb.backward()
    scale = float(a.grad.abs().max())
    assert float((a.grad - b.grad).abs().max()) < 1e-5 * scale
    # Adam: [5000,59] (vector path: 295000 % 4 == 0) and [4999,3] (scalar path)
    for rows, c0, c1 in ((5000, 0, 59), (4999, 0, 3)):
        lr = mo.default_lr_columns()[c0:c1].contiguous().to(dev)
        p0 = packed[:rows, c0:c1].contiguous()
        p1, p2 = p0.clone(), p0.clone()
        m1, v1 = torch.zeros_like(p1), torch.zeros_like(p1)
        m2, v2 = torch.zeros_like(p1), torch.zeros_like(p1)
        for step in (1, 2, 3):
            gr = torch.randn(p1.shape, generator=gen).to(dev) * 0.01
            mo._adam_hip(p1, gr, m1, v1, lr, step, 1e-15)
            adam_reference(p2, gr, m2, v2, lr, step, 1e-15)
        assert float((p1 - p2).abs().max()) < 1e-6
        assert float((p1 - p0).abs().max()) > 1e-4


def test_long_lists_and_depth_ties():
    """Thousands of low-opacity Gaussians per tile (so pixels walk deep into the lists) with
    bit-equal depths: exercises every LDS tile-sort class and the equal-depth -> id ordering."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam = SMALL
    g, s = ru.make_scene(60000, cam, seed=21, r_range=(0.02, 0.1))
    g["opacity"] = torch.full_like(g["opacity"], 0.03)
    for k in ru.FIELDS:                       # 5000 exact duplicates -> equal depth bits, different ids
        g[k][55000:] = g[k][:5000]
    out_a, _ = ru.hip_run(s, g)
    st = (__import__("ctypes").c_int64 * 8)()
    lib.rtgs_raster_last_stats(st)
    assert st[6] == 1 and st[7] > 3072, (st[6], st[7])     # LDS path taken, longest list in the top radix class
    lib.rtgs_raster_force_sort_path(1)
    try:
        out_b, _ = ru.hip_run(s, g)
    finally:
        lib.rtgs_raster_force_sort_path(0)
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)
    # against the oracle on every other tile (the oracle walks thousands of entries per tile: 35 s of CPU for all 24)
    mask = torch.zeros((cam.H + 15) // 16, (cam.W + 15) // 16, dtype=torch.int32)
    mask.view(-1)[::2] = 1
    out_m, _ = ru.hip_run(s, g, tile_mask=mask)
    px = mask.bool().repeat_interleave(16, 0).repeat_interleave(16, 1)[:cam.H, :cam.W]
    for a, b in zip(out_m, out_a):                     # a masked render IS the full render on its tiles
        assert torch.equal(a[:, px], b[:, px])
    out_o, _, _ = ru.oracle_run(s, g, tile_mask=mask)
    check_forward(out_m, out_o)


def test_matches_committed_golden(golden_dir):
    """HIP path vs the committed oracle fixture (tests/golden/raster_small.npz)."""
    import os
    import numpy as np
    from oracle import raster_oracle as ro
    z = np.load(os.path.join(golden_dir, "raster_small.npz"))
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
    g = {k: torch.from_numpy(z[f"in_{k}"]) for k in ru.FIELDS}
    s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, viewmatrix=torch.from_numpy(z["viewmatrix"]))
    grads = (torch.from_numpy(z["g_color"]), torch.from_numpy(z["g_depth"]))
    out_h, gd_h = ru.hip_run(s, g, grads=grads)
    ref = tuple(torch.from_numpy(z[f"out_{n}"]) for n in ["color", "depth", "cidx", "didx", "cw", "dw", "T"])
    check_forward(out_h, ref)
    for k in ru.FIELDS:
        r = torch.from_numpy(z[f"grad_{k}"])
        assert float((gd_h[k] - r).abs().max()) / (float(r.abs().max()) + 1e-12) < 1e-3, k


@pytest.mark.parametrize("masked,H,W", [(False, 37, 53), (True, 37, 53), (False, 40, 52), (True, 40, 52)])
def test_fused_loss_matches_torch(masked, H, W):
    """rtgs_slam_loss (HIP) vs map_optim.slam_losses (torch; pinned to the oracle's restatement of mapper.py:402-448 and
    to the reference's own ssim in tests/test_oracle_slam_ops.py / test_dist_cpu.py): value, the four reported terms and
    both image gradients - with a render mask (masked L1 + gated depth) and without (all pixels + the SSIM term)."""
    from rtg_slam_amd import map_optim as mo
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(4)
    # 37 x 53 pixels: the general (one pixel per lane) kernels; 40 x 52, a multiple of four: the 16-B load path
    color = torch.rand(3, H, W, generator=gen).to(dev).requires_grad_(True)
    depth = (torch.rand(1, H, W, generator=gen) * 3).to(dev).requires_grad_(True)
    didx = (torch.randint(-1, 5, (1, H, W), generator=gen, dtype=torch.int32)).to(dev)
    gt_c = torch.rand(3, H, W, generator=gen).to(dev)
    gt_d = (depth.detach() + 0.3 * torch.randn(1, H, W, generator=gen).to(dev)).clamp_min(0)      # errors on both sides of 0.1
    gt_d[0, ::5] = 0
    rm = (torch.rand(H, W, generator=gen) < 0.6).to(dev) if masked else None
    render = (color, depth, None, didx)
    la = td.slam_losses(render, gt_c, gt_d, render_mask=rm)
    ga = torch.autograd.grad(la, [color, depth])
    lb = mo.slam_losses_hip(render, gt_c, gt_d, render_mask=rm)
    gb = torch.autograd.grad(lb * 2.0, [color, depth])
    assert abs(float(la.detach()) - float(lb.detach())) < 2e-6 * max(1.0, abs(float(la.detach())))
    sc = float(ga[0].abs().max())
    assert float((ga[0] * 2 - gb[0]).abs().max()) < (1e-5 * sc if not masked else 1e-9)       # SSIM gradient: conv order
    assert float((ga[1] * 2 - gb[1]).abs().max()) < 1e-7
    if masked:
        assert float(gb[0][:, ~rm].abs().max()) == 0 and float(gb[1][0][~rm].abs().max()) == 0
    else:
        from oracle import slam_ops_oracle as so
        want = 1 - so.ssim(color.detach().cpu(), gt_c.cpu())
        terms = lb.grad_fn.terms.cpu() if hasattr(lb.grad_fn, "terms") else None
        assert terms is None or abs(float(terms[3]) - float(want)) < 2e-6


def test_step_slam_attach_regulariser_and_confidence():
    """The one-call step with the remaining pieces of loss_update: attach regulariser on Gaussians whose initial opacity
    is below 0.9 (mapper.py:384-401) and the confidence increment from non-zero f_dc gradients (:454-456), against the
    autograd path with the same loss written in torch."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from rtg_slam_amd import map_optim as mo
    dev = "cuda:0"
    N = 4000
    g, s = ru.make_scene(N, SMALL, seed=9, pose_seed=1)            # opacity 0.99 w.p. 0.8 else 0.1: ~20 % are attached
    packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
    gen = torch.Generator().manual_seed(2)
    gt_c = torch.rand(3, SMALL.H, SMALL.W, generator=gen).to(dev)
    gt_d = (1.0 + torch.rand(1, SMALL.H, SMALL.W, generator=gen)).to(dev)
    rm = (torch.rand(SMALL.H, SMALL.W, generator=gen) < 0.7).to(dev)
    rs = ru.hip_settings(s, dev)
    rast = GaussianRasterizer(raster_settings=rs)
    oa, ob = mo.ShardedMapOptimizer(packed.clone()), mo.ShardedMapOptimizer(packed.clone())
    ob.begin_local_optimization()
    init = packed.clone()
    sel = torch.sigmoid(init[:, 51]) < 0.9
    assert 0 < int(sel.sum()) < N
    conf = torch.zeros(N, device=dev)
    conf_ref = torch.zeros(N, device=dev)
    for step in range(4):
        # autograd reference: image loss + attach regulariser in torch on the same leaves, dense fused Adam
        leaves = {n: oa.state[n]["p"][:N].detach().clone().requires_grad_(True) for n in ("xyz", "shs", "raw8")}
        gd = mo.activate8_hip(leaves["raw8"])
        gd["xyz"], gd["shs"] = leaves["xyz"], leaves["shs"].view(N, 16, 3)
        out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None, scales=gd["scales"],
                   rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"], tile_mask=None)
        l2 = lambda a, b: ((a - b) ** 2).mean()
        attach = 1000 * (l2(leaves["raw8"][sel][:, 1:4], init[sel][:, 52:55]) + l2(leaves["xyz"][sel], init[sel][:, 0:3])
                         + l2(leaves["raw8"][sel][:, 4:8], init[sel][:, 55:59]))
        total = mo.slam_losses_hip(out, gt_c, gt_d, render_mask=rm) + attach
        grads = torch.autograd.grad(total, [leaves["xyz"], leaves["shs"], leaves["raw8"]])
        conf_ref += (grads[1].view(N, 16, 3)[:, 0].abs() != 0).any(-1).float()
        oa.step_count += 1
        for (name, _, _), gr in zip(mo.BLOCKS, grads):
            st = oa.state[name]
            mo._adam_hip(st["p"][:N], gr.contiguous(), st["m"], st["v"], st["lr"], oa.step_count, oa.eps)
        att_b = float(ob.attach_loss())                              # value at the parameters the step starts from
        lb = ob.step_slam(rs, gt_c, gt_d, None, render_mask=rm, confidence=conf)
        assert abs(float(lb) - float((total - attach).detach())) <= 1e-4 * max(1.0, abs(float(total.detach()))), step
        assert abs(att_b - float(attach.detach())) <= 1e-3 * max(1e-9, abs(float(attach.detach()))) + 1e-12, step
    # step_slam skips the full activation pass after its first call: its tail re-activates the rows it steps, and
    # the persistent activated arrays must equal a fresh activation of the current raw8 bit for bit
    fresh = mo.activate8_hip(ob.state["raw8"]["p"][:N])
    for k in ("opacity", "scales", "rotations", "normal"):
        assert torch.equal(ob.act[k][:N], fresh[k].reshape(ob.act[k][:N].shape)), k
    assert ob._act_valid
    pa, pb = oa.params.cpu(), ob.params.cpu()
    assert ru.frac_bad(pa, pb, 1e-5) < 2e-3
    assert torch.equal(conf.cpu(), conf_ref.cpu()) and 0 < float(conf.max()) <= 4
    moved = (pb - packed.cpu()).abs().max(dim=1).values > 0
    assert bool(moved[sel.cpu()].any())


def test_automatic_fallbacks():
    """The two conditions that leave the LDS-resident binning path: a tile grid above 16 000 tiles and a single
    tile list above 16 384 entries (the latter is discovered after counting and re-runs the preprocess)."""
    import ctypes
    from rtg_slam_amd import _lib
    lib = _lib.load()
    st = (ctypes.c_int64 * 8)()
    # (a) 2064 x 2064 image -> 129 x 129 = 16 641 tiles
    big = synth.CameraSpec(2064, 2064, 1500.0, 1500.0, 1031.5, 1031.5)
    g, s = ru.make_scene(400, big, seed=31)
    out_h, _ = ru.hip_run(s, g)
    lib.rtgs_raster_last_stats(st)
    assert st[6] == 0 and st[2] == 129 * 129
    out_o, _, _ = ru.oracle_run(s, g)
    check_forward(out_h, out_o)
    # (b) 20 000 low-opacity discs piled onto one tile
    cam = SMALL
    g, s = ru.make_scene(20000, cam, seed=32)
    z = 1.0 + 2.0 * torch.rand(20000)
    g["xyz"][:, 0] = ((40.0 - cam.cx) / cam.fx) * z + 0.004 * torch.randn(20000)      # pixel (40, 24): middle of a tile
    g["xyz"][:, 1] = ((24.0 - cam.cy) / cam.fy) * z + 0.004 * torch.randn(20000)
    g["xyz"][:, 2] = z
    g["scales"] = torch.full((20000, 3), 0.004)
    g["opacity"] = torch.full((20000, 1), 0.02)
    gen = torch.Generator().manual_seed(9)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    out_h, gd_h = ru.hip_run(s, g, grads=grads)
    lib.rtgs_raster_last_stats(st)
    assert st[6] == 0 and st[0] > 16384, (st[6], st[0], st[7])
    out_o, gd_o, _ = ru.oracle_run(s, g, grads=grads)
    check_forward(out_h, out_o)
    for k in ru.FIELDS:
        sc = float(gd_o[k].abs().max()) + 1e-12
        assert float((gd_h[k] - gd_o[k]).abs().max()) / sc < 1e-3, k


def test_row_skipping_adam_is_bit_identical_to_dense():
    """rtgs_fused_adam_rows vs rtgs_fused_adam on sparse gradients (changing support over the steps)."""
    from rtg_slam_amd import map_optim as mo
    dev = "cuda:0"
    gen = torch.Generator().manual_seed(7)
    for rows, cols, c0 in ((5003, 3, 0), (5003, 48, 3), (5003, 8, 51)):
        lr = (mo.default_lr_columns()[c0:c0 + cols] + 1e-4).contiguous().to(dev)
        p0 = torch.randn(rows, cols, generator=gen).to(dev)
        pa, pb = p0.clone(), p0.clone()
        ma, va = torch.zeros_like(pa), torch.zeros_like(pa)
        mb, vb = torch.zeros_like(pa), torch.zeros_like(pa)
        ever = torch.zeros(rows, dtype=torch.uint8, device=dev)
        for step in range(1, 6):
            g = torch.randn(rows, cols, generator=gen).to(dev)
            keep = (torch.rand(rows, generator=gen) < 0.15).to(dev)
            g = g * keep[:, None]
            mo._adam_hip(pa, g, ma, va, lr, step, 1e-15)
            mo._adam_rows_hip(pb, g, mb, vb, lr, step, 1e-15, ever)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), cols
        assert 0 < int(ever.sum()) < rows


def test_row_state_backward_matches_dense_over_changing_views():
    """rtgs_raster_backward_rows + the row-state activation / Adam kernels against the dense path, over a
    sequence of views so that rows enter (state 1), leave (state 2) and stay out (state 0)."""
    import ctypes as C
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from rtg_slam_amd import _lib, map_optim as mo
    from rtg_slam_amd.rasterizer import RowGradArena
    lib = _lib.load()
    dev = "cuda:0"
    N = 6000
    P = lambda t: C.c_void_p(t.data_ptr())
    stream = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
    arena = RowGradArena(N, 16, dev)
    g0, _ = ru.make_scene(N, SMALL, seed=5, pose_seed=1)
    raw8 = torch.randn(N, 8, generator=torch.Generator().manual_seed(3)).to(dev)
    lr8 = (mo.default_lr_columns()[51:59] + 1e-4).contiguous().to(dev)
    pa, pb = raw8.clone(), raw8.clone()
    ma, va, mb, vb = (torch.zeros_like(raw8) for _ in range(4))
    ever_a = torch.zeros(N, dtype=torch.uint8, device=dev)
    ever_b = torch.zeros(N, dtype=torch.uint8, device=dev)
    seen_states = set()
    for step, pose in enumerate((1, 2, 3, 1), start=1):
        _, s = ru.make_scene(N, SMALL, seed=5, pose_seed=pose)
        gen = torch.Generator().manual_seed(40 + pose)
        gc, gdp = torch.randn(3, SMALL.H, SMALL.W, generator=gen), torch.randn(1, SMALL.H, SMALL.W, generator=gen)
        _, gd_dense = ru.hip_run(s, g0, grads=(gc, gdp), dev=dev)
        leaves = {k: g0[k].detach().to(dev).clone().requires_grad_(True) for k in ru.FIELDS}
        rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
        arena.begin_step()
        outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                    scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None,
                    normal_w=leaves["normal"], tile_mask=None, grad_rows=arena)
        ((outs[0] * gc.to(dev)).sum() + (outs[1] * gdp.to(dev)).sum()).backward()
        assert arena.calls == 1
        assert leaves["xyz"].grad.data_ptr() == arena.d_means.data_ptr() or torch.equal(leaves["xyz"].grad, arena.d_means)
        state = arena.row_state.cpu()
        seen_states |= set(state.unique().tolist())
        nz = torch.zeros(N, dtype=torch.bool)
        for k in ru.FIELDS:
            got, want = leaves[k].grad.detach().cpu(), gd_dense[k]
            scale = float(want.abs().max()) + 1e-12
            assert ru.frac_bad(got, want, 1e-3 * scale) < 1e-3, (k, step)
            row_nz = got.reshape(N, -1).ne(0).any(1)
            assert torch.equal(row_nz, want.reshape(N, -1).ne(0).any(1)), (k, step)
            nz |= row_nz
        assert torch.equal(nz, state == 1), step                    # state 1 <=> the row carries gradient
        assert int(arena.scratch.count_nonzero()) == 0, step         # scratch handed back clean
        # activation backward: row-state kernel vs dense kernel on the very same incoming gradients
        dense8 = torch.empty(N, 8, device=dev)
        assert lib.rtgs_map_activate8_backward(P(raw8), N, P(arena.d_opac), P(arena.d_scales), P(arena.d_rots),
                                               P(arena.d_normal), P(dense8), stream()) == 0
        assert lib.rtgs_map_activate8_backward_rows(P(raw8), N, P(arena.d_opac), P(arena.d_scales), P(arena.d_rots),
                                                    P(arena.d_normal), P(arena.row_state), P(arena.d_raw8), stream()) == 0
        assert torch.equal(dense8, arena.d_raw8), step
        # Adam: gradient-scanning rows kernel vs state-driven rows kernel vs dense kernel
        mo._adam_hip(pa, dense8, ma, va, lr8, step, 1e-15)
        mo._adam_rows_hip(pb, arena.d_raw8, mb, vb, lr8, step, 1e-15, ever_b, arena.row_state)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), step
    assert seen_states == {0, 1, 2}
    del ever_a


def test_optimizer_row_state_path_matches_dense_path():
    """ShardedMapOptimizer with the persistent-row backward vs the same optimizer forced onto the dense
    backward: same loss trajectory, parameters equal up to the atomics' summation-order noise."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from rtg_slam_amd import map_optim as mo
    dev = "cuda:0"
    N = 5000
    g, _ = ru.make_scene(N, SMALL, seed=9, pose_seed=1)
    packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
    gen = torch.Generator().manual_seed(2)
    gt_c = torch.rand(3, SMALL.H, SMALL.W, generator=gen).to(dev)
    gt_d = (1.0 + torch.rand(1, SMALL.H, SMALL.W, generator=gen)).to(dev)
    opts = [mo.ShardedMapOptimizer(packed.clone()), mo.ShardedMapOptimizer(packed.clone())]
    assert opts[0].grad_rows is not None
    opts[1].grad_rows = None
    used = []
    for step, pose in enumerate((1, 2, 3, 2, 1)):
        _, s = ru.make_scene(N, SMALL, seed=9, pose_seed=pose)
        rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))

        def loss_fn(gd):
            used.append(gd.get("grad_rows") is not None)
            out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None,
                       scales=gd["scales"], rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"],
                       tile_mask=None, grad_rows=gd.get("grad_rows"))
            return mo.slam_losses_hip(out, gt_c, gt_d)
        la, lb = opts[0].step(loss_fn), opts[1].step(loss_fn)
        assert opts[0].grad_rows.calls == 1
        assert abs(float(la) - float(lb)) <= 1e-4 * max(1.0, abs(float(lb))), step
    assert used == [True, False] * 5
    pa, pb = opts[0].params.cpu(), opts[1].params.cpu()
    # Adam's first steps move by +-lr whatever |g| is, so a gradient that cancels to ~0 in a different atomic order can
    # flip a step: bound the fraction of such entries instead of demanding equality
    assert ru.frac_bad(pa, pb, 1e-5) < 2e-3
    moved = (pa - packed.cpu()).abs().max(dim=1).values > 0
    assert 0 < int(moved.sum()) < N            # untouched rows never moved
    assert torch.equal(moved, (pb - packed.cpu()).abs().max(dim=1).values > 0)


def _slice_stats(lib):
    import ctypes as C
    st = (C.c_int64 * 4)()
    assert lib.rtgs_raster_last_slice_stats(st) == 0
    return [int(v) for v in st]


@pytest.mark.parametrize("cam,N,budget,masked", [(SMALL, 4000, 24, False), (ODD, 6000, 48, True), (SMALL, 4000, 4, False),
                                               (ODD, 2500, 2000, False)])
def test_near_slice_forward_is_bit_identical(cam, N, budget, masked):
    """Two-pass forward (near slice first, the rest only for unfinished tiles) vs the single-pass forward:
    every output bit-identical, gradients equal up to the atomics' summation order - for budgets that finish
    none, some and all of the tiles, with and without a tile mask."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    dev = "cuda:0"
    # an opaque, overlapping scene: many tiles saturate early, some never do
    g, s = ru.make_scene(N, cam, seed=21, pose_seed=2, r_range=(0.02, 0.12))
    mask = None
    if masked:
        gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
        mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(4)) < 0.7).int()
    gen = torch.Generator().manual_seed(5)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    try:
        lib.rtgs_raster_set_near_slice(0, 0)
        out_a, gd_a = ru.hip_run(s, g, tile_mask=mask, grads=grads, dev=dev)
        assert _slice_stats(lib)[0] == 0
        lib.rtgs_raster_set_near_slice(1, budget)
        out_b, gd_b = ru.hip_run(s, g, tile_mask=mask, grads=grads, dev=dev)
        used, r1, fin, left = _slice_stats(lib)
    finally:
        lib.rtgs_raster_set_near_slice(2, 384)
    assert used == 1
    for k, (a, b) in enumerate(zip(out_a, out_b)):
        assert torch.equal(a, b), (k, fin, left)
    for k in ru.FIELDS:
        scale = float(gd_a[k].abs().max()) + 1e-12
        assert ru.frac_bad(gd_b[k], gd_a[k], 1e-4 * scale) < 1e-3, (k, fin, left)
        assert torch.equal(gd_a[k].reshape(N, -1).ne(0).any(1), gd_b[k].reshape(N, -1).ne(0).any(1)), k
    if budget == 4:
        assert left > 0                # a starved slice leaves work for pass 2
    if budget == 2000:
        assert r1 > 0 and fin > 0      # a slice that holds every Gaussian finishes the saturating tiles
    print("near slice:", dict(budget=budget, r1=r1, finished=fin, left=left))


@pytest.mark.parametrize("budget,r_range,masked", [(384, (0.02, 0.12), True), (6000, (0.02, 0.12), False), (0, (0.01, 0.05), False)])
def test_onepass_binning_is_bit_identical(budget, r_range, masked):
    """One-pass placement into per-tile segments (bin_place_kernel; maps >= 100 k Gaussians) vs count + scan + scatter:
    every forward output bit-identical (the tile sort's order is total), same slice statistics - with a slice that fits,
    with one whose tile lists outgrow their segments (those tiles are left to pass 2 either way) and in automatic mode."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam, N, dev = synth.CONFIG2, 120_000, "cuda:0"
    g, s = ru.make_scene(N, cam, seed=7, pose_seed=None if budget == 6000 else 2, r_range=r_range)
    if budget == 6000:
        # 5 000 faint specks right in front of the camera (identity view), all on the four tiles round the principal
        # point: their near-slice lists outgrow the 3 072-entry segment
        n = 5000
        gen0 = torch.Generator().manual_seed(11)
        g["xyz"][:n] = torch.cat([(torch.rand(n, 2, generator=gen0) - 0.5) * 0.003, 0.3 + 0.01 * torch.rand(n, 1, generator=gen0)], 1)
        g["scales"][:n] = 0.002
        g["opacity"][:n] = 0.02
    mask = None
    if masked:
        gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
        mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(4)) < 0.7).int()
    gen = torch.Generator().manual_seed(5)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    try:
        lib.rtgs_raster_set_near_slice(1 if budget else 2, budget if budget else 384)
        lib.rtgs_raster_set_onepass_ctx(None, 0)
        out_a, gd_a = ru.hip_run(s, g, tile_mask=mask, grads=grads, dev=dev)
        st_a = _slice_stats(lib)
        lib.rtgs_raster_set_onepass_ctx(None, 1)
        out_b, gd_b = ru.hip_run(s, g, tile_mask=mask, grads=grads, dev=dev)
        st_b = _slice_stats(lib)
    finally:
        lib.rtgs_raster_set_near_slice(2, 384)
        lib.rtgs_raster_set_onepass_ctx(None, 1)
    assert st_a == st_b, (st_a, st_b)
    for k, (a, b) in enumerate(zip(out_a, out_b)):
        assert torch.equal(a, b), (k, st_a)
    for k in ru.FIELDS:
        scale = float(gd_a[k].abs().max()) + 1e-12
        assert ru.frac_bad(gd_b[k], gd_a[k], 1e-4 * scale) < 1e-3, (k, st_a)
    if budget == 6000:
        assert st_a[3] >= 4, st_a          # overgrown lists: tiles left to pass 2
    print("one-pass binning:", dict(budget=budget, slice=st_a))


def test_near_slice_automatic_on_large_map():
    """Automatic mode: a 200 k map takes the two-pass forward; result identical to the single pass."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam = synth.CONFIG2
    g, s = ru.make_scene(200_000, cam, seed=2024)
    try:
        lib.rtgs_raster_set_near_slice(2, 0)
        out_b, _ = ru.hip_run(s, g)
        used, r1, fin, left = _slice_stats(lib)
        lib.rtgs_raster_set_near_slice(0, 0)
        out_a, _ = ru.hip_run(s, g)
    finally:
        lib.rtgs_raster_set_near_slice(2, 384)
    assert used == 1 and fin + left > 0
    for k, (a, b) in enumerate(zip(out_a, out_b)):
        assert torch.equal(a, b), (k, fin, left)


def test_one_call_slam_step_matches_autograd_step():
    """ShardedMapOptimizer.step_slam (rtgs_slam_map_step, one C call) vs step(loss_fn) through autograd:
    same loss values, parameters equal up to the atomics' summation order, same rows moved."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from rtg_slam_amd import map_optim as mo
    dev = "cuda:0"
    N = 5000
    g, _ = ru.make_scene(N, SMALL, seed=9, pose_seed=1)
    packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
    gen = torch.Generator().manual_seed(2)
    gt_c = torch.rand(3, SMALL.H, SMALL.W, generator=gen).to(dev)
    gt_d = (1.0 + torch.rand(1, SMALL.H, SMALL.W, generator=gen)).to(dev)
    gy, gx = (SMALL.H + 15) // 16, (SMALL.W + 15) // 16
    mask = (torch.rand(gy, gx, generator=gen) < 0.8).int().to(dev)
    oa, ob = mo.ShardedMapOptimizer(packed.clone()), mo.ShardedMapOptimizer(packed.clone())
    for step, pose in enumerate((1, 2, 3, 2, 1)):
        _, s = ru.make_scene(N, SMALL, seed=9, pose_seed=pose)
        rs = ru.hip_settings(s, dev)
        rast = GaussianRasterizer(raster_settings=rs)
        tm = mask if step % 2 else None

        def loss_fn(gd):
            out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None,
                       scales=gd["scales"], rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"],
                       tile_mask=tm, grad_rows=gd.get("grad_rows"))
            return mo.slam_losses_hip(out, gt_c, gt_d)
        la = float(oa.step(loss_fn))
        lb = float(ob.step_slam(rs, gt_c, gt_d, tm))
        assert abs(la - lb) <= 1e-4 * max(1.0, abs(la)), step
        assert ob.last_num_rendered > 0 and ob.last_render[0].shape == (3, SMALL.H, SMALL.W)
    # step_slam skips the full activation pass after its first call: its tail re-activates the rows it steps, and
    # the persistent activated arrays must equal a fresh activation of the current raw8 bit for bit
    fresh = mo.activate8_hip(ob.state["raw8"]["p"][:N])
    for k in ("opacity", "scales", "rotations", "normal"):
        assert torch.equal(ob.act[k][:N], fresh[k].reshape(ob.act[k][:N].shape)), k
    assert ob._act_valid
    pa, pb = oa.params.cpu(), ob.params.cpu()
    assert ru.frac_bad(pa, pb, 1e-5) < 2e-3
    moved = (pa - packed.cpu()).abs().max(dim=1).values > 0
    assert 0 < int(moved.sum()) < N
    assert torch.equal(moved, (pb - packed.cpu()).abs().max(dim=1).values > 0)


def test_wall_of_bit_equal_depths():
    """A wall seen head-on: hundreds of Gaussians per tile share ONE float32 depth, so the tile order is decided by the
    Gaussian index alone.  Every LDS sort class (bitonic <= 256, radix above) and the global-sort fallback agree with
    the oracle's stable (depth, index) order."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam = SMALL
    g = synth.surface_gaussians(40000, cam, seed=3, half=(0.6, 0.4, 1.0))      # a small room: the front wall fills the view
    _, s = ru.make_scene(1, cam, seed=1)
    z = (g["xyz"][:, 2] == 1.0)
    assert int(z.sum()) > 3000                                               # thousands of bit-equal depths
    gen = torch.Generator().manual_seed(3)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    out_a, gd_a = ru.hip_run(s, g, grads=grads)
    st = (__import__("ctypes").c_int64 * 8)()
    lib.rtgs_raster_last_stats(st)
    assert st[6] == 1 and st[7] > 256, (st[6], st[7])                        # LDS path, lists in the radix classes
    lib.rtgs_raster_force_sort_path(1)
    try:
        out_b, _ = ru.hip_run(s, g)
    finally:
        lib.rtgs_raster_force_sort_path(0)
    for a, b in zip(out_a, out_b):
        assert torch.equal(a, b)
    out_o, gd_o, _ = ru.oracle_run(s, g, grads=grads)
    check_forward(out_a, out_o)
    for k in ru.FIELDS:
        sc = float(gd_o[k].abs().max()) + 1e-12
        assert float((gd_a[k] - gd_o[k]).abs().max()) / sc < 1e-3, k


def test_normal_loss_term_matches_oracle():
    """The normal term of Mapping.loss_update (mapper.py:433-442, `normal_weight`): value against the oracle's
    restatement, gradient to the Gaussians' normals against autograd through torch indexing (render.py:130-133)."""
    from rtg_slam_amd import map_optim as mo
    from oracle import slam_ops_oracle as so
    dev = "cuda:0"
    cam = SMALL
    N = 1500
    g, s = ru.make_scene(N, cam, seed=12, pose_seed=3, r_range=(0.03, 0.12))
    out_h, _ = ru.hip_run(s, g)
    render = tuple(o.to(dev) for o in out_h)
    gen = torch.Generator().manual_seed(8)
    gt_n = torch.nn.functional.normalize(torch.randn(cam.H, cam.W, 3, generator=gen), dim=-1)
    gt_n[::7, ::5] = 0.0                                         # invalid normals are excluded
    mask = torch.rand(cam.H, cam.W, generator=gen) < 0.8
    gt_c = torch.rand(3, cam.H, cam.W, generator=gen)
    gt_d = out_h[1] + 0.02 * torch.randn(1, cam.H, cam.W, generator=gen)
    assert int((out_h[3] >= 0).sum()) > 200
    for rm in (None, mask):
        nw = g["normal"].to(dev).clone().requires_grad_(True)
        total = mo.slam_losses_hip(render, gt_c.to(dev), gt_d.to(dev), render_mask=None if rm is None else rm.to(dev),
                                   normal_weight=0.3, normal_w=nw, gt_normal=gt_n.to(dev))
        base = mo.slam_losses_hip(render, gt_c.to(dev), gt_d.to(dev), render_mask=None if rm is None else rm.to(dev))
        # oracle: render normal by literal indexing, channels first
        idx = out_h[3]
        rn = torch.zeros(3, cam.H, cam.W)
        nref = g["normal"].clone().requires_grad_(True)
        rn[:, idx[0] > -1] = nref[idx[idx > -1].long()].permute(1, 0)
        tot_o, terms = so.slam_loss(out_h, gt_c, gt_d, gt_normal=gt_n.permute(2, 0, 1), render_mask=rm, normal_weight=0.3,
                                    render_normal=rn)
        assert abs(float(total.detach()) - float(tot_o.detach())) < 1e-5 * max(1.0, abs(float(tot_o.detach())))
        assert abs((float(total.detach()) - float(base)) - 0.3 * float(terms["normal"].detach())) < 1e-5
        (g_hip,) = torch.autograd.grad(total, nw)
        (g_ref,) = torch.autograd.grad(0.3 * terms["normal"], nref)
        assert float(g_ref.abs().max()) > 0
        assert float((g_hip.cpu() - g_ref).abs().max()) <= 1e-4 * float(g_ref.abs().max())


def test_one_call_step_with_the_normal_term_matches_the_autograd_step():
    """`normal_weight > 0` inside rtgs_slam_map_step (rtgs_slam_normal_loss: value into the total, gradient into the depth
    owners' d_normal rows of the row-state arena) vs step(loss_fn) with slam_losses_hip(..., normal_weight) through
    autograd and the gather kernel: same loss values, same parameters, same rows moved - with and without a render mask
    (i.e. without and with the SSIM term)."""
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    from rtg_slam_amd import map_optim as mo
    dev = "cuda:0"
    N = 4000
    g, _ = ru.make_scene(N, SMALL, seed=9, pose_seed=1, r_range=(0.03, 0.12))
    packed = mo.pack_from_activated({k: v.to(dev) for k, v in g.items()})
    gen = torch.Generator().manual_seed(2)
    gt_c = torch.rand(3, SMALL.H, SMALL.W, generator=gen).to(dev)
    gt_d = (1.0 + torch.rand(1, SMALL.H, SMALL.W, generator=gen)).to(dev)
    gt_n = torch.nn.functional.normalize(torch.randn(SMALL.H, SMALL.W, 3, generator=gen), dim=-1)
    gt_n[::7, ::5] = 0.0
    gt_n = gt_n.to(dev)
    rmask = (torch.rand(SMALL.H, SMALL.W, generator=gen) < 0.8).to(dev)
    for rm in (None, rmask):
        oa, ob = mo.ShardedMapOptimizer(packed.clone()), mo.ShardedMapOptimizer(packed.clone())
        for step, pose in enumerate((1, 2, 3, 2)):
            _, s = ru.make_scene(N, SMALL, seed=9, pose_seed=pose)
            rs = ru.hip_settings(s, dev)
            rast = GaussianRasterizer(raster_settings=rs)

            def loss_fn(gd):
                out = rast(means3D=gd["xyz"], opacities=gd["opacity"], shs=gd["shs"], colors_precomp=None,
                           scales=gd["scales"], rotations=gd["rotations"], cov3D_precomp=None, normal_w=gd["normal"],
                           tile_mask=None, grad_rows=gd.get("grad_rows"))
                return mo.slam_losses_hip(out, gt_c, gt_d, render_mask=rm, normal_weight=0.3, normal_w=gd["normal"], gt_normal=gt_n)
            la = float(oa.step(loss_fn))
            lb = float(ob.step_slam(rs, gt_c, gt_d, None, render_mask=rm, normal_weight=0.3, gt_normal=gt_n))
            lc = float(mo.ShardedMapOptimizer(ob.params.clone()).step_slam(rs, gt_c, gt_d, None, render_mask=rm))
            assert abs(la - lb) <= 1e-4 * max(1.0, abs(la)), (step, la, lb)
            assert abs(lb - lc) > 1e-4, "the normal term must be in the total"
        pa, pb = oa.params.cpu(), ob.params.cpu()
        assert ru.frac_bad(pa, pb, 1e-5) < 2e-3
        moved = (pa - packed.cpu()).abs().max(dim=1).values > 0
        assert torch.equal(moved, (pb - packed.cpu()).abs().max(dim=1).values > 0)


@pytest.mark.parametrize("kind,N,cam_i,mode,masked", [("volume", 3000, 0, 0, False), ("volume", 20000, 1, 1, True), ("surface", 60000, 2, 2, False),
                                                      ("volume", 150000, 3, 1, True), ("surface", 150000, 3, 0, False), ("surface", 400000, 1, 0, False)])
def test_tile_cache_backward_equals_the_gather_backward(kind, N, cam_i, mode, masked):
    """Round 6: blend_fwd leaves the records, block masks and plane words of every tile's first 256 list positions where the
    backward finds them from the tile index alone (TileCache, raster_common.h); the entry-per-lane backward then needs ONE memory
    round trip before its walk.  rtgs_raster_set_bwd_debug(8) makes the same kernel ignore the cache and take the gather path
    (tile range -> list ids -> Splat records -> quadrant test; what positions >= 256 and cache-less forwards take).  Same
    entries, same arithmetic: the gradients differ by the order of the LDS float adds only.  The last case has lists longer
    than 256 entries (60 Gaussians per pixel column on 192 x 128): both paths inside one tile."""
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam = [synth.CameraSpec(70, 90, 80.0, 80.0, 44.5, 34.5), synth.CameraSpec(128, 192, 160.0, 160.0, 95.5, 63.5),
           synth.CameraSpec(240, 320, 200.0, 200.0, 159.5, 119.5), synth.CameraSpec(339, 601, 300.0, 300.0, 300.0, 169.0)][cam_i]
    g, s = ru.make_scene(N, cam, seed=33, pose_seed=4, r_range=(0.01, 0.08))
    if kind == "surface":
        g = synth.surface_gaussians(N, cam, seed=11)
    mask = None
    if masked:
        gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
        mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(8)) < 0.6).int()
    gen = torch.Generator().manual_seed(9)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    try:
        lib.rtgs_raster_set_near_slice(mode, 0)
        lib.rtgs_raster_set_bwd_debug(8)
        out_a, gd_a = ru.hip_run(s, g, tile_mask=mask, grads=grads)
        lib.rtgs_raster_set_bwd_debug(0)
        out_b, gd_b = ru.hip_run(s, g, tile_mask=mask, grads=grads)
    finally:
        lib.rtgs_raster_set_bwd_debug(0)
        lib.rtgs_raster_set_near_slice(2, 384)
    assert float((out_a[6] != 1).float().mean()) > 0.2
    for k, (a, b) in enumerate(zip(out_a, out_b)):
        assert torch.equal(a, b), k
    for k in ru.FIELDS:
        scale = float(gd_a[k].abs().max()) + 1e-12
        assert ru.frac_bad(gd_b[k], gd_a[k], 2e-5 * scale) < 1e-4, (k, float((gd_b[k] - gd_a[k]).abs().max()) / scale)
        assert torch.equal(gd_a[k].reshape(N, -1).ne(0).any(1), gd_b[k].reshape(N, -1).ne(0).any(1)), k
