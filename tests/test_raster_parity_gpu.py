"""Parity of the HIP rasterizer at BASELINE.json's own sizes, and against the independent pixel reference.

* configs[1]: 200 000 Gaussians, 640x480, forward + backward vs oracle/raster_oracle.py;
* the headline scene: 1 200 000 Gaussians, 1200x680, forward + backward vs the oracle;
  both on a tile-masked subset (the oracle blends only masked tiles, so it stays a few seconds of CPU), with the
  near-slice two-pass forward off, forced on and in automatic mode - the HIP path that produces the headline
  number is compared with the ORACLE, not only with the single-pass HIP forward;
* the unmasked HIP render of the same scene is bit-identical to the masked one on the masked tiles (so the
  oracle comparison on the subset speaks for the full-image render);
* small scenes against oracle/raster_pixel_ref.py (float64, per pixel, no tiles, hand-written backward), including
  opacity 1.0 where o G > 0.99 and the alpha clamp is active.

Tolerances: north_star - 1e-4 abs on RGB / depth, 1e-3 relative (to the tensor max) on gradients; discontinuous
decisions bound the FRACTION of differing pixels (<= 2e-3)."""
import numpy as np
import pytest
import torch

from rtg_slam_amd import synth
from tests import margins
from tests import raster_util as ru

pytestmark = pytest.mark.gpu

SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
ODD = synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0)


def _spread_mask(cam, n_tiles):
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    m = torch.zeros(gy * gx, dtype=torch.int32)
    m[torch.linspace(0, gy * gx - 1, n_tiles).long()] = 1
    return m.view(gy, gx)


def _grads(cam, seed):
    gen = torch.Generator().manual_seed(seed)
    return torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen)


_cache = {}


def _big_case(name):
    """(scene, settings, mask, upstream grads, oracle outputs, oracle grads), oracle evaluated once per size."""
    if name not in _cache:
        cam, N, n_tiles = {"config2": (synth.CONFIG2, 200_000, 96), "headline": (synth.REPLICA, 1_200_000, 128),
                           "surface": (synth.REPLICA, 1_200_000, 128)}[name]
        g, s = ru.make_scene(N, cam, seed=2024)
        if name == "surface":          # the single-layer map of bench.py's surface leg: the near slice declines itself
            g = synth.surface_gaussians(N, cam, seed=7)
        mask = _spread_mask(cam, n_tiles)
        grads = _grads(cam, 11)
        if name == "config2":
            # BASELINE configs[1] against the DEFINITIONAL oracle (raster_oracle.py + autograd)
            out_o, gd_o, aux = ru.oracle_run(s, g, tile_mask=mask, grads=grads)
            assert aux["num_rendered"] > 0
        else:
            # the two 1.2 M scenes against oracle/raster_oracle_fast.py (the same definition, its tile blend differentiated by
            # hand; pinned to raster_oracle.py + autograd in float64 by tests/test_oracle_raster.py): a quarter of the time
            from oracle import raster_oracle_fast as rf
            out_o, gd_o = rf.forward_backward(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], mask,
                                              grads[0], grads[1])
            assert float(out_o[6].min()) < 1.0
        _cache[name] = (cam, g, s, mask, grads, out_o, gd_o)
    return _cache[name]


def _check_maps(out_h, out_o, max_bad=None, tag="maps"):
    """Bounds derived from what round 4 recorded (profiles/r04_parity_margins.json: at most 2.0e-6 of the pixels of any map
    over 1e-4, on one scene; 0 elsewhere): a map may hold max(4 pixels, 1e-5 of its pixels) over 1e-4 - north_star's 1e-4 is a
    max-norm bound that discontinuous decisions (T-threshold stop, 1/255 skip, depth gates) break on isolated pixels; every
    such pixel is printed with the decision that flipped (ru.explain_outliers) and counted in the margins file."""
    names = ["color", "depth", "color_index", "depth_index", "color_weight", "depth_weight", "T"]
    npix = out_o[0].shape[-1] * out_o[0].shape[-2]
    if max_bad is None:
        max_bad = max(4.0 / npix, 1e-5)
    seen = {}
    for k in (0, 1, 4, 5, 6):
        bad = ru.frac_bad(out_h[k], out_o[k], 1e-4)
        err = (out_h[k] - out_o[k]).abs()
        # the bulk of the error (all but the flipped pixels): the 99.9th percentile would need a sort of 816 k values;
        # the mean over the pixels INSIDE the tolerance says the same thing cheaply
        inside = err <= 1e-4
        seen[names[k]] = {"frac_over_1e-4": bad, "max_abs_err": float(err.max()),
                          "max_abs_err_of_pixels_inside_tol": float(err[inside].max()) if inside.any() else 0.0}
    for k in (2, 3):
        seen[names[k]] = {"frac_different": float((out_h[k] != out_o[k]).float().mean())}
    if any(seen[names[k]]["frac_over_1e-4"] > 0 for k in (0, 1, 4, 5, 6)):
        seen["flipped_decisions"] = ru.explain_outliers([t.cpu() for t in out_h], [t.cpu() for t in out_o])
    margins.record(tag, **seen)
    for k in (0, 1, 4, 5, 6):
        assert seen[names[k]]["frac_over_1e-4"] <= max_bad, (names[k], seen[names[k]])
    for k in (2, 3):
        assert seen[names[k]]["frac_different"] <= max_bad, names[k]


def _check_grads(gd_h, gd_o, tag="grads"):
    seen = {}
    fails = []
    for k in ru.FIELDS:
        ref = gd_o[k]
        scale = float(ref.abs().max()) + 1e-12
        # a pixel that flips a discontinuous decision moves the gradient of the few Gaussians it sees: bound the
        # fraction of rows outside tolerance instead of the max
        row_err = (gd_h[k] - ref).abs().reshape(ref.shape[0], -1).max(dim=1).values / scale
        touched = ref.reshape(ref.shape[0], -1).abs().sum(1) > 0
        untouched = ~touched
        leak = float(gd_h[k].reshape(ref.shape[0], -1)[untouched].abs().max() if untouched.any() else 0.0) / scale
        n_over = float((row_err > 1e-3).float().sum())
        allowed = 2.0          # observed in every recorded run: 0 rows (r04 margins); a flipped pixel may move a row or two
        srt = torch.sort(row_err, descending=True).values
        seen[k] = {"max_rel_err": float(row_err.max()), "rows_over_1e-3": n_over, "rows_allowed": allowed,
                   "touched_rows": float(touched.sum()),
                   "max_rel_err_without_the_3_worst_rows": float(srt[3]) if srt.numel() > 3 else 0.0,
                   "untouched_rows_max_rel": leak}
        if n_over > allowed or leak > 1e-3:
            fails.append((k, seen[k]))
    margins.record(tag, **seen)
    assert not fails, fails


@pytest.mark.parametrize("mode", [0, 1, 2])
@pytest.mark.parametrize("name", ["config2", "headline"])
def test_parity_at_baseline_sizes(name, mode):
    from rtg_slam_amd import _lib
    lib = _lib.load()
    cam, g, s, mask, grads, out_o, gd_o = _big_case(name)
    try:
        lib.rtgs_raster_set_near_slice(mode, 0)
        out_h, gd_h = ru.hip_run(s, g, tile_mask=mask, grads=grads)
        import ctypes as C
        st = (C.c_int64 * 4)()
        lib.rtgs_raster_last_slice_stats(st)
        if mode == 0:
            assert st[0] == 0
        if mode == 1:
            assert st[0] == 1
    finally:
        lib.rtgs_raster_set_near_slice(2, 384)
    _check_maps(out_h, out_o)
    _check_grads(gd_h, gd_o)
    print(name, "mode", mode, "slice stats", [int(v) for v in st])


@pytest.mark.parametrize("name", ["config2", "headline"])
def test_unmasked_render_equals_masked_render_on_the_masked_tiles(name):
    cam, g, s, mask, grads, out_o, _ = _big_case(name)
    out_full, _ = ru.hip_run(s, g)
    out_mask, _ = ru.hip_run(s, g, tile_mask=mask)
    pix = torch.nn.functional.interpolate(mask[None, None].float(), scale_factor=16, mode="nearest")[0, 0, :cam.H, :cam.W] > 0
    for k, (a, b) in enumerate(zip(out_full, out_mask)):
        assert torch.equal(a[:, pix], b[:, pix]), k
    # and the full render agrees with the oracle there too (bit-identical to the masked render, which is checked above)
    for k in (0, 1, 6):
        assert ru.frac_bad(out_full[k][:, pix], out_o[k][:, pix], 1e-4) <= 2e-3, k


def _pixel_ref(s, g, mask, grads):
    from oracle import raster_pixel_ref as pr
    out_p, gd_p = pr.render(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], mask,
                            g_color=grads[0], g_depth=grads[1])
    out_p = tuple(torch.from_numpy(np.ascontiguousarray(o)) for o in out_p)
    out_p = tuple(o.float() if o.dtype == torch.float64 else o for o in out_p)
    gd_p = {k: torch.from_numpy(v).float().reshape(g[k].shape) for k, v in gd_p.items()}
    return out_p, gd_p


@pytest.mark.parametrize("cam,N,seed,pose,opaque", [(SMALL, 300, 1, None, False), (ODD, 1500, 2, 7, False),
                                                    (SMALL, 80, 8, None, True), (ODD, 2000, 4, 3, True)])
def test_hip_matches_independent_pixel_reference(cam, N, seed, pose, opaque):
    """HIP (float32) vs the float64 per-pixel evaluator that shares no code with raster_oracle.py.  `opaque`
    scenes set every opacity to 1.0 so that o G > 0.99 near the centres: the clamp's gradient passes through."""
    kw = dict(r_range=(0.2, 0.6)) if (opaque and N < 200) else {}
    g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose, **kw)
    if opaque:
        g["opacity"] = torch.ones_like(g["opacity"])
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = None if seed % 2 else (torch.rand(gy, gx, generator=torch.Generator().manual_seed(seed)) < 0.7).int()
    grads = _grads(cam, seed)
    out_p, gd_p = _pixel_ref(s, g, mask, grads)
    out_h, gd_h = ru.hip_run(s, g, tile_mask=mask, grads=grads)
    _check_maps(out_h, out_p)
    _check_grads(gd_h, gd_p)
    if opaque:
        assert int((out_p[4] >= 0.99 - 1e-6).sum()) > 0          # the clamp was active somewhere


def test_parity_on_the_surface_map_through_the_declined_slice_paths():
    """The 1.2 M-disc surface map of bench.py's second leg against the oracle: the automatic near slice declines it on
    the device; the FIRST call learns that after launching the slice's (empty) kernels, the SECOND asks first and
    renders through the compact list of visible Gaussians (visible_compact) - both, and the slice switched off, must
    agree with the oracle, forward and backward."""
    from rtg_slam_amd.rasterizer import RasterContext
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    cam, g, s, mask, grads, out_o, gd_o = _big_case("surface")
    dev = "cuda:0"

    def run(ctx):
        leaves = {k: g[k].detach().to(dev).clone().requires_grad_(True) for k in ru.FIELDS}
        rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
        outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                    scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"],
                    tile_mask=mask.to(dev), context=ctx)
        ((outs[0] * grads[0].to(dev)).sum() + (outs[1] * grads[1].to(dev)).sum()).backward()
        return tuple(o.detach().cpu() for o in outs), {k: leaves[k].grad.detach().cpu() for k in ru.FIELDS}

    auto = RasterContext.create()
    auto.set_near_slice(2, 0)
    off = RasterContext.create()
    off.set_near_slice(0, 0)
    for label, ctx in (("off", off), ("auto, blind", auto), ("auto, asked", auto), ("auto, asked again", auto)):
        out_h, gd_h = run(ctx)
        if ctx is auto:
            st = ctx.last_slice_stats()
            assert st["used"] == 1 and st["instances"] == 0 and st["tiles_finished"] == 0, (label, st)
        _check_maps(out_h, out_o, tag="maps [" + label + "]")
        _check_grads(gd_h, gd_o, tag="grads [" + label + "]")


@pytest.mark.parametrize("name", ["headline", "surface"])
def test_whole_image_parity_at_the_headline_size(name):
    """EVERY tile of the 1200x680 render of the two 1.2 M bench scenes against the oracle, forward and backward - not a
    tile-masked subset.  The oracle side is oracle/raster_oracle_fast.py: raster_oracle.py's own per-Gaussian stage and
    binning, the tile blend without an autograd graph and its backward written out by hand (pinned to raster_oracle.py +
    autograd in float64 by tests/test_oracle_raster.py) - about a minute of CPU per scene, where autograd over the
    chunked tile walk needs ~20 (and tens of GB).  The HIP side is ONE unmasked forward + backward: the path bench.py
    times, near slice in automatic mode, per-tile choice of the backward walk."""
    from oracle import raster_oracle_fast as rf
    cam = synth.REPLICA
    g, s = ru.make_scene(1_200_000, cam, seed=2024)
    if name == "surface":
        g = synth.surface_gaussians(1_200_000, cam, seed=7)
    grads = _grads(cam, 11)
    out_h, gd_h = ru.hip_run(s, g, grads=grads)
    out_o, gd_o = rf.forward_backward(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], None,
                                      grads[0], grads[1])
    _check_maps(out_h, out_o)
    _check_grads(gd_h, gd_o)


@pytest.mark.parametrize("kind", ["volume", "surface"])
def test_parity_at_config5_size(kind):
    """BASELINE.json configs[4]'s map size on ONE GPU (the N = 1 anchor of the 1 -> 8 curve): 5 000 000 Gaussians,
    1200x680, forward + backward against the oracle on a spread of 64 tiles (the oracle's per-Gaussian stage alone is
    ~15 s of CPU at this size), volume generator and single-layer surface map."""
    cam = synth.REPLICA
    N = 5_000_000
    g, s = ru.make_scene(N, cam, seed=2024)
    if kind == "surface":
        g = synth.surface_gaussians(N, cam, seed=7)
    from oracle import raster_oracle_fast as rf
    mask = _spread_mask(cam, 64)
    grads = _grads(cam, 11)
    out_o, gd_o = rf.forward_backward(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], mask,
                                      grads[0], grads[1])
    assert float(out_o[6].min()) < 1.0          # something was rendered
    out_h, gd_h = ru.hip_run(s, g, tile_mask=mask, grads=grads)
    _check_maps(out_h, out_o)
    _check_grads(gd_h, gd_o)
