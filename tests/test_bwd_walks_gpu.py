"""The backward's three tile walks: the entry-per-lane walk (raster_bwd_entry.hip, the default; lane sums since round 5), the tile-uniform
strip walk and the row-granular walk (raster_bwd.hip; mode 4 chooses between those two per tile from the measured share of
the list the 4x4 blocks need).  Here each is forced on every tile of the same scenes and compared

* with each other (same per-pixel arithmetic, different summation order: 1e-4 of the tensor max),
* with the oracle (1e-3 of the tensor max, north_star),
* for the exact-zero rows (mapper.py:455): the same Gaussians untouched on both walks,

on a single-layer surface map (small footprints: the row-granular case), a volume of large Gaussians (shared lists),
masked tiles, image sizes that are not multiples of the tile, with and without depth gradient, and under the forced
near slice.  The per-tile choice itself: a surface map takes the row-granular walk, large footprints the strip walk."""
import pytest
import torch

from rtg_slam_amd import rasterizer as rz
from rtg_slam_amd import synth
from tests import margins
from tests import raster_util as ru

pytestmark = pytest.mark.gpu

SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
ODD = synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0)
MID = synth.CameraSpec(160, 240, 200.0, 200.0, 119.5, 79.5)


def _grads(cam, seed, depth=True):
    gen = torch.Generator().manual_seed(seed)
    gc, gd = torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen)
    return gc, (gd if depth else torch.zeros_like(gd))


def _scene(kind, N, cam):
    g, s = ru.make_scene(N, cam, seed=5)
    if kind == "surface":
        g = synth.surface_gaussians(N, cam, seed=3)
    return g, s


def _run(s, g, grads, walk, mask=None, slice_mode=None):
    ctx = rz.current_context()
    ctx.set_bwd_walk(walk)
    if slice_mode is not None:
        ctx.set_near_slice(*slice_mode)
    try:
        from diff_gaussian_rasterization_depth import GaussianRasterizer
        dev = "cuda:0"
        leaves = {k: g[k].detach().to(dev).clone().requires_grad_(True) for k in ru.FIELDS}
        rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
        outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                    scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None,
                    normal_w=leaves["normal"], tile_mask=None if mask is None else mask.to(dev))
        img = outs[0].grad_fn.saved_tensors[8]
        mode = rz.image_buffer_views(img, s.image_height, s.image_width)["tile_mode"].clone().cpu()
        loss = (outs[0] * grads[0].to(dev)).sum() + (outs[1] * grads[1].to(dev)).sum()
        loss.backward()
        gd = {k: leaves[k].grad.detach().cpu() for k in ru.FIELDS}
        return tuple(o.detach().cpu() for o in outs), gd, mode
    finally:
        ctx.set_bwd_walk(0)
        if slice_mode is not None:
            ctx.set_near_slice(2, 384)


def _rel(a, b):
    return float((a - b).abs().max()) / (float(b.abs().max()) + 1e-12)


@pytest.mark.parametrize("kind,N,cam,depth,masked", [
    ("surface", 60_000, MID, True, False), ("surface", 60_000, MID, False, True), ("volume", 3000, SMALL, True, False),
    ("volume", 500, ODD, True, True), ("surface", 4000, ODD, True, False), ("volume", 20_000, MID, True, False)])
def test_the_two_walks_agree_and_match_the_oracle(kind, N, cam, depth, masked):
    g, s = _scene(kind, N, cam)
    grads = _grads(cam, 2, depth)
    mask = None
    if masked:
        gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
        mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(4)) < 0.6).int()
    out_s, gd_s, m_s = _run(s, g, grads, 1, mask)
    out_r, gd_r, m_r = _run(s, g, grads, 2, mask)
    out_a, gd_a, m_a = _run(s, g, grads, 4, mask)              # per-tile choice between strip and rows (round 3's default)
    out_d, gd_d, m_d = _run(s, g, grads, 0, mask)              # the default: entry-per-lane walk
    out_m, gd_m, m_m = _run(s, g, grads, 3, mask)              # entry-per-lane walk on every tile
    assert int(m_s.sum()) == 0 and int(m_r.sum()) == m_r.numel() and bool((m_m == 2).all()) and bool((m_d == 2).all())
    for a, b in zip(out_s, out_r):
        assert torch.equal(a, b)                       # the forward does not depend on the backward's walk
    for k in ru.FIELDS:
        assert _rel(gd_r[k], gd_s[k]) < 1e-4, k
        assert _rel(gd_a[k], gd_s[k]) < 1e-4, k
        assert _rel(gd_m[k], gd_s[k]) < 1e-4, (k, _rel(gd_m[k], gd_s[k]))
        rows = lambda t: t.reshape(t.shape[0], -1).abs().sum(1) == 0
        assert torch.equal(rows(gd_r[k]), rows(gd_s[k])), k      # untouched Gaussians: exact zeros on every walk
        assert torch.equal(rows(gd_a[k]), rows(gd_s[k])), k
        assert torch.equal(rows(gd_m[k]), rows(gd_s[k])), k
    _, gd_o, _ = ru.oracle_run(s, g, tile_mask=mask, grads=grads)
    margins.record("max gradient error relative to the tensor max",
                   **{k: {"rows_vs_strip": _rel(gd_r[k], gd_s[k]), "auto_vs_strip": _rel(gd_a[k], gd_s[k]),
                          "mfma_vs_strip": _rel(gd_m[k], gd_s[k]), "mfma_vs_oracle": _rel(gd_m[k], gd_o[k]),
                          "rows_vs_oracle": _rel(gd_r[k], gd_o[k]), "strip_vs_oracle": _rel(gd_s[k], gd_o[k])}
                      for k in ru.FIELDS})
    for k in ru.FIELDS:
        assert _rel(gd_r[k], gd_o[k]) < 1e-3, k
        assert _rel(gd_s[k], gd_o[k]) < 1e-3, k
        assert _rel(gd_m[k], gd_o[k]) < 1e-3, k


def test_per_tile_choice_follows_the_footprints():
    cam = MID
    g, s = _scene("surface", 60_000, cam)
    _, _, mode = _run(s, g, _grads(cam, 1), 4)
    assert float(mode.float().mean()) > 0.8, "small discs: the blocks of a tile need a fraction of its list"
    # the same map with every disc blown up to cover whole tiles: the blocks share the list
    g2 = dict(g)
    g2["scales"] = g["scales"] * 12.0
    _, _, mode2 = _run(s, g2, _grads(cam, 1), 4)
    assert float(mode2.float().mean()) < 0.2


def test_walks_under_the_forced_near_slice():
    cam = MID
    g, s = ru.make_scene(20_000, cam, seed=9)
    grads = _grads(cam, 3)
    _, gd_r, _ = _run(s, g, grads, 2, None, slice_mode=(1, 48))
    _, gd_m, _ = _run(s, g, grads, 3, None, slice_mode=(1, 48))
    _, gd_s, _ = _run(s, g, grads, 1, None, slice_mode=(1, 48))
    _, gd_0, _ = _run(s, g, grads, 1, None, slice_mode=(0, 0))
    for k in ru.FIELDS:
        assert _rel(gd_r[k], gd_s[k]) < 1e-4, k
        assert _rel(gd_m[k], gd_s[k]) < 1e-4, k
        assert _rel(gd_r[k], gd_0[k]) < 1e-4, k
