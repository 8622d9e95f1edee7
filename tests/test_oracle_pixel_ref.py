"""The two CPU restatements of the rasterizer against each other, and both against closed forms.

oracle/raster_oracle.py (torch, tiled, chunked, autograd) and oracle/raster_pixel_ref.py (numpy float64, one
global order per pixel, hand-written blend backward, finite-difference per-Gaussian chain) share no code; the
reference holds no source or golden vector for this op (parity unpinned), so agreement of two
independent restatements plus analytic cases is the strongest pin available."""
import math

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from oracle import raster_pixel_ref as pr
from rtg_slam_amd import synth
from tests import raster_util as ru

SMALL = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
ODD = synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0)


def _both(s, g, mask=None, seed=0, dtype=torch.float64):
    gen = torch.Generator().manual_seed(seed)
    H, W = s.image_height, s.image_width
    grads = (torch.randn(3, H, W, generator=gen), torch.randn(1, H, W, generator=gen))
    out_o, gd_o, _ = ru.oracle_run(s, g, tile_mask=mask, grads=grads, dtype=dtype)
    out_p, gd_p = pr.render(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], mask,
                            g_color=grads[0], g_depth=grads[1])
    return out_o, gd_o, out_p, gd_p


def _check(out_o, gd_o, out_p, gd_p, fwd_tol=1e-9, grad_tol=1e-5, max_bad=2e-3):
    names = ["color", "depth", "cidx", "didx", "cw", "dw", "T"]
    for k in (0, 1, 4, 5, 6):
        bad = float((np.abs(out_o[k].double().numpy() - out_p[k]) > fwd_tol).mean())
        assert bad <= max_bad, (names[k], bad)
    for k in (2, 3):
        assert float((out_o[k].numpy() != out_p[k]).mean()) <= max_bad, names[k]
    for k in ru.FIELDS:
        a, b = gd_o[k].double().numpy().reshape(gd_p[k].shape), gd_p[k]
        scale = float(np.abs(b).max()) + 1e-30
        assert float(np.abs(a - b).max()) / scale < grad_tol, (k, float(np.abs(a - b).max()) / scale)
        assert np.array_equal(np.abs(a).reshape(a.shape[0], -1).sum(1) > 0, np.abs(b).reshape(b.shape[0], -1).sum(1) > 0), k


@pytest.mark.parametrize("cam,N,seed,pose", [(SMALL, 300, 1, None), (ODD, 500, 2, 7), (SMALL, 1500, 3, 11)])
def test_oracles_agree_f64(cam, N, seed, pose):
    g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose)
    g = {k: v.double() for k, v in g.items()}
    _check(*_both(s, g, seed=seed))


def test_oracles_agree_with_tile_mask_and_low_opacity():
    g, s = ru.make_scene(800, ODD, seed=5, pose_seed=3, r_range=(0.01, 0.08))
    g = {k: v.double() for k, v in g.items()}
    g["opacity"] = torch.where(torch.rand(800, 1, generator=torch.Generator().manual_seed(1)) < 0.5,
                               torch.tensor(0.3, dtype=torch.float64), g["opacity"])
    gy, gx = (ODD.H + 15) // 16, (ODD.W + 15) // 16
    mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(4)) < 0.6).int()
    _check(*_both(s, g, mask=mask, seed=9))


def test_clamp_gradient_passes_through_above_099():
    """Opacity 1.0: o G > 0.99 around every centre, alpha is clamped there.  Both restatements pass the
    gradient through the clamp (upstream 3DGS backward: dL/do = G dL/dalpha without a clamp mask); a
    restatement that zeroed it would lose most of dL/d opacity on this scene."""
    g, s = ru.make_scene(80, SMALL, seed=8, r_range=(0.2, 0.6))
    g = {k: v.double() for k, v in g.items()}
    g["opacity"] = torch.ones_like(g["opacity"])
    out_o, gd_o, out_p, gd_p = _both(s, g, seed=2)
    _check(out_o, gd_o, out_p, gd_p)
    # the clamp is really active on this scene: pixels whose strongest contributor has alpha T = 0.99 (T = 1)
    assert int((out_p[4] >= 0.99 - 1e-12).sum()) >= 20


def test_single_isotropic_gaussian_closed_form():
    """One fronto-parallel isotropic disc on the optical axis: Sigma2D = (f s / z)^2 + 0.3 on the diagonal,
    alpha(x, y) = o exp(-r^2 / (2 Sigma2D)), colour = c alpha, T = 1 - alpha; dL/do for L = sum(colour) is
    sum(c G)."""
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 48.0, 32.0)
    z, sc, o = 2.0, 0.08, 0.5
    g = dict(xyz=torch.tensor([[0.0, 0.0, z]], dtype=torch.float64), opacity=torch.tensor([[o]], dtype=torch.float64),
             shs=torch.zeros(1, 16, 3, dtype=torch.float64), scales=torch.tensor([[sc, sc, sc]], dtype=torch.float64),
             rotations=torch.tensor([[1.0, 0, 0, 0]], dtype=torch.float64),
             normal=torch.tensor([[0.0, 0.0, -1.0]], dtype=torch.float64))
    rgb = torch.tensor([0.9, 0.5, 0.2], dtype=torch.float64)
    g["shs"][0, 0] = (rgb - 0.5) / 0.28209479177387814
    s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, dtype=torch.float64)
    var = (cam.fx * sc / z) ** 2 + 0.3
    ys, xs = np.meshgrid(np.arange(cam.H), np.arange(cam.W), indexing="ij")
    r2 = (xs - cam.cx) ** 2 + (ys - cam.cy) ** 2
    G = np.exp(-0.5 * r2 / var)
    alpha = np.minimum(0.99, o * G)
    radius = math.ceil(3.0 * math.sqrt(var))
    inrect = ((xs // 16 >= int((cam.cx - radius) // 16)) & (xs // 16 < int((cam.cx + radius + 15) // 16))
              & (ys // 16 >= int((cam.cy - radius) // 16)) & (ys // 16 < int((cam.cy + radius + 15) // 16)))
    alpha = np.where((alpha >= 1 / 255.0) & inrect, alpha, 0.0)
    want = rgb.numpy().reshape(3, 1, 1) * alpha
    ones = (torch.ones(3, cam.H, cam.W, dtype=torch.float64), torch.zeros(1, cam.H, cam.W, dtype=torch.float64))
    out_o, gd_o, _ = ru.oracle_run(s, g, grads=ones, dtype=torch.float64)
    out_p, gd_p = pr.render(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], None,
                            g_color=ones[0], g_depth=ones[1])
    for out in (tuple(o_.numpy() for o_ in out_o), out_p):
        assert np.abs(out[0] - want).max() < 1e-12
        assert np.abs(out[6][0] - (1 - alpha)).max() < 1e-12
        assert out[1][0, 32, 48] == 0.0 and out[3][0, 32, 48] == -1      # alpha 0.5 <= opaque_threshold 0.6: no depth
    # raise the opacity above the threshold and the centre pixel gets the plane depth
    g2 = dict(g, opacity=torch.tensor([[0.7]], dtype=torch.float64))
    out_p2, _ = pr.render(s, g2["xyz"], g2["opacity"], g2["shs"], g2["scales"], g2["rotations"], g2["normal"])
    assert out_p2[1][0, 32, 48] == pytest.approx(z, abs=1e-12) and out_p2[3][0, 32, 48] == 0
    want_do = float((rgb.numpy().reshape(3, 1, 1) * np.where(alpha > 0, G, 0.0)).sum())
    assert float(gd_o["opacity"][0, 0]) == pytest.approx(want_do, rel=1e-10)
    assert float(gd_p["opacity"][0, 0]) == pytest.approx(want_do, rel=1e-10)
    # by symmetry the centre does not want to move, and dL/d(SH dc) = C0 * sum(alpha) per channel
    assert abs(float(gd_p["xyz"][0, 0])) < 1e-6 * want_do and abs(float(gd_o["xyz"][0, 0])) < 1e-6 * want_do
    assert float(gd_p["shs"][0, 0, 1]) == pytest.approx(0.28209479177387814 * alpha.sum(), rel=1e-6)
    assert float(gd_o["shs"][0, 0, 1]) == pytest.approx(0.28209479177387814 * alpha.sum(), rel=1e-10)


def test_f32_oracle_stays_within_product_tolerance_of_pixel_ref():
    """The float32 oracle (what the GPU tests compare the kernels with) against the float64 pixel reference:
    inside the north-star tolerances (1e-4 on maps, 1e-3 relative on gradients)."""
    g, s = ru.make_scene(600, SMALL, seed=12, pose_seed=4)
    out_o, gd_o, out_p, gd_p = _both(s, g, seed=5, dtype=torch.float32)
    _check(out_o, gd_o, out_p, gd_p, fwd_tol=1e-4, grad_tol=1e-3)
