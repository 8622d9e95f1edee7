"""CPU tests of the rasterizer oracle: committed golden, float64 finite differences, sentinels."""
import os

import numpy as np
import pytest
import torch

from oracle import raster_oracle as ro
from rtg_slam_amd import synth
from tests import raster_util as ru


def test_oracle_matches_committed_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "raster_small.npz"))
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
    g = {k: torch.from_numpy(z[f"in_{k}"]) for k in ru.FIELDS}
    s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, viewmatrix=torch.from_numpy(z["viewmatrix"]))
    grads = (torch.from_numpy(z["g_color"]), torch.from_numpy(z["g_depth"]))
    outs, gd, _ = ru.oracle_run(s, g, grads=grads)
    for i, n in enumerate(["color", "depth", "cidx", "didx", "cw", "dw", "T"]):
        ref = torch.from_numpy(z[f"out_{n}"])
        if ref.dtype == torch.int32:
            assert torch.equal(outs[i], ref), n
        else:
            assert float((outs[i] - ref).abs().max()) < 1e-6, n
    for k in ru.FIELDS:
        ref = torch.from_numpy(z[f"grad_{k}"])
        assert float((gd[k] - ref).abs().max()) <= 1e-5 * float(ref.abs().max()) + 1e-9, k


def test_oracle_autograd_vs_finite_differences_fp64():
    """Central differences in float64 on a 12-Gaussian scene; a parameter whose perturbation flips a
    discontinuous decision (alpha < 1/255, stop rule, depth gates) is skipped by comparing the index maps."""
    cam = synth.CameraSpec(32, 32, 40.0, 40.0, 15.5, 15.5)
    g, s = ru.make_scene(12, cam, seed=7, r_range=(0.02, 0.08))
    dt = torch.float64
    gen = torch.Generator().manual_seed(1)
    wc = torch.randn(3, cam.H, cam.W, generator=gen, dtype=dt)
    wd = torch.randn(1, cam.H, cam.W, generator=gen, dtype=dt)

    def run(vals):
        s2 = s._replace(bg=s.bg.to(dt), viewmatrix=s.viewmatrix.to(dt), campos=s.campos.to(dt))
        out = ro.rasterize(s2, vals["xyz"], vals["opacity"], vals["shs"], vals["scales"], vals["rotations"], vals["normal"])
        return (out[0] * wc).sum() + (out[1] * wd).sum(), out

    base = {k: g[k].to(dt).clone().requires_grad_(True) for k in ru.FIELDS}
    loss, out0 = run(base)
    loss.backward()
    checked = 0
    rng = np.random.RandomState(0)
    for k in ru.FIELDS:
        flat = base[k].detach().reshape(-1)
        for idx in rng.choice(flat.numel(), size=min(6, flat.numel()), replace=False):
            eps = 1e-6
            vals_p = {q: base[q].detach().clone() for q in ru.FIELDS}
            vals_m = {q: base[q].detach().clone() for q in ru.FIELDS}
            vals_p[k].reshape(-1)[idx] += eps
            vals_m[k].reshape(-1)[idx] -= eps
            lp, op = run(vals_p)
            lm, om = run(vals_m)
            if not (torch.equal(op[2], om[2]) and torch.equal(op[3], om[3]) and torch.equal(op[3], out0[3])):
                continue
            fd = float(lp - lm) / (2 * eps)
            an = float(base[k].grad.reshape(-1)[idx])
            assert abs(fd - an) <= 1e-4 * max(1.0, abs(an), abs(fd)), (k, int(idx), fd, an)
            checked += 1
    assert checked >= 20


def test_oracle_sentinels():
    cam = synth.CameraSpec(40, 50, 40.0, 40.0, 24.5, 19.5)
    s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy, bg=torch.tensor([0.1, 0.2, 0.3]))
    e = torch.zeros(0, 3)
    out = ro.rasterize(s, e, torch.zeros(0, 1), torch.zeros(0, 16, 3), e, torch.zeros(0, 4), e)
    assert out[0].shape == (3, cam.H, cam.W) and torch.allclose(out[0][:, 0, 0], torch.tensor([0.1, 0.2, 0.3]))
    assert torch.all(out[6] == 1) and torch.all(out[2] == -1) and torch.all(out[1] == 0)
    g, s2 = ru.make_scene(60, cam, seed=2)
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    out = ro.rasterize(s2, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"],
                       torch.zeros(gy, gx, dtype=torch.int32))
    assert torch.all(out[6] == 1) and torch.all(out[3] == -1) and float(out[0].abs().max()) == 0


def test_scene_generator_is_deterministic_and_wellformed():
    a = synth.random_gaussians(500, synth.CONFIG2, seed=2024)
    b = synth.random_gaussians(500, synth.CONFIG2, seed=2024)
    for k in a:
        assert torch.equal(a[k], b[k])
    assert torch.allclose(a["rotations"].norm(dim=-1), torch.ones(500), atol=1e-5)
    assert torch.allclose(a["normal"].norm(dim=-1), torch.ones(500), atol=1e-5)
    assert float(a["scales"].min()) >= 1e-4 * 0.99 and float(a["scales"].max()) <= 0.05 * 1.01
    assert {round(float(x), 4) for x in np.unique(a["opacity"].numpy())} <= {0.1, 0.99}


def test_depth_prefix_renders_terminated_tiles_identically():
    """The property the HIP forward's near-slice pass rests on, checked on the oracle: render only
    the Gaussians in front of a depth cut; every tile whose pixels all terminated inside that subset comes out exactly
    as in the render of the whole map (the subset's tile list is a prefix of the full depth-ordered list)."""
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
    g = synth.random_gaussians(3000, cam, seed=21, r_range=(0.03, 0.15))
    s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy)
    args = lambda d: (d["xyz"], d["opacity"], d["shs"], d["scales"], d["rotations"], d["normal"])
    full = ro.rasterize(s, *args(g))
    z = g["xyz"][:, 2]                                          # identity camera: view depth
    checked = 0
    for q in (0.15, 0.4):
        keep = z <= torch.quantile(z, q)
        ids = torch.nonzero(keep).reshape(-1)
        sub_g = {k: v[keep] for k, v in g.items()}
        sub, aux = ro.rasterize(s, *args(sub_g), return_aux=True)
        done_tiles = [t for t, ok in aux["tile_terminated"].items() if ok]
        assert 0 < len(done_tiles) < len(aux["tile_terminated"]) or q > 0.3
        gx = (cam.W + 15) // 16
        for t in done_tiles:
            ty, tx = divmod(t, gx)
            sl = (slice(None), slice(ty * 16, min(ty * 16 + 16, cam.H)), slice(tx * 16, min(tx * 16 + 16, cam.W)))
            for k in (0, 1, 4, 5, 6):                           # colour, depth, weights, T
                assert torch.equal(sub[k][sl], full[k][sl]), (q, t, k)
            for k in (2, 3):                                    # index maps: subset ids -> map ids
                si = sub[k][sl].long()
                mapped = torch.where(si >= 0, ids[si.clamp_min(0)], si)
                assert torch.equal(mapped.to(torch.int32), full[k][sl]), (q, t, k)
            checked += 1
    assert checked > 0


@pytest.mark.parametrize("N,seed,pose,masked,opaque", [(300, 1, None, False, False), (800, 2, 7, True, False), (200, 3, 4, False, True)])
def test_hand_written_tile_backward_equals_autograd(N, seed, pose, masked, opaque):
    """oracle/raster_oracle_fast.py (no autograd graph over the tiles; the blend's backward written out by hand, one
    autograd pass through the per-Gaussian stage) against raster_oracle.rasterize + autograd, float64: the same maps to
    1e-12 and the same gradients to 1e-9 of the tensor max - it is what makes a WHOLE-image parity test at 1.2 M
    Gaussians affordable (tests/test_raster_parity_gpu.py)."""
    from oracle import raster_oracle_fast as rf
    from rtg_slam_amd import synth
    from tests import raster_util as ru
    cam = synth.CameraSpec(70, 101, 90.0, 85.0, 49.0, 36.0)
    g, s = ru.make_scene(N, cam, seed=seed, pose_seed=pose)
    if opaque:
        g["opacity"] = torch.ones_like(g["opacity"])
    gy, gx = (cam.H + 15) // 16, (cam.W + 15) // 16
    mask = (torch.rand(gy, gx, generator=torch.Generator().manual_seed(seed)) < 0.7).int() if masked else None
    gen = torch.Generator().manual_seed(seed + 5)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen).double(), torch.randn(1, cam.H, cam.W, generator=gen).double())
    out_a, gd_a, _ = ru.oracle_run(s, g, tile_mask=mask, grads=grads, dtype=torch.float64)
    g64 = {k: g[k].double() for k in ru.FIELDS}
    s64 = s._replace(bg=s.bg.double(), viewmatrix=s.viewmatrix.double(), campos=s.campos.double())
    out_f, gd_f = rf.forward_backward(s64, g64["xyz"], g64["opacity"], g64["shs"], g64["scales"], g64["rotations"],
                                      g64["normal"], mask, grads[0], grads[1])
    for k in (0, 1, 4, 5, 6):
        assert float((out_a[k] - out_f[k]).abs().max()) < 1e-12, k
    for k in (2, 3):
        assert torch.equal(out_a[k], out_f[k]), k
    for k in ru.FIELDS:
        scale = float(gd_a[k].abs().max()) + 1e-300
        assert float((gd_a[k] - gd_f[k]).abs().max()) / scale < 1e-9, (k, float((gd_a[k] - gd_f[k]).abs().max()) / scale)
