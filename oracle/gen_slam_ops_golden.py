"""Generates tests/golden/slam_ops.npz by running the REFERENCE's own functions (SLAM/utils.py, utils/loss_utils.py,
imported from /root/reference through oracle/ref_shim.py) on seeded inputs, on CPU.  Build container only; the vectors
are committed so the GPU box can check the HIP kernels against the reference's outputs.

    python oracle/gen_slam_ops_golden.py
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def reference_outputs():
    from oracle import ref_shim
    from rtg_slam_amd import synth
    ru = ref_shim.load("SLAM.utils")
    lu = ref_shim.load("utils.loss_utils")
    torch.Tensor.cuda = lambda self, *a, **k: self          # compute_confidence_map hard-codes .cuda()
    g = torch.Generator().manual_seed(11)
    out = {}
    H, W = 70, 101                                           # not multiples of 16
    pm = torch.rand(H, W, generator=g) < 0.45
    err = torch.rand(H, W, generator=g) ** 3
    out["pixelmask"] = pm.numpy()
    out["color_error"] = err.numpy()
    out["t2t"] = ru.transmission2tilemask(pm, 16, 0.5).numpy()
    out["p2t"] = ru.pixelmask2tilemask(pm, 16).numpy()
    out["c2t"] = ru.colorerror2tilemask(err, 16, 0.4).numpy()
    cam = synth.CameraSpec(96, 128, 110.0, 108.0, 63.5, 47.5)
    depth = synth.tum_noise(synth.box_room_depth(cam, synth.look_at_pose(seed=4, max_angle_deg=8, max_trans=0.3)), seed=3,
                            hole_frac=0.03)
    K = torch.tensor([[cam.fx, 0, cam.cx], [0, cam.fy, cam.cy], [0, 0, 1]], dtype=torch.float32)
    out["depth"] = depth.numpy()
    out["K"] = K.numpy()
    out["bilateral"] = ru.bilateralFilter_torch(depth.clone(), 5, 2, 2).numpy()
    for tag, filt in (("raw", False), ("filt", True)):       # tracker.py:104-131 restated over reference helpers
        d = ru.bilateralFilter_torch(depth.clone(), 5, 2, 2) if filt else depth.clone()
        d[~((d > 0.3) & (d < 5.0))] = 0.0
        v = ru.compute_vertex_map(d, K)
        n = ru.compute_normal_map(v)
        c = ru.compute_confidence_map(n, K)
        bad = ((n == 0).all(dim=-1)) | (c < 0.2)[..., 0]
        d, n, v, c = d.clone(), n.clone(), v.clone(), c.clone()
        d[bad] = 0; n[bad] = 0; v[bad] = 0; c[bad] = 0
        out[f"pre_{tag}_depth"], out[f"pre_{tag}_normal"] = d.numpy(), n.numpy()
        out[f"pre_{tag}_vertex"], out[f"pre_{tag}_conf"] = v.numpy(), c.numpy()
        out[f"pre_{tag}_bad"] = bad.numpy()
    a = torch.rand(3, 40, 56, generator=g)
    b = (a + 0.1 * torch.randn(3, 40, 56, generator=g)).clamp(0, 1)
    out["ssim_a"], out["ssim_b"] = a.numpy(), b.numpy()
    out["ssim"] = np.float32(lu.ssim(a, b))
    out["l1"] = np.float32(lu.l1_loss(a, b))
    out["l2"] = np.float32(lu.l2_loss(a, b))
    return out


def main():
    out = reference_outputs()
    path = os.path.join(ROOT, "tests", "golden", "slam_ops.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: getattr(v, "shape", ()) for k, v in out.items()})


if __name__ == "__main__":
    main()
