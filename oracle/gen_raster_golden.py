"""Writes tests/golden/raster_small.npz: outputs + gradients of oracle/raster_oracle.py on a seeded
96x64 scene.  The reference holds no golden vectors for the rasterizer (its source is an
un-vendored submodule), so this fixture pins the ORACLE against regressions and gives the GPU
tests a committed target that does not depend on re-running the oracle.
    python oracle/gen_raster_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from rtg_slam_amd import synth
    from tests import raster_util as ru
    cam = synth.CameraSpec(64, 96, 80.0, 80.0, 47.5, 31.5)
    g, s = ru.make_scene(300, cam, seed=1, pose_seed=4)
    gen = torch.Generator().manual_seed(0)
    grads = (torch.randn(3, cam.H, cam.W, generator=gen), torch.randn(1, cam.H, cam.W, generator=gen))
    outs, gd, aux = ru.oracle_run(s, g, grads=grads)
    save = {f"in_{k}": g[k].numpy() for k in ru.FIELDS}
    save.update(viewmatrix=s.viewmatrix.numpy(), campos=s.campos.numpy(), g_color=grads[0].numpy(), g_depth=grads[1].numpy(),
                cam=np.array([cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy]))
    for i, n in enumerate(["color", "depth", "cidx", "didx", "cw", "dw", "T"]):
        save[f"out_{n}"] = outs[i].numpy()
    for k in ru.FIELDS:
        save[f"grad_{k}"] = gd[k].numpy()
    path = os.path.join(ROOT, "tests", "golden", "raster_small.npz")
    np.savez_compressed(path, **save)
    print(path, os.path.getsize(path) // 1024, "KiB", aux["num_rendered"])


if __name__ == "__main__":
    main()
