"""CPU emulation of the near slice's automatic decision (raster_common.h: slice_cut) on a scene, from the oracle's
per-Gaussian stage: which bin the cut lands on, the slice's share of the instances and its cover per pixel.
usage: python tests/diag/slice_decision_sim.py [headline|config2|surface|surface200k|surface5m]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import raster_oracle as ro
from rtg_slam_amd import synth

which = sys.argv[1] if len(sys.argv) > 1 else "headline"
cam, N, kind = {"headline": (synth.REPLICA, 1_200_000, "vol"), "config2": (synth.CONFIG2, 200_000, "vol"),
                "surface": (synth.REPLICA, 1_200_000, "surf"), "surface200k": (synth.CONFIG2, 200_000, "surf"),
                "surface5m": (synth.REPLICA, 5_000_000, "surf")}[which]
g = synth.random_gaussians(N, cam, seed=2024) if kind == "vol" else synth.surface_gaussians(N, cam, seed=7)
s = ro.make_settings(cam.H, cam.W, cam.fx, cam.fy, cam.cx, cam.cy)
with torch.no_grad():
    pre = ro.preprocess(s, g["xyz"], g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"])
v = pre["valid"]
x0, y0, x1, y1 = pre["rect"]
area = ((x1 - x0) * (y1 - y0))[v].numpy()
rad = pre["radius"][v].numpy()
z = pre["depth"][v].numpy().astype(np.float32)
zb = np.clip((z.view(np.uint32) >> 18).astype(np.int64) - (0x3E4CCCCD >> 18), 0, 254)
ntiles = pre["gx"] * pre["gy"]
cap = ntiles * 384
hist = np.bincount(zb, weights=area, minlength=256)
cnt = np.bincount(zb, minlength=256)
cov = np.bincount(zb, weights=np.minimum(rad, 256) ** 2, minlength=256)
cum, cumn = np.cumsum(hist), np.cumsum(cnt)
ok = (cum <= cap) & (cumn <= min(cap, 65536))
cut = int(ok.sum()) - 1
ins = cum[cut] if cut >= 0 else 0
cover = cov[:cut + 1].sum() / (ntiles * 256)
print(f"{which}: visible {v.sum().item()}  rect instances {int(hist.sum())}  cap {cap}  cut bin {cut}  slice instances {int(ins)} "
      f"({int(cumn[cut]) if cut >= 0 else 0} Gaussians)  cover/pixel {cover:.1f}  total/slice {hist.sum() / max(ins, 1):.2f}  "
      f"-> use slice: {bool(cover >= 24 and hist.sum() >= 2 * ins)}")
