"""Tile-walk statistics of the two 1.2 M bench scenes (needs a GPU):
    python tools/walk_stats.py [headline|surface|both]
Prints how many tiles take the row-granular backward walk, the distribution of the measured list share, list lengths."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import synth, rasterizer as rz
from tests import raster_util as ru

which = sys.argv[1] if len(sys.argv) > 1 else "both"
N = int(os.environ.get("RTGS_N", 1_200_000))
cam = synth.REPLICA
for name in (["headline", "surface"] if which == "both" else [which]):
    g, s = ru.make_scene(N, cam, seed=2024)
    if name == "surface":
        g = synth.surface_gaussians(N, cam, seed=7)
    dev = "cuda:0"
    from diff_gaussian_rasterization_depth import GaussianRasterizer
    leaves = {k: g[k].to(dev).requires_grad_(True) for k in ru.FIELDS}
    rast = GaussianRasterizer(raster_settings=ru.hip_settings(s, dev))
    for _ in range(2):
        outs = rast(means3D=leaves["xyz"], opacities=leaves["opacity"], shs=leaves["shs"], colors_precomp=None,
                    scales=leaves["scales"], rotations=leaves["rotations"], cov3D_precomp=None, normal_w=leaves["normal"],
                    tile_mask=None)
    v = rz.image_buffer_views(outs[0].grad_fn.saved_tensors[8], cam.H, cam.W)
    mode, share = v["tile_mode"].cpu(), v["tile_share"].cpu()
    rng = v["ranges"].cpu()
    ln = rng[:, 1] - rng[:, 0]
    nc = v["n_contrib"].cpu()
    print(name, "tiles", mode.numel(), "row-granular tiles", int(mode.sum()), "list len mean/max", float(ln.float().mean()), int(ln.max()),
          "slice", rz.current_context().last_slice_stats())
    print("  share of the list a 4x4 block needs, deciles:", [round(float(q), 3) for q in torch.quantile(share, torch.linspace(0, 1, 11))])
    print("  n_contrib (last position + 1) mean/max", float(nc.float().mean()), int(nc.max()))
