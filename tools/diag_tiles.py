"""Diagnostic: distribution of tile list lengths and consumed entries for the bench scene."""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rtg_slam_amd import synth
from rtg_slam_amd.rasterizer import GaussianRasterizationSettings, _RasterizeGaussians
cam = synth.REPLICA
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_200_000
dev = "cuda:0"
g = {k: v.to(dev) for k, v in synth.random_gaussians(N, cam, seed=2024).items()}
rs = GaussianRasterizationSettings(cam.H, cam.W, cam.W/(2*cam.fx), cam.H/(2*cam.fy), torch.zeros(3, device=dev), 1.0,
     torch.eye(4, device=dev), torch.eye(4, device=dev), 3, torch.zeros(3, device=dev), 0.6, 1.0, 0.5, 3.0, False, False, cam.cx, cam.cy, 1e-4)
leaf = g["xyz"].clone().requires_grad_(True)
mask = torch.ones((cam.H+15)//16, (cam.W+15)//16, dtype=torch.int32, device=dev)
outs = _RasterizeGaussians.apply(leaf, g["opacity"], g["shs"], g["scales"], g["rotations"], g["normal"], mask, rs)
img = outs[0].grad_fn.saved_tensors[8]
gy, gx = (cam.H+15)//16, (cam.W+15)//16
nt = gy*gx
off = (nt*8 + 255)//256*256
ranges = img[:nt*8].view(torch.int32).view(nt, 2).cpu()
ncon = img[off:off+cam.H*cam.W*4].view(torch.int32).view(cam.H, cam.W).cpu()
lens = (ranges[:,1]-ranges[:,0]).float()
pad = torch.zeros(gy*16, gx*16, dtype=torch.int32); pad[:cam.H,:cam.W] = ncon
tmax = pad.view(gy,16,gx,16).permute(0,2,1,3).reshape(nt,256).max(dim=1).values.float()
# per wave (4 rows x 16) max
wmax = pad.view(gy,4,4,gx,16).permute(0,3,1,2,4).reshape(nt,4,64).max(dim=2).values.float()
q = torch.tensor([0.5,0.9,0.99,0.999,1.0])
print("list length   mean %.0f quantiles" % lens.mean(), torch.quantile(lens, q).tolist())
print("tile consumed mean %.0f quantiles" % tmax.mean(), torch.quantile(tmax, q).tolist())
print("wave consumed mean %.0f quantiles" % wmax.mean(), torch.quantile(wmax.reshape(-1), q).tolist())
print("pixel n_contrib mean %.1f" % ncon.float().mean(), torch.quantile(ncon.float().reshape(-1)[::7], q).tolist())
print("T<1e-4 stop frac", float((outs[6] < 1e-3).float().mean()))
