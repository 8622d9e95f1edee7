"""Seeded synthetic inputs for the hot path: random Gaussian maps with the
statistics of RTG-SLAM's config constants, Replica/TUM-shaped cameras, and analytic
box-room RGB-D streams for the ICP tracker.  CPU torch only (deterministic across hosts);
callers move the tensors to the device.

Constants follow /root/reference/configs/base.yaml:30-36 (SH degree 3, xyz_factor
[1,1,0.1], init_opacity 0.99, radius 0.001..0.05 m) and the dataset intrinsics
quoted in SURVEY.md §8 (Replica 1200x680 f=600; TUM fr1 640x480).
"""
from __future__ import annotations

import math
from typing import Dict, NamedTuple

import torch

SH_C0 = 0.28209479177387814


class CameraSpec(NamedTuple):
    H: int
    W: int
    fx: float
    fy: float
    cx: float
    cy: float


REPLICA = CameraSpec(680, 1200, 600.0, 600.0, 599.5, 339.5)
TUM_FR1 = CameraSpec(480, 640, 517.3, 516.5, 318.6, 255.3)
CONFIG2 = CameraSpec(480, 640, 517.0, 517.0, 319.5, 239.5)


def rotmat_to_quat(R: torch.Tensor) -> torch.Tensor:
    """[P,3,3] proper rotations -> (w,x,y,z), robust branch-free (Shepperd via max component)."""
    m00, m11, m22 = R[:, 0, 0], R[:, 1, 1], R[:, 2, 2]
    q = torch.stack([
        1 + m00 + m11 + m22,
        1 + m00 - m11 - m22,
        1 - m00 + m11 - m22,
        1 - m00 - m11 + m22,
    ], dim=-1).clamp_min(0)
    k = q.argmax(dim=-1)
    w0 = torch.stack([q[:, 0], R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], -1)
    w1 = torch.stack([R[:, 2, 1] - R[:, 1, 2], q[:, 1], R[:, 0, 1] + R[:, 1, 0], R[:, 0, 2] + R[:, 2, 0]], -1)
    w2 = torch.stack([R[:, 0, 2] - R[:, 2, 0], R[:, 0, 1] + R[:, 1, 0], q[:, 2], R[:, 1, 2] + R[:, 2, 1]], -1)
    w3 = torch.stack([R[:, 1, 0] - R[:, 0, 1], R[:, 0, 2] + R[:, 2, 0], R[:, 1, 2] + R[:, 2, 1], q[:, 3]], -1)
    cand = torch.stack([w0, w1, w2, w3], dim=1)                     # [P,4,4]
    out = cand[torch.arange(R.shape[0], device=R.device), k]
    out = out / out.norm(dim=-1, keepdim=True)
    return torch.where(out[:, :1] < 0, -out, out)


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
    ], dim=-1).reshape(-1, 3, 3)


def normal_from_scale_rot(scales: torch.Tensor, rotations: torch.Tensor) -> torch.Tensor:
    """`GaussianPointCloud.get_normal` rule (SLAM/gaussian_pointcloud.py:538-550):
    the column of R belonging to the smallest scale, re-normalised with +1e-8."""
    R = quat_to_rotmat(rotations)
    k = scales.argmin(dim=1)
    n = R[torch.arange(R.shape[0]), :, k]
    return n / (n.norm(dim=-1, keepdim=True) + 1e-8)


def random_gaussians(N: int, cam: CameraSpec, seed: int = 2024, z_range=(0.5, 5.0),
                     r_range=(0.001, 0.05), margin: float = 0.05,
                     c2w: torch.Tensor | None = None) -> Dict[str, torch.Tensor]:
    """SURVEY.md §8d "Config 2 -> concrete" generator.  Returns the `gaussian_data` dict of
    SLAM/render.py:93-98 (post-activation values, float32, CPU)."""
    g = torch.Generator().manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    z = z_range[0] + (z_range[1] - z_range[0]) * U(N)
    u = (-margin + (1 + 2 * margin) * U(N)) * cam.W
    v = (-margin + (1 + 2 * margin) * U(N)) * cam.H
    xyz_c = torch.stack([(u - cam.cx) / cam.fx * z, (v - cam.cy) / cam.fy * z, z], -1)

    r = torch.exp(math.log(r_range[0]) + (math.log(r_range[1]) - math.log(r_range[0])) * U(N))
    aniso = 0.6 + 0.4 * U(N)
    scales = torch.stack([r * aniso, r, 0.1 * r], -1)            # xyz_factor [1,1,0.1]

    # disc normal within 60 deg of the ray back to the camera, random spin about it
    back = -xyz_c / xyz_c.norm(dim=-1, keepdim=True)
    helper = torch.where((back[:, :1].abs() < 0.9), torch.tensor([[1.0, 0, 0]], dtype=torch.float64),
                         torch.tensor([[0, 1.0, 0]], dtype=torch.float64))
    e1 = torch.linalg.cross(back, helper)
    e1 = e1 / e1.norm(dim=-1, keepdim=True)
    e2 = torch.linalg.cross(back, e1)
    tilt = math.radians(60.0) * U(N)
    azim = 2 * math.pi * U(N)
    n = (torch.cos(tilt)[:, None] * back
         + torch.sin(tilt)[:, None] * (torch.cos(azim)[:, None] * e1 + torch.sin(azim)[:, None] * e2))
    h2 = torch.where((n[:, :1].abs() < 0.9), torch.tensor([[1.0, 0, 0]], dtype=torch.float64),
                     torch.tensor([[0, 1.0, 0]], dtype=torch.float64))
    t1 = torch.linalg.cross(n, h2)
    t1 = t1 / t1.norm(dim=-1, keepdim=True)
    spin = 2 * math.pi * U(N)
    t2 = torch.linalg.cross(n, t1)
    a1 = torch.cos(spin)[:, None] * t1 + torch.sin(spin)[:, None] * t2
    a2 = torch.linalg.cross(n, a1)
    R = torch.stack([a1, a2, n], dim=-1)                           # columns = local axes, det +1

    opacity = torch.where(U(N) < 0.8, torch.tensor(0.99, dtype=torch.float64),
                          torch.tensor(0.1, dtype=torch.float64))[:, None]
    dc = (U(N, 1, 3) - 0.5) / SH_C0                                 # RGB2SH, utils/sh_utils.py:123
    rest = 0.05 * torch.randn(N, 15, 3, generator=g, dtype=torch.float64)
    shs = torch.cat([dc, rest], dim=1)

    if c2w is not None:
        c2w = c2w.double()
        xyz = xyz_c @ c2w[:3, :3].t() + c2w[:3, 3]
        R = c2w[:3, :3] @ R
    else:
        xyz = xyz_c
    rot = rotmat_to_quat(R)
    out = dict(xyz=xyz, opacity=opacity, scales=scales, rotations=rot, shs=shs)
    out = {k: t.to(torch.float32).contiguous() for k, t in out.items()}
    out["normal"] = normal_from_scale_rot(out["scales"], out["rotations"]).contiguous()
    return out


def surface_gaussians(N: int, cam: CameraSpec, seed: int = 2024, half=(2.5, 1.5, 3.0),
                      c2w: torch.Tensor | None = None) -> Dict[str, torch.Tensor]:
    """A single-layer SURFACE map, the shape RTG-SLAM's mapper actually builds (mapper.py:715-829 samples new
    Gaussians on depth-map pixels, gaussian_pointcloud.py:366-405 sets their radius to the spacing of their three
    nearest neighbours): N discs spread uniformly over the six walls of the box room of `box_room_depth`
    (|x|<hx, |y|<hy, |z|<hz, camera inside), flat against the wall (normal = smallest axis, xyz_factor [1,1,0.1]),
    radius = sqrt(area / N) clipped to [min_radius, max_radius] = [0.001, 0.05] (configs/base.yaml:32-36),
    opacity init_opacity = 0.99, smooth position-dependent colour.  Only the part of the room inside the frustum is
    visible from one view, as in a real map.  `cam` is unused by the geometry (kept for a uniform signature)."""
    g = torch.Generator().manual_seed(seed)
    U = lambda *s: torch.rand(*s, generator=g, dtype=torch.float64)
    hx, hy, hz = half
    areas = torch.tensor([4 * hy * hz, 4 * hy * hz, 4 * hx * hz, 4 * hx * hz, 4 * hx * hy, 4 * hx * hy], dtype=torch.float64)
    wall = torch.multinomial(areas / areas.sum(), N, replacement=True, generator=g)
    a, b = 2 * U(N) - 1, 2 * U(N) - 1
    ax = wall // 2                                     # axis the wall is perpendicular to
    sgn = (wall % 2).double() * 2 - 1                  # which of the two walls
    hv = torch.tensor([hx, hy, hz], dtype=torch.float64)
    o0 = torch.tensor([1, 0, 0])[ax]                   # the two in-plane axes
    o1 = torch.tensor([2, 2, 1])[ax]
    xyz = torch.zeros(N, 3, dtype=torch.float64)
    idx = torch.arange(N)
    xyz[idx, ax] = sgn * hv[ax]
    xyz[idx, o0] = a * hv[o0]
    xyz[idx, o1] = b * hv[o1]
    n = torch.zeros(N, 3, dtype=torch.float64)
    n[idx, ax] = -sgn                                  # into the room
    t1 = torch.zeros(N, 3, dtype=torch.float64)
    t1[idx, o0] = 1.0
    spin = 2 * math.pi * U(N)
    t2 = torch.linalg.cross(n, t1)
    a1 = torch.cos(spin)[:, None] * t1 + torch.sin(spin)[:, None] * t2
    a2 = torch.linalg.cross(n, a1)
    R = torch.stack([a1, a2, n], dim=-1)               # columns = local axes, det +1
    r = float(min(0.05, max(0.001, math.sqrt(float(areas.sum()) / N))))
    rr = r * (0.85 + 0.3 * U(N))
    scales = torch.stack([rr, rr * (0.8 + 0.2 * U(N)), 0.1 * rr], -1)
    opacity = torch.full((N, 1), 0.99, dtype=torch.float64)
    col = 0.5 + 0.35 * torch.stack([torch.sin(1.3 * xyz[:, 0] + 0.7 * xyz[:, 2]), torch.cos(1.1 * xyz[:, 1] - 0.4 * xyz[:, 0]),
                                    torch.sin(0.9 * xyz[:, 2] + 0.5 * xyz[:, 1])], -1)
    dc = ((col - 0.5) / SH_C0)[:, None, :]
    rest = 0.02 * torch.randn(N, 15, 3, generator=g, dtype=torch.float64)
    shs = torch.cat([dc, rest], dim=1)
    if c2w is not None:
        c2w = c2w.double()
        xyz = xyz @ c2w[:3, :3].t() + c2w[:3, 3]
        R = c2w[:3, :3] @ R
    out = dict(xyz=xyz, opacity=opacity, scales=scales, rotations=rotmat_to_quat(R), shs=shs)
    out = {k: t.to(torch.float32).contiguous() for k, t in out.items()}
    out["normal"] = normal_from_scale_rot(out["scales"], out["rotations"]).contiguous()
    return out


def look_at_pose(seed: int = 0, max_angle_deg: float = 10.0, max_trans: float = 0.2) -> torch.Tensor:
    """A random camera-to-world pose near identity (float64 4x4)."""
    g = torch.Generator().manual_seed(seed)
    w = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 2 * math.radians(max_angle_deg)
    t = (torch.rand(3, generator=g, dtype=torch.float64) - 0.5) * 2 * max_trans
    th = w.norm()
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
    Rm = torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * (K @ K)
    T = torch.eye(4, dtype=torch.float64)
    T[:3, :3] = Rm
    T[:3, 3] = t
    return T


# ---------------------------------------------------------------------------------------------
# analytic box-room RGB-D stream (ICP inputs): ray / axis-aligned-plane intersections
# ---------------------------------------------------------------------------------------------

def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    w, v = xi[:3].double(), xi[3:].double()
    th = w.norm()
    K = torch.tensor([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=torch.float64)
    T = torch.eye(4, dtype=torch.float64)
    if th < 1e-12:
        T[:3, 3] = v
        return T
    K2 = K @ K
    T[:3, :3] = torch.eye(3, dtype=torch.float64) + torch.sin(th) / th * K + (1 - torch.cos(th)) / th ** 2 * K2
    Vm = torch.eye(3, dtype=torch.float64) + (1 - torch.cos(th)) / th ** 2 * K + (th - torch.sin(th)) / th ** 3 * K2
    T[:3, 3] = Vm @ v
    return T


def box_room_depth(cam: CameraSpec, c2w: torch.Tensor, half=(2.5, 1.5, 3.0),
                   bump: float = 0.05, device=None) -> torch.Tensor:
    """Depth map [H,W,1] (camera z, metres, float32) of a camera inside an axis-aligned box
    |x|<hx, |y|<hy, |z|<hz whose walls carry a smooth sinusoidal relief (so ICP is
    well-conditioned in all 6 DoF).  `device`: where to evaluate it (default CPU: bit-reproducible across hosts; a long
    synthetic sequence renders its frames on the GPU)."""
    H, W = cam.H, cam.W
    c2w = c2w.double().to(device) if device is not None else c2w.double()
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=device), torch.arange(W, dtype=torch.float64, device=device), indexing="ij")
    rays_c = torch.stack([(xs - cam.cx) / cam.fx, (ys - cam.cy) / cam.fy, torch.ones_like(xs)], -1)
    rays_w = rays_c @ c2w[:3, :3].t()
    o = c2w[:3, 3]
    hv = torch.tensor(half, dtype=torch.float64, device=device)
    tbest = torch.full((H, W), float("inf"), dtype=torch.float64, device=device)
    for ax in range(3):
        for sgn in (-1.0, 1.0):
            d = rays_w[..., ax]
            t = (sgn * hv[ax] - o[ax]) / torch.where(d.abs() < 1e-12, torch.full_like(d, 1e-12), d)
            ok = (t > 1e-6) & (d * sgn > 0)
            hit = o + t[..., None] * rays_w
            oth = [a for a in range(3) if a != ax]
            # relief: push the wall in/out along the ray by a smooth function of the hit point
            rel = bump * (torch.sin(2.1 * hit[..., oth[0]] + 0.3 * ax) * torch.cos(1.7 * hit[..., oth[1]] - 0.2 * sgn))
            t2 = t * (1 - rel / (hv[ax] + 1.0))
            tbest = torch.where(ok & (t2 < tbest), t2, tbest)
    depth = tbest                                           # rays_c has z = 1 -> t is camera z
    depth = torch.where(torch.isfinite(depth), depth, torch.zeros_like(depth))
    return depth.to(torch.float32)[..., None].contiguous()


def box_room_color(cam: CameraSpec, c2w: torch.Tensor, depth: torch.Tensor) -> torch.Tensor:
    """Colour image [3,H,W] of the box room: a smooth function of the WORLD point every pixel sees (the same one
    `surface_gaussians` paints its discs with), so colour is consistent across views.  `depth` = box_room_depth (evaluated
    on the depth's device)."""
    H, W = cam.H, cam.W
    device = depth.device
    c2w = c2w.double().to(device)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64, device=device), torch.arange(W, dtype=torch.float64, device=device), indexing="ij")
    z = depth.reshape(H, W).double()
    pc = torch.stack([(xs - cam.cx) / cam.fx * z, (ys - cam.cy) / cam.fy * z, z], -1)
    pw = pc @ c2w[:3, :3].t() + c2w[:3, 3]
    col = 0.5 + 0.35 * torch.stack([torch.sin(1.3 * pw[..., 0] + 0.7 * pw[..., 2]), torch.cos(1.1 * pw[..., 1] - 0.4 * pw[..., 0]),
                                    torch.sin(0.9 * pw[..., 2] + 0.5 * pw[..., 1])], 0)
    return col.to(torch.float32).contiguous()


def tum_noise(depth: torch.Tensor, seed: int = 0, hole_frac: float = 0.05, scale: float = 5000.0) -> torch.Tensor:
    """sigma_z = 0.0012 + 0.0019 (z-0.4)^2, 5 % holes, 1/5000 m quantisation."""
    g = torch.Generator().manual_seed(seed)
    z = depth.double()
    sig = 0.0012 + 0.0019 * (z - 0.4) ** 2
    z = z + sig * torch.randn(z.shape, generator=g, dtype=torch.float64)
    z = torch.round(z * scale) / scale
    holes = torch.rand(z.shape, generator=g, dtype=torch.float64) < hole_frac
    z = torch.where(holes | (depth <= 0), torch.zeros_like(z), z)
    return z.to(torch.float32)


def room_tour(n_frames: int, seed: int = 0, half=(2.5, 1.5, 3.0), max_trans: float = 0.02, max_rot_deg: float = 1.0):
    """A bounded camera-to-world trajectory for LONG sequences in the box room (BASELINE configs[2] is a 2000-frame
    sequence; `trajectory` drifts out of the room after ~150 frames): the position follows a Lissajous curve well inside
    the walls, the camera pans steadily and nods, and every frame moves <= max_trans and turns <= max_rot_deg."""
    g = torch.Generator().manual_seed(seed)
    ph = (torch.rand(3, generator=g, dtype=torch.float64) * 2 * math.pi).tolist()
    amp = [0.45 * half[0], 0.3 * half[1], 0.45 * half[2]]
    w = [1.0, 0.7, 1.3]
    # angular speed of the curve so that |dp/di| <= 0.8 max_trans: |dp/di| <= k * sqrt(sum (amp w)^2)
    k = 0.8 * max_trans / math.sqrt(sum((a * ww) ** 2 for a, ww in zip(amp, w)))
    yaw_rate = math.radians(max_rot_deg) * 0.6
    poses = []
    yaw = 0.0
    for i in range(n_frames):
        p = [amp[c] * math.sin(w[c] * k * i + ph[c]) for c in range(3)]
        yaw += yaw_rate * (0.75 + 0.25 * math.sin(0.01 * i))
        pitch = math.radians(12.0) * math.sin(0.013 * i + ph[0])
        cy, sy, cp, sp = math.cos(yaw), math.sin(yaw), math.cos(pitch), math.sin(pitch)
        Ry = torch.tensor([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=torch.float64)      # pan about the world's vertical (y down)
        Rx = torch.tensor([[1, 0, 0], [0, cp, -sp], [0, sp, cp]], dtype=torch.float64)
        T = torch.eye(4, dtype=torch.float64)
        T[:3, :3] = Ry @ Rx
        T[:3, 3] = torch.tensor(p, dtype=torch.float64)
        poses.append(T)
    return poses


def trajectory(n_frames: int, seed: int = 0, max_trans: float = 0.02, max_rot_deg: float = 1.0):
    """Smooth 6-DoF camera-to-world trajectory, <= 2 cm and <= 1 degree per frame."""
    g = torch.Generator().manual_seed(seed)
    dirs = torch.randn(6, generator=g, dtype=torch.float64)
    dirs[:3] = dirs[:3] / dirs[:3].norm() * math.radians(max_rot_deg) * 0.8
    dirs[3:] = dirs[3:] / dirs[3:].norm() * max_trans * 0.8
    poses = [torch.eye(4, dtype=torch.float64)]
    for i in range(1, n_frames):
        wob = 0.6 + 0.4 * math.sin(0.37 * i)
        poses.append(poses[-1] @ se3_exp(dirs * wob))
    return poses
