"""CPU oracle for the ICP tracker: a plain-PyTorch (float32, CPU) restatement of
/root/reference/SLAM/icp.py and the helpers it pulls from SLAM/utils.py.

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product (rtg_slam_amd/).

PARITY PINNED: the reference's ICP is importable in the build container through a stub shim
(oracle/gen_icp_golden.py).  That script ran the reference code itself on seeded inputs and
committed its outputs under tests/golden/icp_*.npz; tests/test_oracle_icp.py checks every
function below against those vectors.

Function <- reference
  vertex_map            <- compute_vertex_map           SLAM/utils.py:65-75
  sobel_gradients       <- feature_gradient             SLAM/utils.py:77-98 (normalize_gradient=False)
  normal_map            <- compute_normal_map           SLAM/utils.py:100-122
  depth_pyramid         <- ImagePyramids(...,'max')     SLAM/icp.py:337-355, :374
  vertex_pyramid        <- build_vertex_pyramid         SLAM/utils.py:511-521
  normal_pyramid        <- build_normal_pyramid         SLAM/utils.py:523-527
  nearest_sample        <- warp_features                SLAM/icp.py:132-148
  residuals_jacobian    <- ICP.compute_residuals_jacobian  SLAM/icp.py:52-104
  normal_equations      <- compute_jtj + compute_jtr    SLAM/icp.py:107-119
  gauss_newton_update   <- GN_solver / lev_mar_H / least_square_solve / exp_se3  SLAM/icp.py:122-129, 248-334
  icp_level             <- ICP.icp                      SLAM/icp.py:33-48
  track                 <- IcpTracker.predict_pose      SLAM/icp.py:417-452 (level loop + p2p loss)
  fill_model_depth      <- IcpTracker.update_last_status SLAM/icp.py:397-415
"""
from __future__ import annotations

import math
from typing import List, Sequence

import torch
import torch.nn.functional as F


def vertex_map(depth: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """depth [H,W,1] -> [H,W,3]: ((x-cx)/fx, (y-cy)/fy, 1) * d."""
    H, W = depth.shape[:2]
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    xs = torch.arange(W, dtype=torch.float32).reshape(1, W).expand(H, W)
    ys = torch.arange(H, dtype=torch.float32).reshape(H, 1).expand(H, W)
    rays = torch.stack([(xs - cx) / fx, (ys - cy) / fy, torch.ones(H, W)], dim=-1)
    return rays * depth.reshape(H, W, 1)


def sobel_gradients(img: torch.Tensor):
    """[H,W,C] -> (d/dx, d/dy) with 3x3 Sobel taps, replicate padding, no normalisation."""
    H, W, Cn = img.shape
    kx = torch.tensor([[-1., 0., 1.], [-2., 0., 2.], [-1., 0., 1.]]).reshape(1, 1, 3, 3)
    ky = torch.tensor([[-1., -2., -1.], [0., 0., 0.], [1., 2., 1.]]).reshape(1, 1, 3, 3)
    x = F.pad(img.permute(2, 0, 1).reshape(Cn, 1, H, W), (1, 1, 1, 1), mode="replicate")
    gx = F.conv2d(x, kx).reshape(Cn, H, W).permute(1, 2, 0)
    gy = F.conv2d(x, ky).reshape(Cn, H, W).permute(1, 2, 0)
    return gx, gy


def normal_map(vertex: torch.Tensor) -> torch.Tensor:
    H, W, _ = vertex.shape
    gx, gy = sobel_gradients(vertex)
    n = torch.linalg.cross(gy.reshape(-1, 3), gx.reshape(-1, 3)).reshape(H, W, 3)
    n = n / (n.norm(dim=-1, keepdim=True) + 1e-8)
    d = vertex[..., 2]
    bad = (d <= d.min()) | (d >= d.max())
    return torch.where(bad[..., None], torch.zeros_like(n), n)


def depth_pyramid(depth: torch.Tensor, levels: int) -> List[torch.Tensor]:
    H, W = depth.shape[:2]
    x = depth.reshape(1, 1, H, W)
    return [F.max_pool2d(x, 1 << (levels - 1 - l), 1 << (levels - 1 - l)) for l in range(levels)]


def vertex_pyramid(depth: torch.Tensor, K: torch.Tensor, levels: int = 3) -> List[torch.Tensor]:
    out = []
    for l, d in enumerate(depth_pyramid(depth, levels)):
        Hs, Ws = d.shape[2:]
        Kl = K * (1.0 / 2 ** (levels - 1 - l))
        Kl[2, 2] = 1.0
        out.append(vertex_map(d.reshape(Hs, Ws, 1), Kl))
    return out


def normal_pyramid(vp: Sequence[torch.Tensor]) -> List[torch.Tensor]:
    return [normal_map(v) for v in vp]


def nearest_sample(feat: torch.Tensor, u: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """grid_sample(nearest, border, align_corners=True) of a [H,W,C] map at pixel coords (u, v)."""
    H, W, Cn = feat.shape
    grid = torch.stack([u / ((W - 1) / 2) - 1, v / ((H - 1) / 2) - 1], dim=-1).reshape(1, H, W, 2)
    out = F.grid_sample(feat.permute(2, 0, 1)[None], grid, mode="nearest", padding_mode="border", align_corners=True)
    return out[0].permute(1, 2, 0)


def residuals_jacobian(v_src, v_tgt, n_src, n_tgt, pose, K, dist_thr, cos_thr):
    """Returns (res [HW], J [HW,6] ordered (rot, trs), valid [H,W])."""
    H, W, _ = v_src.shape
    R, t = pose[:3, :3], pose[:3, 3]
    p = (R @ v_src.reshape(-1, 3).t()).t().reshape(H, W, 3) + t
    n = (R @ n_src.reshape(-1, 3).t()).t().reshape(H, W, 3)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    u = (p[..., 0] / p[..., 2]) * fx + cx
    v = (p[..., 1] / p[..., 2]) * fy + cy
    inview = (u > 0) & (u < W - 1) & (v > 0) & (v < H - 1)
    q = nearest_sample(v_tgt, u, v)
    m = nearest_sample(n_tgt, u, v)
    diff = p - q
    res = (m * diff).sum(-1)
    J = torch.cat([torch.linalg.cross(p.reshape(-1, 3), m.reshape(-1, 3)), m.reshape(-1, 3)], dim=-1)
    valid = (inview & (diff.norm(dim=-1) <= dist_thr) & (v_src[..., 2] > 0) & (q[..., 2] > 0)
             & ((n * m).sum(-1) > cos_thr))
    res = torch.where(valid, res, torch.zeros_like(res)).reshape(-1)
    J = torch.where(valid.reshape(-1, 1), J, torch.zeros_like(J))
    return res, J, valid


def normal_equations(J: torch.Tensor, res: torch.Tensor):
    """compute_jtj / compute_jtr (SLAM/icp.py:107-119), literally: the per-pixel 6x6 (6x1) products are materialised
    by bmm and summed by torch.sum over the pixel axis - the same float32 reduction the reference performs (a J^T J
    GEMM sums in another order: 4e-5 on the final pose of the noisy full-size frame)."""
    jac = J.reshape(-1, 1, 6)
    jacT = jac.transpose(-1, -2)
    jtj = torch.bmm(jacT, jac).sum(0)
    jtr = torch.bmm(jacT, res.reshape(-1, 1, 1)).sum(0)
    return jtj, jtr.reshape(6)


def se3_exp(xi: torch.Tensor) -> torch.Tensor:
    w, v = xi[:3], xi[3:]
    Wh = torch.zeros(3, 3, dtype=xi.dtype)
    Wh[0, 1], Wh[0, 2], Wh[1, 0], Wh[1, 2], Wh[2, 0], Wh[2, 1] = -w[2], w[1], w[2], -w[0], -w[1], w[0]
    W2 = Wh @ Wh
    th = torch.linalg.norm(w)
    I = torch.eye(3, dtype=xi.dtype)
    if th <= 1e-8:
        Rm, Jl = I, I
    else:
        Rm = I + Wh * torch.sin(th) / th + W2 * (1.0 - torch.cos(th)) / th ** 2
        Jl = I + Wh * (1 - torch.cos(th)) / th ** 2 + W2 * (th - torch.sin(th)) / th ** 3
    T = torch.eye(4, dtype=xi.dtype)
    T[:3, :3] = Rm
    T[:3, 3] = Jl @ v
    return T


def gauss_newton_update(JtJ, Jtr, pose, damping):
    Hm = JtJ + torch.eye(6, dtype=JtJ.dtype) * (torch.trace(JtJ) * damping)
    xi = -(torch.inverse(Hm) @ Jtr.reshape(6, 1)).reshape(6)
    return se3_exp(xi) @ pose


def icp_level(pose, v_src, v_tgt, n_src, n_tgt, K, iters, dist_thr, cos_thr, damping, exact_sums=False):
    """`exact_sums`: the per-pixel arithmetic (projection, association, gates, residual, Jacobian) stays the
    reference's float32, but the 27 sums over the pixels and the 6x6 solve are done in float64 - the reference
    algorithm without the rounding noise of its own float32 reductions (which an ill-conditioned, noisy frame
    amplifies).  The HIP tracker reduces in float64 too, so this is the variant it should match most closely."""
    valid = None
    for _ in range(iters):
        res, J, valid = residuals_jacobian(v_src, v_tgt, n_src, n_tgt, pose, K, dist_thr, cos_thr)
        if exact_sums:
            JtJ, Jtr = normal_equations(J.double(), res.double())
            pose = gauss_newton_update(JtJ, Jtr, pose.double(), damping).float()
        else:
            JtJ, Jtr = normal_equations(J, res)
            pose = gauss_newton_update(JtJ, Jtr, pose, damping)
    H, W = v_src.shape[:2]
    return pose, valid.sum() / H / W


def p2p_loss(p_t0, p_t1, n_t0):
    l = ((p_t1 - p_t0) * n_t0).sum(-1)
    return (l * l).mean()


def track(vp_t1, np_t1, vp_t0, np_t0, K, downscales=(0.25, 0.5, 1.0), iters=(5, 5, 5),
          dist_thr=0.1, normal_thr_deg=20.0, damping=1e-4, exact_sums=False):
    """Level loop + loss of predict_pose; source = current frame t1, target = t0 (icp.py:438-441)."""
    cos_thr = math.cos(math.radians(normal_thr_deg))
    pose = torch.eye(4, dtype=torch.float32)
    ratio = torch.tensor(0.0)
    for l, ds in enumerate(downscales):
        Kl = K * ds
        Kl[2, 2] = 1.0
        pose, ratio = icp_level(pose, vp_t1[l], vp_t0[l], np_t1[l], np_t0[l], Kl, iters[l], dist_thr, cos_thr, damping,
                                exact_sums)
    loss = p2p_loss(vp_t0[-1], vp_t1[-1] @ pose[:3, :3].t() + pose[:3, 3], np_t0[-1])
    return pose, float(ratio), float(loss)


def fill_model_depth(render_depth, frame_depth, render_normal, frame_normal, dist_thr, normal_thr):
    """Returns the filled copy of render_depth [H,W,1]."""
    nm = (1 - F.cosine_similarity(render_normal, frame_normal, dim=-1)) > normal_thr
    fill = (((render_depth - frame_depth).abs() > dist_thr)[..., 0] | (render_depth == 0)[..., 0] | nm) \
        & (frame_depth > 0)[..., 0]
    out = render_depth.clone()
    out[fill] = frame_depth[fill]
    return out
