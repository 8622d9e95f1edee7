"""CPU restatements (plain torch) of the callers and data producers either side of the hot path - SURVEY.md 8 rows R9
and (f-1)..(f-3).  TEST INFRASTRUCTURE ONLY: imported by tests/ (and bench.py's cpu_baseline leg), never by the product.

Pinned: every function that restates in-tree reference Python is checked against the reference itself, imported from
/root/reference through oracle/ref_shim.py, in tests/test_oracle_slam_ops.py (build container), and the vectors that
test produces travel to the GPU box as tests/golden/slam_ops.npz (oracle/gen_slam_ops_golden.py).

Unpinned (source absent from the reference tree; semantics FROZEN here from the call sites):
  * simple_knn._C.distCUDA2          - un-vendored submodule; call site SLAM/gaussian_pointcloud.py:376-389
  * cuda_utils._C.accumulate_gaussian_error - un-vendored submodule; call site SLAM/multiprocess/mapper.py:541-565
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F


# ------------------------------------------------------------------ R9: tile-mask producers (SLAM/utils.py:681-734)
def _pad_to(x: torch.Tensor, stride: int, value=0):
    H, W = x.shape[:2]
    return F.pad(x, (0, (W + stride - 1) // stride * stride - W, 0, (H + stride - 1) // stride * stride - H), value=value)


def pixelmask2tilemask(pixelmask: torch.Tensor, stride: int = 16) -> torch.Tensor:
    """SLAM/utils.py:681-692: a tile is on if ANY of its pixels is (max-pool)."""
    p = _pad_to(pixelmask, stride)[None, None].float()
    return F.max_pool2d(p, stride, stride)[0, 0].int()


def transmission2tilemask(pixelmask: torch.Tensor, stride: int = 16, tile_mask_ratio: float = 0.5) -> torch.Tensor:
    """SLAM/utils.py:695-705: a tile is on if more than `tile_mask_ratio` of its (zero-padded) pixels are."""
    p = _pad_to(pixelmask, stride)[None, None].float()
    return (F.avg_pool2d(p, stride, stride)[0, 0] > tile_mask_ratio).int()


def colorerror2tilemask(color_error: torch.Tensor, stride: int = 16, top_ratio: float = 0.4) -> torch.Tensor:
    """SLAM/utils.py:708-734: the int(tiles * top_ratio) tiles with the largest mean (zero-padded) error."""
    p = _pad_to(color_error, stride, value=0)[None, None].float()
    mean = F.avg_pool2d(p, stride, stride)[0, 0]
    k = int(mean.numel() * top_ratio)
    _, idx = torch.topk(mean.reshape(-1), k=k)
    out = torch.zeros(mean.numel(), dtype=torch.int32)
    out[idx] = 1
    return out.reshape(mean.shape)


def tile_mean(x: torch.Tensor, stride: int = 16) -> torch.Tensor:
    return F.avg_pool2d(_pad_to(x, stride, value=0)[None, None].float(), stride, stride)[0, 0]


# ------------------------------------------------------------------ (f-1) simple_knn.distCUDA2  [FROZEN, unpinned]
def dist2_knn(points: torch.Tensor, chunk: int = 2048):
    """For every point: squared distances to its three nearest OTHER points (ascending), their indices, and the mean
    of the three - what `simple_knn._C.distCUDA2` returns in RTG-SLAM's fork: `(mean_dist2[N], idx[N,3])`
    (gaussian_pointcloud.py:376-389 indexes `total_xyz[knn_indices[:, k]]`).  Brute force; d^2 = dx*dx + dy*dy + dz*dz
    in float32, in that order.  Fewer than three other points: the missing distances are FLT_MAX, indices -1."""
    N = points.shape[0]
    p = points.float()
    best_d = torch.full((N, 3), torch.finfo(torch.float32).max)
    best_i = torch.full((N, 3), -1, dtype=torch.int64)
    x, y, z = p[:, 0], p[:, 1], p[:, 2]
    for s in range(0, N, chunk):
        e = min(N, s + chunk)
        dx, dy, dz = x[s:e, None] - x[None, :], y[s:e, None] - y[None, :], z[s:e, None] - z[None, :]
        d = dx * dx + dy * dy + dz * dz
        d[torch.arange(e - s), torch.arange(s, e)] = float("inf")
        k = min(3, N - 1)
        if k > 0:
            v, i = torch.topk(d, k=k, dim=1, largest=False)
            best_d[s:e, :k], best_i[s:e, :k] = v, i
    return best_d.sum(1) / 3.0, best_i.int(), best_d


def knn_query(ref: torch.Tensor, query: torch.Tensor, self_offset: int = -1, ref_box=None, chunk: int = 2048):
    """Three nearest REFERENCE points of every query point, brute force: (dist2 [Nq,3] ascending, idx [Nq,3]) - what
    pytorch3d.ops.knn_points(query[None], ref[None], K=3) returns (squared distances; mapper.py:811-818 takes their sqrt),
    with the float32 distance formula of dist2_knn.  self_offset >= 0: query i is ref[self_offset + i] and is skipped
    (the new-point rows of update_geometry's distCUDA2, gaussian_pointcloud.py:366-381)."""
    Nq, Nr = query.shape[0], ref.shape[0]
    q, r = query.float(), ref.float()
    best_d = torch.full((Nq, 3), torch.finfo(torch.float32).max)
    best_i = torch.full((Nq, 3), -1, dtype=torch.int64)
    for s in range(0, Nq, chunk):
        e = min(Nq, s + chunk)
        dx, dy, dz = q[s:e, 0, None] - r[None, :, 0], q[s:e, 1, None] - r[None, :, 1], q[s:e, 2, None] - r[None, :, 2]
        d = dx * dx + dy * dy + dz * dz
        if self_offset >= 0:
            d[torch.arange(e - s), self_offset + torch.arange(s, e)] = float("inf")
        if ref_box is not None:                                   # bbox_filter (SLAM/utils.py:737-744): strict inequalities
            inb = (r > ref_box[:3]).all(-1) & (r < ref_box[3:]).all(-1)
            d[:, ~inb] = float("inf")
        k = min(3, Nr)
        if k > 0:
            v, i = torch.topk(d, k=k, dim=1, largest=False)
            ok = torch.isfinite(v)
            best_d[s:e, :k] = torch.where(ok, v, best_d[s:e, :k])
            best_i[s:e, :k] = torch.where(ok, i, best_i[s:e, :k])
    return best_d, best_i.int()


# ------------------------------------------------------------------ per-frame masks / errors of the mapper (mapper.py)
def add_masks(T_map, depth, render_depth, render_color, frame_color, depth_index, thr_T, thr_d, thr_c):
    """Mapping.temp_points_init, mapper.py:728-768, line by line on [H,W,C] maps.  Inputs: T_map / render_depth /
    depth_index [1,H,W], depth [H,W,1], colours [3,H,W].  Returns (transmission mask, error mask) bool [H,W], counts."""
    T = T_map.permute(1, 2, 0)
    rd, di = render_depth.permute(1, 2, 0), depth_index.permute(1, 2, 0)
    rc, fc = render_color.permute(1, 2, 0), frame_color.permute(1, 2, 0)
    d = depth.reshape(rd.shape)
    transmission_sample_mask = (T > thr_T) & (d > 0)
    depth_error = torch.abs(d - rd)
    color_error = torch.abs(fc - rc).mean(dim=-1, keepdim=True)
    depth_sample_mask = (depth_error > thr_d) & (d > 0) & (di > -1)
    color_sample_mask = (color_error > thr_c) & (d > 0) & (T < thr_T)
    sample_mask = (color_sample_mask | depth_sample_mask) & (~transmission_sample_mask)
    tm, em = transmission_sample_mask[..., 0], sample_mask[..., 0]
    return tm, em, torch.stack([tm.sum(), em.sum()]).int()


def frame_errors(depth, render_depth, render_color, frame_color, depth_index):
    """Mapping.error_gaussians_remove, mapper.py:527-540 -> (color_error [H,W], depth_error [H,W])."""
    rd, di = render_depth.permute(1, 2, 0), depth_index.permute(1, 2, 0)
    color, cm_color = render_color.permute(1, 2, 0), frame_color.permute(1, 2, 0)
    d = depth.reshape(rd.shape)
    depth_error = torch.abs(d - rd)
    depth_error[(d - rd) < 0] = 0
    image_error = torch.abs(cm_color - color)
    color_error = torch.sum(image_error, dim=-1, keepdim=True)
    invalid_mask = ((d == 0) | (di == -1)).squeeze()
    depth_error[invalid_mask] = 0
    color_error[d == 0] = 0
    return color_error[..., 0], depth_error[..., 0]


def attach_test(points, w2c, fx, fy, cx, cy, H, W, stable_color_index, stable_xyz, stable_normal, max_plane_dist):
    """Mapping.temp_points_attach, mapper.py:838-872 (all new points have opacity > 0.1), as a flag per point."""
    K = torch.tensor([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], dtype=torch.float32)
    xyz_c = points @ w2c[:3, :3].T + w2c[:3, 3]
    uv = xyz_c @ K.T
    uv = (uv[:, :2] / uv[:, 2:]).long()
    inside = (uv[:, 0] >= 0) & (uv[:, 0] < W) & (uv[:, 1] >= 0) & (uv[:, 1] < H)
    out = torch.zeros(points.shape[0], dtype=torch.uint8)
    idx = torch.nonzero(inside).reshape(-1)
    if idx.numel() == 0 or stable_xyz.shape[0] == 0:
        return out
    sidx = stable_color_index.reshape(H, W)[uv[idx, 1], uv[idx, 0]].long()
    hit = sidx >= 0
    idx, sidx = idx[hit], sidx[hit]
    p2p = ((stable_xyz[sidx] - points[idx]) * stable_normal[sidx]).sum(dim=-1)
    out[idx[p2p.abs() < max_plane_dist]] = 1
    return out


# ------------------------------------------------------------------ (f-1) cuda_utils.accumulate_gaussian_error  [FROZEN]
def accumulate_gaussian_error(H, W, P, color_err, depth_err, normal_err, color_index, depth_index, thr_c, thr_d, thr_n,
                              mean=True):
    """FROZEN definition (mapper.py:541-565 is the only evidence): colour error is attributed to the Gaussian in
    `color_index`, depth and normal error to the Gaussian in `depth_index`; index -1 = nobody.  Per Gaussian: the mean
    (flag True; the sum if False) of each error over the pixels attributed to it, 0 where none, and
    outlier_count = number of attributed pixels whose error exceeds its threshold (colour, depth and normal counted
    together).  The caller compares the means with 2x the add_*_thres (mapper.py:567-571)."""
    ce, de, ne = color_err.reshape(-1).float(), depth_err.reshape(-1).float(), normal_err.reshape(-1).float()
    ci, di = color_index.reshape(-1).long(), depth_index.reshape(-1).long()
    sums = [torch.zeros(P) for _ in range(3)]
    cnts = [torch.zeros(P) for _ in range(2)]
    out = torch.zeros(P, dtype=torch.int32)
    mc, md = ci >= 0, di >= 0
    sums[0].index_add_(0, ci[mc], ce[mc]); cnts[0].index_add_(0, ci[mc], torch.ones(int(mc.sum())))
    sums[1].index_add_(0, di[md], de[md]); sums[2].index_add_(0, di[md], ne[md])
    cnts[1].index_add_(0, di[md], torch.ones(int(md.sum())))
    out.index_add_(0, ci[mc], (ce[mc] > thr_c).int())
    out.index_add_(0, di[md], (de[md] > thr_d).int() + (ne[md] > thr_n).int())
    if mean:
        return (sums[0] / cnts[0].clamp_min(1), sums[1] / cnts[1].clamp_min(1), sums[2] / cnts[1].clamp_min(1), out)
    return sums[0], sums[1], sums[2], out


# ------------------------------------------------------------------ (f-3) frame preprocessing (tracker.py:97-159)
def bilateral_filter(depth: torch.Tensor, radius: int, sigma_color: float, sigma_space: float) -> torch.Tensor:
    """SLAM/utils.py:550-589: taps inside the disc i^2 + j^2 <= radius^2, weight exp(spatial + range) where the range
    term is the squared depth difference to the CENTRE, taps whose depth is 0 ignored, zero padding, 0 where no tap
    counted."""
    h, w = depth.shape[:2]
    d = depth.reshape(h, w).float()
    pad = F.pad(d, (radius, radius, radius, radius))
    wsum, psum = torch.zeros_like(d), torch.zeros_like(d)
    for i in range(-radius, radius + 1):
        for j in range(-radius, radius + 1):
            if i * i + j * j > radius * radius:
                continue
            tap = pad[radius + i:radius + i + h, radius + j:radius + j + w]
            wgt = torch.exp(-(i * i + j * j) / (2 * sigma_space ** 2) + -((d - tap) ** 2) / (2 * sigma_color ** 2))
            wgt = wgt * (tap != 0)
            wsum += wgt
            psum += wgt * tap
    out = psum / wsum
    out[wsum == 0] = 0
    return out.reshape(h, w, 1)


def confidence_map(normal_map: torch.Tensor, K: torch.Tensor) -> torch.Tensor:
    """SLAM/utils.py:124-139: |cos| between the normal and the (normalised, +1e-8) viewing ray of the pixel."""
    H, W, _ = normal_map.shape
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    ray = torch.ones(H, W, 3)
    ray[..., 0] = (xs - K[0, 2]) / K[0, 0]
    ray[..., 1] = (ys - K[1, 2]) / K[1, 1]
    ray = ray / (ray.norm(dim=-1, keepdim=True) + 1e-8)
    return F.cosine_similarity(normal_map, ray, dim=-1).abs()[..., None]


def frame_preprocess(depth_map, K, min_depth=0.3, max_depth=5.0, depth_filter=False, invalid_confidence_thresh=0.2):
    """The map part of Tracker.map_preprocess (tracker.py:104-131): optional bilateral filter (radius 5, sigma 2 / 2),
    range mask, vertex / normal / confidence maps, and the invalid-confidence mask zeroing all four."""
    from oracle import icp_oracle as io
    d = bilateral_filter(depth_map, 5, 2, 2) if depth_filter else depth_map.clone().float()
    d[~((d > min_depth) & (d < max_depth))] = 0.0
    vertex = io.vertex_map(d, K)
    normal = io.normal_map(vertex)
    conf = confidence_map(normal, K)
    bad = (normal == 0).all(dim=-1) | (conf < invalid_confidence_thresh)[..., 0]
    d, normal, vertex, conf = d.clone(), normal.clone(), vertex.clone(), conf.clone()
    d[bad] = 0
    normal[bad] = 0
    vertex[bad] = 0
    conf[bad] = 0
    return dict(depth_map=d, normal_map_c=normal, vertex_map_c=vertex, confidence_map=conf, invalid_confidence_mask=bad)


def sample_pixels_mask(normal_map: torch.Tensor, select_mask: torch.Tensor | None) -> torch.Tensor:
    """The deterministic half of sample_pixels (SLAM/utils.py:141-183): which pixels are candidates - select_mask
    (all ones if None) minus the pixels whose normal sums to exactly 0.  The other half is a uniform draw without
    replacement (torch.randperm) of min(n, #candidates) of them."""
    H, W = normal_map.shape[:2]
    m = torch.ones(H, W, dtype=torch.bool) if select_mask is None else select_mask.reshape(H, W).bool().clone()
    m[normal_map.sum(dim=-1) == 0] = False
    return m


# ------------------------------------------------------------------ (f-2) loss of one optimisation step (mapper.py:371-445)
def _gauss_window(size=11, sigma=1.5):
    g = torch.tensor([math.exp(-((x - size // 2) ** 2) / float(2 * sigma ** 2)) for x in range(size)])
    g = g / g.sum()
    return (g[:, None] @ g[None, :]).float()


def ssim(img1: torch.Tensor, img2: torch.Tensor, window_size: int = 11) -> torch.Tensor:
    """utils/loss_utils.py:58-100: 11x11 Gaussian window (sigma 1.5), zero padding, per channel, mean of the map."""
    C = img1.shape[-3]
    win = _gauss_window(window_size).to(img1.dtype)[None, None].expand(C, 1, window_size, window_size).contiguous()
    a, b = img1.reshape(1, C, *img1.shape[-2:]), img2.reshape(1, C, *img2.shape[-2:])
    pad = window_size // 2
    conv = lambda t: F.conv2d(t, win, padding=pad, groups=C)
    mu1, mu2 = conv(a), conv(b)
    s11, s22, s12 = conv(a * a) - mu1 * mu1, conv(b * b) - mu2 * mu2, conv(a * b) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s11 + s22 + C2))
    return m.mean()


def slam_loss(render, gt_color, gt_depth, gt_normal=None, render_mask=None, color_weight=0.8, depth_weight=1.0,
              ssim_weight=0.2, normal_weight=0.0, add_depth_thres=0.1, render_normal=None):
    """The image terms of Mapping.loss_update (mapper.py:402-448); `render` = the rasterizer's 7-tuple, channels first;
    gt_color [3,H,W], gt_depth [1,H,W], render_mask bool [H,W] or None.  Returns (total, dict of terms).
      * render_mask None -> all pixels, and the SSIM term 1 - ssim(render, gt) is live (:411-415);
      * colour: L1 mean over the masked pixels (x3 channels) (:421);
      * depth: mean |D - D_gt| over depth_index != -1 & D_gt > 0 & (D - D_gt) < add_depth_thres & render_mask (:423-431)
        - the threshold is on the SIGNED error, as the reference writes it; nan if the mask is empty, as torch's mean;
      * normal: mean (1 - cos) over render_mask & depth_index != -1 & gt normal != 0 (:433-442), only if weighted."""
    color, depth, didx = render[0], render[1], render[3]
    H, W = color.shape[-2:]
    ssim_loss = color.new_zeros(())
    if render_mask is None:
        mask = torch.ones(H, W, dtype=torch.bool)
        ssim_loss = 1 - ssim(color, gt_color)
    else:
        mask = render_mask.bool()
    color_loss = (color - gt_color).abs()[:, mask].mean()
    depth_loss = color.new_zeros(())
    if depth_weight > 0:
        err = depth[0] - gt_depth[0]
        vm = (didx[0] != -1) & (gt_depth[0] > 0) & (err < add_depth_thres) & mask
        depth_loss = err.abs()[vm].mean()
    normal_loss = color.new_zeros(())
    if normal_weight > 0 and render_normal is not None and gt_normal is not None:
        cosd = 1 - F.cosine_similarity(render_normal, gt_normal, dim=0)
        vn = mask & (didx[0] != -1) & ~((gt_normal == 0).all(dim=0))
        normal_loss = cosd[vn].mean()
    total = depth_weight * depth_loss + normal_weight * normal_loss + color_weight * color_loss + ssim_weight * ssim_loss
    return total, dict(color=color_loss, depth=depth_loss, ssim=ssim_loss, normal=normal_loss)


def attach_loss(opacity_act, scaling, xyz, rotation, init_scaling, init_xyz, init_rotation):
    """mapper.py:384-401: Gaussians whose activated opacity is below 0.9 are tied to their initial raw scaling, position
    and raw rotation: 1000 x (mean-squared difference of each), means over the selected rows x columns."""
    m = (opacity_act < 0.9).reshape(-1)
    if int(m.sum()) == 0:
        return xyz.new_zeros(())
    l2 = lambda a, b: ((a - b) ** 2).mean()
    return 1000 * (l2(scaling[m], init_scaling[m]) + l2(xyz[m], init_xyz[m]) + l2(rotation[m], init_rotation[m]))


def slerp(v0: torch.Tensor, v1: torch.Tensor, t: torch.Tensor, dot_threshold: float = 0.9995) -> torch.Tensor:
    """SLAM/utils.py:593-651 for v0, v1 [N,4] and t [N,1]: spherical interpolation, linear where the (re-normalised)
    vectors are colinear (|dot| > 0.9995) or the dot product is NaN."""
    a = v0 / torch.norm(v0, dim=-1, keepdim=True)
    b = v1 / torch.norm(v1, dim=-1, keepdim=True)
    dot = (a * b).sum(-1)
    lerp = dot.abs().isnan() | (dot.abs() > dot_threshold)
    th0 = dot.arccos().unsqueeze(-1)
    tht = th0 * t
    sl = (th0 - tht).sin() / th0.sin() * v0 + tht.sin() / th0.sin() * v1
    return torch.where(lerp.unsqueeze(-1), torch.lerp(v0, v1, t), sl)


def history_merge(xyz, shs, raw8, then_xyz, then_shs, then_raw8, conf_then, conf_now, max_weight: float = 0.5):
    """Mapping.history_merge (mapper.py:212-251) on the block-SoA map: xyz [N,3], shs [N,48], raw8 [N,8] = opacity |
    scaling | rotation (raw), confidences [N,1].  Returns the merged (xyz, shs, raw8).  Faithful to the reference's
    `history_weight[0]`: features and scaling of EVERY row are blended with the weight of row 0; xyz and the rotation
    use the row's own weight; the opacity is not merged; the rotation is slerp(get_rotation then, get_rotation now,
    1 - w) with get_rotation = F.normalize (gaussian_pointcloud.py:19, 519-521)."""
    if max_weight <= 0:
        return xyz, shs, raw8
    w = max_weight * conf_then / (conf_now + 1e-6)                     # [N,1]
    out_xyz = then_xyz * w + (1 - w) * xyz
    out_shs = then_shs * w[0] + (1 - w[0]) * shs
    out_raw8 = raw8.clone()
    out_raw8[:, 1:4] = then_raw8[:, 1:4] * w[0] + (1 - w[0]) * raw8[:, 1:4]
    out_raw8[:, 4:8] = slerp(F.normalize(then_raw8[:, 4:8]), F.normalize(raw8[:, 4:8]), 1 - w)
    return out_xyz, out_shs, out_raw8
