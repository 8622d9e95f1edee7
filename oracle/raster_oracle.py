"""CPU oracle for the RTG-SLAM rasterizer (`diff_gaussian_rasterization_depth`).

TEST INFRASTRUCTURE ONLY.  Nothing under ``rtg_slam_amd/`` or
``diff_gaussian_rasterization_depth/`` may import this module; only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg do, as the
checker / the timed CPU baseline - never as the product path.

PARITY UNPINNED.  The reference's rasterizer source is an empty, un-vendored git
submodule (/root/reference/.gitmodules:1-4 -> CJAPPLE5/RTG-SLAM-cuda_utils,
branch main, no pinned commit) and the reference holds no tests or golden
vectors for it.  This file therefore restates (a) the *contract* the in-tree
call sites pin, and (b) the frozen arithmetic of SURVEY.md Appendix B:

  contract   SLAM/render.py:68-88   19 settings fields
             SLAM/render.py:110-120 9 call arguments
             SLAM/render.py:122-128 7 outputs, their order
             SLAM/render.py:101-108 16x16 tiles, tile_mask int32[ceil(H/16), ceil(W/16)]
  sentinels  mapper.py:501,504 (T_map == 1 untouched), render.py:131 (index -1),
             icp.py:410 (depth 0 = no surface), mapper.py:455 (exact-zero grads)
  math       utils/general_utils.py:108-150 quaternion (w,x,y,z) -> R, Sigma = (R S)(R S)^T
             utils/sh_utils.py:26-128       SH basis, +0.5 offset
             utils/graphics_utils.py:66-90  fx = W / (2 tanfovx)
             scene/cameras.py:96-111        viewmatrix = W2C^T (row-vector convention)

Everything is plain PyTorch on CPU (float32 by default, float64 for finite
difference checks); the backward pass is autograd of this forward.

Frozen decisions (each one is a choice, because the source is absent):
  * cull p_c.z <= 0.2; Sigma2D dilation +0.3; radius = ceil(color_sigma*sqrt(lambda_max))
  * pixel centre u = fx*x/z + cx (cx <= 0 -> (W-1)/2, same for cy)
  * the 1.3*tanfov clamp of t.x/t.z, t.y/t.z makes t.x / t.y constants in the
    backward pass (upstream 3DGS behaviour)
  * alpha = min(0.99, o*G); skip power > 0 or alpha < 1/255; stop BEFORE the
    Gaussian that would drive T below T_threshold; in the backward pass the
    clamp is transparent (d alpha / d(oG) = 1 also where oG > 0.99), as in the
    upstream 3DGS backward the module derives from
  * colour index/weight = first arg-max of alpha*T over blended Gaussians
  * depth = ray/plane intersection of the FIRST blended Gaussian with
    alpha > opaque_threshold, |cos(ray, n)| > normal_threshold, z_hit > 0 and
    |z_hit - z_mu| < depth_threshold
  * sort key (tile, f32 depth bits); ties keep Gaussian-index order (stable)
  * quaternions are used as given (the caller normalises; SLAM/gaussian_pointcloud.py:520)
"""
from __future__ import annotations

import math
from typing import NamedTuple, Optional

import torch

TILE = 16

SH_C0 = 0.28209479177387814
SH_C1 = 0.4886025119029199
SH_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
         -1.0925484305920792, 0.5462742152960396)
SH_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
         0.3731763325901154, -0.4570457994644658, 1.445305721320277,
         -0.5900435899266435)


class OracleSettings(NamedTuple):
    """Same 19 names as SLAM/render.py:68-88."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    opaque_threshold: float
    depth_threshold: float
    normal_threshold: float
    color_sigma: float
    prefiltered: bool
    debug: bool
    cx: float
    cy: float
    T_threshold: float


def quat_to_rotmat(q: torch.Tensor) -> torch.Tensor:
    """(w,x,y,z) -> R, utils/general_utils.py:108-131 without the normalisation."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y),
    ], dim=-1).reshape(-1, 3, 3)
    return R


def eval_sh_color(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """sh [P, M, 3] (coefficient-major, channel-minor), dirs [P,3] unit.
    utils/sh_utils.py:57-120 basis; returns SH + 0.5 before the clamp."""
    res = SH_C0 * sh[:, 0]
    if deg > 0:
        x, y, z = dirs[:, 0:1], dirs[:, 1:2], dirs[:, 2:3]
        res = res - SH_C1 * y * sh[:, 1] + SH_C1 * z * sh[:, 2] - SH_C1 * x * sh[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + SH_C2[0] * xy * sh[:, 4] + SH_C2[1] * yz * sh[:, 5]
                   + SH_C2[2] * (2.0 * zz - xx - yy) * sh[:, 6]
                   + SH_C2[3] * xz * sh[:, 7] + SH_C2[4] * (xx - yy) * sh[:, 8])
            if deg > 2:
                res = (res + SH_C3[0] * y * (3.0 * xx - yy) * sh[:, 9]
                       + SH_C3[1] * xy * z * sh[:, 10]
                       + SH_C3[2] * y * (4.0 * zz - xx - yy) * sh[:, 11]
                       + SH_C3[3] * z * (2.0 * zz - 3.0 * xx - 3.0 * yy) * sh[:, 12]
                       + SH_C3[4] * x * (4.0 * zz - xx - yy) * sh[:, 13]
                       + SH_C3[5] * z * (xx - yy) * sh[:, 14]
                       + SH_C3[6] * x * (xx - 3.0 * yy) * sh[:, 15])
    return res + 0.5


def preprocess(s: OracleSettings, means3D, opacities, shs, scales, rotations, normal_w):
    """Per-Gaussian stage. Differentiable."""
    dt = means3D.dtype
    H, W = int(s.image_height), int(s.image_width)
    fx = W / (2.0 * s.tanfovx)
    fy = H / (2.0 * s.tanfovy)
    cx = s.cx if s.cx > 0 else (W - 1) / 2.0
    cy = s.cy if s.cy > 0 else (H - 1) / 2.0
    V = s.viewmatrix.to(dt).t()                     # world -> camera, column-vector convention
    Wr = V[:3, :3]
    p_c = means3D @ Wr.t() + V[:3, 3]
    z = p_c[:, 2]
    valid = z > 0.2

    R = quat_to_rotmat(rotations)
    M = R * (scales * s.scale_modifier)[:, None, :]  # R @ diag(s)
    Sigma = M @ M.transpose(1, 2)

    zs = torch.where(valid, z, torch.ones_like(z))   # keep culled rows finite
    limx, limy = 1.3 * s.tanfovx, 1.3 * s.tanfovy
    txtz, tytz = p_c[:, 0] / zs, p_c[:, 1] / zs
    in_x = (txtz >= -limx) & (txtz <= limx)
    in_y = (tytz >= -limy) & (tytz <= limy)
    tx = torch.where(in_x, p_c[:, 0], (txtz.clamp(-limx, limx) * zs).detach())
    ty = torch.where(in_y, p_c[:, 1], (tytz.clamp(-limy, limy) * zs).detach())
    zero = torch.zeros_like(zs)
    J = torch.stack([
        torch.stack([fx / zs, zero, -fx * tx / (zs * zs)], dim=-1),
        torch.stack([zero, fy / zs, -fy * ty / (zs * zs)], dim=-1),
    ], dim=1)                                        # [P,2,3]
    Tm = J @ Wr                                      # [P,2,3]
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    valid = valid & (det != 0)
    dets = torch.where(det != 0, det, torch.ones_like(det))
    conic = torch.stack([c / dets, -b / dets, a / dets], dim=-1)
    mid = 0.5 * (a + c)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(s.color_sigma * torch.sqrt(lam)).detach()
    u = fx * p_c[:, 0] / zs + cx
    v = fy * p_c[:, 1] / zs + cy

    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    ud, vd = u.detach(), v.detach()
    big = 1 << 28
    def tdiv(t):  # C (int)(float) truncation, saturating
        return torch.trunc(t / TILE).clamp(-big, big).to(torch.int64)
    rminx = tdiv(ud - radius).clamp(0, gx)
    rmaxx = tdiv(ud + radius + (TILE - 1)).clamp(0, gx)
    rminy = tdiv(vd - radius).clamp(0, gy)
    rmaxy = tdiv(vd + radius + (TILE - 1)).clamp(0, gy)
    valid = valid & ((rmaxx - rminx) * (rmaxy - rminy) > 0)

    d = means3D - s.campos.to(dt)
    dirs = d / d.norm(dim=-1, keepdim=True)
    rgb_raw = eval_sh_color(int(s.sh_degree), shs, dirs)
    rgb = rgb_raw.clamp_min(0.0)

    n_c = normal_w @ Wr.t()
    plane_d = (n_c * p_c).sum(-1)
    return dict(p_c=p_c, depth=z, valid=valid, conic=conic, u=u, v=v, radius=radius,
                rect=(rminx, rminy, rmaxx, rmaxy), rgb=rgb, n_c=n_c, plane_d=plane_d,
                opacity=opacities.reshape(-1), fx=fx, fy=fy, cx=cx, cy=cy, gx=gx, gy=gy)


def bin_tiles(pre, tile_mask: torch.Tensor):
    """Instances (Gaussian, tile) for mask != 0 tiles, sorted by (tile, f32 depth bits),
    ties in Gaussian-index order (stable sort over index-ordered emission)."""
    gx, gy = pre["gx"], pre["gy"]
    rminx, rminy, rmaxx, rmaxy = pre["rect"]
    valid = pre["valid"]
    idx = torch.nonzero(valid).reshape(-1)
    w = (rmaxx - rminx)[idx]
    h = (rmaxy - rminy)[idx]
    cnt = w * h
    total = int(cnt.sum())
    if total == 0:
        e = torch.zeros(0, dtype=torch.int64)
        return e, e, torch.zeros(gx * gy, 2, dtype=torch.int64)
    gid = torch.repeat_interleave(idx, cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(total) - torch.repeat_interleave(start, cnt)
    wrep = torch.repeat_interleave(w, cnt)
    ty = rminy[gid] + local // wrep
    tx = rminx[gid] + local % wrep
    tile = ty * gx + tx
    keep = tile_mask.reshape(-1)[tile] != 0
    gid, tile = gid[keep], tile[keep]
    depth_bits = pre["depth"].detach().to(torch.float32)[gid].view(torch.int32).to(torch.int64)
    key = (tile << 32) | depth_bits              # depth > 0.2 -> sign bit clear
    key_sorted, order = torch.sort(key, stable=True)
    gid_sorted = gid[order]
    tile_sorted = key_sorted >> 32
    ranges = torch.zeros(gx * gy, 2, dtype=torch.int64)
    if tile_sorted.numel():
        t_ids = torch.arange(gx * gy)
        ranges[:, 0] = torch.searchsorted(tile_sorted, t_ids, right=False)
        ranges[:, 1] = torch.searchsorted(tile_sorted, t_ids, right=True)
    return gid_sorted, tile_sorted, ranges


def rasterize(s: OracleSettings, means3D, opacities, shs, scales, rotations, normal_w,
              tile_mask: Optional[torch.Tensor] = None, chunk: int = 64, return_aux: bool = False):
    """Forward of the 9-argument rasterizer call (SLAM/render.py:110-120);
    returns the 7-tuple of SLAM/render.py:122-128 (+ aux dict if asked)."""
    dt = means3D.dtype
    H, W = int(s.image_height), int(s.image_width)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    if tile_mask is None:
        tile_mask = torch.ones(gy, gx, dtype=torch.int32)
    bg = s.bg.to(dt)
    P = means3D.shape[0]

    color_tiles = bg.reshape(3, 1, 1).expand(3, gy * TILE, gx * TILE).clone()
    # masked-out tiles render as if empty (bg colour, T = 1)
    depth_t = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt)
    cidx = torch.full((1, gy * TILE, gx * TILE), -1, dtype=torch.int32)
    didx = torch.full((1, gy * TILE, gx * TILE), -1, dtype=torch.int32)
    cw = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt)
    dw = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt)
    Tmap = torch.ones(1, gy * TILE, gx * TILE, dtype=dt)
    aux = dict(num_rendered=0, consumed=0, evaluated_pairs=0, tile_terminated={})
    n_blended = torch.zeros(gy * TILE, gx * TILE, dtype=torch.int32)     # entries each pixel blended (diagnostic)

    color_parts = {}
    depth_parts = {}
    if P > 0:
        pre = preprocess(s, means3D, opacities, shs, scales, rotations, normal_w)
        gid_sorted, _, ranges = bin_tiles(pre, tile_mask)
        aux["num_rendered"] = int(gid_sorted.numel())
        aux["radii"] = torch.where(pre["valid"], pre["radius"], torch.zeros_like(pre["radius"])).to(torch.int32)
        fx, fy, cx, cy = pre["fx"], pre["fy"], pre["cx"], pre["cy"]
        thr = torch.tensor(s.T_threshold, dtype=dt)
        ly, lx = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
        for t in torch.nonzero(ranges[:, 1] > ranges[:, 0]).reshape(-1).tolist():
            ty, tx = divmod(t, gx)
            px = (tx * TILE + lx).reshape(-1)
            py = (ty * TILE + ly).reshape(-1)
            inside = (px < W) & (py < H)
            pxf, pyf = px.to(dt), py.to(dt)
            rx, ry = (pxf - cx) / fx, (pyf - cy) / fy
            rnorm = torch.sqrt(rx * rx + ry * ry + 1.0)

            T = torch.ones(TILE * TILE, dtype=dt)
            done = ~inside
            C = torch.zeros(TILE * TILE, 3, dtype=dt)
            best_w = torch.zeros(TILE * TILE, dtype=dt)
            best_id = torch.full((TILE * TILE,), -1, dtype=torch.int64)
            D = torch.zeros(TILE * TILE, dtype=dt)
            d_id = torch.full((TILE * TILE,), -1, dtype=torch.int64)
            d_w = torch.zeros(TILE * TILE, dtype=dt)
            d_found = torch.zeros(TILE * TILE, dtype=torch.bool)
            nb = torch.zeros(TILE * TILE, dtype=torch.int32)

            lo, hi = int(ranges[t, 0]), int(ranges[t, 1])
            pos = lo
            while pos < hi and not bool(done.all()):
                ids = gid_sorted[pos:min(pos + chunk, hi)]
                B = ids.numel()
                aux["consumed"] += B
                aux["evaluated_pairs"] += int(B * (~done).sum())
                pos += B
                dx = pre["u"][ids][:, None] - pxf[None, :]
                dy = pre["v"][ids][:, None] - pyf[None, :]
                con = pre["conic"][ids]
                power = (-0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy)
                         - con[:, 1:2] * dx * dy)
                o = pre["opacity"][ids][:, None]
                raw = o * torch.exp(power.clamp(max=0.0))
                # alpha = min(0.99, o G); the clamp's gradient is PASSED THROUGH (upstream 3DGS backward computes
                # dL/dG = o dL/dalpha, dL/do = G dL/dalpha with no clamp mask - SURVEY.md Appendix B "Backward")
                alpha = raw + (raw.clamp(max=0.99) - raw).detach()
                ok = (power <= 0) & (alpha >= 1.0 / 255.0) & (~done)[None, :]
                a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
                one_m = 1.0 - a_eff
                T_after = T[None, :] * torch.cumprod(one_m, dim=0)
                T_before = torch.cat([T[None, :], T_after[:-1]], dim=0)
                stopped = ok & (T_after.detach() < thr)
                alive = torch.cumsum(stopped.to(torch.int32), dim=0) == 0   # includes rows before the stopper
                contrib = ok & alive
                wgt = torch.where(contrib, a_eff * T_before, torch.zeros_like(a_eff))
                C = C + (wgt[:, :, None] * pre["rgb"][ids][:, None, :]).sum(0)
                nb += contrib.sum(0).to(torch.int32)

                # colour arg-max (first max wins)
                wd = wgt.detach()
                carg = torch.argmax(wd, dim=0)
                cmax = wd.gather(0, carg[None, :])[0]
                upd = cmax > best_w.detach()
                best_id = torch.where(upd, ids[carg], best_id)
                best_w = torch.where(upd, wgt.gather(0, carg[None, :])[0], best_w)

                # opaque-surface depth: first blended Gaussian passing the three gates
                nc = pre["n_c"][ids]
                den = nc[:, 0:1] * rx[None, :] + nc[:, 1:2] * ry[None, :] + nc[:, 2:3]
                gate_n = (den.detach().abs() / rnorm[None, :]) > s.normal_threshold
                den_s = torch.where(gate_n, den, torch.ones_like(den))
                zhit = pre["plane_d"][ids][:, None] / den_s
                zmu = pre["depth"][ids][:, None]
                cand = (contrib & (alpha.detach() > s.opaque_threshold) & gate_n
                        & (zhit.detach() > 0) & ((zhit - zmu).detach().abs() < s.depth_threshold))
                has = cand.any(dim=0) & ~d_found
                first = torch.argmax(cand.to(torch.int32), dim=0)
                D = torch.where(has, zhit.gather(0, first[None, :])[0], D)
                d_w = torch.where(has, alpha.gather(0, first[None, :])[0], d_w)
                d_id = torch.where(has, ids[first], d_id)
                d_found = d_found | has

                # carry T: T before the stopper if one fired, else T after the last row
                any_stop = stopped.any(dim=0)
                first_stop = torch.argmax(stopped.to(torch.int32), dim=0)
                T = torch.where(any_stop, T_before.gather(0, first_stop[None, :])[0], T_after[-1])
                done = done | any_stop

            aux["tile_terminated"][t] = bool(done.all())     # every pixel of the tile hit the stop rule (or is off-image)
            ys, xs = ty * TILE, tx * TILE
            color_parts[t] = (C + T[:, None] * bg[None, :]).t().reshape(3, TILE, TILE)
            depth_parts[t] = D.reshape(1, TILE, TILE)
            cidx[0, ys:ys + TILE, xs:xs + TILE] = best_id.reshape(TILE, TILE).to(torch.int32)
            didx[0, ys:ys + TILE, xs:xs + TILE] = d_id.reshape(TILE, TILE).to(torch.int32)
            cw[0, ys:ys + TILE, xs:xs + TILE] = best_w.detach().reshape(TILE, TILE)
            dw[0, ys:ys + TILE, xs:xs + TILE] = d_w.detach().reshape(TILE, TILE)
            Tmap[0, ys:ys + TILE, xs:xs + TILE] = T.detach().reshape(TILE, TILE)
            n_blended[ys:ys + TILE, xs:xs + TILE] = nb.reshape(TILE, TILE)

    if color_parts:
        # assemble differentiably: one scatter of all rendered tiles
        tl = sorted(color_parts)
        cstack = torch.stack([color_parts[t] for t in tl])        # [n,3,16,16]
        dstack = torch.stack([depth_parts[t] for t in tl])        # [n,1,16,16]
        full_c = color_tiles.reshape(3, gy, TILE, gx, TILE).permute(1, 3, 0, 2, 4).reshape(gy * gx, 3, TILE, TILE)
        full_d = depth_t.reshape(1, gy, TILE, gx, TILE).permute(1, 3, 0, 2, 4).reshape(gy * gx, 1, TILE, TILE)
        ti = torch.tensor(tl)
        full_c = full_c.index_copy(0, ti, cstack)
        full_d = full_d.index_copy(0, ti, dstack)
        color_tiles = full_c.reshape(gy, gx, 3, TILE, TILE).permute(2, 0, 3, 1, 4).reshape(3, gy * TILE, gx * TILE)
        depth_t = full_d.reshape(gy, gx, 1, TILE, TILE).permute(2, 0, 3, 1, 4).reshape(1, gy * TILE, gx * TILE)

    out = (color_tiles[:, :H, :W], depth_t[:, :H, :W], cidx[:, :H, :W].contiguous(),
           didx[:, :H, :W].contiguous(), cw[:, :H, :W].contiguous(), dw[:, :H, :W].contiguous(),
           Tmap[:, :H, :W].contiguous())
    if return_aux:
        aux["n_blended"] = n_blended[:H, :W].contiguous()
        return out, aux
    return out


def make_settings(H, W, fx, fy, cx, cy, viewmatrix=None, campos=None, sh_degree=3,
                  opaque_threshold=0.6, depth_threshold=1.0, normal_threshold_deg=60.0,
                  color_sigma=3.0, bg=None, T_threshold=1e-4, dtype=torch.float32):
    """Settings exactly as SLAM/render.py:66-88 would build them from a Camera with
    FoVx = 2 atan(W / 2fx) (utils/graphics_utils.py:93-94) and Replica/TUM intrinsics."""
    if viewmatrix is None:
        viewmatrix = torch.eye(4, dtype=dtype)
    V = viewmatrix.t()                                   # W2C
    if campos is None:
        campos = torch.linalg.inv(V.double())[:3, 3].to(dtype)
    tanfovx, tanfovy = W / (2.0 * fx), H / (2.0 * fy)
    znear, zfar = 0.01, 100.0
    Pm = torch.zeros(4, 4, dtype=dtype)                  # utils/graphics_utils.py:66-86
    Pm[0, 0] = 1.0 / tanfovx
    Pm[1, 1] = 1.0 / tanfovy
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    proj = viewmatrix @ Pm.t()                           # scene/cameras.py:106-110
    if bg is None:
        bg = torch.zeros(3, dtype=dtype)
    return OracleSettings(
        image_height=H, image_width=W, tanfovx=tanfovx, tanfovy=tanfovy, bg=bg,
        scale_modifier=1.0, viewmatrix=viewmatrix, projmatrix=proj, sh_degree=sh_degree,
        campos=campos, opaque_threshold=opaque_threshold, depth_threshold=depth_threshold,
        normal_threshold=math.cos(math.radians(normal_threshold_deg)), color_sigma=color_sigma,
        prefiltered=False, debug=False, cx=cx, cy=cy, T_threshold=T_threshold)
