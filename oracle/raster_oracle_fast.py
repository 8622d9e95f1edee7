"""TEST INFRASTRUCTURE - not part of the product (only tests/ may import this).

Whole-image forward + backward of the rasterizer oracle in minutes instead of hours: the per-Gaussian stage and the
binning are oracle/raster_oracle.py's own (`preprocess`, `bin_tiles`, differentiable); the tile blend is evaluated
without an autograd graph and its backward is written out by hand per tile (SURVEY.md Appendix B "Backward"), producing
the gradient with respect to the per-Gaussian splat quantities (u, v, conic, opacity, rgb, n_c, plane_d); ONE autograd
pass through `preprocess` then carries those to the parameters.  raster_oracle.rasterize + autograd builds a graph per
64-entry chunk of every tile: 140 s for an eighth of the tiles of the 1.2 M surface scene; this takes ~1 minute for all.

Pinned by tests/test_oracle_raster.py against raster_oracle.rasterize + autograd (float64: 1e-9 on the maps, 1e-8 on
the gradients) - it restates nothing new about the reference, it is the same definition differentiated by hand.
Parity status of the rasterizer oracle itself: unpinned (the reference's source is absent) - see raster_oracle.py."""
from __future__ import annotations

from typing import Optional

import torch

from . import raster_oracle as ro

TILE = ro.TILE


def forward_backward(s: ro.OracleSettings, means3D, opacities, shs, scales, rotations, normal_w,
                     tile_mask: Optional[torch.Tensor], g_color: torch.Tensor, g_depth: torch.Tensor, chunk: int = 128):
    """Returns (the 7 output maps, dict of gradients of sum(color * g_color) + sum(depth * g_depth) with respect to
    xyz / opacity / shs / scales / rotations / normal)."""
    dt = means3D.dtype
    H, W = int(s.image_height), int(s.image_width)
    gx, gy = (W + TILE - 1) // TILE, (H + TILE - 1) // TILE
    if tile_mask is None:
        tile_mask = torch.ones(gy, gx, dtype=torch.int32)
    bg = s.bg.to(dt)
    P = means3D.shape[0]
    leaves = [t.detach().clone().requires_grad_(True) for t in (means3D, opacities, shs, scales, rotations, normal_w)]
    color = bg.reshape(3, 1, 1).expand(3, gy * TILE, gx * TILE).clone()
    depth = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt)
    cidx = torch.full((1, gy * TILE, gx * TILE), -1, dtype=torch.int32)
    didx = torch.full((1, gy * TILE, gx * TILE), -1, dtype=torch.int32)
    cw = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt)
    dw = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt)
    Tmap = torch.ones(1, gy * TILE, gx * TILE, dtype=dt)
    names = ("xyz", "opacity", "shs", "scales", "rotations", "normal")
    if P == 0:
        outs = (color[:, :H, :W], depth[:, :H, :W], cidx[:, :H, :W], didx[:, :H, :W], cw[:, :H, :W], dw[:, :H, :W], Tmap[:, :H, :W])
        return outs, {k: torch.zeros_like(t) for k, t in zip(names, leaves)}

    pre = ro.preprocess(s, *leaves)
    # The tile loop works on [<= 128, 256] tensors: with many intra-op threads the fork / join of every small op costs more
    # than the op (a whole 1.2 M frame: 53 s with 8 threads, 23 s with 4, 41 s with 1) - capped for the loop only.
    n_threads = torch.get_num_threads()
    torch.set_num_threads(min(n_threads, 4))
    with torch.no_grad():
        gid_sorted, _, ranges = ro.bin_tiles(pre, tile_mask)
        U, V = pre["u"].detach(), pre["v"].detach()
        CON, OP, RGB = pre["conic"].detach(), pre["opacity"].detach(), pre["rgb"].detach()
        NC, PD, ZMU = pre["n_c"].detach(), pre["plane_d"].detach(), pre["depth"].detach()
        fx, fy, cx, cy = pre["fx"], pre["fy"], pre["cx"], pre["cy"]
        dU = torch.zeros(P, dtype=dt); dV = torch.zeros(P, dtype=dt)
        dCON = torch.zeros(P, 3, dtype=dt); dOP = torch.zeros(P, dtype=dt); dRGB = torch.zeros(P, 3, dtype=dt)
        dNC = torch.zeros(P, 3, dtype=dt); dPD = torch.zeros(P, dtype=dt)
        gC_full = torch.zeros(3, gy * TILE, gx * TILE, dtype=dt); gC_full[:, :H, :W] = g_color.to(dt)
        gD_full = torch.zeros(1, gy * TILE, gx * TILE, dtype=dt); gD_full[:, :H, :W] = g_depth.to(dt)
        thr = torch.tensor(s.T_threshold, dtype=dt)
        ly, lx = torch.meshgrid(torch.arange(TILE), torch.arange(TILE), indexing="ij")
        npx = TILE * TILE
        for t in torch.nonzero(ranges[:, 1] > ranges[:, 0]).reshape(-1).tolist():
            ty, tx = divmod(t, gx)
            px = (tx * TILE + lx).reshape(-1); py = (ty * TILE + ly).reshape(-1)
            inside = (px < W) & (py < H)
            pxf, pyf = px.to(dt), py.to(dt)
            rx, ry = (pxf - cx) / fx, (pyf - cy) / fy
            rnorm = torch.sqrt(rx * rx + ry * ry + 1.0)
            T = torch.ones(npx, dtype=dt)
            done = ~inside
            best_w = torch.zeros(npx, dtype=dt); best_id = torch.full((npx,), -1, dtype=torch.int64)
            D = torch.zeros(npx, dtype=dt); d_id = torch.full((npx,), -1, dtype=torch.int64)
            d_w = torch.zeros(npx, dtype=dt); d_found = torch.zeros(npx, dtype=torch.bool)
            d_row = torch.full((npx,), -1, dtype=torch.int64)       # list position of the depth owner
            d_den = torch.ones(npx, dtype=dt)
            lo, hi = int(ranges[t, 0]), int(ranges[t, 1])
            pos = lo
            A_l, C_l, Tb_l, G_l, dx_l, dy_l = [], [], [], [], [], []
            while pos < hi and not bool(done.all()):
                ids = gid_sorted[pos:min(pos + chunk, hi)]
                B = ids.numel()
                dx = U[ids][:, None] - pxf[None, :]
                dy = V[ids][:, None] - pyf[None, :]
                con = CON[ids]
                power = -0.5 * (con[:, 0:1] * dx * dx + con[:, 2:3] * dy * dy) - con[:, 1:2] * dx * dy
                G = torch.exp(power.clamp(max=0.0))
                raw = OP[ids][:, None] * G
                alpha = raw.clamp(max=0.99)
                ok = (power <= 0) & (alpha >= 1.0 / 255.0) & (~done)[None, :]
                a_eff = torch.where(ok, alpha, torch.zeros_like(alpha))
                T_after = T[None, :] * torch.cumprod(1.0 - a_eff, dim=0)
                T_before = torch.cat([T[None, :], T_after[:-1]], dim=0)
                stopped = ok & (T_after < thr)
                alive = torch.cumsum(stopped.to(torch.int32), dim=0) == 0
                contrib = ok & alive
                wgt = torch.where(contrib, a_eff * T_before, torch.zeros_like(a_eff))
                carg = torch.argmax(wgt, dim=0)
                cmax = wgt.gather(0, carg[None, :])[0]
                upd = cmax > best_w
                best_id = torch.where(upd, ids[carg], best_id)
                best_w = torch.where(upd, cmax, best_w)
                nc = NC[ids]
                den = nc[:, 0:1] * rx[None, :] + nc[:, 1:2] * ry[None, :] + nc[:, 2:3]
                gate_n = (den.abs() / rnorm[None, :]) > s.normal_threshold
                den_s = torch.where(gate_n, den, torch.ones_like(den))
                zhit = PD[ids][:, None] / den_s
                cand = contrib & (alpha > s.opaque_threshold) & gate_n & (zhit > 0) & ((zhit - ZMU[ids][:, None]).abs() < s.depth_threshold)
                has = cand.any(dim=0) & ~d_found
                first = torch.argmax(cand.to(torch.int32), dim=0)
                D = torch.where(has, zhit.gather(0, first[None, :])[0], D)
                d_w = torch.where(has, alpha.gather(0, first[None, :])[0], d_w)
                d_id = torch.where(has, ids[first], d_id)
                d_row = torch.where(has, first + (pos - lo), d_row)
                d_den = torch.where(has, den_s.gather(0, first[None, :])[0], d_den)
                d_found = d_found | has
                any_stop = stopped.any(dim=0)
                first_stop = torch.argmax(stopped.to(torch.int32), dim=0)
                T = torch.where(any_stop, T_before.gather(0, first_stop[None, :])[0], T_after[-1])
                done = done | any_stop
                A_l.append(a_eff); C_l.append(contrib); Tb_l.append(T_before); G_l.append(G); dx_l.append(dx); dy_l.append(dy)
                pos += B
            n_used = pos - lo
            ids = gid_sorted[lo:pos]
            A = torch.cat(A_l); Cn = torch.cat(C_l); Tb = torch.cat(Tb_l); G = torch.cat(G_l); dx = torch.cat(dx_l); dy = torch.cat(dy_l)
            Wt = torch.where(Cn, A * Tb, torch.zeros_like(A))                      # [n, 256]
            c = RGB[ids]                                                            # [n, 3]
            Ctot = Wt.t() @ c                                                       # [256, 3]
            ys, xs = ty * TILE, tx * TILE
            color[:, ys:ys + TILE, xs:xs + TILE] = (Ctot + T[:, None] * bg[None, :]).t().reshape(3, TILE, TILE)
            depth[0, ys:ys + TILE, xs:xs + TILE] = D.reshape(TILE, TILE)
            cidx[0, ys:ys + TILE, xs:xs + TILE] = best_id.reshape(TILE, TILE).to(torch.int32)
            didx[0, ys:ys + TILE, xs:xs + TILE] = d_id.reshape(TILE, TILE).to(torch.int32)
            cw[0, ys:ys + TILE, xs:xs + TILE] = best_w.reshape(TILE, TILE)
            dw[0, ys:ys + TILE, xs:xs + TILE] = d_w.reshape(TILE, TILE)
            Tmap[0, ys:ys + TILE, xs:xs + TILE] = T.reshape(TILE, TILE)
            # ---- backward of the tile (Appendix B): colour
            gC = gC_full[:, ys:ys + TILE, xs:xs + TILE].reshape(3, npx).t()          # [256, 3]
            gD = gD_full[0, ys:ys + TILE, xs:xs + TILE].reshape(npx)
            cg = c @ gC.t()                                                         # [n, 256]: c_k . g
            # colour strictly behind entry k, seen through T: S_k = C_total - sum_{m <= k} c_m w_m
            cum = torch.cumsum(Wt[:, :, None] * c[:, None, :], dim=0)               # [n, 256, 3]
            Sg = ((Ctot[None] - cum) * gC[None]).sum(-1) + (T * (gC @ bg))[None, :]
            dL_dalpha = torch.where(Cn, Tb * cg - Sg / (1.0 - A), torch.zeros_like(A))
            o = OP[ids][:, None]
            dOP.index_add_(0, ids, (G * dL_dalpha).sum(1))                          # clamp of alpha: gradient passes through
            dL_dpow = o * dL_dalpha * G
            con = CON[ids]
            dU.index_add_(0, ids, (-(con[:, 0:1] * dx + con[:, 1:2] * dy) * dL_dpow).sum(1))
            dV.index_add_(0, ids, (-(con[:, 2:3] * dy + con[:, 1:2] * dx) * dL_dpow).sum(1))
            dCON.index_add_(0, ids, torch.stack([(-0.5 * dx * dx * dL_dpow).sum(1), (-dx * dy * dL_dpow).sum(1),
                                                 (-0.5 * dy * dy * dL_dpow).sum(1)], dim=-1))
            dRGB.index_add_(0, ids, Wt @ gC)
            # ---- depth: z = plane_d / (n_c . r) of the pixel's owner
            own = d_found & (gD != 0)
            if bool(own.any()):
                oi = d_id[own]
                den = d_den[own]
                pd = PD[oi]
                dPD.index_add_(0, oi, gD[own] / den)
                k = -gD[own] * pd / (den * den)
                dNC.index_add_(0, oi, torch.stack([k * rx[own], k * ry[own], k], dim=-1))
            del n_used
    torch.set_num_threads(n_threads)
    proxy = ((pre["u"] * dU).sum() + (pre["v"] * dV).sum() + (pre["conic"] * dCON).sum() + (pre["opacity"] * dOP).sum()
             + (pre["rgb"] * dRGB).sum() + (pre["n_c"] * dNC).sum() + (pre["plane_d"] * dPD).sum())
    proxy.backward()
    grads = {k: (t.grad if t.grad is not None else torch.zeros_like(t)) for k, t in zip(names, leaves)}
    outs = (color[:, :H, :W].contiguous(), depth[:, :H, :W].contiguous(), cidx[:, :H, :W].contiguous(),
            didx[:, :H, :W].contiguous(), cw[:, :H, :W].contiguous(), dw[:, :H, :W].contiguous(), Tmap[:, :H, :W].contiguous())
    return outs, grads
