"""Second, independent restatement of the RTG-SLAM rasterizer contract: a per-pixel float64 evaluator.

TEST INFRASTRUCTURE ONLY (same rule as oracle/raster_oracle.py: imported by tests/ only, never by the product).

Why it exists.  The reference's rasterizer source is absent (/root/reference/.gitmodules:1-4), so
oracle/raster_oracle.py and the HIP kernels were both written from SURVEY.md Appendix B by the same hand; a
shared misreading would pass every HIP-vs-oracle test.  This file shares NOTHING with raster_oracle.py:

  * numpy float64, no torch, no autograd;
  * no tiles, no tile lists, no chunks: every pixel walks ALL visible Gaussians in one global (depth, index)
    order.  The only trace of tiling is the per-pixel predicate "pixel's 16x16 tile lies inside the Gaussian's
    tile rectangle and tile_mask there is non-zero", which is part of the semantics (Appendix B item 6: a
    Gaussian cannot reach pixels outside the tiles of its 3-sigma rect, even where alpha >= 1/255);
  * the alpha-blend backward is written out analytically (Appendix B "Backward": back-to-front recursion on the
    colour behind each Gaussian), not differentiated by a tool;
  * the per-Gaussian chain (projection, EWA covariance, conic, SH colour, plane) is differentiated by float64
    CENTRAL DIFFERENCES of the forward projection, with the two non-smooth decisions frozen exactly as
    Appendix B prescribes (the 1.3 tan(fov) clamp makes t.x / t.y constants; colour channels clamped at 0 get
    zero gradient).  It is therefore independent of both autograd and the hand-derived chain rule in
    rtg_slam_amd/csrc/raster_bwd.hip.

Frozen decisions it shares with Appendix B by construction (they ARE the specification): cull z <= 0.2,
+0.3 dilation, radius = ceil(color_sigma sqrt(lambda_max)), u = fx x/z + cx, alpha = min(0.99, o G) with the
clamp's gradient PASSED THROUGH (upstream 3DGS backward does not mask it), skip alpha < 1/255 and power > 0,
stop before the Gaussian that would drive T below T_threshold, first arg-max colour index, first opaque
Gaussian passing the normal / depth gates owns the depth.

Reference call contract: /root/reference/SLAM/render.py:68-128.  SH constants: utils/sh_utils.py:26-45.
Quaternion convention (w,x,y,z): utils/general_utils.py:108-131.
"""
from __future__ import annotations

import numpy as np

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435]

_NAMES = ("xyz", "opacity", "shs", "scales", "rotations", "normal")


def _np(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t, dtype=np.float64)


class _Cam:
    def __init__(self, s):
        self.H, self.W = int(s.image_height), int(s.image_width)
        self.fx = self.W / (2.0 * float(s.tanfovx))
        self.fy = self.H / (2.0 * float(s.tanfovy))
        self.cx = float(s.cx) if float(s.cx) > 0 else (self.W - 1) / 2.0
        self.cy = float(s.cy) if float(s.cy) > 0 else (self.H - 1) / 2.0
        self.limx, self.limy = 1.3 * float(s.tanfovx), 1.3 * float(s.tanfovy)
        self.w2c = _np(s.viewmatrix).T            # the settings carry W2C transposed (scene/cameras.py:96-98)
        self.campos = _np(s.campos)
        self.bg = _np(s.bg)
        self.deg = int(s.sh_degree)
        self.mod = float(s.scale_modifier)
        self.sigma = float(s.color_sigma)
        self.opaque = float(s.opaque_threshold)
        self.cos_n = float(s.normal_threshold)
        self.dthr = float(s.depth_threshold)
        self.Tthr = float(s.T_threshold)


def _sh_basis(deg, d):
    """[n, 16] real SH basis of utils/sh_utils.py:57-120 evaluated at unit directions d (columns beyond the
    active degree are zero)."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    B = np.zeros((d.shape[0], 16))
    B[:, 0] = _C0
    if deg >= 1:
        B[:, 1], B[:, 2], B[:, 3] = -_C1 * y, _C1 * z, -_C1 * x
    if deg >= 2:
        B[:, 4] = _C2[0] * x * y
        B[:, 5] = _C2[1] * y * z
        B[:, 6] = _C2[2] * (2 * z * z - x * x - y * y)
        B[:, 7] = _C2[3] * x * z
        B[:, 8] = _C2[4] * (x * x - y * y)
    if deg >= 3:
        B[:, 9] = _C3[0] * y * (3 * x * x - y * y)
        B[:, 10] = _C3[1] * x * y * z
        B[:, 11] = _C3[2] * y * (4 * z * z - x * x - y * y)
        B[:, 12] = _C3[3] * z * (2 * z * z - 3 * x * x - 3 * y * y)
        B[:, 13] = _C3[4] * x * (4 * z * z - x * x - y * y)
        B[:, 14] = _C3[5] * z * (x * x - y * y)
        B[:, 15] = _C3[6] * x * (x * x - 3 * y * y)
    return B


def _project(cam, xyz, scales, quats, shs, normals, frozen=None):
    """Per-Gaussian stage for the rows given (all assumed in front of the camera).  Returns the 14 quantities
    the blend consumes, stacked [n,14] = u v A B C r g b pcx pcy pcz ncx ncy ncz, plus the raw pieces the
    caller freezes (clamp decisions) or needs once (covariance for the radius)."""
    Rw = cam.w2c[:3, :3]
    p = xyz @ Rw.T + cam.w2c[:3, 3]
    px, py, pz = p[:, 0], p[:, 1], p[:, 2]
    qw, qx, qy, qz = quats[:, 0], quats[:, 1], quats[:, 2], quats[:, 3]
    n = xyz.shape[0]
    R = np.empty((n, 3, 3))
    R[:, 0, 0] = 1 - 2 * (qy * qy + qz * qz); R[:, 0, 1] = 2 * (qx * qy - qw * qz); R[:, 0, 2] = 2 * (qx * qz + qw * qy)
    R[:, 1, 0] = 2 * (qx * qy + qw * qz); R[:, 1, 1] = 1 - 2 * (qx * qx + qz * qz); R[:, 1, 2] = 2 * (qy * qz - qw * qx)
    R[:, 2, 0] = 2 * (qx * qz - qw * qy); R[:, 2, 1] = 2 * (qy * qz + qw * qx); R[:, 2, 2] = 1 - 2 * (qx * qx + qy * qy)
    L = R * (scales * cam.mod)[:, None, :]                       # R diag(s)
    cov3 = L @ np.transpose(L, (0, 2, 1))
    rx, ry = px / pz, py / pz
    if frozen is None:
        in_x = (rx >= -cam.limx) & (rx <= cam.limx)
        in_y = (ry >= -cam.limy) & (ry <= cam.limy)
        tx_c = np.clip(rx, -cam.limx, cam.limx) * pz
        ty_c = np.clip(ry, -cam.limy, cam.limy) * pz
    else:
        in_x, in_y, tx_c, ty_c = frozen["in_x"], frozen["in_y"], frozen["tx"], frozen["ty"]
    tx = np.where(in_x, px, tx_c)                                # clamped: a constant of the backward pass
    ty = np.where(in_y, py, ty_c)
    Jm = np.zeros((n, 2, 3))
    Jm[:, 0, 0] = cam.fx / pz; Jm[:, 0, 2] = -cam.fx * tx / (pz * pz)
    Jm[:, 1, 1] = cam.fy / pz; Jm[:, 1, 2] = -cam.fy * ty / (pz * pz)
    Tm = Jm @ Rw
    cov2 = Tm @ cov3 @ np.transpose(Tm, (0, 2, 1))
    a = cov2[:, 0, 0] + 0.3
    b = cov2[:, 0, 1]
    c = cov2[:, 1, 1] + 0.3
    det = a * c - b * b
    A, Bc, C = c / det, -b / det, a / det
    u = cam.fx * px / pz + cam.cx
    v = cam.fy * py / pz + cam.cy
    d = xyz - cam.campos
    d = d / np.sqrt((d * d).sum(1, keepdims=True))
    basis = _sh_basis(cam.deg, d)
    raw = np.einsum("nk,nkc->nc", basis[:, :shs.shape[1]], shs) + 0.5
    neg = (raw < 0) if frozen is None else frozen["neg"]
    rgb = np.where(neg, 0.0, raw)
    nc = normals @ Rw.T
    out = np.stack([u, v, A, Bc, C, rgb[:, 0], rgb[:, 1], rgb[:, 2], px, py, pz, nc[:, 0], nc[:, 1], nc[:, 2]], axis=1)
    aux = dict(in_x=in_x, in_y=in_y, tx=tx_c, ty=ty_c, neg=neg, a=a, b=b, c=c, det=det)
    return out, aux


def render(settings, means3D, opacities, shs, scales, rotations, normal_w, tile_mask=None,
           g_color=None, g_depth=None, fd_step=1e-6):
    """Forward (and, if g_color / g_depth are given, backward) of the 9-argument rasterizer call.
    Returns (outs, grads): outs = (color[3,H,W], depth[1,H,W], color_index, depth_index, color_weight,
    depth_weight, T) as float64 / int32 numpy arrays; grads = dict over xyz / opacity / shs / scales / rotations /
    normal (None without upstream gradients)."""
    cam = _Cam(settings)
    H, W = cam.H, cam.W
    gx, gy = (W + 15) // 16, (H + 15) // 16
    xyz, op, sh = _np(means3D).reshape(-1, 3), _np(opacities).reshape(-1), _np(shs)
    sc, qt, nw = _np(scales).reshape(-1, 3), _np(rotations).reshape(-1, 4), _np(normal_w).reshape(-1, 3)
    N = xyz.shape[0]
    sh = sh.reshape(N, -1, 3) if N else sh.reshape(0, 16, 3)
    tmask = np.ones((gy, gx), dtype=np.int32) if tile_mask is None else np.asarray(
        tile_mask.detach().cpu().numpy() if hasattr(tile_mask, "detach") else tile_mask).astype(np.int32)

    ys, xs = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pxf, pyf = xs.reshape(-1).astype(np.float64), ys.reshape(-1).astype(np.float64)
    ptx, pty = (xs.reshape(-1) // 16), (ys.reshape(-1) // 16)
    pix_on = tmask[pty, ptx] != 0
    HW = H * W
    ray_x, ray_y = (pxf - cam.cx) / cam.fx, (pyf - cam.cy) / cam.fy
    ray_n = np.sqrt(ray_x * ray_x + ray_y * ray_y + 1.0)

    color = np.tile(cam.bg.reshape(3, 1), (1, HW))
    depth = np.zeros(HW)
    cidx = np.full(HW, -1, dtype=np.int32)
    didx = np.full(HW, -1, dtype=np.int32)
    cwt, dwt, Tfin = np.zeros(HW), np.zeros(HW), np.ones(HW)
    grads = None
    want_grad = g_color is not None or g_depth is not None
    if want_grad:
        grads = {"xyz": np.zeros((N, 3)), "opacity": np.zeros((N, 1)), "shs": np.zeros_like(sh),
                 "scales": np.zeros((N, 3)), "rotations": np.zeros((N, 4)), "normal": np.zeros((N, 3))}
    if N == 0:
        return _pack(color, depth, cidx, didx, cwt, dwt, Tfin, H, W), grads

    # ---- per-Gaussian stage on the Gaussians in front of the camera
    pz_all = xyz @ cam.w2c[2, :3] + cam.w2c[2, 3]
    front = np.nonzero(pz_all > 0.2)[0]
    proj, aux = _project(cam, xyz[front], sc[front], qt[front], sh[front], nw[front])
    ok = aux["det"] != 0
    mid = 0.5 * (aux["a"] + aux["c"])
    lam = mid + np.sqrt(np.maximum(0.1, mid * mid - aux["det"]))
    radius = np.ceil(cam.sigma * np.sqrt(lam))
    u, v = proj[:, 0], proj[:, 1]
    big = float(1 << 28)
    cdiv = lambda t: np.clip(np.trunc(t / 16.0), -big, big).astype(np.int64)       # C (int) cast, then clamp
    x0 = np.clip(cdiv(u - radius), 0, gx); x1 = np.clip(cdiv(u + radius + 15.0), 0, gx)
    y0 = np.clip(cdiv(v - radius), 0, gy); y1 = np.clip(cdiv(v + radius + 15.0), 0, gy)
    ok &= (x1 - x0) * (y1 - y0) > 0
    vis = np.nonzero(ok)[0]                                       # positions inside `front`
    # one global order: float32 depth bits, then Gaussian index
    zkey = proj[vis, 10].astype(np.float32)
    order = np.lexsort((front[vis], zkey))
    vis = vis[order]
    K = vis.shape[0]

    T = np.ones(HW)
    done = ~pix_on.copy()
    C = np.zeros((3, HW))
    has_depth = np.zeros(HW, dtype=bool)
    keep_a = np.zeros((K, HW)) if want_grad else None             # alpha of contributing pairs, else 0
    keep_T = np.zeros((K, HW)) if want_grad else None             # transmittance in front of the pair
    for k in range(K):
        j = vis[k]
        gid = int(front[j])
        uu, vv, A, Bc, Cc = proj[j, 0], proj[j, 1], proj[j, 2], proj[j, 3], proj[j, 4]
        reach = (ptx >= x0[j]) & (ptx < x1[j]) & (pty >= y0[j]) & (pty < y1[j]) & ~done
        if not reach.any():
            continue
        dx, dy = uu - pxf, vv - pyf
        power = -0.5 * (A * dx * dx + Cc * dy * dy) - Bc * dx * dy
        G = np.exp(np.minimum(power, 0.0))
        alpha = np.minimum(0.99, op[gid] * G)
        cand = reach & (power <= 0) & (alpha >= 1.0 / 255.0)
        Tn = T * (1.0 - alpha)
        stop = cand & (Tn < cam.Tthr)
        done |= stop
        con = cand & ~stop
        if not con.any():
            continue
        w = np.where(con, alpha * T, 0.0)
        C += proj[j, 5:8].reshape(3, 1) * w
        better = w > cwt
        cwt = np.where(better, w, cwt)
        cidx = np.where(better, gid, cidx).astype(np.int32)
        # opaque-surface depth: first contributing Gaussian passing the gates
        pc, ncv = proj[j, 8:11], proj[j, 11:14]
        den = ncv[0] * ray_x + ncv[1] * ray_y + ncv[2]
        gate = con & ~has_depth & (alpha > cam.opaque) & (np.abs(den) / ray_n > cam.cos_n)
        if gate.any():
            zh = np.where(gate, (ncv @ pc) / np.where(gate, den, 1.0), 0.0)
            gate &= (zh > 0) & (np.abs(zh - pc[2]) < cam.dthr)
            depth = np.where(gate, zh, depth)
            dwt = np.where(gate, alpha, dwt)
            didx = np.where(gate, gid, didx).astype(np.int32)
            has_depth |= gate
        if want_grad:
            keep_a[k] = np.where(con, alpha, 0.0)
            keep_T[k] = T
        T = np.where(con, Tn, T)
    Tfin = T
    color = C + cam.bg.reshape(3, 1) * Tfin
    outs = _pack(color, depth, cidx, didx, cwt, dwt, Tfin, H, W)
    if not want_grad:
        return outs, None

    gC = np.zeros((3, HW)) if g_color is None else _np(g_color).reshape(3, HW)
    gD = np.zeros(HW) if g_depth is None else _np(g_depth).reshape(HW)
    # ---- blend backward, back to front.  behind = colour accumulated behind the current Gaussian (starts as bg)
    behind = np.tile(cam.bg.reshape(3, 1), (1, HW))
    dproj = np.zeros((front.shape[0], 14))                        # dL / d(u v A B C r g b pc nc) per front Gaussian
    for k in range(K - 1, -1, -1):
        a = keep_a[k]
        live = a > 0
        j = vis[k]
        gid = int(front[j])
        if live.any():
            Tk = keep_T[k]
            col = proj[j, 5:8].reshape(3, 1)
            dL_dalpha = np.where(live, Tk * ((col - behind) * gC).sum(0), 0.0)
            dproj[j, 5:8] += (np.where(live, a * Tk, 0.0) * gC).sum(1)
            behind = np.where(live, a * col + (1.0 - a) * behind, behind)
            uu, vv, A, Bc, Cc = proj[j, 0], proj[j, 1], proj[j, 2], proj[j, 3], proj[j, 4]
            dx, dy = uu - pxf, vv - pyf
            G = np.exp(np.minimum(-0.5 * (A * dx * dx + Cc * dy * dy) - Bc * dx * dy, 0.0))
            # alpha = min(0.99, o G): the clamp's gradient is passed through (upstream 3DGS)
            grads["opacity"][gid, 0] += (G * dL_dalpha).sum()
            dL_dpow = op[gid] * dL_dalpha * G
            dproj[j, 0] += (dL_dpow * (-A * dx - Bc * dy)).sum()
            dproj[j, 1] += (dL_dpow * (-Cc * dy - Bc * dx)).sum()
            dproj[j, 2] += (dL_dpow * (-0.5 * dx * dx)).sum()
            dproj[j, 3] += (dL_dpow * (-dx * dy)).sum()
            dproj[j, 4] += (dL_dpow * (-0.5 * dy * dy)).sum()
        own = (didx == gid) & (gD != 0)
        if own.any():                                             # z = (n.p) / (n.r)
            pc, ncv = proj[j, 8:11], proj[j, 11:14]
            den = np.where(own, ncv[0] * ray_x + ncv[1] * ray_y + ncv[2], 1.0)
            g = np.where(own, gD, 0.0)
            zz = (ncv @ pc) / den
            dproj[j, 8:11] += ncv * (g / den).sum()
            dproj[j, 11] += (g * (pc[0] - zz * ray_x) / den).sum()
            dproj[j, 12] += (g * (pc[1] - zz * ray_y) / den).sum()
            dproj[j, 13] += (g * (pc[2] - zz) / den).sum()

    # ---- per-Gaussian chain by central differences of _project (non-smooth decisions frozen at the base point)
    act = np.nonzero(np.abs(dproj).sum(1) > 0)[0]
    if act.size:
        rows = front[act]
        frozen = {k2: aux[k2][act] for k2 in ("in_x", "in_y", "tx", "ty", "neg")}
        base = dict(xyz=xyz[rows], scales=sc[rows], rotations=qt[rows], shs=sh[rows], normal=nw[rows])
        w14 = dproj[act]

        def run(th):
            return _project(cam, th["xyz"], th["scales"], th["rotations"], th["shs"], th["normal"], frozen)[0]
        for name in ("xyz", "scales", "rotations", "shs", "normal"):
            flat = base[name].reshape(act.size, -1)
            for col in range(flat.shape[1]):
                h = fd_step * np.maximum(1.0, np.abs(flat[:, col]))
                hi, lo = dict(base), dict(base)
                fp, fm = flat.copy(), flat.copy()
                fp[:, col] += h; fm[:, col] -= h
                hi[name], lo[name] = fp.reshape(base[name].shape), fm.reshape(base[name].shape)
                dy = (run(hi) - run(lo)) / (2.0 * h)[:, None]
                grads[name].reshape(N, -1)[rows, col] += (dy * w14).sum(1)
    return outs, grads


def _pack(color, depth, cidx, didx, cwt, dwt, T, H, W):
    return (color.reshape(3, H, W), depth.reshape(1, H, W), cidx.reshape(1, H, W).astype(np.int32),
            didx.reshape(1, H, W).astype(np.int32), cwt.reshape(1, H, W), dwt.reshape(1, H, W), T.reshape(1, H, W))
