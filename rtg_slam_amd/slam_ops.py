"""Host side of the SLAM-side ops around the hot path (include/rtgs_slam.h), under the reference's own function names
and signatures so that its callers bind unchanged:

    pixelmask2tilemask / transmission2tilemask / colorerror2tilemask      SLAM/utils.py:681-734
    render_range                                                           mapper.py:471-508 (T_map -> masks, fused)
    distCUDA2                (also importable as simple_knn._C.distCUDA2)   gaussian_pointcloud.py:3, 376
    accumulate_gaussian_error (also cuda_utils._C.accumulate_gaussian_error) mapper.py:15, 541-565
    bilateralFilter_torch / frame_preprocess / sample_pixels               SLAM/utils.py:550-589, tracker.py:104-131,
                                                                           SLAM/utils.py:141-183
torch supplies device memory and the current HIP stream; there is no CPU path."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib


def _dev(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("rtg_slam_amd.slam_ops: tensors must live on a HIP device; this build has no CPU path.")
    return t.device


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _u8(mask: torch.Tensor) -> torch.Tensor:
    return (mask != 0).to(torch.uint8).contiguous() if mask.dtype != torch.uint8 else mask.contiguous()


def _grid(H, W, stride):
    return (H + stride - 1) // stride, (W + stride - 1) // stride


def pixelmask2tilemask(pixelmask: torch.Tensor, stride: int = 16) -> torch.Tensor:
    lib, dev = _lib.load(), _dev(pixelmask)
    H, W = int(pixelmask.shape[0]), int(pixelmask.shape[1])
    gy, gx = _grid(H, W, stride)
    out = torch.empty(gy, gx, dtype=torch.int32, device=dev)
    tmp = torch.empty(gy * gx, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_pixelmask2tilemask(_p(_u8(pixelmask)), H, W, int(stride), _p(out), _p(tmp), _stream(dev))
    _lib.check(rc, "rtgs_pixelmask2tilemask")
    return out


def transmission2tilemask(pixelmask: torch.Tensor, stride: int = 16, tile_mask_ratio: float = 0.5) -> torch.Tensor:
    lib, dev = _lib.load(), _dev(pixelmask)
    H, W = int(pixelmask.shape[0]), int(pixelmask.shape[1])
    gy, gx = _grid(H, W, stride)
    out = torch.empty(gy, gx, dtype=torch.int32, device=dev)
    tmp = torch.empty(gy * gx, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_transmission2tilemask(_p(_u8(pixelmask)), H, W, int(stride), float(tile_mask_ratio), _p(out), _p(tmp),
                                            _stream(dev))
    _lib.check(rc, "rtgs_transmission2tilemask")
    return out


def colorerror2tilemask(color_error: torch.Tensor, stride: int = 16, top_ratio: float = 0.4) -> torch.Tensor:
    lib, dev = _lib.load(), _dev(color_error)
    H, W = int(color_error.shape[0]), int(color_error.shape[1])
    gy, gx = _grid(H, W, stride)
    out = torch.empty(gy, gx, dtype=torch.int32, device=dev)
    tmp = torch.empty(gy * gx, dtype=torch.float32, device=dev)
    err = color_error.float().contiguous()
    k = int(gy * gx * top_ratio)        # in double, as the reference's int(torch.numel(...) * top_ratio) (SLAM/utils.py:708-734)
    with torch.cuda.device(dev):
        rc = lib.rtgs_colorerror2tilemask_k(_p(err), H, W, int(stride), k, _p(out), _p(tmp), _stream(dev))
    _lib.check(rc, "rtgs_colorerror2tilemask")
    return out


def render_range(T_map: torch.Tensor, tile_mask_ratio: float = 0.5):
    """mapper.py:500-508 in one call: (render_mask bool[H,W], tile_mask int32[gy,gx], count uint32[1] on device);
    render_ratio = count / (H*W) - left on the device so the caller decides when (if ever) to synchronise."""
    lib, dev = _lib.load(), _dev(T_map)
    H, W = int(T_map.shape[-2]), int(T_map.shape[-1])
    T = T_map.float().contiguous()
    gy, gx = _grid(H, W, 16)
    mask = torch.empty(H, W, dtype=torch.uint8, device=dev)
    tile = torch.empty(gy, gx, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    tmp = torch.empty(gy * gx, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_render_range(_p(T), H, W, float(tile_mask_ratio), _p(mask), _p(tile), _p(count), _p(tmp), _stream(dev))
    _lib.check(rc, "rtgs_render_range")
    return mask.bool(), tile, count


def distCUDA2(points: torch.Tensor, return_dist2: bool = False):
    """`simple_knn._C.distCUDA2` of RTG-SLAM's fork: (mean squared distance to the 3 nearest other points [N],
    their indices [N,3] int32).  Exact.  With fewer than 4 points the missing neighbours come back as index -1 with
    distance FLT_MAX (and the mean accordingly): a caller that indexes with the result, as gaussian_pointcloud.py:376-389
    does, must not pass fewer than 4 points - a -1 would silently wrap to the last row."""
    lib, dev = _lib.load(), _dev(points)
    pts = points.float().contiguous()
    N = int(pts.shape[0])
    mean = torch.empty(N, dtype=torch.float32, device=dev)
    idx = torch.empty(N, 3, dtype=torch.int32, device=dev)
    d3 = torch.empty(N, 3, dtype=torch.float32, device=dev) if return_dist2 else None
    scratch = torch.empty(lib.rtgs_knn3_scratch_bytes(N), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_knn3(_p(pts), N, _p(mean), _p(idx), _p(d3), _p(scratch), _stream(dev))
    _lib.check(rc, "rtgs_knn3")
    return (mean, idx, d3) if return_dist2 else (mean, idx)


def knn_query(ref_points: torch.Tensor, query_points: torch.Tensor, self_offset: int = -1,
              ref_box: Optional[torch.Tensor] = None):
    """The three nearest REFERENCE points of every query point -> (dist2 [Nq,3] ascending, idx [Nq,3] int32 into
    ref_points; -1 / FLT_MAX where fewer than three exist).  Exact.  Stands where the reference calls
    pytorch3d.ops.knn_points(temp_xyz, exist_xyz, K=3) (mapper.py:803-827; note knn_points returns SQUARED distances
    too).  self_offset >= 0: query i is reference self_offset + i and is not its own neighbour.  ref_box float[6]
    (lo xyz, hi xyz, device): references outside the open box are ignored - bbox_filter (SLAM/utils.py:737-744) without
    the compaction."""
    lib, dev = _lib.load(), _dev(query_points)
    ref = ref_points.float().contiguous()
    q = query_points.float().contiguous()
    Nr, Nq = int(ref.shape[0]), int(q.shape[0])
    idx = torch.empty(Nq, 3, dtype=torch.int32, device=dev)
    d3 = torch.empty(Nq, 3, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.rtgs_knn3_query_scratch_bytes(Nr, Nq), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        box = None if ref_box is None else ref_box.to(device=dev, dtype=torch.float32).reshape(6).contiguous()
        rc = lib.rtgs_knn3_query(_p(ref), Nr, _p(q), Nq, int(self_offset), _p(box), _p(idx), _p(d3), _p(scratch), _stream(dev))
    _lib.check(rc, "rtgs_knn3_query")
    return d3, idx


def knn_build_ref(ref_points: torch.Tensor) -> torch.Tensor:
    """The search structure of `ref_points` [Nr >= 1, 3] as an opaque byte tensor (include/rtgs_slam.h: rtgs_knn3_build_ref); it
    holds a copy of the points - build again when they change."""
    lib, dev = _lib.load(), _dev(ref_points)
    ref = ref_points.float().contiguous()
    Nr = int(ref.shape[0])
    built = torch.empty(lib.rtgs_knn3_built_bytes(Nr), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_knn3_build_ref(_p(ref), Nr, _p(built), _stream(dev))
    _lib.check(rc, "rtgs_knn3_build_ref")
    return built


def knn_query_built(built: torch.Tensor, Nr: int, query_points: torch.Tensor, ref_box: Optional[torch.Tensor] = None):
    """knn_query(ref, query, -1, ref_box) against a structure knn_build_ref left -> (dist2 [Nq,3], idx [Nq,3] into ref)."""
    lib, dev = _lib.load(), _dev(query_points)
    q = query_points.float().contiguous()
    Nq = int(q.shape[0])
    idx = torch.empty(Nq, 3, dtype=torch.int32, device=dev)
    d3 = torch.empty(Nq, 3, dtype=torch.float32, device=dev)
    qs = torch.empty(lib.rtgs_knn3_query_built_scratch_bytes(Nq), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        box = None if ref_box is None else ref_box.to(device=dev, dtype=torch.float32).reshape(6).contiguous()
        rc = lib.rtgs_knn3_query_built(_p(built), int(Nr), _p(q), Nq, _p(box), _p(idx), _p(d3), _p(qs), _stream(dev))
    _lib.check(rc, "rtgs_knn3_query_built")
    return d3, idx


def knn_dynamic_merge(query_points, unstable_points, n_stable: int, d2_stable, idx_stable, ref_box=None):
    """The three nearest of every query among its stable neighbours (knn_query_built over the stable rows), the other queries
    and the unstable points, the last two compared directly (include/rtgs_slam.h: rtgs_knn3_dynamic_merge) -> (dist2, idx) with
    idx into cat(queries, stable rows, unstable points) - what knn_query(cat(query, existing), query, 0, box) returns."""
    lib, dev = _lib.load(), _dev(query_points)
    q = query_points.float().contiguous()
    u = unstable_points.float().contiguous()
    Nq, Nu = int(q.shape[0]), int(u.shape[0])
    d2s, ids = d2_stable.float().contiguous(), idx_stable.to(torch.int32).contiguous()
    idx = torch.empty(Nq, 3, dtype=torch.int32, device=dev)
    d3 = torch.empty(Nq, 3, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        box = None if ref_box is None else ref_box.to(device=dev, dtype=torch.float32).reshape(6).contiguous()
        rc = lib.rtgs_knn3_dynamic_merge(_p(q), Nq, _p(u) if Nu else None, Nu, int(n_stable), _p(d2s), _p(ids), _p(box), _p(idx), _p(d3),
                                         _stream(dev))
    _lib.check(rc, "rtgs_knn3_dynamic_merge")
    return d3, idx


def accumulate_gaussian_error(H, W, P, color_error, depth_error, normal_error, color_index, depth_index, color_thres,
                              depth_thres, normal_thres, mean=True):
    """`cuda_utils._C.accumulate_gaussian_error` (frozen semantics, include/rtgs_slam.h) ->
    (gaussian_color_error[P], gaussian_depth_error[P], gaussian_normal_error[P], outlier_count[P] int32)."""
    lib, dev = _lib.load(), _dev(color_error)
    H, W, P = int(H), int(W), int(P)
    f = lambda t: t.float().contiguous()
    i = lambda t: t.to(torch.int32).contiguous()
    ce, de, ne, ci, di = f(color_error), f(depth_error), f(normal_error), i(color_index), i(depth_index)
    for t in (ce, de, ne, ci, di):
        if t.numel() != H * W:
            raise RuntimeError(f"accumulate_gaussian_error: expected {H}x{W} maps, got {tuple(t.shape)}")
    buf = torch.empty(6 * max(P, 1), dtype=torch.float32, device=dev)      # one allocation: the C side clears it in one go
    gc, gd, gn = buf[0:P], buf[P:2 * P], buf[2 * P:3 * P]
    oc = buf[3 * P:4 * P].view(torch.int32)
    scratch = buf[4 * P:]
    with torch.cuda.device(dev):
        rc = lib.rtgs_accumulate_error(H, W, P, _p(ce), _p(de), _p(ne), _p(ci), _p(di), float(color_thres),
                                       float(depth_thres), float(normal_thres), int(bool(mean)), _p(gc), _p(gd), _p(gn),
                                       _p(oc), _p(scratch), _stream(dev))
    _lib.check(rc, "rtgs_accumulate_error")
    return gc, gd, gn, oc


def bilateralFilter_torch(depth: torch.Tensor, radius: int, sigma_color: float, sigma_space: float) -> torch.Tensor:
    lib, dev = _lib.load(), _dev(depth)
    H, W = int(depth.shape[0]), int(depth.shape[1])
    d = depth.reshape(H, W).float().contiguous()
    out = torch.empty(H, W, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_bilateral_filter(_p(d), H, W, int(radius), float(sigma_color), float(sigma_space), _p(out), _stream(dev))
    _lib.check(rc, "rtgs_bilateral_filter")
    return out.reshape(H, W, 1)


def frame_preprocess(depth_map: torch.Tensor, K: torch.Tensor, min_depth: float = 0.3, max_depth: float = 5.0,
                     depth_filter: bool = False, invalid_confidence_thresh: float = 0.2):
    """The map part of Tracker.map_preprocess (tracker.py:104-131): dict with depth_map [H,W,1], vertex_map_c [H,W,3],
    normal_map_c [H,W,3], confidence_map [H,W,1], invalid_confidence_mask bool [H,W]."""
    lib, dev = _lib.load(), _dev(depth_map)
    H, W = int(depth_map.shape[0]), int(depth_map.shape[1])
    d = depth_map.reshape(H, W).float().contiguous()
    if depth_filter:
        d = bilateralFilter_torch(d, 5, 2, 2).reshape(H, W)
    Kd = K.to(device=dev, dtype=torch.float32).contiguous()
    f = dict(dtype=torch.float32, device=dev)
    dout, vout, nout, cout = torch.empty(H, W, 1, **f), torch.empty(H, W, 3, **f), torch.empty(H, W, 3, **f), torch.empty(H, W, 1, **f)
    bad = torch.empty(H, W, dtype=torch.uint8, device=dev)
    scratch = torch.empty(lib.rtgs_frame_preprocess_scratch_bytes(H, W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_frame_preprocess(_p(d), H, W, _p(Kd), float(min_depth), float(max_depth), float(invalid_confidence_thresh),
                                       _p(dout), _p(vout), _p(nout), _p(cout), _p(bad), _p(scratch), _stream(dev))
    _lib.check(rc, "rtgs_frame_preprocess")
    return dict(depth_map=dout, vertex_map_c=vout, normal_map_c=nout, confidence_map=cout, invalid_confidence_mask=bad.bool())


def add_masks(T_map, depth, render_depth, render_color, frame_color, depth_index, thr_transmission, thr_depth, thr_color):
    """Mapping.temp_points_init's two sampling masks (mapper.py:728-775) in one pass ->
    (transmission_mask uint8 [H,W], error_mask uint8 [H,W], counts int32[2] on the device).  Maps: any layout with H*W
    elements per plane; colours [3,H,W]."""
    lib, dev = _lib.load(), _dev(T_map)
    H, W = int(render_color.shape[-2]), int(render_color.shape[-1])
    f = lambda t: t.float().contiguous()
    tm = torch.empty(H, W, dtype=torch.uint8, device=dev)
    em = torch.empty(H, W, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    T, d, rd, rc, fc, di = f(T_map), f(depth), f(render_depth), f(render_color), f(frame_color), depth_index.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        rc_ = lib.rtgs_add_masks(_p(T), _p(d), _p(rd), _p(rc), _p(fc), _p(di), H, W, float(thr_transmission), float(thr_depth),
                                 float(thr_color), _p(tm), _p(em), _p(counts), _stream(dev))
    _lib.check(rc_, "rtgs_add_masks")
    return tm, em, counts


def frame_errors(depth, render_depth, render_color, frame_color, depth_index):
    """The error maps of Mapping.error_gaussians_remove (mapper.py:527-540) -> (color_error [H,W], depth_error [H,W])."""
    lib, dev = _lib.load(), _dev(depth)
    H, W = int(render_color.shape[-2]), int(render_color.shape[-1])
    f = lambda t: t.float().contiguous()
    ce = torch.empty(H, W, dtype=torch.float32, device=dev)
    de = torch.empty(H, W, dtype=torch.float32, device=dev)
    d, rd, rc, fc, di = f(depth), f(render_depth), f(render_color), f(frame_color), depth_index.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        rc_ = lib.rtgs_frame_errors(_p(d), _p(rd), _p(rc), _p(fc), _p(di), H, W, _p(ce), _p(de), _stream(dev))
    _lib.check(rc_, "rtgs_frame_errors")
    return ce, de


def attach_test(points, w2c, fx, fy, cx, cy, H, W, stable_color_index, stable_xyz, stable_normal, max_plane_dist):
    """Mapping.temp_points_attach's test (mapper.py:830-883) -> uint8 [n]: 1 where the point projects onto a pixel owned by a
    stable Gaussian and lies within `max_plane_dist` of its plane."""
    lib, dev = _lib.load(), _dev(points)
    pts = points.float().contiguous()
    n = int(pts.shape[0])
    out = torch.empty(n, dtype=torch.uint8, device=dev)
    T = w2c.to(device=dev, dtype=torch.float32).contiguous()
    ci = stable_color_index.to(torch.int32).contiguous()
    sx, sn = stable_xyz.float().contiguous(), stable_normal.float().contiguous()
    with torch.cuda.device(dev):
        rc = lib.rtgs_attach_test(_p(pts), n, _p(T), float(fx), float(fy), float(cx), float(cy), int(H), int(W), _p(ci), _p(sx), _p(sn),
                                  float(max_plane_dist), _p(out), _stream(dev))
    _lib.check(rc, "rtgs_attach_test")
    return out


def error_counters(g_color, g_depth, nf, color_strike_thr, depth_strike_thr, depth_counter, color_counter, limit=10, sync=True):
    """Strikes, delete / release decisions and their counts for the first `nf` rows in one kernel (include/rtgs_slam.h:
    rtgs_error_counters; mapper.py:541-565).  The int32 counters [>= nf, 1] are updated IN PLACE.
    -> (delete_mask uint8 [nf], release_mask uint8 [nf], (n_delete, n_release))  - one host synchronisation."""
    lib, dev = _lib.load(), _dev(g_color)
    nf = int(nf)
    ddel = torch.empty(nf, dtype=torch.uint8, device=dev)
    crel = torch.empty(nf, dtype=torch.uint8, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    assert depth_counter.dtype == torch.int32 and color_counter.dtype == torch.int32 and depth_counter.is_contiguous() and color_counter.is_contiguous()
    with torch.cuda.device(dev):
        rc = lib.rtgs_error_counters(nf, _p(g_color.float().contiguous()), _p(g_depth.float().contiguous()), float(color_strike_thr),
                                     float(depth_strike_thr), _p(depth_counter), _p(color_counter), int(limit), _p(ddel), _p(crel),
                                     _p(counts), _stream(dev))
    _lib.check(rc, "rtgs_error_counters")
    if not sync:
        return ddel, crel, counts                    # int32[2] on the device: the caller reads it with its other counts
    n_del, n_rel = counts.tolist()
    return ddel, crel, (n_del, n_rel)


def delete_mask(scales, add_tick, time_now, window, sync=True):
    """Mapping.gaussians_delete's mask for one cloud (rtgs_delete_mask) -> (mask uint8 [n], count)."""
    lib, dev = _lib.load(), _dev(scales)
    sc = scales.float().contiguous()
    n = int(sc.shape[0])
    mask = torch.empty(n, dtype=torch.uint8, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    tick = None if add_tick is None else add_tick.to(torch.int32).contiguous()
    with torch.cuda.device(dev):
        rc = lib.rtgs_delete_mask(n, _p(sc), _p(tick), int(time_now), int(window), _p(mask), _p(count), _stream(dev))
    _lib.check(rc, "rtgs_delete_mask")
    return mask, (int(count.item()) if sync else count)


def gather_new_points(pick, vertex_map, normal_map, color_map, identity_rot: bool):
    """The sampled pixels of one pass -> (xyz [n,3], unit normal [n,3], colour [n,3], rotation [n,4]) in one kernel
    (include/rtgs_slam.h: rtgs_gather_new_points).  Not for a pass of exactly three points (Mapping keeps the torch form there)."""
    lib, dev = _lib.load(), _dev(vertex_map)
    pick = pick.to(torch.int64).contiguous()
    n = int(pick.shape[0])
    v, nm, c = (t.float().contiguous() for t in (vertex_map, normal_map, color_map))
    xyz, nrm, col = (torch.empty(n, 3, dtype=torch.float32, device=dev) for _ in range(3))
    rot = torch.empty(n, 4, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_gather_new_points(_p(pick), n, _p(v), _p(nm), _p(c), int(bool(identity_rot)), _p(xyz), _p(nrm), _p(col), _p(rot),
                                        _stream(dev))
    _lib.check(rc, "rtgs_gather_new_points")
    return xyz, nrm, col, rot


def draw_new_points(passes, vertex_map, normal_map, color_map, identity_rot: bool, want_pick: bool = False):
    """Every sampling pass of a frame - `passes` = [(cand int32[>= n_cand], n_cand, k, key)] - into ONE set of arrays (xyz, unit
    normal, colour, rotation; pass after pass), one launch per pass (include/rtgs_slam.h: rtgs_draw_new_points): the draw without
    replacement happens inside the kernel (a keyed bijection of [0, n_cand)), so there is no randperm, no index gather and no
    concatenation.  No pass may have k == 3 (Mapping keeps the torch form there)."""
    lib, dev = _lib.load(), _dev(vertex_map)
    total = sum(int(p[2]) for p in passes)
    v, nm, c = (t.float().contiguous() for t in (vertex_map, normal_map, color_map))
    xyz, nrm, col = (torch.empty(total, 3, dtype=torch.float32, device=dev) for _ in range(3))
    rot = torch.empty(total, 4, dtype=torch.float32, device=dev)
    pick = torch.empty(total, dtype=torch.int32, device=dev) if want_pick else None
    off = 0
    with torch.cuda.device(dev):
        st = _stream(dev)
        for cand, n_cand, k, key in passes:
            k = int(k)
            if k <= 0:
                continue
            assert cand.dtype == torch.int32 and cand.is_contiguous()
            rc = lib.rtgs_draw_new_points(_p(cand), int(n_cand), k, int(key) & 0xFFFFFFFFFFFFFFFF, _p(v), _p(nm), _p(c),
                                          int(bool(identity_rot)), xyz.data_ptr() + 12 * off, nrm.data_ptr() + 12 * off,
                                          col.data_ptr() + 12 * off, rot.data_ptr() + 16 * off,
                                          None if pick is None else pick.data_ptr() + 4 * off, st)
            _lib.check(rc, "rtgs_draw_new_points")
            off += k
    return (xyz, nrm, col, rot, pick) if want_pick else (xyz, nrm, col, rot)


def filter_keep(d2, idx, scales, ratio: float = 0.6):
    """temp_points_filter's decision (mapper.py:812-826) from the neighbour query's output and the unstable Gaussians' activated
    scales -> keep bool[n] (include/rtgs_slam.h: rtgs_filter_keep)."""
    lib, dev = _lib.load(), _dev(d2)
    n = int(d2.shape[0])
    d2, idx, scales = d2.float().contiguous(), idx.to(torch.int32).contiguous(), scales.float().contiguous()
    keep = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_filter_keep(n, _p(d2), _p(idx), _p(scales), float(ratio), _p(keep), _stream(dev))
    _lib.check(rc, "rtgs_filter_keep")
    return keep.view(torch.bool)


def bbox_pad(xyz, pad: float):
    """[min - pad | max + pad] float32[6] of the points (include/rtgs_slam.h: rtgs_bbox_pad)."""
    lib, dev = _lib.load(), _dev(xyz)
    xyz = xyz.float().contiguous()
    out = torch.empty(6, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_bbox_pad(int(xyz.shape[0]), _p(xyz), float(pad), _p(out), _stream(dev))
    _lib.check(rc, "rtgs_bbox_pad")
    return out


def compact_points(keep, xyz, color, opacity_raw, rots):
    """The candidates with keep != 0, in order, and how many (ONE host synchronisation) - Mapping.temp_to_optimize's first
    compaction in one launch (include/rtgs_slam.h: rtgs_compact_points).  -> (xyz, color, opacity_raw, rots, n)"""
    lib, dev = _lib.load(), _dev(xyz)
    n = int(xyz.shape[0])
    k8 = keep.view(torch.uint8) if keep.dtype == torch.bool else _u8(keep)
    f = lambda t: t.float().contiguous()
    xyz, color, opacity_raw, rots = f(xyz), f(color), f(opacity_raw), f(rots)
    o_xyz, o_col = torch.empty(n, 3, dtype=torch.float32, device=dev), torch.empty(n, 3, dtype=torch.float32, device=dev)
    o_op, o_rot = torch.empty(n, 1, dtype=torch.float32, device=dev), torch.empty(n, 4, dtype=torch.float32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_compact_points(n, _p(k8.contiguous()), _p(xyz), _p(color), _p(opacity_raw), _p(rots), _p(o_xyz), _p(o_col), _p(o_op),
                                     _p(o_rot), _p(count), _stream(dev))
    _lib.check(rc, "rtgs_compact_points")
    m = int(count.item())
    return o_xyz[:m], o_col[:m], o_op[:m], o_rot[:m], m


def new_rows(xyz, color, opacity_raw, rots, d2, idx, exist_scales, min_radius, max_radius, scale_factor, xyz_factor):
    """update_geometry + row packing for the new Gaussians of a frame in one kernel (include/rtgs_slam.h: rtgs_new_rows) ->
    (packed [n,59] raw rows of every candidate, valid uint8 [n])."""
    lib, dev = _lib.load(), _dev(xyz)
    n = int(xyz.shape[0])
    f = lambda t: t.float().contiguous()
    xyz, color, opacity_raw, rots, d2 = f(xyz), f(color), f(opacity_raw), f(rots), f(d2)
    idx = idx.to(torch.int32).contiguous()
    es = f(exist_scales) if exist_scales is not None and exist_scales.numel() else None
    packed = torch.empty(n, 59, dtype=torch.float32, device=dev)
    valid = torch.empty(n, dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_new_rows(n, _p(xyz), _p(color), _p(opacity_raw), _p(rots), _p(d2), _p(idx), _p(es), float(min_radius),
                               float(max_radius), float(scale_factor), float(xyz_factor[0]), float(xyz_factor[1]),
                               float(xyz_factor[2]), _p(packed), _p(valid), _stream(dev))
    _lib.check(rc, "rtgs_new_rows")
    return packed, valid


def transform_map(map3: torch.Tensor, transform: torch.Tensor) -> torch.Tensor:
    """SLAM/utils.py:56-63: every 3-vector of `map3` [..., 3] through the 4x4 `transform` (its rotation only when the
    caller passes get_rot(c2w), as for normal maps).  One streaming kernel (a `@` would be a K = 3 GEMM)."""
    lib, dev = _lib.load(), _dev(map3)
    m = map3.float().contiguous()
    T = transform.to(device=dev, dtype=torch.float32).contiguous()
    out = torch.empty_like(m)
    with torch.cuda.device(dev):
        rc = lib.rtgs_transform_map(_p(m), m.numel() // 3, _p(T), _p(out), _stream(dev))
    _lib.check(rc, "rtgs_transform_map")
    return out


def sample_candidates(normal_map: torch.Tensor, select_mask: Optional[torch.Tensor] = None):
    """Flat indices (ascending) of the pixels sample_pixels may draw from, and their number (device int32[1])."""
    lib, dev = _lib.load(), _dev(normal_map)
    H, W = int(normal_map.shape[0]), int(normal_map.shape[1])
    n = normal_map.float().contiguous()
    sel = None if select_mask is None else _u8(select_mask.reshape(H, W))
    idx = torch.empty(H * W, dtype=torch.int32, device=dev)
    count = torch.empty(1, dtype=torch.int32, device=dev)
    flags = torch.empty(H * W, dtype=torch.uint8, device=dev)
    scratch = torch.empty(lib.rtgs_compact_scratch_bytes(H * W), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        rc = lib.rtgs_sample_candidates(_p(n), _p(sel), H, W, _p(idx), _p(count), _p(flags), _p(scratch), _stream(dev))
    _lib.check(rc, "rtgs_sample_candidates")
    return idx, count


def sample_pixels(vertex_map, normal_map, color_map, uniform_sample_num, select_mask=None, generator=None):
    """SLAM/utils.py:141-183: a uniform draw without replacement of min(uniform_sample_num, #candidates) pixels among
    select_mask (all if None) minus zero-normal pixels -> (points [n,3], normals [n,3], colors [n,3])."""
    assert uniform_sample_num >= 0
    dev = _dev(vertex_map)
    if uniform_sample_num == 0:
        e = torch.empty(0, device=dev)
        return e, e.clone(), e.clone()
    idx, count = sample_candidates(normal_map, select_mask)
    n_cand = int(count.item())                       # the reference synchronises here too (boolean-mask indexing)
    n = min(int(uniform_sample_num), n_cand)
    pick = idx[:n_cand][torch.randperm(n_cand, device=dev, generator=generator)[:n]].long()
    return (vertex_map.reshape(-1, 3)[pick].view(n, 3), normal_map.reshape(-1, 3)[pick].view(n, 3),
            color_map.reshape(-1, 3)[pick].view(n, 3))
