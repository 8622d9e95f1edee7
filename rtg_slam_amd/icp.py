"""Host side of the ICP frame-to-model tracker: an `IcpTracker` with the public surface of
/root/reference/SLAM/icp.py:357-452 (same constructor argument, same five methods, same
return values), over the C ABI of include/rtgs_icp.h.  The whole 3-level x 5-iteration
Gauss-Newton loop runs on the device with the pose resident there; `predict_pose` performs
exactly one device->host copy (the 4x4 pose + the two scalars it must return / test).

Function-level ops for monkey-patch style parity tests: `build_pyramids`, `icp_step`,
`icp_track`, `fill_model_depth`."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np
import torch

from . import _lib


def _require_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("rtg_slam_amd.icp: tensors must live on a HIP device; this build has no CPU path.")


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _vp(t: torch.Tensor):
    return C.c_void_p(t.data_ptr())


_scratch = {}
_builds = {}            # pyramid builds per scratch: the min / max set alternates (include/rtgs_icp.h)
FLAG_PERSISTENT, FLAG_CLUSTER, FLAG_SCRATCH_READY, FLAG_FROM_IDENTITY, FLAG_F32_SOLVE = 1, 2, 4, 8, 16
PYR_SCRATCH_READY, PYR_SECOND_SET = 1, 2


def _scratch_key(dev):
    return (dev.type, dev.index, torch.cuda.current_stream(dev).cuda_stream)


def _get_scratch(dev) -> torch.Tensor:
    """One scratch per (device, stream), armed once (rtgs_icp_scratch_init): the calls below then issue no memsets."""
    key = _scratch_key(dev)
    s = _scratch.get(key)
    if s is None:
        lib = _lib.load()
        s = torch.empty(lib.rtgs_icp_scratch_bytes(), dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            _lib.check(lib.rtgs_icp_scratch_init(_vp(s), C.c_void_p(key[2])), "rtgs_icp_scratch_init")
        _scratch[key] = s
        _builds[key] = 0
    return s


def build_pyramids(depth: torch.Tensor, K: torch.Tensor, levels: int = 3) -> Tuple[List[torch.Tensor], List[torch.Tensor]]:
    """depth [H,W,1] or [H,W] -> (vertex_pyramid, normal_pyramid), coarsest first, each [H_l,W_l,3].
    = build_vertex_pyramid(depth, ImagePyramids([L-1..0], 'max'), K) + build_normal_pyramid
    (SLAM/utils.py:511-527, icp.py:374)."""
    lib = _lib.load()
    _require_device(depth)
    dev = depth.device
    depth = _f32c(depth)
    H, W = int(depth.shape[0]), int(depth.shape[1])
    K = _f32c(K.to(dev))
    verts, norms = [], []
    for l in range(levels):
        sh = levels - 1 - l
        verts.append(torch.empty(H >> sh, W >> sh, 3, dtype=torch.float32, device=dev))
        norms.append(torch.empty(H >> sh, W >> sh, 3, dtype=torch.float32, device=dev))
    vp = (C.c_void_p * levels)(*[v.data_ptr() for v in verts])
    npp = (C.c_void_p * levels)(*[n.data_ptr() for n in norms])
    stream = torch.cuda.current_stream(dev).cuda_stream
    scratch = _get_scratch(dev)
    key = _scratch_key(dev)
    flags = PYR_SCRATCH_READY | (PYR_SECOND_SET if _builds[key] & 1 else 0)
    with torch.cuda.device(dev):
        rc = lib.rtgs_icp_build_pyramids_ex(_vp(depth), H, W, _vp(K), levels, vp, npp, _vp(scratch), flags,
                                            C.c_void_p(stream))
    _lib.check(rc, "rtgs_icp_build_pyramids_ex")
    _builds[key] += 1          # only a build that was launched flips the min / max set (a refused call re-armed nothing)
    return verts, norms


def icp_step(v_src, n_src, v_tgt, n_tgt, K, pose, dist_thr: float, cos_thr: float):
    """One evaluation of ICP.compute_residuals_jacobian + compute_jtj + compute_jtr
    (icp.py:52-119) -> (JtJ[6,6], Jtr[6], n_valid[1]) device tensors.  K is the level's K."""
    lib = _lib.load()
    _require_device(v_src)
    dev = v_src.device
    v_src, n_src, v_tgt, n_tgt = map(_f32c, (v_src, n_src, v_tgt, n_tgt))
    H, W = int(v_src.shape[0]), int(v_src.shape[1])
    K = _f32c(K.to(dev))
    pose = _f32c(pose.to(dev))
    JtJ = torch.empty(6, 6, dtype=torch.float32, device=dev)
    Jtr = torch.empty(6, dtype=torch.float32, device=dev)
    nv = torch.empty(1, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        rc = lib.rtgs_icp_step(_vp(v_src), _vp(n_src), _vp(v_tgt), _vp(n_tgt), H, W, _vp(K), _vp(pose),
                               float(dist_thr), float(cos_thr), _vp(JtJ), _vp(Jtr), _vp(nv),
                               _vp(_get_scratch(dev)), C.c_void_p(stream))
    _lib.check(rc, "rtgs_icp_step")
    return JtJ, Jtr, nv


def icp_track(vertex_src: Sequence[torch.Tensor], normal_src, vertex_tgt, normal_tgt, K: torch.Tensor,
              downscales: Sequence[float], iters: Sequence[int], dist_thr: float, cos_thr: float,
              damping: float, pose0: torch.Tensor | None = None, persistent: bool = False,
              f32_solve: bool = False) -> torch.Tensor:
    """The level loop of IcpTracker.predict_pose (icp.py:428-447) on the device.
    Returns a device float32[20]: pose (16, row-major) + [valid_ratio, p2p_loss, n_singular, aborted].
    f32_solve: the Gauss-Newton update in float32 in the reference's order of operations (RTGS_ICP_FLAG_F32_SOLVE,
    include/rtgs_icp.h) - a measurement aid, launch-per-iteration chain only."""
    lib = _lib.load()
    dev = vertex_src[0].device
    _require_device(vertex_src[0])
    n = len(downscales)
    keep = []
    lv = (_lib.IcpLevelC * n)()
    for l in range(n):
        ts = [_f32c(t[l]) for t in (vertex_src, normal_src, vertex_tgt, normal_tgt)]
        keep.append(ts)
        lv[l] = _lib.IcpLevelC(int(ts[0].shape[0]), int(ts[0].shape[1]), float(downscales[l]), int(iters[l]),
                               ts[0].data_ptr(), ts[1].data_ptr(), ts[2].data_ptr(), ts[3].data_ptr())
    K = _f32c(K.to(dev))
    out = torch.empty(20, dtype=torch.float32, device=dev)
    flags = (FLAG_PERSISTENT if persistent else 0) | FLAG_SCRATCH_READY | (FLAG_F32_SOLVE if f32_solve else 0)
    if pose0 is None:
        flags |= FLAG_FROM_IDENTITY            # the first iteration's last workgroup writes the pose: no fill / copy launches
    else:
        out[:16] = _f32c(pose0.to(dev)).reshape(-1)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        rc = lib.rtgs_icp_track(lv, n, _vp(K), float(dist_thr), float(cos_thr), float(damping),
                                C.c_void_p(out.data_ptr()), C.c_void_p(out.data_ptr() + 64),
                                _vp(_get_scratch(dev)), flags, C.c_void_p(stream))
    _lib.check(rc, "rtgs_icp_track")
    return out


def fill_model_depth(render_depth, frame_depth, render_normal, frame_normal, dist_thr: float, normal_thr: float):
    """In-place hole filling of the rendered model depth (icp.py:397-415)."""
    lib = _lib.load()
    _require_device(render_depth)
    if render_depth.dtype != torch.float32 or not render_depth.is_contiguous():
        raise RuntimeError("render_depth must be a contiguous float32 tensor (it is updated in place)")
    dev = render_depth.device
    H, W = int(render_depth.shape[0]), int(render_depth.shape[1])
    fd, rn, fn = _f32c(frame_depth), _f32c(render_normal), _f32c(frame_normal)
    stream = torch.cuda.current_stream(dev).cuda_stream
    with torch.cuda.device(dev):
        rc = lib.rtgs_icp_fill_model_depth(_vp(render_depth), _vp(fd), _vp(rn), _vp(fn), H, W, float(dist_thr),
                                           float(normal_thr), C.c_void_p(stream))
    _lib.check(rc, "rtgs_icp_fill_model_depth")
    return render_depth


class IcpTracker:
    """Drop-in for SLAM/icp.py:357-452.  `args` needs the attributes the reference reads
    (icp.py:358-383): icp_downscales, icp_warmup_frames, icp_use_model_depth,
    icp_downscale_iters, icp_distance_threshold, icp_normal_threshold (degrees), icp_damping,
    verbose, icp_sample_distance_threshold, icp_sample_normal_threshold, icp_fail_threshold."""

    def __init__(self, args):
        self.icp_downscales = list(args.icp_downscales)
        self.icp_warmup_frames = args.icp_warmup_frames
        self.icp_use_model_depth = args.icp_use_model_depth
        self.icp_downscale_iters = list(args.icp_downscale_iters)
        self.icp_distance_threshold = float(args.icp_distance_threshold)
        self.icp_normal_threshold = float(np.cos(np.deg2rad(args.icp_normal_threshold)))
        self.icp_damping = float(args.icp_damping)
        self.icp_sample_distance_threshold = args.icp_sample_distance_threshold
        self.icp_sample_normal_threshold = args.icp_sample_normal_threshold
        self.icp_fail_threshold = args.icp_fail_threshold
        self.verbose = getattr(args, "verbose", False)
        self.persistent = bool(getattr(args, "icp_persistent", False))     # extension: one persistent kernel per track
        n = len(self.icp_downscales)
        # the reference's pyramid builder pools by 2^(n-1-l) (icp.py:374); its level loop scales K
        # by icp_downscales[l] (icp.py:431-433) - the two agree for the shipped [0.25, 0.5, 1.0]
        for l, ds in enumerate(self.icp_downscales):
            if abs(ds - 1.0 / (1 << (n - 1 - l))) > 1e-12:
                raise ValueError("icp_downscales must be [2^-(n-1), ..., 0.5, 1.0] (SLAM/icp.py:374)")
        self.normal_pyramid_t0 = None
        self.vertex_pyramid_t0 = None
        self.normal_pyramid_t1 = None
        self.vertex_pyramid_t1 = None
        self.last_model_depth = None
        self.depth_t1 = None
        self.K = None
        self.last_p2ploss = None
        self.last_valid_ratio = None

    def update_curr_status(self, depth_t1, K):
        if self.K is None:
            self.K = K
        self.depth_t1 = depth_t1
        self.vertex_pyramid_t1, self.normal_pyramid_t1 = build_pyramids(depth_t1, self.K, len(self.icp_downscales))

    def move_last_status(self):
        self.vertex_pyramid_t0 = self.vertex_pyramid_t1
        self.normal_pyramid_t0 = self.normal_pyramid_t1
        self.last_model_depth = self.depth_t1

    def update_last_status(self, frame, render_depth, frame_depth, render_normal, frame_normal):
        fill_model_depth(render_depth, frame_depth, render_normal, frame_normal,
                         self.icp_sample_distance_threshold, self.icp_sample_normal_threshold)
        self.last_model_depth = render_depth

    def predict_pose(self, frame):
        K = frame["K"]
        frame_id = frame["frame_id"]
        if self.vertex_pyramid_t0 is None:
            self.K = K
            return np.eye(4), True
        if self.icp_use_model_depth and frame_id >= self.icp_warmup_frames:
            self.vertex_pyramid_t0, self.normal_pyramid_t0 = build_pyramids(
                self.last_model_depth, self.K, len(self.icp_downscales))
        out = icp_track(self.vertex_pyramid_t1, self.normal_pyramid_t1, self.vertex_pyramid_t0,
                        self.normal_pyramid_t0, K, self.icp_downscales, self.icp_downscale_iters,
                        self.icp_distance_threshold, self.icp_normal_threshold, self.icp_damping,
                        persistent=self.persistent)
        host = out.cpu().numpy()                      # the single device->host copy of the frame
        if host[19] != 0:
            # the persistent kernel needs all of its workgroups co-resident for its grid barrier; when another stream (the
            # mapper) holds the CUs the bounded spin gives up - the pose was not touched, so take the track again with one
            # launch per Gauss-Newton iteration instead of failing the frame
            if not getattr(self, "_warned_persistent", False):
                import warnings
                warnings.warn("rtgs_icp_track: the persistent tracking kernel timed out at a grid barrier (device shared "
                              "with another stream?); falling back to one launch per Gauss-Newton iteration")
                self._warned_persistent = True
            out = icp_track(self.vertex_pyramid_t1, self.normal_pyramid_t1, self.vertex_pyramid_t0,
                            self.normal_pyramid_t0, K, self.icp_downscales, self.icp_downscale_iters,
                            self.icp_distance_threshold, self.icp_normal_threshold, self.icp_damping, persistent=False)
            host = out.cpu().numpy()
            if host[19] != 0:
                raise RuntimeError("rtgs_icp_track failed with and without the persistent kernel")
        pose_t1_t0 = host[:16].reshape(4, 4).copy()
        if host[18] != 0 or not np.isfinite(pose_t1_t0).all():
            # no valid correspondence at some iteration: J^T J = 0, the damped system H + trace(H) * damping * I stays
            # singular (the kernel counts the Gauss-Newton steps it had to skip in stats[2]) - the reference raises here
            # too (torch.inverse of a singular matrix, icp.py:313-325)
            raise RuntimeError("rtgs_icp_track: singular normal equations (no valid correspondences); "
                               "the reference's torch.inverse raises on the same input")
        self.last_valid_ratio = float(host[16])
        self.last_p2ploss = float(host[17])
        if self.verbose:
            print(self.last_p2ploss, self.last_valid_ratio)
        tracking_success = not (self.last_p2ploss > self.icp_fail_threshold)
        return pose_t1_t0, tracking_success
