"""The single-process SLAM loop of the reference (/root/reference/slam.py:56-95) on this package's pieces: per frame
`Tracker.map_preprocess` -> `Tracker.tracking` (IcpTracker.predict_pose, frame-to-model) -> `Mapping.mapping` ->
`Mapping.get_render_output` -> `Tracker.update_last_status`, with the reference's time recorder semantics
(utils/monitor.py:22-34: `tracking` and `mapping` are running means of the per-frame wall time, fps = 1 / mapping).

`Tracker` mirrors SLAM/multiprocess/tracker.py:97-290 without the ORB backend: the frame maps come from one fused kernel
(slam_ops.frame_preprocess), the pose from rtg_slam_amd.icp.IcpTracker.  BASELINE.json configs[2] is `run_sequence` over a
Replica-shaped stream; bench.py's `sequence` leg and tests/test_sequence_gpu.py call it.  HIP tensors only."""
from __future__ import annotations

import time
from typing import Callable, Iterable, Optional

import numpy as np
import torch

from .mapping import Frame, Mapping


class Tracker:
    def __init__(self, args, device):
        from . import slam_ops
        from .icp import IcpTracker
        self.args, self.device, self.so = args, device, slam_ops
        self.icp_tracker = IcpTracker(args)
        self.pose_es, self.pose_gt = [], []
        self.initialized = False
        self.curr_frame = None

    def map_preprocess(self, frame: Frame, depth: torch.Tensor, color: torch.Tensor, frame_id: int):
        """tracker.py:97-159: range mask, vertex / normal / confidence maps, confidence mask; the tracker's current status."""
        a = self.args
        fm = self.so.frame_preprocess(depth, frame.K, a.min_depth, a.max_depth, a.depth_filter, a.invalid_confidence_thresh)
        fm["color_map"] = color.permute(1, 2, 0).contiguous()
        fm["color_chw"] = color.contiguous()
        fm["depth_chw"] = fm["depth_map"].permute(2, 0, 1).contiguous()
        fm["time"] = frame_id
        self.curr_frame = {"K": frame.K, "frame_id": frame_id}
        self.icp_tracker.update_curr_status(fm["depth_map"], frame.K)
        return fm

    def tracking(self, frame: Frame, frame_map, pose_gt=None, init_pose=None):
        """tracker.py:259-290."""
        ok = True
        if pose_gt is not None:
            self.pose_gt.append(np.asarray(pose_gt, dtype=np.float64))
        if self.args.use_gt_pose:
            pose = self.pose_gt[-1]
        elif not self.initialized:
            self.initialized = True
            pose = np.eye(4) if init_pose is None else np.asarray(init_pose, dtype=np.float64)
        else:
            rel, ok = self.icp_tracker.predict_pose(self.curr_frame)
            pose = self.pose_es[-1] @ rel.astype(np.float64)
        self.icp_tracker.move_last_status()
        self.pose_es.append(pose)
        frame.updatePose(pose)
        c2w = frame.get_c2w                                                   # transform_map, SLAM/utils.py:56-63; tracker.py:283-288
        rot = frame.get_rot                                                   # get_rot(c2w): translation cleared
        frame_map["vertex_map_w"] = self.so.transform_map(frame_map["vertex_map_c"], c2w)
        frame_map["normal_map_w"] = self.so.transform_map(frame_map["normal_map_c"], rot)
        # transform_map moves the zero vertices of invalid pixels too; they are never sampled (depth 0 / zero normal masks)
        return ok

    def update_last_status(self, frame, render_depth, frame_depth, render_normal, frame_normal):
        self.icp_tracker.update_last_status(frame, render_depth, frame_depth, render_normal, frame_normal)

    def get_new_poses(self):
        """tracker.py:69-74: the trajectory as a back end (ORB-SLAM2 in the reference, out of scope here) has corrected it,
        or None without one - Mapping.update_poses(None) then leaves every frame where it is."""
        return None


def ate_rmse(pose_es, pose_gt, align: bool = False) -> float:
    """RMSE of the translation error; align=True first fits the rigid transform (Horn) the reference's eval applies."""
    E = np.stack([p[:3, 3] for p in pose_es])
    G = np.stack([p[:3, 3] for p in pose_gt])
    if align and len(E) >= 3:
        me, mg = E.mean(0), G.mean(0)
        U, _, Vt = np.linalg.svd((G - mg).T @ (E - me))
        S = np.eye(3)
        if np.linalg.det(U @ Vt) < 0:
            S[2, 2] = -1
        R = U @ S @ Vt
        E = (E - me) @ R.T + mg
    return float(np.sqrt(((E - G) ** 2).sum(1).mean()))


def run_sequence(cam, stream: Iterable, args, device, mapper: Optional[Mapping] = None, lr_scale: float = 1.0,
                 capacity: Optional[int] = None, on_frame: Optional[Callable] = None, final_global: bool = False):
    """slam.py:56-95 over `stream` = iterable of (depth [H,W] metres, colour [3,H,W] in 0..1, ground-truth c2w 4x4) on the
    device.  Returns (mapper, tracker, report): report["fps"] is the reference's definition, 1 / mean(mapping seconds per
    frame) (utils/monitor.py:22-24), next to the frame rate of the whole loop (tracking + mapping, sequential)."""
    mapper = mapper if mapper is not None else Mapping(args, device, capacity=capacity, lr_scale=lr_scale)
    tracker = Tracker(args, device)
    t_track, t_map, per_frame = 0.0, 0.0, []
    n = 0
    torch.cuda.synchronize(device)
    t_all = time.perf_counter()
    for frame_id, (depth, color, gt_c2w) in enumerate(stream):
        t0 = time.perf_counter()
        frame = Frame(cam, gt_c2w, device, uid=frame_id)
        frame_map = tracker.map_preprocess(frame, depth, color, frame_id)
        tracker.tracking(frame, frame_map, pose_gt=gt_c2w, init_pose=gt_c2w if frame_id == 0 else None)
        t1 = time.perf_counter()                                # predict_pose returned a host pose: the tracker is done
        mapper.update_poses(tracker.get_new_poses())            # slam.py:76-77
        mapper.mapping(frame, frame_map, frame_id)
        mm = mapper.get_render_output(frame)
        # update_last_status fills holes of the model depth IN PLACE (icp.py:397-415): hand it a copy, the render is cached
        tracker.update_last_status(frame, mm["render_depth"].clone(), frame_map["depth_map"],
                                   mm["render_normal"].contiguous(), frame_map["normal_map_w"])
        torch.cuda.synchronize(device)                          # the mapper's frame is over when its kernels are
        t2 = time.perf_counter()
        t_track += t1 - t0
        t_map += t2 - t1
        per_frame.append((t1 - t0, t2 - t1, mapper.opt.N, mapper.opt.n_frozen))
        if on_frame is not None:
            on_frame(frame_id, frame, frame_map, mapper, tracker)
        mapper.time += 1
        n += 1
    if final_global and n > 0:
        mapper.global_optimization(select_keyframe_num=-1, is_end=True)
        torch.cuda.synchronize(device)
    wall = time.perf_counter() - t_all
    es, gt = tracker.pose_es, tracker.pose_gt
    report = {
        "frames": n, "tracking_s_mean": t_track / max(n, 1), "mapping_s_mean": t_map / max(n, 1),
        "fps": (n / t_map) if t_map > 0 else None,                        # monitor.py:22-24: 1 / mean mapping time
        "fps_tracking_plus_mapping": (n / (t_track + t_map)) if n else None,
        "wall_s": wall,
        "ate_rmse_m": ate_rmse(es, gt) if n else None, "ate_rmse_aligned_m": ate_rmse(es, gt, True) if n else None,
        "final_translation_error_m": float(np.linalg.norm(es[-1][:3, 3] - gt[-1][:3, 3])) if n else None,
        "gaussians": int(mapper.opt.N), "stable": int(mapper.opt.n_frozen), "unstable": int(mapper.opt.n_train),
        "keyframes": mapper.get_keyframe_num, "stats": dict(mapper.stats),
        "stable_fraction_over_time": [round(p[3] / max(p[2], 1), 4) for p in per_frame[::max(1, n // 20)]],
        "gaussians_over_time": [p[2] for p in per_frame[::max(1, n // 20)]],
        "per_frame": per_frame,
        "stage_profile_ms_per_frame": None if mapper.prof is None else {k: round(1e3 * v / max(n, 1), 3) for k, v in mapper.prof.items()},
    }
    return mapper, tracker, report
