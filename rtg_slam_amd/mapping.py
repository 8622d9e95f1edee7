"""`Mapping`: the lifecycle of RTG-SLAM's map around the hot path - the callers of the rasterizer and of the one-call map
step - under the reference's own method names (/root/reference/SLAM/multiprocess/mapper.py), on the product's map object
(`ShardedMapOptimizer`: rows [stable | unstable], side arrays, local / global optimisation modes) and the HIP ops.

    mapping                  mapper.py:97-126     per frame: add, optimise every `gaussian_update_frame`-th frame, fix, prune
    gaussians_add            :128-132, 709-883    temp_points_init / _filter / _attach / temp_to_optimize
    local_optimize           :134-210             evaluate_render_range on the UNSTABLE rows of each window frame, a fresh
                                                  Adam, `gaussian_update_iter` loss_update iterations, history_merge
    global_optimization      :594-707             the stable rows alone, rescaled learning rates, colour-error tile masks
    gaussians_fix            :253-271             confidence > stable_confidence_thres -> stable (freeze_rows)
    gaussians_delete         :298-335             too big / unstable for too long -> removed (remove_rows)
    error_gaussians_remove   :510-592             per-Gaussian error accumulation, counters, delete / release
    check_keyframe           :337-369
    evaluate_render_range    :471-508
    get_render_output        :943-958

What is NOT here: dataset IO, the ORB backend, tensorboard / model snapshots (io_formats.py writes the PLYs), the
multi-process plumbing.  What differs, deliberately:
  * row order is [stable | unstable] (the reference concatenates [unstable, stable], :1069-1108); indices of a render of
    the whole map therefore address stable rows first.  Only tie-breaks between equal depths can see the order.
  * shape changes cost O(moved rows) in place (append / freeze / remove of the map object) instead of re-concatenating
    every tensor; boolean-mask compactions - each a device-to-host synchronisation in the reference - are kept to the
    ones whose result changes a shape: two per added frame, one per fix / delete that finds something.
  * bbox_filter (SLAM/utils.py:737-744) runs inside the neighbour query (rtgs_knn3_query's box), not as a compaction.
  * keyframe images stay on the device (the reference parks them on the CPU, :349-367) - the four maps the global
    optimisation reads, 33 MB per 1200x680 keyframe.
  * a render of (frame, map version) is remembered: error_gaussians_remove and get_render_output ask for the same frame
    of the same map when nothing was deleted in between.
  * a missing neighbour is "not inside": with fewer than three unstable Gaussians inside the new points' box, pytorch3d's
    knn_points ZERO-PADS distances and indices (dist 0 < 0.6 radius: the reference then drops EVERY temp point of the frame,
    mapper.py:812-826); rtgs_knn3_query reports -1 / FLT_MAX and temp_points_filter keeps the points (the shim's
    _knn_points does the same, so neither the goldens nor the fuzzer exercise the padded form).  Deliberate: the padded
    behaviour is an artefact of the library call, not of the filter's rule.
There is no CPU path: `ops` defaults to the HIP modules; tests inject torch doubles to exercise the host logic."""
from __future__ import annotations

import math
import random
from collections import deque
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np
import torch

from . import map_optim as mo

SH_C0 = 0.28209479177387814


def replica_args(**over) -> SimpleNamespace:
    """configs/base.yaml overlaid with configs/replica_base.yaml (the values Mapping / Tracker / Renderer / IcpTracker read)."""
    a = dict(
        # gaussian params (base.yaml:30-36)
        active_sh_degree=3, max_sh_degree=3, xyz_factor=[1.0, 1.0, 0.1], init_opacity=0.99, scale_factor=1.0,
        max_radius=0.05, min_radius=0.001,
        # map preprocess (:38-43)
        min_depth=0.3, max_depth=5.0, depth_filter=False, invalid_confidence_thresh=0.2, global_keyframe_num=3,
        # map params (:45-54; replica_base.yaml:9,14)
        memory_length=5, uniform_sample_num=40800, add_transmission_thres=0.5, transmission_sample_ratio=1.0,
        error_sample_ratio=0.05, add_depth_thres=0.1, add_color_thres=0.1, add_normal_thres=1000.0,
        history_merge_max_weight=0.5,
        # state manage (:56-62; replica_base.yaml:12-13)
        keyframe_trans_thes=0.3, keyframe_theta_thes=30.0, stable_confidence_thres=100.0, unstable_time_window=120,
        KNN_num=15, KNN_threshold=-1,
        # render params (:64-70)
        renderer_opaque_threshold=0.6, renderer_normal_threshold=60.0, renderer_depth_threshold=1.0, color_sigma=3.0,
        global_opt_top_ratio=0.4,
        # optimize params (:73-91; replica_base.yaml:16-24, 38-40)
        gaussian_update_iter=50, gaussian_update_frame=6, final_global_iter=20, color_weight=0.8, depth_weight=1.0,
        ssim_weight=0.2, normal_weight=0.0, position_lr=0.001, feature_lr=0.0005, opacity_lr=0.0, scaling_lr=0.004,
        rotation_lr=0.001, feature_lr_coef=4.0, scaling_lr_coef=4.0, rotation_lr_coef=4.0,
        # ICP (:93-103; replica_base.yaml:26-35)
        use_gt_pose=False, icp_use_model_depth=True, icp_downscales=[0.25, 0.5, 1.0], icp_damping=0.0001,
        icp_downscale_iters=[5, 5, 5], icp_distance_threshold=0.1, icp_normal_threshold=20.0,
        icp_sample_distance_threshold=0.01, icp_sample_normal_threshold=0.01, icp_warmup_frames=0, icp_fail_threshold=0.02,
        verbose=False, type="Replica")
    a.update(over)
    return SimpleNamespace(**a)


def tum_args(**over) -> SimpleNamespace:
    """configs/base.yaml overlaid with configs/tum_base.yaml: stable_confidence_thres 200, unstable_time_window 150,
    memory_length 5, gaussian_update_iter 50, gaussian_update_frame 4, feature_lr 0.001, scaling_lr 0.02 (tum_base.yaml:10-22);
    everything replica_base.yaml overrides falls back to base.yaml (uniform_sample_num 50000, final_global_iter 10, the three
    *_lr_coef 1.0).  (tum_base.yaml also switches the ORB backend on - out of scope here: ICP only, as BASELINE configs[3].)"""
    a = dict(uniform_sample_num=50000, memory_length=5, stable_confidence_thres=200.0, unstable_time_window=150,
             gaussian_update_iter=50, gaussian_update_frame=4, feature_lr=0.001, scaling_lr=0.02, final_global_iter=10,
             feature_lr_coef=1.0, scaling_lr_coef=1.0, rotation_lr_coef=1.0, type="TUM")
    a.update(over)
    return replica_args(**a)


def scannetpp_args(**over) -> SimpleNamespace:
    """configs/base.yaml overlaid with configs/scannetpp_base.yaml: uniform_sample_num 68620, max_depth 10, stable_confidence_thres
    400, unstable_time_window 200, gaussian_update_iter 75, gaussian_update_frame 3, ground-truth poses (scannetpp_base.yaml:3-23);
    what replica_base.yaml overrides falls back to base.yaml (memory_length 1, final_global_iter 10, the three *_lr_coef 1.0,
    icp_use_model_depth False).  On this dataset
    type every optimised frame runs the local optimisation AND - on a keyframe - the global one (mapper.py:107-113), and the loss
    leaves out the pixels without depth (:419-420)."""
    a = dict(uniform_sample_num=68620, max_depth=10.0, stable_confidence_thres=400.0, unstable_time_window=200,
             gaussian_update_iter=75, gaussian_update_frame=3, use_gt_pose=True, final_global_iter=10, feature_lr_coef=1.0,
             scaling_lr_coef=1.0, rotation_lr_coef=1.0, memory_length=1, icp_use_model_depth=False, type="Scannetpp")
    a.update(over)
    return replica_args(**a)


class Frame:
    """What the reference's Camera (scene/cameras.py) offers the mapper, tracker and renderer: the attributes
    Renderer.render reads (render.py:66-86), the pose (R = W2C[:3,:3]^T, T = W2C[:3,3]; cameras.py:96-137), intrinsics
    (:147-154) and get_uv (:161-168).  `camera_center` is NOT refreshed by updatePose - cameras.py:125-137 does not either
    (SURVEY.md Appendix A): SH view directions use the pose the frame was constructed with."""

    _serial = 0

    def __init__(self, cam, c2w, device, uid: int = 0):
        Frame._serial += 1
        self.serial, self.pose_version = Frame._serial, 0                 # identity of (frame, pose) for render caches
        self.image_height, self.image_width = int(cam.H), int(cam.W)
        self.fx, self.fy, self.cx, self.cy = float(cam.fx), float(cam.fy), float(cam.cx), float(cam.cy)
        self.FoVx = 2 * math.atan(cam.W / (2 * cam.fx))
        self.FoVy = 2 * math.atan(cam.H / (2 * cam.fy))
        self.device, self.uid = device, uid
        self._rs = None
        self._first = True
        self.updatePose(c2w)

    def updatePose(self, pose_c2w):
        """Everything the device reads of a pose - W2C^T (the view matrix), W2C, C2W, and at construction the intrinsics and
        the camera centre - travels as ONE host-to-device copy (separately: five small copies, ~20 us of host time each)."""
        c2w = np.ascontiguousarray(np.asarray(pose_c2w.numpy() if torch.is_tensor(pose_c2w) else pose_c2w, dtype=np.float64))
        self.c2w = torch.from_numpy(c2w.copy())
        self.pose_version += 1
        w2c = np.linalg.inv(c2w)
        self.R = w2c[:3, :3].T.copy()
        self.T = w2c[:3, 3].copy()
        pack = np.zeros((5, 16), dtype=np.float32)
        pack[0] = w2c.T.astype(np.float32).reshape(-1)
        pack[1] = w2c.astype(np.float32).reshape(-1)
        pack[2] = c2w.astype(np.float32).reshape(-1)
        pack[4] = pack[2]
        pack[4, 3::4] = 0                          # get_rot(c2w): the translation column cleared (SLAM/utils.py: get_rot)
        pack[4, 15] = pack[2, 15]
        if self._first:
            pack[3, :9] = np.array([[self.fx, 0, self.cx], [0, self.fy, self.cy], [0, 0, 1]], dtype=np.float32).reshape(-1)
            pack[3, 9:12] = c2w[:3, 3].astype(np.float32)
        dev = torch.from_numpy(pack).to(self.device)
        self.world_view_transform = dev[0].view(4, 4)
        self.full_proj_transform = self.world_view_transform
        self._w2c = dev[1].view(4, 4)
        self._c2w_dev = dev[2].view(4, 4)
        self._rot_dev = dev[4].view(4, 4)
        if self._first:
            self.K = dev[3, :9].view(3, 3)
            self.camera_center = dev[3, 9:12]          # NOT refreshed by later poses (see the class docstring)
            self._first = False
        self._rs = None

    @property
    def get_intrinsic(self):
        return self.K

    @property
    def get_c2w(self):
        return self._c2w_dev

    @property
    def get_rot(self):
        return self._rot_dev

    def get_w2c(self):
        return self._w2c

    def get_uv(self, xyz_w):
        xyz_c = xyz_w @ self._w2c[:3, :3].T + self._w2c[:3, 3]
        uv = xyz_c @ self.K.T
        return (uv[:, :2] / uv[:, 2:]).long()

    def raster_settings(self, args):
        if self._rs is None:
            from .rasterizer import GaussianRasterizationSettings
            deg = args.max_sh_degree if args.active_sh_degree < 0 else args.active_sh_degree
            self._rs = GaussianRasterizationSettings(
                image_height=self.image_height, image_width=self.image_width, tanfovx=math.tan(self.FoVx * 0.5),
                tanfovy=math.tan(self.FoVy * 0.5), bg=torch.zeros(3, device=self.device), scale_modifier=1.0,
                viewmatrix=self.world_view_transform, projmatrix=self.full_proj_transform, sh_degree=deg,
                campos=self.camera_center, opaque_threshold=args.renderer_opaque_threshold,
                depth_threshold=args.renderer_depth_threshold,
                normal_threshold=math.cos(math.radians(args.renderer_normal_threshold)), color_sigma=args.color_sigma,
                prefiltered=False, debug=False, cx=self.cx, cy=self.cy, T_threshold=0.0001)
        return self._rs


class HipOps:
    """The HIP kernels behind Mapping (the only implementation the product ships)."""

    def __init__(self, args, device):
        from . import slam_ops
        from .render import Renderer
        self.so = slam_ops
        self.renderer = Renderer(args)
        self.args = args
        self.gen = torch.Generator(device=device).manual_seed(int(getattr(args, "seed", 0)))
        self.keys = random.Random(int(getattr(args, "seed", 0)))           # one 64-bit key per sampling pass (rtgs_draw_new_points)
        self.kernel_draw = True
        self._stable_built, self.stable_builds = None, 0
        if __import__("os").environ.get("RTGS_ADD_KERNELS", "1") == "0":     # A/B aid: the tensor forms of the draw, the filter and the box
            self.kernel_draw, self.filter_keep, self.bbox_pad, self.compact_points = False, None, None, None

    def make_optimizer(self, packed, lr_col, capacity):
        return mo.ShardedMapOptimizer(packed, lr_col=lr_col, capacity=capacity)

    def render(self, frame, gd, tile_mask=None):
        with torch.no_grad():
            return self.renderer.render(frame, gd, tile_mask=tile_mask)

    def render_range(self, T_map, ratio):
        return self.so.render_range(T_map, ratio)

    def colorerror2tilemask(self, err, stride, ratio):
        return self.so.colorerror2tilemask(err, stride, ratio)

    def sample_pixels(self, vertex, normal, color, n, mask):
        return self.so.sample_pixels(vertex, normal, color, n, mask, self.gen)

    def knn_query(self, ref, query, self_offset=-1, ref_box=None):
        return self.so.knn_query(ref, query, self_offset, ref_box)

    def sample_new_points(self, vertex, normal, color, n, mask, identity_rot):
        """sample_pixels + the per-point part of add_empty_points (unit normal, rotation z -> normal) as ONE gather kernel
        behind the draw (rtgs_gather_new_points) instead of three gathers and ~20 elementwise launches.  -> (xyz, unit
        normal, colour, rotation), or None when nothing can be drawn."""
        so = self.so
        idx, count = so.sample_candidates(normal, mask)
        n_cand = int(count.item())                       # the reference synchronises here too (boolean-mask indexing)
        k = min(int(n), n_cand)
        if k <= 0:
            return None
        pick = idx[:n_cand][torch.randperm(n_cand, device=idx.device, generator=self.gen)[:k]].long()
        if k == 3:                                       # the reference's torch.cross quirk (compute_rot): keep the torch form
            nrm = normal.reshape(-1, 3)[pick]
            nrm = nrm / (torch.norm(nrm, p=2, dim=-1, keepdim=True) + 1e-8)
            if identity_rot:
                rot = torch.zeros(3, 4, device=nrm.device)
                rot[:, 0] = 1
            else:
                rot = compute_rot(nrm)
            return vertex.reshape(-1, 3)[pick], nrm, color.reshape(-1, 3)[pick], rot
        return so.gather_new_points(pick, vertex, normal, color, identity_rot)

    def draw_two(self, vertex, normal, color, tmask, emask, counts, n_pixels, uniform_sample_num, transmission_ratio, error_ratio,
                 identity_rot):
        """Both sampling passes of temp_points_init behind ONE host synchronisation: the candidate lists of the two masks are
        compacted before their sizes are known to the host, and the two mask counts and the two candidate counts come back in
        one copy (separately: one synchronisation for the masks' counts and one per pass for its candidates).  The draw sizes
        follow mapper.py:737-741, 772 in float32 as before; the random stream is consumed in the same order."""
        so = self.so
        idx_t, c_t = so.sample_candidates(normal, tmask)
        idx_e, c_e = so.sample_candidates(normal, emask)
        n_t, n_e, cand_t, cand_e = torch.cat([counts.reshape(-1), c_t.reshape(-1), c_e.reshape(-1)]).tolist()
        ratio = np.float32(n_t) / np.float32(n_pixels)
        n_trans = int(np.float32(transmission_ratio) * ratio * np.float32(uniform_sample_num))
        n_err = int(np.float32(n_e) * np.float32(error_ratio))
        ks = [min(int(n_trans), int(cand_t)), min(int(n_err), int(cand_e))]
        if self.kernel_draw and 3 not in ks:
            # the draw itself inside the gather kernel (rtgs_draw_new_points): no randperm, no index gathers, no concatenation
            passes = [(idx, n_cand, k, self.keys.getrandbits(64)) for idx, n_cand, k in ((idx_t, cand_t, ks[0]), (idx_e, cand_e, ks[1]))
                      if k > 0]
            return [so.draw_new_points(passes, vertex, normal, color, identity_rot)] if passes else []
        parts = []
        for n, idx, n_cand in ((n_trans, idx_t, cand_t), (n_err, idx_e, cand_e)):
            k = min(int(n), int(n_cand))
            if k <= 0:
                continue
            pick = idx[:n_cand][torch.randperm(n_cand, device=idx.device, generator=self.gen)[:k]].long()
            if k == 3:                                   # the reference's torch.cross quirk (compute_rot): keep the torch form
                nrm = normal.reshape(-1, 3)[pick]
                nrm = nrm / (torch.norm(nrm, p=2, dim=-1, keepdim=True) + 1e-8)
                if identity_rot:
                    rot = torch.zeros(3, 4, device=nrm.device)
                    rot[:, 0] = 1
                else:
                    rot = compute_rot(nrm)
                parts.append((vertex.reshape(-1, 3)[pick], nrm, color.reshape(-1, 3)[pick], rot))
            else:
                parts.append(so.gather_new_points(pick, vertex, normal, color, identity_rot))
        return parts

    def new_rows(self, *a):
        return self.so.new_rows(*a)

    def filter_keep(self, *a):
        return self.so.filter_keep(*a)

    def compact_points(self, *a):
        return self.so.compact_points(*a)

    def knn_stable(self, opt, all_xyz, nf, query, box):
        """knn_query(cat(query, all_xyz), query, 0, box) with the structure over the stable prefix all_xyz[:nf] remembered
        while opt.frozen_key stands (rtgs_knn3_build_ref / _query_built / _dynamic_merge)."""
        key = opt.frozen_key
        if self._stable_built is None or self._stable_built[0] != key:
            self._stable_built = (key, self.so.knn_build_ref(all_xyz[:nf]))
            self.stable_builds += 1
        d2s, ids = self.so.knn_query_built(self._stable_built[1], nf, query, box)
        return self.so.knn_dynamic_merge(query, all_xyz[nf:], nf, d2s, ids, box)

    def bbox_pad(self, *a):
        return self.so.bbox_pad(*a)

    def error_counters(self, *a, **kw):
        return self.so.error_counters(*a, **kw)

    def delete_mask(self, *a, **kw):
        return self.so.delete_mask(*a, **kw)

    def accumulate_gaussian_error(self, *a):
        return self.so.accumulate_gaussian_error(*a)

    def add_masks(self, *a):
        return self.so.add_masks(*a)

    def frame_errors(self, *a):
        return self.so.frame_errors(*a)

    def attach_test(self, *a):
        return self.so.attach_test(*a)

    def history_merge(self, opt, confidence, max_weight):
        opt.history_merge(confidence, max_weight)

    def step(self, opt, frame, gt_color, gt_depth, tile_mask, render_mask, confidence, w, gt_normal=None):
        """gt_normal: the step's image's world normals [H, W, 3] - read only when the normal term is weighted (mapper.py:433-443;
        0.0 in every configuration file of the reference)."""
        nw = float(getattr(w, "normal_weight", 0.0))
        return opt.step_slam(frame.raster_settings(self.args), gt_color, gt_depth, tile_mask, color_weight=w.color_weight,
                             depth_weight=w.depth_weight, ssim_weight=w.ssim_weight, add_depth_thres=w.add_depth_thres,
                             render_mask=render_mask, confidence=confidence, normal_weight=nw,
                             gt_normal=gt_normal if nw > 0 else None)


def RGB2SH(rgb):
    return (rgb - 0.5) / SH_C0


def inverse_sigmoid(x):
    return math.log(x / (1 - x))


def compute_rot(target_vec: torch.Tensor) -> torch.Tensor:
    """compute_rot((0,0,1), normal) of SLAM/utils.py:216-221 + quaternion_from_axis_angle (general_utils.py:185-191):
    the quaternion (w, x, y, z) that turns the z axis onto `target_vec`.
    One quirk is kept because the maps would differ otherwise: the reference calls `torch.cross(init_vec, target_vec)`
    WITHOUT a dim, and torch's legacy rule takes the FIRST dimension of size 3 - for a batch of exactly three points that is
    the batch dimension, not the vector one (found by oracle/fuzz_mapping_vs_reference.py; the three Gaussians then start
    turned about the z axis).  add_empty_points is called per sampling pass, so `target_vec` here is ONE pass's points."""
    z = torch.zeros_like(target_vec)
    z[:, 2] = 1.0
    axis = torch.linalg.cross(z, target_vec, dim=0 if target_vec.shape[0] == 3 else 1)
    axis = axis / (torch.norm(axis, p=2, dim=-1, keepdim=True) + 1e-8)
    angle = torch.acos(target_vec[:, 2:3])
    axis = axis / (torch.norm(axis, p=2, dim=-1, keepdim=True) + 1e-8)
    return torch.cat([torch.cos(angle / 2), axis * torch.sin(angle / 2)], dim=1)


class Mapping:
    def __init__(self, args, device, ops=None, capacity: Optional[int] = None, lr_scale: float = 1.0):
        self.args, self.device = args, device
        self.ops = ops if ops is not None else HipOps(args, device)
        self.masked_append = isinstance(self.ops, HipOps) and __import__("os").environ.get("RTGS_ADD_KERNELS", "1") != "0"
        self.stable_search = __import__("os").environ.get("RTGS_STABLE_SEARCH", "1") != "0"      # A/B aid
        self.time = 0
        self.iter = 0
        self.processed_frames = deque(maxlen=args.memory_length)
        self.processed_map = deque(maxlen=args.memory_length)
        self.keyframe_list, self.keymap_list, self.keyframe_ids, self.optimize_frames_ids = [], [], [], []
        self.lr_col = mo.default_lr_columns(args.position_lr, args.feature_lr, args.opacity_lr, args.scaling_lr,
                                            args.rotation_lr) * float(lr_scale)
        cap = int(capacity) if capacity is not None else 8 * int(args.uniform_sample_num)
        self.opt = self.ops.make_optimizer(torch.zeros(0, mo.COLS, dtype=torch.float32, device=device), self.lr_col, cap)
        for name, dtype, fill in (("confidence", torch.float32, 0.0), ("add_tick", torch.int32, 0),
                                  ("depth_error_counter", torch.int32, 0), ("color_error_counter", torch.int32, 0)):
            self.opt.add_aux(name, 1, dtype, fill)
        self.frame_map: Dict[str, torch.Tensor] = {}
        self.model_map: Dict[str, torch.Tensor] = {}
        self._render_cache = None
        self._zero_map = None
        self.xyz_factor = torch.tensor(args.xyz_factor, dtype=torch.float32, device=device)
        self.rng = random.Random(int(getattr(args, "seed", 0)))
        self.stats = dict(added=0, fixed=0, deleted_unstable=0, deleted_stable=0, released=0, local_opts=0, global_opts=0,
                          iterations=0, renders=0, renders_reused=0)
        self.prof = {} if __import__("os").environ.get("RTGS_MAP_PROFILE") else None
        self.weights = SimpleNamespace(color_weight=args.color_weight, depth_weight=args.depth_weight,
                                       ssim_weight=args.ssim_weight, add_depth_thres=args.add_depth_thres,
                                       normal_weight=float(getattr(args, "normal_weight", 0.0)))

    # ------------------------------------------------------------------ sizes and views
    @property
    def get_stable_num(self) -> int:
        return self.opt.n_frozen

    @property
    def get_unstable_num(self) -> int:
        return self.opt.n_train

    @property
    def get_pixel_num(self) -> int:
        return int(self.frame_map["depth_map"].shape[0] * self.frame_map["depth_map"].shape[1])

    @property
    def get_keyframe_num(self) -> int:
        return len(self.keyframe_list)

    @property
    def get_total_num(self) -> int:                                       # mapper.py:1128-1130
        return self.get_stable_num + self.get_unstable_num

    @property
    def get_total_iter(self) -> int:                                      # mapper.py:1116-1118
        return self.iter + self.time * self.args.gaussian_update_iter

    @property
    def get_curr_frame(self):                                             # mapper.py:1132-1134
        return self.optimize_frames_ids[-1]

    def update_poses(self, new_poses):
        """mapper.py:134-141: poses corrected by the tracker's back end (tracker.py:69-74; None without one) replace those of
        the frames the map still optimises against - the window and the keyframes.  Frame.updatePose bumps the frame's pose
        version, so renders cached for the old pose are not reused."""
        if new_poses is None:
            return
        for frame in list(self.processed_frames) + list(self.keyframe_list):
            frame.updatePose(new_poses[frame.uid])

    def aux(self, name, rows="all"):
        o = self.opt
        r0, r1 = {"all": (0, o.N), "stable": (0, o.n_frozen), "unstable": (o.n_frozen, o.N)}[rows]
        return o.aux[name][r0:r1]

    def params(self, rows="all"):
        """global_params / stable_params / unstable_params (mapper.py:984-1108) incl. `radius` and `confidence`."""
        gd = self.opt.gaussian_data(rows)
        s = gd["scales"]
        gd["radius"] = (s.sum(dim=1) - s.min(dim=1).values) / 2 if s.shape[0] else s.new_zeros(0)   # get_radius, gaussian_pointcloud.py:515-519
        gd["confidence"] = self.aux("confidence", rows)
        return gd

    @property
    def global_params(self):
        return self.params("all")

    @property
    def stable_params(self):
        return self.params("stable")

    @property
    def unstable_params(self):
        return self.params("unstable")

    def _render(self, frame, rows="all", tile_mask=None):
        key = (frame.serial, frame.pose_version, rows, self.opt.version)
        if tile_mask is None and self._render_cache is not None and self._render_cache[0] == key:
            self.stats["renders_reused"] += 1
            return self._render_cache[1]
        out = self.ops.render(frame, self.opt.gaussian_data(rows), tile_mask=tile_mask)
        self.stats["renders"] += 1
        if tile_mask is None:
            self._render_cache = (key, out)
        return out

    # ------------------------------------------------------------------ per frame (mapper.py:97-126)
    def _stage(self, name, fn, *a, **kw):
        """Diagnosis aid (RTGS_MAP_PROFILE=1): wall time per stage WITH a device synchronisation after each - the sum is
        larger than the unprofiled frame, the shares say where a frame goes."""
        if self.prof is None:
            return fn(*a, **kw)
        import time
        torch.cuda.synchronize(self.device)
        t = time.perf_counter()
        r = fn(*a, **kw)
        torch.cuda.synchronize(self.device)
        self.prof[name] = self.prof.get(name, 0.0) + time.perf_counter() - t
        return r

    def mapping(self, frame, frame_map, frame_id, optimization_params=None):
        self.frame_map = frame_map
        if "color_chw" not in frame_map:                                 # the [C,H,W] form the loss kernels read, once per frame
            frame_map["color_chw"] = frame_map["color_map"].permute(2, 0, 1).contiguous()
            frame_map["depth_chw"] = frame_map["depth_map"].permute(2, 0, 1).contiguous()
        self._stage("gaussians_add", self.gaussians_add, frame)
        self.processed_frames.append(frame)
        self.processed_map.append(frame_map)
        if (self.time + 1) % self.args.gaussian_update_frame == 0 or self.time == 0:
            self.optimize_frames_ids.append(frame_id)
            is_keyframe = self.check_keyframe(frame, frame_id)
            if self.args.type == "Scannetpp":
                self._stage("local_optimize", self.local_optimize, frame)
                if is_keyframe:
                    self._stage("global_optimization", self.global_optimization, select_keyframe_num=self.args.global_keyframe_num)
            else:
                if not is_keyframe or self.get_stable_num <= 0:
                    self._stage("local_optimize", self.local_optimize, frame)
                else:
                    self._stage("global_optimization", self.global_optimization, select_keyframe_num=self.args.global_keyframe_num)
                self._stage("gaussians_delete_stable", self.gaussians_delete, unstable=False)
        self._stage("gaussians_fix", self.gaussians_fix)
        if self.prof is None and __import__("os").environ.get("RTGS_MAP_ONE_SYNC", "1") != "0" and getattr(self.ops, "delete_mask", None) is not None and getattr(self.ops, "error_counters", None) is not None \
                and 0 < self.opt.n_train <= 65536:
            self._error_remove_and_delete()
        else:
            self._stage("error_gaussians_remove", self.error_gaussians_remove)
            self._stage("gaussians_delete", self.gaussians_delete)

    def gaussians_add(self, frame):
        temp = self._stage("add.temp_points_init", self.temp_points_init, frame)
        if temp is None:
            return
        keep = self._stage("add.temp_points_filter", self.temp_points_filter, temp)
        self._stage("add.temp_points_attach", self.temp_points_attach, frame, temp)
        self._stage("add.temp_to_optimize", self.temp_to_optimize, temp, keep)

    # ------------------------------------------------------------------ new Gaussians (mapper.py:709-883)
    def _new_points(self, parts):
        """GaussianPointCloud.add_empty_points (gaussian_pointcloud.py:305-364) for the sampled pixels of this frame: unit
        normals, SH dc from the colour, raw scale log(1e-6) until update_geometry, rotation z -> normal, init opacity.
        (sample_pixels never returns a pixel whose normal sums to zero, which is the only thing :319-322 filters.)"""
        parts = [p for p in parts if p is not None and p[0].shape[0] > 0]
        if not parts:
            return None
        a = self.args
        if len(parts[0]) == 4:                           # ops.sample_new_points: normals are unit, rotations are there
            cat = lambda k: parts[0][k] if len(parts) == 1 else torch.cat([p[k] for p in parts], 0)
            xyz = cat(0)
            return dict(xyz=xyz.contiguous(), normal=cat(1), color=cat(2), rots=cat(3),
                        opacity_raw=torch.full((xyz.shape[0], 1), inverse_sigmoid(a.init_opacity), device=xyz.device))
        same = a.xyz_factor[0] == 1 and a.xyz_factor[1] == 1 and a.xyz_factor[2] == 1
        xyz, color = (torch.cat([p[k] for p in parts], 0) for k in (0, 2))
        # normals and rotations PER sampling pass, as add_empty_points is called (mapper.py:754, 794)
        normals, rot_parts = [], []
        for p in parts:
            nrm = p[1] / (torch.norm(p[1], p=2, dim=-1, keepdim=True) + 1e-8)
            normals.append(nrm)
            if same:
                r = torch.zeros(nrm.shape[0], 4, device=nrm.device)
                r[:, 0] = 1
            else:
                r = compute_rot(nrm)
            rot_parts.append(r)
        normal, rots = torch.cat(normals, 0), torch.cat(rot_parts, 0)
        n = xyz.shape[0]
        return dict(xyz=xyz.contiguous(), normal=normal, color=color, rots=rots,
                    opacity_raw=torch.full((n, 1), inverse_sigmoid(a.init_opacity), device=xyz.device))

    def temp_points_init(self, frame):
        a, fm = self.args, self.frame_map
        fused = getattr(self.ops, "sample_new_points", None) is not None
        same = a.xyz_factor[0] == 1 and a.xyz_factor[1] == 1 and a.xyz_factor[2] == 1

        def draw(n, m):
            if fused:
                return self.ops.sample_new_points(fm["vertex_map_w"], fm["normal_map_w"], fm["color_map"], n, m, same)
            return self.ops.sample_pixels(fm["vertex_map_w"], fm["normal_map_w"], fm["color_map"], n, m)
        if self.time == 0:
            mask = fm["depth_map"] > 0
            return self._new_points([draw(a.uniform_sample_num, mask)])
        out = self._render(frame, "all")                                    # get_render_output (:727): the model at the new pose
        # transmission mask / error mask of :728-768 and their sizes in ONE pass over the frame (rtgs_add_masks)
        tmask, emask, counts = self.ops.add_masks(out["T_map"], fm["depth_map"], out["depth"], out["render"], fm["color_chw"],
                                                  out["depth_index_map"], a.add_transmission_thres, a.add_depth_thres,
                                                  a.add_color_thres)
        if fused and getattr(self.ops, "draw_two", None) is not None:
            return self._new_points(self.ops.draw_two(fm["vertex_map_w"], fm["normal_map_w"], fm["color_map"], tmask, emask, counts,
                                                      self.get_pixel_num, a.uniform_sample_num, a.transmission_sample_ratio,
                                                      a.error_sample_ratio, same))
        n_t, n_e = counts.tolist()                                          # ONE synchronisation for both counts
        # float32 arithmetic and truncation as devI(...) of mapper.py:737-741, 772
        ratio = np.float32(n_t) / np.float32(self.get_pixel_num)
        n_trans = int(np.float32(a.transmission_sample_ratio) * ratio * np.float32(a.uniform_sample_num))
        n_err = int(np.float32(n_e) * np.float32(a.error_sample_ratio))
        parts = []
        for n, m in ((n_trans, tmask), (n_err, emask)):
            if n > 0:
                parts.append(draw(n, m))
        return self._new_points(parts)

    def temp_points_filter(self, temp, topk=3):
        """Drop temp points that fall within 0.6 radius of one of their 3 nearest existing unstable Gaussians
        (mapper.py:803-827).  Returns the keep mask (the compaction happens once, in temp_to_optimize)."""
        n = temp["xyz"].shape[0]
        if self.get_unstable_num > 0 and getattr(self.ops, "filter_keep", None) is not None:
            # box, query and decision as three launches (rtgs_bbox_pad, rtgs_knn3_query, rtgs_filter_keep)
            ud = self.opt.gaussian_data("unstable")
            d2, idx = self.ops.knn_query(ud["xyz"], temp["xyz"], -1, self.ops.bbox_pad(temp["xyz"], 0.05))
            return self.ops.filter_keep(d2, idx, ud["scales"], 0.6)
        keep = torch.ones(n, dtype=torch.bool, device=temp["xyz"].device)
        if self.get_unstable_num > 0:
            up = self.params("unstable")
            lo, hi = temp["xyz"].min(dim=0)[0] - 0.05, temp["xyz"].max(dim=0)[0] + 0.05          # bbox_filter
            d2, idx = self.ops.knn_query(up["xyz"], temp["xyz"], -1, torch.cat([lo, hi]))
            found = idx >= 0
            rad = up["radius"][idx.clamp_min(0).long()] * 0.6
            keep = ~((torch.sqrt(d2) < rad) & found).any(dim=-1)
        return keep

    def temp_points_attach(self, frame, temp, unstable_opacity_low=0.1):
        """Temp points that project onto a stable Gaussian and lie on its plane start at opacity 0.1 and are tied to their
        initial state by the attach regulariser (mapper.py:830-883).  Mask arithmetic instead of index compactions."""
        if self.get_stable_num == 0:
            return
        xyz = temp["xyz"]
        out = self._render(frame, "stable")
        sp = self.opt.gaussian_data("stable")
        attach = self.ops.attach_test(xyz, frame.get_w2c(), frame.fx, frame.fy, frame.cx, frame.cy, frame.image_height,
                                      frame.image_width, out["color_index_map"], sp["xyz"], sp["normal"],
                                      0.5 * self.args.add_depth_thres)
        attach = attach.view(torch.bool) if attach.dtype == torch.uint8 else attach.bool()      # 0 / 1 bytes: no conversion pass
        temp["opacity_raw"] = temp["opacity_raw"].masked_fill(attach[:, None], inverse_sigmoid(unstable_opacity_low))

    def temp_to_optimize(self, temp, keep):
        """update_geometry (gaussian_pointcloud.py:366-405) + cat into the unstable cloud (mapper.py:886-899): the in-plane
        scale of a new Gaussian = rms of (distance - 3 radius) to its three nearest neighbours among the new points and the
        existing ones inside the new points' box; a new point INSIDE 3 radii of a neighbour is dropped."""
        a = self.args
        n_all = int(keep.shape[0])
        if getattr(self.ops, "compact_points", None) is not None and getattr(self.ops, "new_rows", None) is not None:
            # synchronisation 1 of the add - what the filter left - as one launch + the count (rtgs_compact_points)
            cx, cc, co, cr, n = self.ops.compact_points(keep, temp["xyz"], temp["color"], temp["opacity_raw"], temp["rots"])
            temp = dict(xyz=cx, color=cc, opacity_raw=co, rots=cr)
        else:
            sel = torch.nonzero(keep).reshape(-1)                    # synchronisation 1 of the add: what the filter left
            if int(sel.shape[0]) < n_all:
                fused = getattr(self.ops, "new_rows", None) is not None       # rtgs_new_rows reads no normals
                temp = {k: v[sel] for k, v in temp.items() if not (fused and k == "normal")}
        xyz = temp["xyz"].contiguous()
        n = xyz.shape[0]
        if n == 0:
            return
        if getattr(self.ops, "new_rows", None) is not None:
            # the same arithmetic as below in ONE kernel behind the neighbour query (rtgs_new_rows): ~45 launches fewer per frame
            gd = self.opt.gaussian_data("all")
            box = self.ops.bbox_pad(xyz, 0.05) if getattr(self.ops, "bbox_pad", None) is not None else \
                torch.cat([xyz.min(dim=0)[0] - 0.05, xyz.max(dim=0)[0] + 0.05])
            nf = self.get_stable_num
            if self.stable_search and nf >= 20000 and n + (self.opt.N - nf) <= 32768 and getattr(self.opt, "world", 1) == 1 \
                    and getattr(self.ops, "knn_stable", None) is not None:
                # the structure over the STABLE Gaussians is kept while they do not change (opt.frozen_key): its Morton sort of
                # ~300 000 points was two thirds of this search; the new points and the unstable Gaussians are compared directly
                d2, idx = self.ops.knn_stable(self.opt, gd["xyz"], nf, xyz, box)
            else:
                d2, idx = self.ops.knn_query(torch.cat([xyz, gd["xyz"]]), xyz, 0, box)
            rows, valid = self.ops.new_rows(xyz, temp["color"], temp["opacity_raw"], temp["rots"], d2, idx, gd["scales"],
                                            a.min_radius, a.max_radius, a.scale_factor, a.xyz_factor)
            if self.masked_append and getattr(self.opt, "append_rows_masked", None) is not None and getattr(self.opt, "world", 1) == 1:
                # synchronisation 2 of the add: the accepted rows go straight behind the map's last row, the count comes back
                m = self.opt.append_rows_masked(rows, valid, aux={"add_tick": int(self.time)})
                if m == 0:
                    return
            else:
                good = torch.nonzero(valid).reshape(-1)              # synchronisation 2 of the add
                m = int(good.shape[0])
                if m == 0:
                    return
                self.opt.append_rows(rows if m == n else rows[good], aux={"add_tick": int(self.time)})
            self.stats["added"] += m
            self.stats["last_add"] = (n_all, n, m)
            return
        gp = self.params("all")
        tiny = torch.full((n,), 1e-6, device=xyz.device)             # get_radius of the still unscaled temp points
        total_xyz = torch.cat([xyz, gp["xyz"]])
        total_radius = torch.cat([tiny, gp["radius"]])
        lo, hi = xyz.min(dim=0)[0] - 0.05, xyz.max(dim=0)[0] + 0.05
        d2, idx = self.ops.knn_query(total_xyz, xyz, 0, torch.cat([lo, hi]))
        j = idx.clamp_min(0).long()
        dist = torch.sqrt(d2) - 3 * total_radius[j]                  # [n,3]; a missing neighbour (idx -1) is infinitely far
        dist = torch.where(idx >= 0, dist, torch.full_like(dist, 1e30))
        invalid = (dist < 0).any(dim=-1)
        scales = torch.sqrt((dist ** 2).sum(dim=-1) / 3).clamp(a.min_radius, a.max_radius)
        log_scales = torch.log(a.scale_factor * scales[:, None] * self.xyz_factor[None, :])
        good = torch.nonzero(~invalid).reshape(-1)                   # synchronisation 2 of the add
        m = int(good.shape[0])
        if m == 0:
            return
        packed = torch.zeros(m, mo.COLS, dtype=torch.float32, device=xyz.device)
        packed[:, 0:3] = xyz[good]
        packed[:, 3:6] = RGB2SH(temp["color"][good])
        packed[:, 51:52] = temp["opacity_raw"][good]
        packed[:, 52:55] = log_scales[good]
        packed[:, 55:59] = temp["rots"][good]
        self.opt.append_rows(packed, aux={"add_tick": int(self.time)})
        self.stats["added"] += m
        self.stats["last_add"] = (n_all, n, m)

    # ------------------------------------------------------------------ optimisation
    def evaluate_render_range(self, frame, global_opt=False, sample_ratio=-1, unstable=True, gt_color=None):
        """mapper.py:471-508.  Local: render the UNSTABLE rows, render_mask = T != 1, tile_mask = tiles more than half
        covered.  Global: render the STABLE rows; with a sample ratio the tiles with the largest colour error."""
        out = self._render(frame, "unstable" if unstable else "stable")
        T = out["T_map"]
        if global_opt and sample_ratio > 0:
            render_image = out["render"].permute(1, 2, 0)
            color_error = (render_image - gt_color).abs().sum(dim=-1)
            color_error = torch.where(render_image.sum(dim=-1) == 0, torch.zeros_like(color_error), color_error)
            tile_mask = self.ops.colorerror2tilemask(color_error, 16, sample_ratio)
            H, W = color_error.shape
            render_mask = tile_mask.bool().repeat_interleave(16, 0).repeat_interleave(16, 1)[:H, :W]
            return render_mask.contiguous(), tile_mask, None
        render_mask, tile_mask, count = self.ops.render_range(T, 0.5)
        if global_opt:
            tile_mask = None
        return render_mask, tile_mask, count

    def _loss_mask(self, render_mask, frame_map):
        """mapper.py:419-420: on Scannetpp the loss also leaves out the pixels without depth - of the image the step is GIVEN
        (in a keyframe-triggered global optimisation that is the randomly drawn keyframe even where the masks are the last
        one's, :672-678)."""
        if self.args.type != "Scannetpp" or render_mask is None:
            return render_mask
        return (render_mask.bool() & (frame_map["depth_chw"][0] > 0)).to(torch.uint8)

    def local_optimize(self, frame, update_args=None):
        a, o = self.args, self.opt
        conf = self.aux("confidence", "unstable").reshape(-1)
        o.begin_local_optimization(confidence=conf)                      # history_stat + a fresh Adam (mapper.py:136-156)
        masks = [self._stage("opt.evaluate_render_range", self.evaluate_render_range, f) for f in self.processed_frames]
        masks = [(m[0].to(torch.uint8), m[1], m[2]) for m in masks]     # the loss kernels read bytes: converted once, not per step
        self.stats["local_opts"] += 1
        if o.n_train == 0:
            return
        n_win = len(self.processed_frames)
        for it in range(a.gaussian_update_iter):
            self.iter = it
            j = self.rng.randint(0, n_win - 1)
            if it > a.gaussian_update_iter / 2:
                j = -1
            fm = self.processed_map[j]
            self.ops.step(o, self.processed_frames[j], fm["color_chw"], fm["depth_chw"], masks[j][1],
                          self._loss_mask(masks[j][0], fm), conf, self.weights, gt_normal=fm.get("normal_map_w"))
        self.stats["iterations"] += a.gaussian_update_iter
        self.iter = 0
        self.ops.history_merge(o, conf, a.history_merge_max_weight)

    def global_optimization(self, update_args=None, select_keyframe_num=-1, is_end=False):
        """mapper.py:594-707.  Keyframe-triggered form (select_keyframe_num = global_keyframe_num): the last keyframes,
        position lr 0 and everything else x 0.1, tiles with the largest 40 % colour error.  Final form (-1): everything
        becomes stable first, all keyframes x final_global_iter iterations, no depth term, feature / scaling / rotation
        rates x their coefficients, masks from the stable rows' transmission."""
        a, o = self.args, self.opt
        final = select_keyframe_num == -1
        if final:
            self.gaussians_fix(mask=self.aux("confidence", "unstable").reshape(-1) > -1)
        if self.get_stable_num == 0:
            return
        scale = mo.global_lr_scale(final, a.feature_lr_coef, a.scaling_lr_coef, a.rotation_lr_coef)
        o.begin_global_optimization(scale)
        total_iter = int(a.gaussian_update_iter)
        sample_ratio = 0.4
        w = SimpleNamespace(**vars(self.weights))
        if final:
            total_iter = self.get_keyframe_num * a.final_global_iter
            select_keyframe_num = self.get_keyframe_num
            w.depth_weight = 0.0
            self.weights.depth_weight = 0.0                              # the reference overwrites update_args (:633)
            sample_ratio = -1
        select_keyframe_num = min(select_keyframe_num, self.get_keyframe_num)
        sel = [-i - 1 for i in range(select_keyframe_num)]
        frames = [self.keyframe_list[i] for i in sel]
        maps = [self.keymap_list[i] for i in sel]
        masks = [self.evaluate_render_range(f, global_opt=True, unstable=False, sample_ratio=sample_ratio,
                                            gt_color=m["color_map"]) for f, m in zip(frames, maps)]
        masks = [(m[0].to(torch.uint8), m[1], m[2]) for m in masks]
        conf = self.aux("confidence", "stable").reshape(-1)
        try:
            for it in range(total_iter):
                self.iter = it
                j = self.rng.randint(0, select_keyframe_num - 1)
                fr, im = frames[j], maps[j]
                if it > total_iter / 2 and not final:
                    j = -1           # as the reference (:675-678): the FRAME stays the random one, the MASKS become the last entry's
                self.ops.step(o, fr, im["color_chw"], im["depth_chw"], masks[j][1], self._loss_mask(masks[j][0], im), conf, w,
                              gt_normal=im.get("normal_map_w"))
            self.stats["iterations"] += total_iter
            self.stats["global_opts"] += 1
        finally:
            # an exception inside a step must not leave the map object in the global scope (every later append / permute would
            # refuse: ADVICE r5)
            self.iter = 0
            o.end_global_optimization()

    # ------------------------------------------------------------------ state management
    def gaussians_fix(self, mask=None):
        """mapper.py:253-271: unstable Gaussians whose confidence passed the threshold join the stable cloud."""
        o = self.opt
        if o.n_train == 0:
            return
        if mask is None:
            # Confidence only ever RISES inside an optimisation step (appended rows start at 0, a release resets to 0): if no
            # step has run since this method last looked, nothing can have crossed the threshold - five frames in six.  Saves the
            # compare, the sum and their host synchronisation; the outcome is the same by construction.
            steps = getattr(o, "total_steps", None)
            if steps is not None and steps == getattr(self, "_fix_seen_steps", None):
                return
            self._fix_seen_steps = steps
        conf_u = self.aux("confidence", "unstable").reshape(-1)
        stable_mask = (conf_u > self.args.stable_confidence_thres) if mask is None else mask.reshape(-1)
        k = int(stable_mask.sum())                                       # the reference synchronises here too (:267)
        if k > 0:
            full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
            full[o.n_frozen:] = stable_mask
            nf0 = o.n_frozen
            o.freeze_rows(full)
            c = o.aux["confidence"]
            c[nf0:nf0 + k] = torch.clip(c[nf0:nf0 + k], max=self.args.stable_confidence_thres)
            self.stats["fixed"] += k

    def gaussians_release(self, mask, count=None):
        """mapper.py:286-295: the stable Gaussians under `mask` (one entry per stable row) start over - confidence 0, tick =
        now.  The reference removes them from the stable cloud and re-appends them to its end; here the rows stay where they
        are (row order carries no meaning; tests compare the two in a canonical order)."""
        o = self.opt
        nf = o.n_frozen
        mask = mask.reshape(-1).bool()
        n = int(mask.sum()) if count is None else int(count)
        if n > 0:
            o.aux["confidence"][:nf, 0].masked_fill_(mask, 0)
            o.aux["add_tick"][:nf, 0].masked_fill_(mask, int(self.time))
            self.stats["released"] += n

    def gaussians_delete(self, unstable=True):
        """mapper.py:298-335: too big (radius > 10 x mean), or - unstable only - older than the time window."""
        o = self.opt
        rows = "unstable" if unstable else "stable"
        r0, r1 = (o.n_frozen, o.N) if unstable else (0, o.n_frozen)
        if r1 == r0:
            return
        if unstable and r1 - r0 <= 65536 and getattr(self.ops, "delete_mask", None) is not None:
            # the unstable cloud (a few thousand rows): mean radius, mask and count in one single-workgroup kernel
            # (rtgs_delete_mask) instead of eleven tensor operations
            sc = o.gaussian_data("unstable")["scales"]
            dm, k = self.ops.delete_mask(sc, self.aux("add_tick", rows).reshape(-1), int(self.time), int(self.args.unstable_time_window))
            if k > 0:
                full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
                full[r0:r1] = dm.bool()
                o.remove_rows(full, start=r0)
                self.stats["deleted_unstable"] += k
            return
        radius = self.params(rows)["radius"]
        delete_mask = radius > radius.mean() * 10
        if unstable:
            delete_mask = delete_mask | ((self.time - self.aux("add_tick", rows).reshape(-1)) > self.args.unstable_time_window)
        k = int(delete_mask.sum())
        if k > 0:
            full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
            full[r0:r1] = delete_mask
            o.remove_rows(full)
            self.stats["deleted_unstable" if unstable else "deleted_stable"] += k

    def check_keyframe(self, frame, frame_id):
        def keep():
            # what the global optimisation reads of a keyframe (the reference keeps colour / depth / normal too, on the CPU:
            # mapper.py:340-367) - not the whole frame map (vertex maps, camera-space normals, confidence: 68 MB per 1200x680
            # keyframe for the life of the map)
            self.keyframe_list.append(frame)
            self.keymap_list.append({k: self.frame_map[k] for k in ("color_map", "color_chw", "depth_chw", "normal_map_w", "time")
                                     if k in self.frame_map})
            self.keyframe_ids.append(frame_id)
        if self.time == 0:
            keep()
            return False
        prev = self.keyframe_list[-1]
        rot_diff = prev.R @ frame.R.T                                    # rot_compare(prev.R.T, curr.R.T) = prev_rot.T @ curr_rot, SLAM/utils.py:42-47
        theta = np.rad2deg(np.arccos(np.clip((np.trace(rot_diff) - 1) / 2, -1.0, 1.0)))
        l2 = np.linalg.norm(prev.T - frame.T, ord=2)
        if theta > self.args.keyframe_theta_thes or l2 > self.args.keyframe_trans_thes:
            keep()
            return True
        return False

    def error_gaussians_remove(self):
        """mapper.py:510-592: back-project the errors of the newest frame onto the Gaussians that own its pixels; stable
        Gaussians that were wrong in depth 10 times are deleted, wrong in colour 10 times "released" (confidence 0, new tick;
        the reference re-appends them to the STABLE cloud - :286-295 - so they stay stable)."""
        if self.get_stable_num <= 0:
            return
        o, a = self.opt, self.args
        frame, cm = self.processed_frames[-1], self.processed_map[-1]
        out = self._render(frame, "all")
        color_error, depth_error = self.ops.frame_errors(cm["depth_map"], out["depth"], out["render"], cm["color_chw"],
                                                         out["depth_index_map"])
        H, W = cm["color_map"].shape[:2]
        if self._zero_map is None or self._zero_map.shape != depth_error.shape:
            self._zero_map = torch.zeros_like(depth_error)              # normal_error: zeros (:531), allocated once
        g_color, g_depth, _, _ = self.ops.accumulate_gaussian_error(
            H, W, o.N, color_error, depth_error, self._zero_map, out["color_index_map"], out["depth_index_map"],
            a.add_color_thres, a.add_depth_thres, a.add_normal_thres, True)
        nf = o.n_frozen                                                  # stable rows come FIRST here ([unstable, stable] there)
        dcnt, ccnt = o.aux["depth_error_counter"], o.aux["color_error_counter"]
        if getattr(self.ops, "error_counters", None) is not None:
            # strikes, decisions and both counts in one kernel and one synchronisation (rtgs_error_counters)
            ddel, crel, (n_del, n_rel) = self.ops.error_counters(g_color, g_depth, nf, 2 * a.add_color_thres, 2 * a.add_depth_thres,
                                                                 dcnt, ccnt, 10)
            if n_rel > 0:
                self.gaussians_release(crel, count=n_rel)
            if n_del > 0:
                full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
                full[:nf] = ddel.bool()
                o.remove_rows(full)
                self.stats["deleted_stable"] += n_del
            return
        dcnt[:nf, 0] += (g_depth[:nf] > 2 * a.add_depth_thres).to(dcnt.dtype)
        ccnt[:nf, 0] += (g_color[:nf] > 2 * a.add_color_thres).to(ccnt.dtype)
        delete_thresh = 10
        ddel = dcnt[:nf, 0] >= delete_thresh
        crel = (ccnt[:nf, 0] >= delete_thresh) & ~ddel
        n_del, n_rel = torch.stack([ddel.sum(), crel.sum()]).tolist()
        if n_rel > 0:
            self.gaussians_release(crel, count=n_rel)
        if n_del > 0:
            full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
            full[:nf] = ddel
            o.remove_rows(full)
            self.stats["deleted_stable"] += n_del

    def _error_remove_and_delete(self):
        """error_gaussians_remove followed by gaussians_delete (mapper.py:123-125) behind ONE host synchronisation: which unstable
        Gaussians the delete removes (too big, too old) does not depend on what the error pass does to the STABLE cloud (strikes,
        releases, deletions of stable rows), so its mask is computed first and the three counts come back together."""
        o, a = self.opt, self.args
        dm, kd = self.ops.delete_mask(o.gaussian_data("unstable")["scales"], self.aux("add_tick", "unstable").reshape(-1),
                                      int(self.time), int(a.unstable_time_window), sync=False)
        n_del = n_rel = 0
        ddel = crel = None
        if self.get_stable_num > 0:
            frame, cm = self.processed_frames[-1], self.processed_map[-1]
            out = self._render(frame, "all")
            color_error, depth_error = self.ops.frame_errors(cm["depth_map"], out["depth"], out["render"], cm["color_chw"],
                                                             out["depth_index_map"])
            H, W = cm["color_map"].shape[:2]
            if self._zero_map is None or self._zero_map.shape != depth_error.shape:
                self._zero_map = torch.zeros_like(depth_error)
            g_color, g_depth, _, _ = self.ops.accumulate_gaussian_error(
                H, W, o.N, color_error, depth_error, self._zero_map, out["color_index_map"], out["depth_index_map"],
                a.add_color_thres, a.add_depth_thres, a.add_normal_thres, True)
            ddel, crel, counts = self.ops.error_counters(g_color, g_depth, o.n_frozen, 2 * a.add_color_thres, 2 * a.add_depth_thres,
                                                         o.aux["depth_error_counter"], o.aux["color_error_counter"], 10, sync=False)
            n_del, n_rel, k = torch.cat([counts.reshape(-1), kd.reshape(-1)]).tolist()
        else:
            k = int(kd.item())
        if n_rel > 0:
            self.gaussians_release(crel, count=n_rel)
        if n_del > 0:
            full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
            full[:o.n_frozen] = ddel.bool()
            o.remove_rows(full)
            self.stats["deleted_stable"] += n_del
        if k > 0:
            full = torch.zeros(o.N, dtype=torch.bool, device=self.device)
            full[o.n_frozen:] = dm.bool()
            o.remove_rows(full, start=o.n_frozen)            # only trainable rows go: the stable prefix is not touched
            self.stats["deleted_unstable"] += k

    def save_model(self, path: str, save_data: bool = True, save_sibr: bool = True, save_merge: bool = True):
        """Mapping.save_model (mapper.py:916-941): `<path>.ply` = the unstable cloud, `<path>_stable.ply` = the stable one (raw
        values + confidence), `_sibr` variants without the confidence column, `_merge` = both.  Bytes as the reference's
        writer lays them out (io_formats.py)."""
        from . import io_formats as iof
        o = self.opt
        P = o.params
        conf = o.aux["confidence"][:o.N]
        parts = {"": (o.n_frozen, o.N), "_stable": (0, o.n_frozen)}
        for with_conf, tag in ((True, ""), (False, "_sibr")):
            if not (save_data if with_conf else save_sibr):
                continue
            for name, (r0, r1) in parts.items():
                m = iof.packed_to_model(P[r0:r1].detach().cpu().numpy())
                iof.save_model_ply(path + name + tag + ".ply", m["xyz"], m["features_dc"], m["features_rest"], m["opacity"],
                                   m["scaling"], m["rotation"], conf[r0:r1] if with_conf else None, include_confidence=with_conf)
            if save_merge and o.n_train > 0 and o.n_frozen > 0:
                iof.merge_ply(path + tag + ".ply", path + "_stable" + tag + ".ply", path + "_merge" + tag + ".ply",
                              include_confidence=with_conf)

    def get_render_output(self, frame):
        out = self._render(frame, "all")
        self.model_map = {
            "render_color": out["render"].permute(1, 2, 0), "render_depth": out["depth"].permute(1, 2, 0),
            "render_normal": out["normal"].permute(1, 2, 0), "render_color_index": out["color_index_map"].permute(1, 2, 0),
            "render_depth_index": out["depth_index_map"].permute(1, 2, 0), "render_transmission": out["T_map"].permute(1, 2, 0)}
        return self.model_map
