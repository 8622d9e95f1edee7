"""TEST DOUBLES for rtg_slam_amd.mapping.Mapping: torch / oracle implementations of the `ops` facade (the product's only
implementation is HipOps), so that the lifecycle's HOST LOGIC - indices, masks, row moves, counters, schedules - runs on
the CPU in the `-m "not gpu"` suite.  Not a numerics reference: the renderer is oracle/raster_oracle.py, the step is
ShardedMapOptimizer.step with the torch loss of tests/torch_doubles.py plus the attach term and the confidence rule."""
from __future__ import annotations

import torch

from oracle import raster_oracle as ro
from oracle import slam_ops_oracle as so
from rtg_slam_amd import map_optim as mo
from tests import torch_doubles as td
from tests.dist_util import adam_reference


class TorchOps:
    def __init__(self, args, seed=0):
        self.args = args
        self.gen = torch.Generator().manual_seed(seed)
        self.steps = []                       # (rows rendered, first trainable row, tiles on, mask pixels) per step

    def make_optimizer(self, packed, lr_col, capacity):
        return mo.ShardedMapOptimizer(packed, lr_col=lr_col, capacity=capacity, adam_fn=adam_reference, activate_fn=td.activate8)

    def _settings(self, frame):
        a = self.args
        return ro.make_settings(frame.image_height, frame.image_width, frame.fx, frame.fy, frame.cx, frame.cy,
                                viewmatrix=frame.world_view_transform, campos=frame.camera_center, sh_degree=a.max_sh_degree,
                                opaque_threshold=a.renderer_opaque_threshold, depth_threshold=a.renderer_depth_threshold,
                                normal_threshold_deg=a.renderer_normal_threshold, color_sigma=a.color_sigma)

    def _raster(self, frame, gd, tile_mask):
        return ro.rasterize(self._settings(frame), gd["xyz"], gd["opacity"], gd["shs"], gd["scales"], gd["rotations"],
                            gd["normal"], tile_mask)

    def render(self, frame, gd, tile_mask=None):
        with torch.no_grad():
            res = self._raster(frame, gd, tile_mask)
            didx = res[3][0].long()
            nm = torch.zeros(3, *didx.shape)
            if gd["normal"].shape[0]:
                nm = torch.where((didx >= 0)[None], gd["normal"][didx.clamp_min(0)].permute(2, 0, 1), nm)
        return {"render": res[0], "depth": res[1], "normal": nm, "color_index_map": res[2], "depth_index_map": res[3],
                "color_hit_weight": res[4], "depth_hit_weight": res[5], "T_map": res[6]}

    def render_range(self, T_map, ratio):
        mask = T_map[0] != 1
        return mask, so.transmission2tilemask(mask, 16, ratio), mask.sum().reshape(1)

    def colorerror2tilemask(self, err, stride, ratio):
        return so.colorerror2tilemask(err, stride, ratio)

    def sample_pixels(self, vertex, normal, color, n, mask):
        sel = so.sample_pixels_mask(normal, None if mask is None else mask.reshape(normal.shape[:2]))
        idx = torch.nonzero(sel.reshape(-1)).reshape(-1)
        n = min(int(n), int(idx.numel()))
        pick = idx[torch.randperm(idx.numel(), generator=self.gen)[:n]]
        return vertex.reshape(-1, 3)[pick], normal.reshape(-1, 3)[pick], color.reshape(-1, 3)[pick]

    def knn_query(self, ref, query, self_offset=-1, ref_box=None):
        return so.knn_query(ref, query, self_offset, ref_box)

    def accumulate_gaussian_error(self, *a):
        return so.accumulate_gaussian_error(*a)

    def add_masks(self, *a):
        tm, em, c = so.add_masks(*a)
        return tm.to(torch.uint8), em.to(torch.uint8), c

    def frame_errors(self, *a):
        return so.frame_errors(*a)

    def attach_test(self, *a):
        return so.attach_test(*a)

    def history_merge(self, opt, confidence, max_weight):
        N, t0 = opt._active()
        ai, h, st = opt.attach_init, opt._history, opt.state
        if N == t0:
            return
        x, sh, r8 = so.history_merge(st["xyz"]["p"][t0:N], st["shs"]["p"][t0:N], st["raw8"]["p"][t0:N], ai["xyz"], h["shs"],
                                     ai["raw8"], h["conf"].reshape(-1, 1), confidence.reshape(-1, 1), max_weight)
        st["xyz"]["p"][t0:N], st["shs"]["p"][t0:N], st["raw8"]["p"][t0:N] = x, sh, r8
        opt.version += 1

    def step(self, opt, frame, gt_color, gt_depth, tile_mask, render_mask, confidence, w, gt_normal=None):
        N, t0 = opt._active()
        ai = opt.attach_init
        grads = {}
        self.steps.append((N, t0, None if tile_mask is None else int(tile_mask.sum()),
                           None if render_mask is None else int(render_mask.sum())))

        def loss_fn(gd):
            gd["shs"].register_hook(lambda g: grads.__setitem__("shs", g))
            out = self._raster(frame, gd, tile_mask)
            loss = td.slam_losses(out, gt_color, gt_depth, w.color_weight, w.depth_weight, w.ssim_weight, w.add_depth_thres,
                                  None if render_mask is None else render_mask.bool())
            nw = float(getattr(w, "normal_weight", 0.0))
            if nw > 0 and gt_normal is not None:                          # mapper.py:433-443, on the normals the wrapper gathers
                didx = out[3][0].long()
                rn = torch.where((didx >= 0)[None], gd["normal"][didx.clamp_min(0)].permute(2, 0, 1), torch.zeros(3, *didx.shape))
                gn = gt_normal.permute(2, 0, 1)
                m = torch.ones_like(didx, dtype=torch.bool) if render_mask is None else render_mask.bool()
                vn = m & (didx != -1) & ~((gn == 0).all(dim=0))
                loss = loss + nw * (1 - torch.nn.functional.cosine_similarity(rn, gn, dim=0))[vn].mean()
            if ai is not None and N > t0:                                  # attach regulariser, mapper.py:384-401
                sel = torch.sigmoid(ai["raw8"][:, 0]) < 0.9
                if bool(sel.any()):                       # raw scaling, position AND raw rotation (mapper.py:389-400)
                    raw8 = gd["raw8"][t0:N]
                    l2 = lambda x, y: ((x - y) ** 2).mean()
                    loss = loss + 1000 * (l2(raw8[sel, 1:4], ai["raw8"][sel, 1:4]) + l2(gd["xyz"][t0:N][sel], ai["xyz"][sel]) +
                                          l2(raw8[sel, 4:8], ai["raw8"][sel, 4:8]))
            return loss
        loss = opt.step(loss_fn)
        g = grads.get("shs")
        if g is not None and confidence is not None:
            hit = (g[t0:N, 0, :].abs() != 0).any(dim=-1)                   # mapper.py:455-456
            confidence += hit.to(confidence.dtype)
        return loss
