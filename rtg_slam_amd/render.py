"""`Renderer` with the interface of the reference's SLAM/render.py:21-145 (same constructor argument, same
`render(viewpoint_camera, gaussian_data, tile_mask=None)` and the same result dict), on the HIP rasterizer.

The reference's own SLAM/render.py runs unmodified on this repository's `diff_gaussian_rasterization_depth` package
(tests/test_reference_wrapper.py); this class is the same wrapper without its two device-to-host synchronisations per
call: the normal map `render_normal[:, idx > -1] = normal[idx[idx > -1]].T` (render.py:130-133) is two boolean-mask
indexings there and ONE gather kernel here (`rtgs_gather_rows3`, differentiable with respect to `normal`), and the
default all-ones tile mask is cached per image size instead of rebuilt on the host every call (render.py:101-108).
There is no CPU path."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _lib
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


class _GatherRows3(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rows, index):
        if not rows.is_cuda:
            raise RuntimeError("rtg_slam_amd.render: tensors must live on a HIP device; this build has no CPU path.")
        lib, dev = _lib.load(), rows.device
        rows_c = rows.detach().float().contiguous()
        idx = index.reshape(-1).to(torch.int32).contiguous()
        n = idx.numel()
        out = torch.empty((3,) + tuple(index.shape[-2:]), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.rtgs_gather_rows3(C.c_void_p(rows_c.data_ptr() if rows_c.numel() else 0), C.c_void_p(idx.data_ptr()), n,
                                       C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        _lib.check(rc, "rtgs_gather_rows3")
        ctx.save_for_backward(idx)
        ctx.rows_shape = tuple(rows.shape)
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        lib, dev = _lib.load(), g.device
        grad = torch.zeros(ctx.rows_shape, dtype=torch.float32, device=dev)
        gc = g.float().contiguous()
        if grad.numel():
            with torch.cuda.device(dev):
                rc = lib.rtgs_scatter_rows3(C.c_void_p(gc.data_ptr()), C.c_void_p(idx.data_ptr()), idx.numel(),
                                            C.c_void_p(grad.data_ptr()), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
            _lib.check(rc, "rtgs_scatter_rows3")
        return grad, None


def gather_normal_map(normal: torch.Tensor, depth_index_map: torch.Tensor) -> torch.Tensor:
    """[3,H,W]: the world normal of the Gaussian that owns each pixel's depth, 0 where none (render.py:130-133)."""
    return _GatherRows3.apply(normal, depth_index_map)


class Renderer:
    def __init__(self, args):
        # attributes read: render.py:33-49
        self.raster_settings = None
        self.rasterizer = None
        self.bg_color = None
        self.renderer_opaque_threshold = args.renderer_opaque_threshold
        self.renderer_normal_threshold = math.cos(math.radians(args.renderer_normal_threshold))
        self.scaling_modifier = 1.0
        self.renderer_depth_threshold = args.renderer_depth_threshold
        self.max_sh_degree = args.max_sh_degree
        self.color_sigma = args.color_sigma
        self.active_sh_degree = self.max_sh_degree if args.active_sh_degree < 0 else args.active_sh_degree
        self._ones = {}
        # activations the reference hangs on the renderer (render.py:22-31, 51-58)
        self.scaling_activation = torch.exp
        self.scaling_inverse_activation = torch.log
        self.opacity_activation = torch.sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    def get_scaling(self, scaling):
        return self.scaling_activation(scaling)

    def get_rotation(self, rotaion):
        return self.rotation_activation(rotaion)

    def _default_mask(self, H, W, dev):
        key = (H, W, str(dev))
        if key not in self._ones:
            self._ones[key] = torch.ones((H + 15) // 16, (W + 15) // 16, dtype=torch.int32, device=dev)
        return self._ones[key]

    def render(self, viewpoint_camera, gaussian_data, tile_mask=None):
        means3D = gaussian_data["xyz"]
        dev = means3D.device
        if self.bg_color is None or self.bg_color.device != dev:
            self.bg_color = torch.zeros(3, dtype=torch.float32, device=dev)
        H, W = int(viewpoint_camera.image_height), int(viewpoint_camera.image_width)
        self.raster_settings = GaussianRasterizationSettings(
            image_height=H, image_width=W,
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=self.bg_color, scale_modifier=self.scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=self.active_sh_degree, campos=viewpoint_camera.camera_center,
            opaque_threshold=self.renderer_opaque_threshold, depth_threshold=self.renderer_depth_threshold,
            normal_threshold=self.renderer_normal_threshold, color_sigma=self.color_sigma, prefiltered=False, debug=False,
            cx=viewpoint_camera.cx, cy=viewpoint_camera.cy, T_threshold=0.0001)
        self.rasterizer = GaussianRasterizer(raster_settings=self.raster_settings)
        normal = gaussian_data["normal"]
        if tile_mask is None:
            tile_mask = self._default_mask(H, W, dev)
        res = self.rasterizer(means3D=means3D, opacities=gaussian_data["opacity"], shs=gaussian_data["shs"],
                              colors_precomp=None, scales=gaussian_data["scales"], rotations=gaussian_data["rotations"],
                              cov3D_precomp=None, normal_w=normal, tile_mask=tile_mask)
        return {
            "render": res[0],
            "depth": res[1],
            "normal": gather_normal_map(normal, res[3]),
            "color_index_map": res[2],
            "depth_index_map": res[3],
            "color_hit_weight": res[4],
            "depth_hit_weight": res[5],
            "T_map": res[6],
        }
