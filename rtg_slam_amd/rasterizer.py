"""Host side of the rasterizer: the Python surface of `diff_gaussian_rasterization_depth`
exactly as /root/reference/SLAM/render.py:8-13, 68-128 uses it, over the C ABI of
include/rtgs_raster.h.  torch supplies device memory, the current HIP stream and autograd
plumbing only; all arithmetic runs in the HIP kernels of rtg_slam_amd/csrc."""
from __future__ import annotations

import ctypes as C
import threading
from typing import NamedTuple, Optional

import torch
import torch.nn as nn

from . import _lib


class GaussianRasterizationSettings(NamedTuple):
    """The 19 keyword fields of SLAM/render.py:68-88 (same names, same order)."""
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


class RasterContext:
    """A library context (include/rtgs_raster.h: rtgs_ctx): near-slice mode / budget, last-call statistics, stage
    timings, work counters.  `RasterContext()` wraps the process-wide DEFAULT context (what the plain C entry points
    use); `RasterContext.create()` makes a private one.  One context serves one rendering thread (the autograd
    backward of a forward runs on the forward's context)."""

    def __init__(self, handle=None, owned=False):
        self.handle = handle            # None = the library's default context
        self._owned = owned

    @classmethod
    def create(cls) -> "RasterContext":
        h = _lib.load().rtgs_ctx_create()
        if not h:
            raise RuntimeError("rtgs_ctx_create failed")
        return cls(h, owned=True)

    def __del__(self):
        try:
            if self._owned and self.handle:
                _lib.load().rtgs_ctx_destroy(C.c_void_p(self.handle))
                self.handle = None
        except Exception:
            pass

    @property
    def ptr(self):
        return C.c_void_p(self.handle)

    def set_near_slice(self, mode: int, budget_per_tile: int = 0):
        _lib.load().rtgs_raster_set_near_slice_ctx(self.ptr, int(mode), int(budget_per_tile))

    def force_sort_path(self, enable: bool):
        _lib.load().rtgs_raster_force_sort_path_ctx(self.ptr, int(bool(enable)))

    def set_bwd_walk(self, mode: int):
        """0 / 3 = entry-per-lane walk (default), 1 = strip walk, 2 = row-granular walk, 4 = per-tile choice of 1 / 2 (round 3)."""
        _lib.load().rtgs_raster_set_bwd_walk_ctx(self.ptr, int(mode))

    def set_onepass(self, enable: bool):
        """One-pass binning into per-tile segments (default on); off = count + scan + scatter.  Bit-identical outputs."""
        _lib.load().rtgs_raster_set_onepass_ctx(self.ptr, int(bool(enable)))

    def set_speculation(self, enable: bool):
        """Speculative sizing of the forward inside the one-call map step (RTGS_FWD_SPECULATE); default on."""
        _lib.load().rtgs_raster_set_speculation_ctx(self.ptr, int(bool(enable)))

    def speculation_stats(self):
        out = (C.c_int64 * 3)()
        _lib.check(_lib.load().rtgs_raster_speculation_stats_ctx(self.ptr, out), "rtgs_raster_speculation_stats")
        return dict(speculative=int(out[0]), failed=int(out[1]), not_eligible=int(out[2]))

    def set_plain_onepass(self, enable: bool):
        """Forwards without a backward: one-pass placement with the check inside the call (include/rtgs_raster.h)."""
        _lib.load().rtgs_raster_set_plain_onepass_ctx(self.ptr, int(bool(enable)))

    def plain_stats(self):
        out = (C.c_int64 * 2)()
        _lib.check(_lib.load().rtgs_raster_plain_stats_ctx(self.ptr, out), "rtgs_raster_plain_stats")
        return dict(onepass=int(out[0]), redone=int(out[1]))

    def set_profiling(self, enable: bool):
        _lib.load().rtgs_raster_set_profiling_ctx(self.ptr, int(bool(enable)))

    def set_counters(self, counters: Optional[torch.Tensor]):
        _lib.load().rtgs_raster_set_counters_ctx(self.ptr, C.c_void_p(counters.data_ptr() if counters is not None else 0))

    def last_stats(self):
        out = (C.c_int64 * 8)()
        _lib.check(_lib.load().rtgs_raster_last_stats_ctx(self.ptr, out), "rtgs_raster_last_stats")
        return [int(v) for v in out]

    def last_slice_stats(self):
        out = (C.c_int64 * 4)()
        _lib.check(_lib.load().rtgs_raster_last_slice_stats_ctx(self.ptr, out), "rtgs_raster_last_slice_stats")
        return dict(used=int(out[0]), instances=int(out[1]), tiles_finished=int(out[2]), tiles_left_to_pass2=int(out[3]))

    def last_timings(self):
        out = (C.c_float * 12)()
        _lib.check(_lib.load().rtgs_raster_last_timings_ctx(self.ptr, out), "rtgs_raster_last_timings")
        return [float(v) for v in out]


def image_buffer_views(img: torch.Tensor, H: int, W: int):
    """Views into a forward's image buffer (tests / diagnostics; layout: rtgs_raster_image_offsets): tile ranges
    int32[tiles, 2], n_contrib int32[H*W], the backward's walk per tile int32[tiles] (0 strip, 1 row-granular, 2 entry-per-lane), the list position
    of every pixel's depth owner int32[H*W] and the share of
    the tile's list its 4x4 blocks need on average (what the choice is made from)."""
    off = (C.c_size_t * 6)()
    _lib.check(_lib.load().rtgs_raster_image_offsets(int(H), int(W), off), "rtgs_raster_image_offsets")
    nt = ((H + 15) // 16) * ((W + 15) // 16)
    view = lambda o, nbytes, dt: img[int(o):int(o) + nbytes].view(dt)
    word = view(off[3], nt * 4, torch.int32)
    return dict(ranges=view(off[0], nt * 8, torch.int32).view(nt, 2), n_contrib=view(off[1], H * W * 4, torch.int32),
                tile_mode=word & 3, tile_share=(word >> 8).float() / 1000.0,
                depth_pos=view(off[5], H * W * 4, torch.int32))


_tls = threading.local()


def current_context() -> RasterContext:
    """The context ops of the calling thread use when none is passed: the library's default context on the main
    thread, a private one (created on first use) on every other thread - so two threads never share statistics,
    pinned sync words or near-slice settings by accident."""
    c = getattr(_tls, "ctx", None)
    if c is None:
        c = RasterContext() if threading.current_thread() is threading.main_thread() else RasterContext.create()
        _tls.ctx = c
    return c


def _f32c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class _Keep:
    """Holds the device tensors the settings struct points at alive for the call."""

    def __init__(self, rs: GaussianRasterizationSettings, device):
        self.bg = _f32c(rs.bg.to(device))
        self.view = _f32c(rs.viewmatrix.to(device))
        self.proj = _f32c(rs.projmatrix.to(device))
        self.campos = _f32c(rs.campos.to(device))
        self.c = _lib.RasterSettingsC(
            int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
            self.bg.data_ptr(), float(rs.scale_modifier), self.view.data_ptr(), self.proj.data_ptr(),
            int(rs.sh_degree), self.campos.data_ptr(), float(rs.opaque_threshold), float(rs.depth_threshold),
            float(rs.normal_threshold), float(rs.color_sigma), int(bool(rs.prefiltered)), int(bool(rs.debug)),
            float(rs.cx), float(rs.cy), float(rs.T_threshold))


class _Arena:
    """Resize callback target: allocates through torch's caching allocator on the op's device."""

    def __init__(self, device):
        self.device = device
        self.tensor: Optional[torch.Tensor] = None
        self.cb = _lib.RESIZE_FN(self._resize)

    def _resize(self, _user, nbytes):
        try:
            self.tensor = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=self.device)
            return self.tensor.data_ptr()
        except Exception:  # surfaces as RTGS_E_ALLOC
            return 0


class RowGradArena:
    """Persistent gradient rows for `rtgs_raster_backward_rows` (include/rtgs_raster.h): the six gradient
    tensors of the rasterizer, the raw8 gradient of the activation kernel, the SplatGrad scratch and one state
    byte per Gaussian, all allocated (zero) once.  Pass it as `grad_rows=` to `GaussianRasterizer.forward`; the
    first backward after `begin_step()` then touches only the rows that received gradient.  The tensors handed
    to autograd ARE these buffers - they are overwritten by the next step.

    `capacity` rows are allocated, `P` of them are in use (a map that grows re-uses the arena: `resize`).  `train` =
    (begin, end): only these rows are differentiated (rtgs_raster_backward_range_ctx) - the trainable, "unstable" part
    of an RTG-SLAM map (mapper.py:143-156); None = every row."""

    def __init__(self, P: int, M: int, device, capacity: Optional[int] = None):
        lib = _lib.load()
        f = dict(dtype=torch.float32, device=device)
        cap = max(int(capacity) if capacity is not None else int(P), int(P), 1)
        self.P, self.M, self.capacity = int(P), int(M), cap
        self._full = dict(d_means=torch.zeros(cap, 3, **f), d_opac=torch.zeros(cap, 1, **f), d_shs=torch.zeros(cap, M, 3, **f),
                          d_scales=torch.zeros(cap, 3, **f), d_rots=torch.zeros(cap, 4, **f), d_normal=torch.zeros(cap, 3, **f),
                          d_raw8=torch.zeros(cap, 8, **f))
        self.scratch = torch.zeros(lib.rtgs_raster_backward_scratch_bytes(cap), dtype=torch.uint8, device=device)
        self._state_full = torch.zeros(cap, dtype=torch.uint8, device=device)
        self._train = None
        self._dirty = "all"      # rows of the seven gradient arrays a backward may have written since the last clear(): "all" or (lo, hi)
        self.calls = 0           # rasterizer backward passes since begin_step(); the row states describe exactly one
        self._views()

    @property
    def train(self):
        return self._train

    @train.setter
    def train(self, rng):
        self._train = rng
        if rng is None:
            self._dirty = "all"
        elif self._dirty != "all":
            lo, hi = int(rng[0]), int(rng[1])
            self._dirty = (lo, hi) if self._dirty is None else (min(self._dirty[0], lo), max(self._dirty[1], hi))

    def _views(self):
        P = self.P
        for k, t in self._full.items():
            setattr(self, k, t[:P])                       # same storage, row 0 at the same address
        self.row_state = self._state_full[:max(P, 1)]

    def resize(self, P: int, clear: bool = True):
        """Use `P` rows of the allocation from now on.  clear=True zeroes rows and states (after a permutation the ids
        mean other Gaussians); clear=False keeps them - rows appended behind the old ones are zero already (the arena's
        invariant: a row whose state is not 1 is all-zero, and rows beyond P were never handed out since the last clear)."""
        if P > self.capacity:
            raise ValueError("RowGradArena.resize beyond the allocated capacity")
        self.P = int(P)
        if clear:
            self.clear()
        self._views()

    def clear(self):
        """Zero rows and states.  Of the gradient arrays (70 floats per row, allocated for the capacity: 224 MB at 800 000 rows)
        only the rows a backward could have written since the last clear - the union of the `train` ranges set meanwhile."""
        d = self._dirty
        for t in self._full.values():
            if d == "all":
                t.zero_()
            elif d is not None and d[1] > d[0]:
                t[d[0]:min(d[1], t.shape[0])].zero_()
        self.scratch.zero_()
        self._state_full.zero_()
        self.calls = 0
        self._dirty = None
        self.train = self._train          # the range in force stays in force: it is what the next backward writes

    def begin_step(self):
        self.calls = 0


def _require_device(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError(
            "diff_gaussian_rasterization_depth (rtg_slam_amd): tensors must live on a HIP device; "
            "this build has no CPU path.")


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, opacities, shs, scales, rotations, normal_w, tile_mask, raster_settings, grad_rows=None,
                context=None):
        lib = _lib.load()
        rctx = context if context is not None else current_context()
        _require_device(means3D)
        dev = means3D.device
        rs = raster_settings
        H, W = int(rs.image_height), int(rs.image_width)
        P = int(means3D.shape[0])
        means3D, opacities, shs = _f32c(means3D), _f32c(opacities), _f32c(shs)
        scales, rotations, normal_w = _f32c(scales), _f32c(rotations), _f32c(normal_w)
        M = int(shs.shape[1]) if shs.dim() == 3 and P > 0 else 16
        tile_mask = tile_mask.to(device=dev, dtype=torch.int32).contiguous()
        gy, gx = (H + 15) // 16, (W + 15) // 16
        if tuple(tile_mask.shape) != (gy, gx):
            raise RuntimeError(f"tile_mask must be int32[{gy},{gx}], got {tuple(tile_mask.shape)}")

        f = dict(dtype=torch.float32, device=dev)
        i = dict(dtype=torch.int32, device=dev)
        color = torch.empty(3, H, W, **f)
        depth = torch.empty(1, H, W, **f)
        cidx = torch.empty(1, H, W, **i)
        didx = torch.empty(1, H, W, **i)
        cw = torch.empty(1, H, W, **f)
        dw = torch.empty(1, H, W, **f)
        Tm = torch.empty(1, H, W, **f)
        radii = torch.empty(max(P, 1), **i)
        keep = _Keep(rs, dev)
        geom, binning, img = _Arena(dev), _Arena(dev), _Arena(dev)
        R = C.c_int64(0)
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            rc = lib.rtgs_raster_forward_ctx(
                rctx.ptr, C.byref(keep.c), P, M, _ptr(means3D), _ptr(opacities), _ptr(shs), _ptr(scales), _ptr(rotations),
                _ptr(normal_w), _ptr(tile_mask), _ptr(color), _ptr(depth), _ptr(cidx), _ptr(didx), _ptr(cw),
                _ptr(dw), _ptr(Tm), _ptr(radii), geom.cb, None, binning.cb, None, img.cb, None, C.byref(R),
                0 if any(ctx.needs_input_grad[:6]) else _lib.FWD_NO_BACKWARD, C.c_void_p(stream))
        # the callbacks are bound methods of the arenas they sit in - a reference cycle per arena, i.e. the three buffers of
        # EVERY forward (146 MB of tile segments among them) stayed allocated until Python's cyclic collector came by: 8-9 GB of
        # "live" garbage in the SLAM sequence against 2.5-3.3 GB after a collection (tools/seq_memory.py, round 6)
        geom.cb = binning.cb = img.cb = None
        _lib.check(rc, "rtgs_raster_forward")
        ctx.raster_settings = rs
        ctx.num_rendered = int(R.value)
        ctx.M = M
        ctx.grad_rows = grad_rows
        ctx.rctx = rctx
        ctx.save_for_backward(means3D, opacities, shs, scales, rotations, normal_w, geom.tensor, binning.tensor,
                              img.tensor, color, Tm, didx)
        ctx.mark_non_differentiable(cidx, didx, cw, dw, Tm)
        ctx.set_materialize_grads(False)      # no zero tensors for the five non-differentiable outputs
        return color, depth, cidx, didx, cw, dw, Tm

    @staticmethod
    def backward(ctx, g_color, g_depth, *_unused):
        lib = _lib.load()
        (means3D, opacities, shs, scales, rotations, normal_w, geom, binning, img, color, Tm, didx) = ctx.saved_tensors
        rs = ctx.raster_settings
        dev = means3D.device
        P = int(means3D.shape[0])
        H, W = int(rs.image_height), int(rs.image_width)
        f = dict(dtype=torch.float32, device=dev)
        g_color = torch.zeros(3, H, W, **f) if g_color is None else _f32c(g_color)
        g_depth = torch.zeros(1, H, W, **f) if g_depth is None else _f32c(g_depth)
        arena = ctx.grad_rows
        if arena is not None:
            first = arena.calls == 0
            arena.calls += 1
            if first and P > 0 and arena.P == P and arena.M == ctx.M and arena.d_means.device == dev:
                keep = _Keep(rs, dev)
                stream = torch.cuda.current_stream(dev).cuda_stream
                t0, t1 = arena.train if arena.train is not None else (0, P)
                with torch.cuda.device(dev):
                    rc = lib.rtgs_raster_backward_range_ctx(
                        ctx.rctx.ptr, C.byref(keep.c), P, ctx.M, ctx.num_rendered, _ptr(means3D), _ptr(opacities), _ptr(shs),
                        _ptr(scales), _ptr(rotations), _ptr(normal_w), _ptr(geom), _ptr(binning), _ptr(img),
                        _ptr(color), _ptr(Tm), _ptr(didx), _ptr(g_color), _ptr(g_depth), _ptr(arena.d_means),
                        _ptr(arena.d_opac), _ptr(arena.d_shs), _ptr(arena.d_scales), _ptr(arena.d_rots),
                        _ptr(arena.d_normal), _ptr(arena.scratch), _ptr(arena.row_state), int(t0), int(t1), C.c_void_p(stream))
                _lib.check(rc, "rtgs_raster_backward_range")
                return (arena.d_means, arena.d_opac, arena.d_shs.view_as(shs), arena.d_scales, arena.d_rots,
                        arena.d_normal, None, None, None, None)
        d_means = torch.empty_like(means3D)
        d_opac = torch.empty_like(opacities)
        d_shs = torch.empty_like(shs)
        d_scales = torch.empty_like(scales)
        d_rots = torch.empty_like(rotations)
        d_normal = torch.empty_like(normal_w)
        if P > 0:
            scratch = torch.empty(lib.rtgs_raster_backward_scratch_bytes(P), dtype=torch.uint8, device=dev)
            keep = _Keep(rs, dev)
            stream = torch.cuda.current_stream(dev).cuda_stream
            with torch.cuda.device(dev):
                rc = lib.rtgs_raster_backward_ctx(
                    ctx.rctx.ptr, C.byref(keep.c), P, ctx.M, ctx.num_rendered, _ptr(means3D), _ptr(opacities), _ptr(shs),
                    _ptr(scales), _ptr(rotations), _ptr(normal_w), _ptr(geom), _ptr(binning), _ptr(img), _ptr(color),
                    _ptr(Tm), _ptr(didx), _ptr(g_color), _ptr(g_depth), _ptr(d_means), _ptr(d_opac), _ptr(d_shs),
                    _ptr(d_scales), _ptr(d_rots), _ptr(d_normal), _ptr(scratch), C.c_void_p(stream))
            _lib.check(rc, "rtgs_raster_backward")
        return d_means, d_opac, d_shs, d_scales, d_rots, d_normal, None, None, None, None


class GaussianRasterizer(nn.Module):
    """`GaussianRasterizer(raster_settings=...)(means3D=..., opacities=..., shs=..., colors_precomp=None,
    scales=..., rotations=..., cov3D_precomp=None, normal_w=..., tile_mask=...)` ->
    (color[3,H,W], depth[1,H,W], color_index_map, depth_index_map, color_hit_weight,
    depth_hit_weight, T_map) - SLAM/render.py:89-128."""

    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def forward(self, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None, normal_w=None, tile_mask=None, grad_rows: Optional[RowGradArena] = None,
                context: Optional[RasterContext] = None):
        rs = self.raster_settings
        if colors_precomp is not None or cov3D_precomp is not None:
            raise NotImplementedError(
                "colors_precomp / cov3D_precomp are always None in RTG-SLAM (SLAM/render.py:99-100) and are "
                "not part of this build")
        if shs is None or scales is None or rotations is None:
            raise ValueError("shs, scales and rotations are required")
        P = means3D.shape[0]
        if P == 0:
            # mapper.py:1000-1009 hands torch.empty(0) for every field of an empty sub-map
            dev = means3D.device
            means3D = means3D.reshape(0, 3)
            opacities = opacities.reshape(0, 1)
            shs = shs.reshape(0, 16, 3) if shs.numel() == 0 else shs
            scales = scales.reshape(0, 3)
            rotations = rotations.reshape(0, 4)
            normal_w = torch.zeros(0, 3, device=dev) if normal_w is None else normal_w.reshape(0, 3)
        if normal_w is None:
            raise ValueError("normal_w is required")
        if tile_mask is None:
            tile_mask = torch.ones((int(rs.image_height) + 15) // 16, (int(rs.image_width) + 15) // 16,
                                   dtype=torch.int32, device=means3D.device)
        return _RasterizeGaussians.apply(means3D, opacities, shs, scales, rotations, normal_w, tile_mask, rs, grad_rows,
                                         context)
