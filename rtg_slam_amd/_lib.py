"""ctypes binding of the C-ABI library (include/rtgs_raster.h, include/rtgs_icp.h, include/rtgs_slam.h).

The library is the product: there is NO fallback.  If `librtgs_hip.so` is missing or fails to
load, importing any op raises (build it with `python -c "import __graft_entry__ as g; g.build()"`
or `make -C rtg_slam_amd/csrc`)."""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RTGS_LIB_PATH") or os.path.join(_HERE, "librtgs_hip.so")   # env override: A/B of kernel variants

RESIZE_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
FWD_NO_BACKWARD = 1          # RTGS_FWD_NO_BACKWARD (include/rtgs_raster.h)


class RasterSettingsC(C.Structure):
    """Mirror of `rtgs_raster_settings` (include/rtgs_raster.h), i.e. SLAM/render.py:68-88."""
    _fields_ = [
        ("image_height", C.c_int32), ("image_width", C.c_int32),
        ("tanfovx", C.c_float), ("tanfovy", C.c_float),
        ("bg", C.c_void_p), ("scale_modifier", C.c_float),
        ("viewmatrix", C.c_void_p), ("projmatrix", C.c_void_p),
        ("sh_degree", C.c_int32), ("campos", C.c_void_p),
        ("opaque_threshold", C.c_float), ("depth_threshold", C.c_float),
        ("normal_threshold", C.c_float), ("color_sigma", C.c_float),
        ("prefiltered", C.c_int32), ("debug", C.c_int32),
        ("cx", C.c_float), ("cy", C.c_float), ("T_threshold", C.c_float),
    ]


class IcpLevelC(C.Structure):
    """Mirror of `rtgs_icp_level` (include/rtgs_icp.h)."""
    _fields_ = [
        ("H", C.c_int32), ("W", C.c_int32), ("downscale", C.c_float), ("iters", C.c_int32),
        ("vertex_src", C.c_void_p), ("normal_src", C.c_void_p),
        ("vertex_tgt", C.c_void_p), ("normal_tgt", C.c_void_p),
    ]


_lock = threading.Lock()
_lib = None

_P = C.c_void_p


class LossCfgC(C.Structure):
    """Mirror of `rtgs_loss_cfg` (include/rtgs_raster.h)."""
    _fields_ = [("color_weight", C.c_float), ("depth_weight", C.c_float), ("ssim_weight", C.c_float),
                ("add_depth_thres", C.c_float), ("render_mask", C.c_void_p), ("sums_zeroed", C.c_int32)]


class AttachC(C.Structure):
    """Mirror of `rtgs_attach` (include/rtgs_raster.h)."""
    _fields_ = [("init_xyz", C.c_void_p), ("init_raw8", C.c_void_p), ("attach_info", C.c_void_p)]


class ActivatedC(C.Structure):
    """Mirror of `rtgs_activated` (include/rtgs_raster.h)."""
    _fields_ = [(n, C.c_void_p) for n in ("opacity", "scales", "rotations", "normal")]


class MapStepArgsC(C.Structure):
    """Mirror of `rtgs_map_step_args` (include/rtgs_raster.h)."""
    _fields_ = (
        [("settings", C.POINTER(RasterSettingsC)), ("P", C.c_int32), ("sh_coeffs", C.c_int32)]
        + [(n, C.c_void_p) for n in ("xyz", "shs", "raw8", "tile_mask", "gt_color", "gt_depth")]
        + [("loss", LossCfgC), ("loss_scratch", C.c_void_p)]
        + [(n, C.c_void_p) for n in (
            "opacity", "scales", "rotations", "normal", "out_color", "out_depth", "out_color_index", "out_depth_index",
            "out_color_weight", "out_depth_weight", "out_T", "out_radii", "dL_dcolor", "dL_ddepth", "loss4",
            "d_xyz", "d_opacity", "d_shs", "d_scales", "d_rotations", "d_normal", "d_raw8", "grad_scratch", "row_state",
            "m_xyz", "v_xyz", "m_shs", "v_shs", "m_raw8", "v_raw8", "lr_xyz", "lr_shs", "lr_raw8",
            "ever_xyz", "ever_shs", "ever_raw8")]
        + [("step", C.c_int32), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float)]
        + [("attach", C.POINTER(AttachC)), ("confidence", C.c_void_p), ("activated_valid", C.c_int32)]
        + [("geom_resize", RESIZE_FN), ("geom_user", C.c_void_p), ("binning_resize", RESIZE_FN),
           ("binning_user", C.c_void_p), ("image_resize", RESIZE_FN), ("image_user", C.c_void_p)]
        + [("normal_weight", C.c_float), ("gt_normal", C.c_void_p), ("train_begin", C.c_int32), ("train_end", C.c_int32),
           ("tail_mode", C.c_int32), ("live_counts", C.c_void_p)])


_SIGNATURES = {
    "rtgs_version": (C.c_char_p, []),
    "rtgs_raster_forward": (C.c_int, [C.POINTER(RasterSettingsC), C.c_int32, C.c_int32] + [_P] * 6 + [_P]
                            + [_P] * 8 + [RESIZE_FN, _P, RESIZE_FN, _P, RESIZE_FN, _P,
                                          C.POINTER(C.c_int64), _P]),
    "rtgs_raster_backward": (C.c_int, [C.POINTER(RasterSettingsC), C.c_int32, C.c_int32, C.c_int64] + [_P] * 6
                             + [_P] * 3 + [_P, _P, _P] + [_P, _P] + [_P] * 6 + [_P, _P]),
    "rtgs_raster_backward_rows": (C.c_int, [C.POINTER(RasterSettingsC), C.c_int32, C.c_int32, C.c_int64] + [_P] * 6
                                  + [_P] * 3 + [_P, _P, _P] + [_P, _P] + [_P] * 6 + [_P, _P, _P]),
    "rtgs_raster_backward_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "rtgs_raster_geom_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rtgs_raster_binning_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "rtgs_raster_image_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rtgs_raster_last_stats": (C.c_int, [C.POINTER(C.c_int64)]),
    "rtgs_raster_set_counters": (None, [_P]),
    "rtgs_fused_adam": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, _P]),
    "rtgs_fused_adam_rows": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, _P]),
    "rtgs_map_activate8_forward": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P]),
    "rtgs_map_activate8_backward": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "rtgs_map_activate8_backward_rows": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P, _P, _P, _P]),
    "rtgs_map_tail_rows": (C.c_int, [_P] * 23 + [C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_float,
                                     C.POINTER(AttachC), _P, _P, C.POINTER(ActivatedC), _P]),
    "rtgs_attach_prepare": (C.c_int, [_P, _P, C.POINTER(AttachC), C.c_int64, _P]),
    "rtgs_history_merge": (C.c_int, [_P] * 8 + [C.c_int64, C.c_float, _P]),
    "rtgs_map_fused_tail": (C.c_int, [C.POINTER(RasterSettingsC), C.POINTER(MapStepArgsC), _P, _P, _P, _P, _P]),
    "rtgs_map_fused_tail_hint": (C.c_int, [C.POINTER(RasterSettingsC), C.POINTER(MapStepArgsC), _P, _P, _P, _P, C.c_uint32, _P]),
    "rtgs_raster_last_listed_ctx": (C.c_uint32, [_P]),
    "rtgs_stream_create_reserving": (C.c_void_p, [C.c_int]),
    "rtgs_stream_destroy": (None, [_P]),
    "rtgs_raster_backward_buffers": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "rtgs_raster_last_buffers_ctx": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "rtgs_raster_set_aux_zero_ctx": (None, [_P, _P]),
    "rtgs_raster_backward_walk_ctx": (C.c_int, [_P, C.POINTER(RasterSettingsC), C.c_int32, C.c_int32, C.c_int64] + [_P] * 6
                                      + [_P] * 3 + [_P, _P, _P] + [_P, _P] + [_P] * 6 + [_P, _P, C.c_int32, C.c_int32, _P]),
    "rtgs_slam_loss_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "rtgs_slam_map_step_front": (C.c_int, [C.POINTER(MapStepArgsC), C.POINTER(C.c_int64), _P]),
    "rtgs_rows_pack": (C.c_int, [_P, C.c_int32] + [_P] * 7 + [C.c_int32, _P, _P]),
    "rtgs_rows_overflow": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "rtgs_rows_apply": (C.c_int, [_P, C.c_int32, C.c_int32] + [_P] * 7 + [_P, _P]),
    "rtgs_map_step_args_size": (C.c_size_t, []),
    "rtgs_raster_settings_size": (C.c_size_t, []),
    "rtgs_slam_map_step": (C.c_int, [C.POINTER(MapStepArgsC), C.POINTER(C.c_int64), _P]),
    "rtgs_slam_loss": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(LossCfgC), _P, _P, _P, _P, _P]),
    "rtgs_slam_normal_loss": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, _P, _P, _P]),
    "rtgs_slam_loss_sums": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(LossCfgC), _P, _P]),
    "rtgs_slam_loss_grads": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.POINTER(LossCfgC), _P, _P, _P, _P, _P]),
    "rtgs_raster_set_profiling": (None, [C.c_int]),
    "rtgs_raster_force_sort_path": (None, [C.c_int]),
    "rtgs_raster_set_near_slice": (None, [C.c_int, C.c_int]),
    "rtgs_raster_last_slice_stats": (C.c_int, [C.POINTER(C.c_int64)]),
    "rtgs_raster_last_timings": (C.c_int, [C.POINTER(C.c_float)]),
    "rtgs_ctx_create": (C.c_void_p, []),
    "rtgs_ctx_destroy": (None, [_P]),
    "rtgs_raster_forward_ctx": (C.c_int, [_P, C.POINTER(RasterSettingsC), C.c_int32, C.c_int32] + [_P] * 6 + [_P]
                                + [_P] * 8 + [RESIZE_FN, _P, RESIZE_FN, _P, RESIZE_FN, _P, C.POINTER(C.c_int64),
                                              C.c_int32, _P]),
    "rtgs_raster_backward_ctx": (C.c_int, [_P, C.POINTER(RasterSettingsC), C.c_int32, C.c_int32, C.c_int64] + [_P] * 6
                                 + [_P] * 3 + [_P, _P, _P] + [_P, _P] + [_P] * 6 + [_P, _P]),
    "rtgs_raster_backward_rows_ctx": (C.c_int, [_P, C.POINTER(RasterSettingsC), C.c_int32, C.c_int32, C.c_int64] + [_P] * 6
                                      + [_P] * 3 + [_P, _P, _P] + [_P, _P] + [_P] * 6 + [_P, _P, _P]),
    "rtgs_raster_backward_range_ctx": (C.c_int, [_P, C.POINTER(RasterSettingsC), C.c_int32, C.c_int32, C.c_int64] + [_P] * 6
                                       + [_P] * 3 + [_P, _P, _P] + [_P, _P] + [_P] * 6 + [_P, _P, C.c_int32, C.c_int32, _P]),
    "rtgs_slam_normal_loss_range": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, _P, _P,
                                              C.c_int32, C.c_int32, _P]),
    "rtgs_slam_normal_loss_sums": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "rtgs_slam_normal_loss_grads": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, _P, _P,
                                              C.c_int32, C.c_int32, _P]),
    "rtgs_raster_geom_bytes_ctx": (C.c_size_t, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "rtgs_raster_last_stats_ctx": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "rtgs_raster_set_counters_ctx": (None, [_P, _P]),
    "rtgs_raster_set_profiling_ctx": (None, [_P, C.c_int]),
    "rtgs_raster_force_sort_path_ctx": (None, [_P, C.c_int]),
    "rtgs_raster_set_bwd_walk_ctx": (None, [_P, C.c_int]),
    "rtgs_raster_set_bwd_debug": (None, [C.c_int]),
    "rtgs_raster_set_bwd_stamps": (None, [C.c_void_p]),
    "rtgs_raster_set_fwd_stamps": (None, [C.c_void_p]),
    "rtgs_raster_set_onepass_ctx": (None, [_P, C.c_int]),
    "rtgs_raster_forward_verify_ctx": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "rtgs_raster_spec_fail_ptr_ctx": (C.c_void_p, [_P]),
    "rtgs_raster_set_speculation_ctx": (None, [_P, C.c_int]),
    "rtgs_raster_speculation_stats_ctx": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "rtgs_raster_plain_stats_ctx": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "rtgs_raster_set_plain_onepass_ctx": (None, [_P, C.c_int]),
    "rtgs_raster_image_offsets": (C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_size_t)]),
    "rtgs_raster_set_near_slice_ctx": (None, [_P, C.c_int, C.c_int]),
    "rtgs_raster_last_slice_stats_ctx": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "rtgs_raster_last_timings_ctx": (C.c_int, [_P, C.POINTER(C.c_float)]),
    "rtgs_slam_map_step_ctx": (C.c_int, [_P, C.POINTER(MapStepArgsC), C.POINTER(C.c_int64), _P]),
    "rtgs_slam_map_step_front_ctx": (C.c_int, [_P, C.POINTER(MapStepArgsC), C.POINTER(C.c_int64), _P]),
    "rtgs_icp_build_pyramids": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, C.POINTER(_P), C.POINTER(_P), _P, _P]),
    "rtgs_icp_build_pyramids_ex": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_int32, C.POINTER(_P), C.POINTER(_P), _P, C.c_int32, _P]),
    "rtgs_icp_scratch_init": (C.c_int, [_P, _P]),
    "rtgs_icp_step": (C.c_int, [_P] * 4 + [C.c_int32, C.c_int32, _P, _P, C.c_float, C.c_float, _P, _P, _P, _P, _P]),
    "rtgs_icp_track": (C.c_int, [C.POINTER(IcpLevelC), C.c_int32, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P,
                                 C.c_int32, _P]),
    "rtgs_icp_fill_model_depth": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, C.c_float, _P]),
    "rtgs_icp_scratch_bytes": (C.c_size_t, []),
    # include/rtgs_slam.h
    "rtgs_tile_sum": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "rtgs_transmission2tilemask": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P]),
    "rtgs_pixelmask2tilemask": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "rtgs_colorerror2tilemask": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_float, _P, _P, _P]),
    "rtgs_colorerror2tilemask_k": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "rtgs_render_range": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_float, _P, _P, _P, _P, _P]),
    "rtgs_knn3_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "rtgs_knn3": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P]),
    "rtgs_knn3_query_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rtgs_knn3_query": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "rtgs_knn3_built_bytes": (C.c_size_t, [C.c_int32]),
    "rtgs_knn3_query_built_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "rtgs_knn3_build_ref": (C.c_int, [_P, C.c_int32, _P, _P]),
    "rtgs_knn3_query_built": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "rtgs_knn3_dynamic_merge": (C.c_int, [_P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "rtgs_accumulate_error": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, _P, C.c_float, C.c_float,
                                        C.c_float, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "rtgs_bilateral_filter": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, _P, _P]),
    "rtgs_frame_preprocess_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "rtgs_frame_preprocess": (C.c_int, [_P, C.c_int32, C.c_int32, _P, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P, _P,
                                        _P, _P]),
    "rtgs_compact_scratch_bytes": (C.c_size_t, [C.c_int32]),
    "rtgs_sample_candidates": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P, _P, _P]),
    "rtgs_add_masks": (C.c_int, [_P] * 6 + [C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, _P, _P, _P, _P]),
    "rtgs_frame_errors": (C.c_int, [_P] * 5 + [C.c_int32, C.c_int32, _P, _P, _P]),
    "rtgs_error_counters": (C.c_int, [C.c_int32, _P, _P, C.c_float, C.c_float, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "rtgs_delete_mask": (C.c_int, [C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "rtgs_gather_new_points": (C.c_int, [_P, C.c_int32, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P]),
    "rtgs_draw_new_points": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_uint64, _P, _P, _P, C.c_int32, _P, _P, _P, _P, _P, _P]),
    "rtgs_filter_keep": (C.c_int, [C.c_int32, _P, _P, _P, C.c_float, _P, _P]),
    "rtgs_bbox_pad": (C.c_int, [C.c_int32, _P, C.c_float, _P, _P]),
    "rtgs_compact_points": (C.c_int, [C.c_int32] + [_P] * 11),
    "rtgs_append_valid_rows": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, C.c_int32, _P, _P, _P, _P]),
    "rtgs_new_rows": (C.c_int, [C.c_int32, _P, _P, _P, _P, _P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                C.c_float, _P, _P, _P]),
    "rtgs_attach_test": (C.c_int, [_P, C.c_int32, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int32, C.c_int32, _P, _P, _P,
                                   C.c_float, _P, _P]),
    "rtgs_transform_map": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "rtgs_gather_rows3": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
    "rtgs_scatter_rows3": (C.c_int, [_P, _P, C.c_int32, _P, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


def load():
    """Load (once) and return the ctypes handle; raises RuntimeError when the HIP library is absent."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"rtg_slam_amd: native library not found at {LIB_PATH}. There is no CPU fallback; build it "
                "with `make -C rtg_slam_amd/csrc` (hipcc --offload-arch=gfx950).")
        try:
            lib = C.CDLL(LIB_PATH)
        except OSError as e:  # pragma: no cover
            raise RuntimeError(f"rtg_slam_amd: failed to load {LIB_PATH}: {e}") from e
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        for what, mirror, size in (("rtgs_map_step_args", MapStepArgsC, lib.rtgs_map_step_args_size()),
                                   ("rtgs_raster_settings", RasterSettingsC, lib.rtgs_raster_settings_size())):
            if C.sizeof(mirror) != size:
                raise RuntimeError(f"rtg_slam_amd: ctypes mirror of {what} is {C.sizeof(mirror)} B, the library says "
                                   f"{size} B - include/rtgs_raster.h and rtg_slam_amd/_lib.py are out of step")
        _lib = lib
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = {-1: "invalid argument", -2: "HIP runtime / kernel launch failure", -3: "scratch allocation failed"}.get(rc, "unknown")
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")
