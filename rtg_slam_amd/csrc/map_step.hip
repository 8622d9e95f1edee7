// rtgs_slam_map_step: one map-optimisation iteration (the inner loop body of mapper.py:176-205 with the live loss
// terms of mapper.py:402-442) enqueued by ONE call: raw8 activation -> rasterizer forward -> fused SLAM loss ->
// row-state rasterizer backward -> activation backward + row-skipping Adam on the three block tensors (one kernel).
// Nothing new is computed here - it is the sequence map_optim.ShardedMapOptimizer.step() issues through autograd,
// without the per-launch Python / autograd cost (which exceeds the GPU time of the step on a 1.2 M map).
#include <hip/hip_runtime.h>

#include "../../include/rtgs_raster.h"

// Everything up to and including the rasterizer backward: the gradient rows are in the arena, nothing is stepped yet.
static int map_step_front(rtgs_ctx* ctx, const rtgs_map_step_args* a, int64_t* num_rendered_host, int32_t fwd_flags,
                          void* stream, bool walk_only = false) {
  if (!a || !num_rendered_host || !a->settings) return RTGS_E_INVALID;
  const int32_t P = a->P, M = a->sh_coeffs;
  if (P <= 0 || M != 16 || a->step < 1) return RTGS_E_INVALID;
  if (!a->xyz || !a->shs || !a->raw8 || !a->tile_mask || !a->gt_color || !a->gt_depth) return RTGS_E_INVALID;
  const int32_t H = a->settings->image_height, W = a->settings->image_width;
  int rc = 0;
  if (!a->activated_valid) {
    rc = rtgs_map_activate8_forward(a->raw8, P, a->opacity, a->scales, a->rotations, a->normal, stream);
    if (rc != 0) return RTGS_E_HIP;
  }
  if (!a->loss_scratch || !a->loss4) return RTGS_E_INVALID;
  rtgs_raster_set_aux_zero_ctx(ctx, a->loss_scratch);     // the forward's blend clears the loss sums: no memset launch
  rc = rtgs_raster_forward_ctx(ctx, a->settings, P, M, a->xyz, a->opacity, a->shs, a->scales, a->rotations, a->normal,
                           a->tile_mask, a->out_color, a->out_depth, a->out_color_index, a->out_depth_index,
                           a->out_color_weight, a->out_depth_weight, a->out_T, a->out_radii, a->geom_resize,
                           a->geom_user, a->binning_resize, a->binning_user, a->image_resize, a->image_user,
                           num_rendered_host, fwd_flags, stream);
  if (rc != RTGS_OK) return rc;
  void* bufs[3];                                          // what the forward got from the resize callbacks (no call back)
  if (rtgs_raster_last_buffers_ctx(ctx, bufs) != RTGS_OK) return RTGS_E_ALLOC;
  void *geom = bufs[0], *bin = bufs[1], *img = bufs[2];
  rtgs_loss_cfg lcfg = a->loss;
  lcfg.sums_zeroed = P > 0 ? 1 : 0;
  rc = rtgs_slam_loss(a->out_color, a->out_depth, a->out_depth_index, a->gt_color, a->gt_depth, H, W, &lcfg,
                      a->loss_scratch, a->loss4, a->dL_dcolor, a->dL_ddepth, stream);
  if (rc != 0) return RTGS_E_HIP;
  int32_t t0 = 0, t1 = P;                                 // the trainable rows (rtgs_map_step_args: ONLY 0, 0 = all)
  if (a->train_begin != 0 || a->train_end != 0) { t0 = a->train_begin; t1 = a->train_end; }
  if (t0 < 0 || t1 > P || t1 < t0) return RTGS_E_INVALID;
  if (t1 == t0) return RTGS_OK;                           // an empty range (a fully frozen map): rendered, loss evaluated, nothing differentiated
  rc = (walk_only ? rtgs_raster_backward_walk_ctx : rtgs_raster_backward_range_ctx)(
      ctx, a->settings, P, M, *num_rendered_host, a->xyz, a->opacity, a->shs, a->scales, a->rotations, a->normal, geom, bin, img,
      a->out_color, a->out_T, a->out_depth_index, a->dL_dcolor, a->dL_ddepth, a->d_xyz, a->d_opacity, a->d_shs, a->d_scales,
      a->d_rotations, a->d_normal, a->grad_scratch, a->row_state, t0, t1, stream);
  if (rc != RTGS_OK) return rc;
  if (a->normal_weight > 0.f && a->gt_normal) {
    // the normal term (mapper.py:433-442): value into loss4[0], gradient into the depth owners' d_normal rows; the two
    // floats it needs are words 5-6 of the loss scratch's 8-float header (the image loss uses 0-4 and is done by now)
    rc = rtgs_slam_normal_loss_range(a->normal, a->out_depth_index, a->gt_normal, a->loss.render_mask, H, W, a->normal_weight,
                                     reinterpret_cast<float*>(a->loss_scratch) + 5, a->loss4, a->d_normal, a->row_state,
                                     rtgs_raster_spec_fail_ptr_ctx(ctx), t0, t1, stream);
    if (rc != 0) return RTGS_E_HIP;
  }
  return RTGS_OK;
}

extern "C" int rtgs_slam_map_step_front_ctx(rtgs_ctx* ctx, const rtgs_map_step_args* a, int64_t* num_rendered_host,
                                            void* stream) {
  return map_step_front(ctx, a, num_rendered_host, 0, stream);     // the caller runs the tail: nothing to guard it with
}

extern "C" int rtgs_slam_map_step_front(const rtgs_map_step_args* a, int64_t* num_rendered_host, void* stream) {
  return rtgs_slam_map_step_front_ctx(nullptr, a, num_rendered_host, stream);
}

// One pass over the step.  With `speculate` the forward does not wait for the host in the middle (RTGS_FWD_SPECULATE):
// the whole iteration is enqueued back to back, the tail is guarded by the forward's device word, and the totals are
// verified at the end - while the GPU is still busy with the backward.
static int map_step_once(rtgs_ctx* ctx, const rtgs_map_step_args* a, int64_t* num_rendered_host, bool speculate, void* stream) {
  // one fused kernel for everything per-Gaussian behind the tile walk, unless the caller asks for the three-kernel form or
  // the normal term (which adds to the arena's d_normal rows) is on
  const bool fused = a->tail_mode == 0 && !(a->normal_weight > 0.f && a->gt_normal);
  int rc = map_step_front(ctx, a, num_rendered_host, speculate ? RTGS_FWD_SPECULATE : 0, stream, fused);
  if (rc != RTGS_OK) return rc;
  if ((a->train_begin != 0 || a->train_end != 0) && a->train_end == a->train_begin) return RTGS_OK;   // empty range: no tail
  if (fused) {
    void* bufs[3];
    if (rtgs_raster_last_buffers_ctx(ctx, bufs) != RTGS_OK) return RTGS_E_ALLOC;
    return rtgs_map_fused_tail_hint(a->settings, a, bufs[0], bufs[2], rtgs_raster_spec_fail_ptr_ctx(ctx), a->live_counts,
                                    rtgs_raster_last_listed_ctx(ctx), stream);
  }
  const int32_t P = a->P;
  // activation backward (+ attach gradient) + Adam on the three block tensors (+ confidence increment), one launch;
  // the rows it steps are re-activated, so the next call may skip the full activation pass (activated_valid)
  // ... over the trainable rows only: every per-row pointer moves to row t0 (the Adam state, the attach snapshot and the
  // confidence array already start there, see rtgs_map_step_args)
  int32_t t0 = 0, t1 = P;
  if (a->train_begin != 0 || a->train_end != 0) { t0 = a->train_begin; t1 = a->train_end; }
  const size_t o = (size_t)t0;
  const rtgs_activated act{a->opacity + o, a->scales + 3 * o, a->rotations + 4 * o, a->normal + 3 * o};
  rc = rtgs_map_tail_rows(a->xyz + 3 * o, a->shs + 48 * o, a->raw8 + 8 * o, a->d_opacity + o, a->d_scales + 3 * o,
                          a->d_rotations + 4 * o, a->d_normal + 3 * o, a->d_xyz + 3 * o, a->d_shs + 48 * o, a->d_raw8 + 8 * o,
                          a->row_state + o, a->m_xyz, a->v_xyz, a->m_shs, a->v_shs, a->m_raw8, a->v_raw8,
                          a->lr_xyz, a->lr_shs, a->lr_raw8, a->ever_xyz, a->ever_shs, a->ever_raw8, (int64_t)(t1 - t0), a->step,
                          a->beta1, a->beta2, a->eps, a->attach, a->confidence, rtgs_raster_spec_fail_ptr_ctx(ctx), &act, stream);
  return rc != 0 ? RTGS_E_HIP : RTGS_OK;
}

extern "C" int rtgs_slam_map_step_ctx(rtgs_ctx* ctx, const rtgs_map_step_args* a, int64_t* num_rendered_host, void* stream) {
  int rc = map_step_once(ctx, a, num_rendered_host, true, stream);
  const int v = rtgs_raster_forward_verify_ctx(ctx, num_rendered_host);      // no-op when the forward did not speculate
  if (rc != RTGS_OK) return rc;
  if (v < 0) return v;
  if (v == 1) rc = map_step_once(ctx, a, num_rendered_host, false, stream);  // the guess did not hold: nothing was changed, redo
  return rc;
}

extern "C" int rtgs_slam_map_step(const rtgs_map_step_args* a, int64_t* num_rendered_host, void* stream) {
  return rtgs_slam_map_step_ctx(nullptr, a, num_rendered_host, stream);
}

// Layout guards for foreign-function bindings (ctypes / cgo / JNI mirrors of the two structs): compare with sizeof on the
// binding's side at load time.
extern "C" size_t rtgs_map_step_args_size(void) { return sizeof(rtgs_map_step_args); }
extern "C" size_t rtgs_raster_settings_size(void) { return sizeof(rtgs_raster_settings); }

// ---- a stream for the mapper that leaves some compute units to the tracker ---------------------------------------------
// Tracker || mapper on two streams (rtg_slam_amd/pipeline.py): the tracker is a chain of ~20 short dependent kernels, the
// mapper a chain of long ones with thousands of workgroups.  Priority only orders DISPATCH: a tracker kernel that becomes
// ready still waits for the mapper's resident workgroups to retire before it finds wave slots (icp_reduce: 16 us alone,
// 24 us beside the mapper - and the frame waits for the tracker).  A mapper stream whose CU mask excludes `reserve_cus`
// compute units keeps those free: the tracker's workgroups start at once there (and anywhere else a slot opens).
extern "C" void* rtgs_stream_create_reserving(int32_t reserve_cus) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return nullptr;
  if (reserve_cus <= 0 || reserve_cus >= cus || cus > 1024) return nullptr;
  uint32_t mask[32] = {0};
  const int words = (cus + 31) / 32;
  for (int i = 0; i < cus - reserve_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  if (hipExtStreamCreateWithCUMask(&s, (uint32_t)words, mask) != hipSuccess) return nullptr;
  return (void*)s;
}
extern "C" void rtgs_stream_destroy(void* stream) {
  if (stream) (void)hipStreamDestroy((hipStream_t)stream);
}
