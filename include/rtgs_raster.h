/*
 * rtgs_raster.h - C ABI of the MI355X (gfx950) differentiable Gaussian-splatting rasterizer.
 *
 * Drop-in boundary for the native op behind RTG-SLAM's
 *   diff_gaussian_rasterization_depth.GaussianRasterizer
 * (reference call site /root/reference/SLAM/render.py:68-128; the CUDA sources it binds are an
 * un-vendored submodule, /root/reference/.gitmodules:1-4).  The pybind entry points that
 * submodule exports (`_C.rasterize_gaussians`, `_C.rasterize_gaussians_backward`) map 1:1 onto
 * rtgs_raster_forward / rtgs_raster_backward below; the three resize-callbacks mirror the
 * geometry / binning / image scratch buffers the autograd ctx carries from forward to backward.
 *
 * Conventions: every pointer is a DEVICE pointer unless its name ends in `_host`; all tensors
 * are dense row-major float32 / int32; inputs are borrowed and never written; the call is
 * enqueued on `stream` (a hipStream_t passed as void*).  Return value 0 = success, negative =
 * error (see RTGS_E_*); nothing throws across the boundary.
 */
#ifndef RTGS_RASTER_H
#define RTGS_RASTER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGS_OK 0
#define RTGS_E_INVALID (-1)   /* bad argument (null pointer, negative size, sh_degree > 3 ...) */
#define RTGS_E_HIP (-2)       /* a HIP runtime call or kernel launch failed                     */
#define RTGS_E_ALLOC (-3)     /* a resize callback returned NULL                                */

#define RTGS_TILE 16          /* tile edge in pixels: SLAM/render.py:104-105, mapper.py:488-505  */

/* The 19 fields of GaussianRasterizationSettings, SLAM/render.py:68-88, in that order.
 * bg / viewmatrix / projmatrix / campos stay device tensors exactly as the reference passes
 * them (no host read-back of camera state). */
typedef struct rtgs_raster_settings {
  int32_t image_height;
  int32_t image_width;
  float tanfovx;
  float tanfovy;
  const float* bg;          /* [3]   device                                               */
  float scale_modifier;
  const float* viewmatrix;  /* [4,4] device, = W2C transposed (scene/cameras.py:96-98)     */
  const float* projmatrix;  /* [4,4] device, accepted for API parity, not read             */
  int32_t sh_degree;        /* active degree 0..3                                          */
  const float* campos;      /* [3]   device                                               */
  float opaque_threshold;
  float depth_threshold;
  float normal_threshold;   /* cosine                                                      */
  float color_sigma;
  int32_t prefiltered;
  int32_t debug;            /* non-zero: synchronise + check after every kernel            */
  float cx;                 /* <= 0 -> (W-1)/2                                             */
  float cy;
  float T_threshold;
} rtgs_raster_settings;

/* Context: everything the library remembers between calls - the near-slice mode and budget, the statistics /
 * counters / stage timings of the last forward and backward, the pinned words of the forward's host sync.  The plain
 * entry points below use ONE process-wide default context and are meant for a single calling thread (plus autograd's
 * backward thread, which runs strictly after its forward).  A process that renders from several threads creates one
 * context per thread (rtgs_ctx_create) and calls the *_ctx variants, which take the context as first argument and
 * are otherwise identical; ctx == NULL selects the default context.  A context belongs to one device.  New contexts
 * start from the default context's near-slice mode and budget (environment: RTGS_NEAR_SLICE, RTGS_NEAR_SLICE_BUDGET). */
typedef struct rtgs_ctx rtgs_ctx;
rtgs_ctx* rtgs_ctx_create(void);
void rtgs_ctx_destroy(rtgs_ctx* ctx);

/* Resize callback: return a device allocation of at least `bytes` bytes, 256-byte aligned,
 * that stays alive until the matching backward has run (mirrors the resize-lambdas over
 * torch::empty byte tensors of the reference binding). */
typedef void* (*rtgs_resize_fn)(void* user, size_t bytes);

/* Forward: the 9-argument call of SLAM/render.py:110-120 (colors_precomp / cov3D_precomp are
 * always None in RTG-SLAM and are not part of this ABI).
 *   P            number of Gaussians (may be 0)
 *   sh_coeffs    coefficients per channel held in `shs` (16 for max_sh_degree 3)
 * Outputs (SLAM/render.py:122-128), all fully written:
 *   out_color[3,H,W] out_depth[1,H,W] out_color_index[1,H,W](i32,-1) out_depth_index[1,H,W](i32,-1)
 *   out_color_weight[1,H,W] out_depth_weight[1,H,W] out_T[1,H,W] (exactly 1.0f where untouched)
 *   out_radii[P] (i32, 0 = culled; may be NULL)
 *   num_rendered_host  number of (Gaussian, tile) instances, written to HOST memory
 */
int rtgs_raster_forward(const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                        const float* means3D, const float* opacities, const float* shs,
                        const float* scales, const float* rotations, const float* normal_w,
                        const int32_t* tile_mask,
                        float* out_color, float* out_depth, int32_t* out_color_index,
                        int32_t* out_depth_index, float* out_color_weight,
                        float* out_depth_weight, float* out_T, int32_t* out_radii,
                        rtgs_resize_fn geom_resize, void* geom_user,
                        rtgs_resize_fn binning_resize, void* binning_user,
                        rtgs_resize_fn image_resize, void* image_user,
                        int64_t* num_rendered_host, void* stream);

/* Backward: consumes the three scratch buffers of the matching forward plus the forward's
 * out_color, out_T and out_depth_index.  The geometry and binning buffers are scratch of the forward / backward PAIR:
 * the backward writes into them too (per-Gaussian slot counters, per-(Gaussian, tile) gradient partials), so one
 * forward's buffers serve ONE backward at a time.  Writes (never accumulates into) the gradient tensors:
 *   dL_dmeans3D[P,3] dL_dopacities[P,1] dL_dshs[P,sh_coeffs,3] dL_dscales[P,3]
 *   dL_drotations[P,4] dL_dnormal_w[P,3]
 * Rows of Gaussians that touched no rendered pixel are exactly 0 (mapper.py:455 relies on it).
 * `grad_scratch` must hold rtgs_raster_backward_scratch_bytes(P) bytes. */
int rtgs_raster_backward(const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                         int64_t num_rendered,
                         const float* means3D, const float* opacities, const float* shs,
                         const float* scales, const float* rotations, const float* normal_w,
                         void* geom_buffer, void* binning_buffer,
                         const void* image_buffer, const float* out_color, const float* out_T,
                         const int32_t* out_depth_index,
                         const float* dL_dcolor, const float* dL_ddepth,
                         float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                         float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                         void* grad_scratch, void* stream);

size_t rtgs_raster_backward_scratch_bytes(int32_t P);

/* flags of rtgs_raster_forward_ctx (the plain rtgs_raster_forward passes 0) */
#define RTGS_FWD_NO_BACKWARD 1   /* no backward will follow: skip the backward's bookkeeping (a scan over the Gaussians and
                                    the gradient-slot allocation inside the binning buffer).  A backward called anyway
                                    is still correct - it falls back to global atomics. */
#define RTGS_FWD_SPECULATE 2     /* Speculative sizing - NO host wait inside the call (upstream reads num_rendered back in
                                    the middle of its forward; so does the plain call, through pinned memory).  The host
                                    assumes this call looks like the last verified forward on the context (same pass
                                    structure; capacities = the decaying maxima of the instance / gradient-slot totals + 12.5 %
                                    + 8192, the sort-class ceiling above the longest list + 25 %: raster_api.hip), sizes the binning buffer and picks the sort classes from that, and enqueues
                                    everything.  The kernel that learns the real numbers checks them against those
                                    capacities and raises a device word when they do not hold; every later kernel of the
                                    forward AND of its backward that could overrun a buffer or change persistent state
                                    returns at once on that word.  The caller MUST call rtgs_raster_forward_verify_ctx
                                    before trusting any result or launching another forward on the context, pass
                                    rtgs_raster_spec_fail_ptr_ctx() as the skip flag of whatever it runs on the
                                    gradients (rtgs_map_tail_rows), and redo forward + consumers without this flag when
                                    verify returns 1.  Ignored (plain call) when the context has no verified history for
                                    this (P, H, W) or the pass structure is not one it can guess.  Used by
                                    rtgs_slam_map_step; outputs and num_rendered of a verified call equal the plain
                                    call's (num_rendered_host holds the last call's number until verify). */
int rtgs_raster_forward_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                            const float* means3D, const float* opacities, const float* shs,
                            const float* scales, const float* rotations, const float* normal_w,
                            const int32_t* tile_mask,
                            float* out_color, float* out_depth, int32_t* out_color_index,
                            int32_t* out_depth_index, float* out_color_weight,
                            float* out_depth_weight, float* out_T, int32_t* out_radii,
                            rtgs_resize_fn geom_resize, void* geom_user,
                            rtgs_resize_fn binning_resize, void* binning_user,
                            rtgs_resize_fn image_resize, void* image_user,
                            int64_t* num_rendered_host, int32_t flags, void* stream);
int rtgs_raster_backward_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                             int64_t num_rendered,
                             const float* means3D, const float* opacities, const float* shs,
                             const float* scales, const float* rotations, const float* normal_w,
                             void* geom_buffer, void* binning_buffer,
                             const void* image_buffer, const float* out_color, const float* out_T,
                             const int32_t* out_depth_index,
                             const float* dL_dcolor, const float* dL_ddepth,
                             float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                             float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                             void* grad_scratch, void* stream);

/* Row-state backward (an extension, not part of the reference contract): same arithmetic, but for callers that
 * keep the six gradient tensors, `grad_scratch` and one byte per Gaussian, `row_state[P]`, ALIVE between calls -
 * all zero-initialised once by the caller.  Invariant on entry and exit: a gradient row is all-zero unless its
 * state is 1, and grad_scratch is all-zero.  On exit row_state[i] is
 *   1  row i received gradient in this call (values identical to rtgs_raster_backward's),
 *   2  row i carried gradient from the previous call and was zeroed by this one,
 *   0  row i was zero and stays zero (not written at all).
 * A Gaussian that reaches no pixel then costs two byte reads instead of ~310 B of zero writes, and consumers
 * (rtgs_map_activate8_backward_rows, rtgs_fused_adam_rows) skip state-0 rows the same way. */
int rtgs_raster_backward_rows(const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                              int64_t num_rendered,
                              const float* means3D, const float* opacities, const float* shs,
                              const float* scales, const float* rotations, const float* normal_w,
                              void* geom_buffer, void* binning_buffer,
                              const void* image_buffer, const float* out_color, const float* out_T,
                              const int32_t* out_depth_index,
                              const float* dL_dcolor, const float* dL_ddepth,
                              float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                              float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                              void* grad_scratch, uint8_t* row_state, void* stream);

int rtgs_raster_backward_rows_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                                  int64_t num_rendered,
                                  const float* means3D, const float* opacities, const float* shs,
                                  const float* scales, const float* rotations, const float* normal_w,
                                  void* geom_buffer, void* binning_buffer,
                                  const void* image_buffer, const float* out_color, const float* out_T,
                                  const int32_t* out_depth_index,
                                  const float* dL_dcolor, const float* dL_ddepth,
                                  float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                                  float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                                  void* grad_scratch, uint8_t* row_state, void* stream);
/* rtgs_raster_backward_rows_ctx restricted to the trainable rows [train_begin, train_end): Gaussians outside the range
 * take part in the blend (the forward rendered them) but receive NOTHING - no gradient slot, no record, their rows of the
 * gradient tensors and of row_state are not touched (they stay zero from the allocation).  (0, P) = every row. */
int rtgs_raster_backward_range_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                                  int64_t num_rendered,
                                  const float* means3D, const float* opacities, const float* shs,
                                  const float* scales, const float* rotations, const float* normal_w,
                                  void* geom_buffer, void* binning_buffer,
                                  const void* image_buffer, const float* out_color, const float* out_T,
                                  const int32_t* out_depth_index,
                                  const float* dL_dcolor, const float* dL_ddepth,
                                  float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                                  float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                                  void* grad_scratch, uint8_t* row_state, int32_t train_begin, int32_t train_end, void* stream);
/* ... only its tile WALK (blend_bwd): gradient slots and touched bytes are left for rtgs_map_fused_tail; the dL_d*
 * tensors and row_state are not written (they may be NULL). */
int rtgs_raster_backward_walk_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* settings, int32_t P, int32_t sh_coeffs,
                                  int64_t num_rendered,
                                  const float* means3D, const float* opacities, const float* shs,
                                  const float* scales, const float* rotations, const float* normal_w,
                                  void* geom_buffer, void* binning_buffer,
                                  const void* image_buffer, const float* out_color, const float* out_T,
                                  const int32_t* out_depth_index,
                                  const float* dL_dcolor, const float* dL_ddepth,
                                  float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                                  float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                                  void* grad_scratch, uint8_t* row_state, int32_t train_begin, int32_t train_end, void* stream);

/* Sizes the forward will request through the callbacks (for pre-allocation / accounting).  The geometry buffer's size
 * depends on the context's near-slice budget; its layout is such that a backward never needs to know that budget. */
size_t rtgs_raster_geom_bytes(int32_t P, int32_t image_height, int32_t image_width);
size_t rtgs_raster_geom_bytes_ctx(rtgs_ctx* ctx, int32_t P, int32_t image_height, int32_t image_width);
size_t rtgs_raster_binning_bytes(int64_t num_rendered, int32_t image_height, int32_t image_width);
size_t rtgs_raster_image_bytes(int32_t image_height, int32_t image_width);

/* Per-call statistics of the LAST forward of the context (host values, for roofline
 * accounting): [0] num_rendered, [1] sort bits (fallback path), [2] tiles, [3..5] scratch bytes,
 * [6] 1 = LDS tile-sort binning / 0 = global radix-sort fallback, [7] longest tile list. */
int rtgs_raster_last_stats(int64_t* stats8_host);
int rtgs_raster_last_stats_ctx(rtgs_ctx* ctx, int64_t* stats8_host);

/* Near-slice pass of the last forward (the pass itself: rtgs_debug.h): [0] slice used, [1] instances binned for the slice,
 * [2] tiles it finished, [3] tiles left to the second pass. */
int rtgs_raster_last_slice_stats(int64_t* out4_host);
int rtgs_raster_last_slice_stats_ctx(rtgs_ctx* ctx, int64_t* out4_host);
/* Speculative forward (RTGS_FWD_SPECULATE).  verify: 0 = the guessed sizes held (or nothing was pending), 1 = they did
 * not - nothing persistent was changed, redo without the flag; < 0 = error.  It waits (spinning on pinned memory) only
 * until the kernel that publishes the totals has run.  spec_fail_ptr: device word (non-zero = failed) while a
 * speculative forward is pending, else NULL.  set_speculation 0 makes RTGS_FWD_SPECULATE a no-op on the context
 * (RTGS_SPECULATE=0 at load time); either value also forgets the context's history (the next forward runs plainly).  speculation_stats: [0] speculative forwards, [1] of which failed, [2] requests that
 * could not speculate (no history / different shape). */
int rtgs_raster_forward_verify_ctx(rtgs_ctx* ctx, int64_t* num_rendered_host);
const uint32_t* rtgs_raster_spec_fail_ptr_ctx(rtgs_ctx* ctx);
void rtgs_raster_set_speculation_ctx(rtgs_ctx* ctx, int enable);
int rtgs_raster_speculation_stats_ctx(rtgs_ctx* ctx, int64_t* out3_host);
/* Forwards WITHOUT a backward (RTGS_FWD_NO_BACKWARD; the plain renders of a SLAM frame, SLAM/render.py under torch.no_grad)
 * on a context whose last such forward on the same image declined the near slice place their instances in one pass and check
 * the assumed sort class themselves before returning (round 6): no count / scan / scatter and no host wait before the blend; a
 * wrong guess is redone on the classic path inside the same call, so results never differ.  set_plain_onepass 0 = always the
 * classic path (RTGS_PLAIN_ONEPASS=0 at load time); either value forgets the history.  plain_stats: [0] such forwards, [1] of
 * which were redone. */
void rtgs_raster_set_plain_onepass_ctx(rtgs_ctx* ctx, int enable);
int rtgs_raster_plain_stats_ctx(rtgs_ctx* ctx, int64_t* out2_host);
/* Byte offsets inside the image buffer - [0] tile ranges (uint2 per tile), [1] n_contrib (u32 per pixel), [2] BwdInfo,
 * [3] tile walk (u32 per tile: bits 0..1 = 0 strip / 1 row-granular / 2 MFMA, bits 8.. the measured share in 1/1000),
 * [4] total size, [5] list position of every pixel's depth owner (u32 per pixel). */
int rtgs_raster_image_offsets(int32_t image_height, int32_t image_width, size_t* out6_host);

/* Fused Adam over a packed [rows, cols] float32 parameter shard with one learning rate per
 * column (the six Adam groups of SLAM/gaussian_pointcloud.py:245-284; torch.optim.Adam
 * semantics with eps as given, mapper.py:156).  `step` is the 1-based step count. */
int rtgs_fused_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                    const float* lr_per_column, int64_t rows, int32_t cols, int32_t step,
                    float beta1, float beta2, float eps, void* stream);

/* Same arithmetic as rtgs_fused_adam, but a row whose gradient is all zero and whose moments never left zero
 * (ever_touched[row] == 0, a caller-owned device byte per row, zero-initialised with the optimiser state) is
 * skipped: dense Adam leaves such a row bit-identical, so results are unchanged while untouched rows cost one
 * gradient read.  cols must be 3, 8 or 48 (the block tensors of the map).  `row_state` (nullable) is the byte
 * array rtgs_raster_backward_rows maintains: when given, "gradient row is zero" is read from it (state != 1)
 * instead of scanning the gradient. */
int rtgs_fused_adam_rows(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                         const float* lr_per_column, uint8_t* ever_touched, const uint8_t* row_state, int64_t rows,
                         int32_t cols, int32_t step, float beta1, float beta2, float eps, void* stream);

/* Block-SoA map state (what rtg_slam_amd/map_optim.py keeps): xyz[N,3] and shs[N,48] are stored
 * exactly as the rasterizer reads them (no activation, no copy); only raw8[N,8] =
 * (opacity | scaling xyz | rotation wxyz) is activated:
 *   forward : raw8 -> opacity[N,1] scales[N,3] rotations[N,4] normal[N,3]
 *   backward: the gradients of those four -> g_raw8[N,8]                                        */
int rtgs_map_activate8_forward(const float* raw8, int64_t n, float* opacity, float* scales, float* rotations,
                               float* normal, void* stream);
int rtgs_map_activate8_backward(const float* raw8, int64_t n, const float* g_opacity, const float* g_scales,
                                const float* g_rotations, const float* g_normal, float* g_raw8, void* stream);
/* Row-state variant: g_raw8 is persistent; state 1 rows are computed, state 2 rows zeroed, state 0 rows skipped. */
int rtgs_map_activate8_backward_rows(const float* raw8, int64_t n, const float* g_opacity, const float* g_scales,
                                     const float* g_rotations, const float* g_normal, const uint8_t* row_state,
                                     float* g_raw8, void* stream);

/* Attach regulariser of Mapping.loss_update (mapper.py:384-401): Gaussians whose INITIAL activated opacity
 * sigmoid(init_raw8[:,0]) is below 0.9 are tied to their initial raw scaling, position and raw rotation,
 *   attach = 1000 * ( mean (scaling - scaling0)^2 + mean (xyz - xyz0)^2 + mean (rotation - rotation0)^2 ),
 * means over the selected rows x columns.  rtgs_attach_prepare counts the selected rows and evaluates the loss
 * (attach_info[0..1] = {n_selected, loss}): call it once when the snapshot is taken (the selection, hence n_selected,
 * is fixed by the snapshot) and again whenever the loss VALUE is wanted for reporting; rtgs_map_tail_rows adds the
 * regulariser's gradient to the selected rows using attach_info[0]. */
typedef struct rtgs_attach {
  const float* init_xyz;      /* [rows,3]  snapshot taken when the local optimisation starts (history_stat / init_stat) */
  const float* init_raw8;     /* [rows,8]  opacity | scaling | rotation, raw */
  float* attach_info;         /* device float[6]: [0] n_selected, [1] loss (written by rtgs_attach_prepare), [2..5] scratch */
} rtgs_attach;
int rtgs_attach_prepare(const float* xyz, const float* raw8, const rtgs_attach* attach, int64_t rows, void* stream);

/* Mapping.history_merge (mapper.py:212-251, called at the end of every local_optimize :205-210): the optimised rows are
 * blended with the snapshot taken when the optimisation began, weighted by how much confidence a row had then,
 *   w[r]      = max_weight * conf_then[r] / (conf_now[r] + 1e-6)                       (history_merge_max_weight 0.5)
 *   xyz[r]    = xyz_then[r] w[r] + (1 - w[r]) xyz[r]
 *   shs[r], raw scaling[r] = then w[0] + (1 - w[0]) now      - the reference indexes history_weight[0]: the weight of
 *                                                               ROW 0 scales the features and the scaling of every row
 *   rotation[r] = slerp(normalize(rot_then[r]), normalize(rot[r]), 1 - w[r])  (SLAM/utils.py:593-651: lerp where
 *                 |dot| > 0.9995 or NaN), stored as the raw rotation;  opacity is left alone.
 * rows = the rows the snapshot covers (the trainable range); every pointer starts at its first row.  max_weight <= 0: no-op. */
int rtgs_history_merge(float* xyz, float* shs, float* raw8, const float* then_xyz, const float* then_shs,
                       const float* then_raw8, const float* conf_then, const float* conf_now, int64_t rows,
                       float max_weight, void* stream);

/* rtgs_map_activate8_backward_rows followed by rtgs_fused_adam_rows on xyz[rows,3], shs[rows,48] and raw8[rows,8], as
 * ONE launch (the tail of rtgs_slam_map_step).  g_* are the persistent gradient rows of rtgs_raster_backward_rows,
 * row_state its state bytes; g_raw8 is written for state 1 (value) and state 2 (zero) rows.
 *   attach     (nullable) adds the attach regulariser's gradient to the selected rows (rtgs_attach_prepare must have run)
 *   confidence (nullable) float[rows]: += 1 for every row whose f_dc gradient is non-zero (mapper.py:454-456)
 *   skip_flag  (nullable) device word: non-zero = do nothing (an overflowed multi-GPU exchange, see rtgs_rows_overflow)
 *   refresh    (nullable) the caller's activated copies of raw8 (what rtgs_map_activate8_forward wrote): every row whose
 *              raw8 is stepped is re-activated in place, so they stay equal to a full activation pass of the new raw8
 *              (only rows that moved are touched - a few thousand of 1.2 M on a depth-complex map) */
typedef struct rtgs_activated {
  float *opacity, *scales, *rotations, *normal;           /* [rows,1] [rows,3] [rows,4] [rows,3] */
} rtgs_activated;
int rtgs_map_tail_rows(float* xyz, float* shs, float* raw8, const float* g_opacity, const float* g_scales,
                       const float* g_rotations, const float* g_normal, const float* g_xyz, const float* g_shs,
                       float* g_raw8, const uint8_t* row_state, float* m_xyz, float* v_xyz, float* m_shs, float* v_shs,
                       float* m_raw8, float* v_raw8, const float* lr_xyz, const float* lr_shs, const float* lr_raw8,
                       uint8_t* ever_xyz, uint8_t* ever_shs, uint8_t* ever_raw8, int64_t rows, int32_t step, float beta1,
                       float beta2, float eps, const rtgs_attach* attach, float* confidence, const uint32_t* skip_flag,
                       const rtgs_activated* refresh, void* stream);

/* Fused SLAM loss: the image terms of Mapping.loss_update (mapper.py:402-448) - value and BOTH image gradients
 * (dL/dC [3,H,W], dL/dD [1,H,W]), so the autograd graph of ~40 elementwise launches collapses into a few kernels.
 *   mask   = render_mask (uint8 [H,W]), or every pixel when NULL - and only then the SSIM term is live (:411-417)
 *   colour = mean over mask x 3 channels of |C - C_gt|                                                  (:421)
 *   depth  = mean over {depth_index != -1, D_gt > 0, (D - D_gt) < add_depth_thres, mask} of |D - D_gt|  (:423-431; the
 *            threshold is on the signed error as written there; an empty set contributes 0 where torch gives nan)
 *   ssim   = 1 - mean SSIM(C, C_gt), 11x11 Gaussian window, sigma 1.5, zero padding (utils/loss_utils.py:58-100)
 *   total  = depth_weight depth + color_weight colour + ssim_weight ssim
 * The normal term (normal_weight, 0 in every shipped config) is not part of this kernel: rtgs_slam_normal_loss.
 * loss_out4: device float[4] = {total, colour, depth, ssim}.  scratch: rtgs_slam_loss_scratch_bytes(H, W, with_ssim). */
typedef struct rtgs_loss_cfg {
  float color_weight, depth_weight, ssim_weight, add_depth_thres;
  const uint8_t* render_mask;
  int32_t sums_zeroed;       /* non-zero: the caller guarantees ((float*)scratch)[0..7] are zero on entry (no memset launch) */
} rtgs_loss_cfg;
size_t rtgs_slam_loss_scratch_bytes(int32_t H, int32_t W, int32_t with_ssim);
int rtgs_slam_loss(const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                   const float* gt_depth, int32_t H, int32_t W, const rtgs_loss_cfg* cfg, void* scratch,
                   float* loss_out4, float* g_color, float* g_depth, void* stream);
/* The two halves of rtgs_slam_loss.  _sums leaves {sum |dC|, sum |dD|, #valid-depth, #mask, sum SSIM} in
 * ((float*)scratch)[0..4]; _grads turns them into the loss values and both image gradients.  A multi-GPU caller that
 * splits ONE view into tile bands all-reduces those five floats in between (the normalisers count pixels of the whole
 * image); such a caller must pass a render mask (the SSIM window crosses band boundaries). */
int rtgs_slam_loss_sums(const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                        const float* gt_depth, int32_t H, int32_t W, const rtgs_loss_cfg* cfg, void* scratch, void* stream);
int rtgs_slam_loss_grads(const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                         const float* gt_depth, int32_t H, int32_t W, const rtgs_loss_cfg* cfg, void* scratch,
                         float* loss_out4, float* g_color, float* g_depth, void* stream);

/* One map-optimisation iteration as a single call (the body of local_optimize's inner loop, mapper.py:176-205, with
 * loss_update's image terms, attach regulariser, Adam step and confidence increment, mapper.py:371-456):
 *   raw8 activation (skipped when activated_valid) -> rtgs_raster_forward -> rtgs_slam_loss -> rtgs_raster_backward_rows ->
 *   rtgs_map_tail_rows (activation backward + attach gradient + Adam on xyz[P,3], shs[P,48], raw8[P,8] + confidence).
 * Exactly the sequence the entry points above perform when called one by one (parameters are updated in place, the
 * losses are left in loss4); it exists because the host cost of issuing ~40 launches through an autograd graph
 * exceeds their GPU time on a large map.  Every pointer is a caller-owned DEVICE buffer; the gradient arena (d_*,
 * grad_scratch, row_state) and the Adam state (m_*, v_*, ever_*) persist between calls and follow the
 * rtgs_raster_backward_rows / rtgs_fused_adam_rows contracts.  The three resize callbacks are the ones
 * rtgs_raster_forward takes and must ALSO answer a request of size 0 with the buffer of their last request. */
typedef struct rtgs_map_step_args {
  const rtgs_raster_settings* settings;
  int32_t P, sh_coeffs;                                   /* sh_coeffs must be 16 */
  float *xyz, *shs, *raw8;                                /* parameters, updated in place */
  const int32_t* tile_mask;
  const float *gt_color, *gt_depth;                       /* [3,H,W], [1,H,W] */
  rtgs_loss_cfg loss;                                     /* weights, depth gate, render mask */
  void* loss_scratch;                                     /* rtgs_slam_loss_scratch_bytes(H, W, render_mask == NULL) */
  float *opacity, *scales, *rotations, *normal;           /* activated values: [P,1] [P,3] [P,4] [P,3] */
  float *out_color, *out_depth;                           /* the 7 outputs of the rasterizer (+ radii) */
  int32_t *out_color_index, *out_depth_index;
  float *out_color_weight, *out_depth_weight, *out_T;
  int32_t* out_radii;
  float *dL_dcolor, *dL_ddepth, *loss4;                   /* loss4: {total, colour, depth, ssim} */
  float *d_xyz, *d_opacity, *d_shs, *d_scales, *d_rotations, *d_normal, *d_raw8;
  void* grad_scratch;                                     /* rtgs_raster_backward_scratch_bytes(P), zero-initialised */
  uint8_t* row_state;                                     /* [P], zero-initialised */
  float *m_xyz, *v_xyz, *m_shs, *v_shs, *m_raw8, *v_raw8;
  const float *lr_xyz, *lr_shs, *lr_raw8;                 /* per-column learning rates: [3] [48] [8] */
  uint8_t *ever_xyz, *ever_shs, *ever_raw8;               /* [P] each, zero-initialised */
  int32_t step;                                           /* Adam step count, starts at 1 */
  float beta1, beta2, eps;
  const rtgs_attach* attach;                              /* nullable */
  float* confidence;                                      /* nullable, float[P] */
  int32_t activated_valid;                                /* non-zero: opacity / scales / rotations / normal already hold
                                                             the activation of raw8 (a previous call of this function left
                                                             them so: its tail re-activates the rows it steps) - the full
                                                             activation pass is skipped.  0 after any other change of raw8. */
  rtgs_resize_fn geom_resize; void* geom_user;
  rtgs_resize_fn binning_resize; void* binning_user;
  rtgs_resize_fn image_resize; void* image_user;
  /* The normal term of loss_update (mapper.py:433-442; normal_weight is 0 in every shipped config): when
   * normal_weight > 0 and gt_normal != NULL, total += normal_weight * mean over {render mask & depth_index != -1 &
   * gt normal not all-zero} of 1 - cosine_similarity(normal of the pixel's depth owner, gt normal), and its gradient
   * is added to the owners' d_normal rows (rtgs_slam_normal_loss below) between the rasterizer backward and the tail. */
  float normal_weight;
  const float* gt_normal;                                 /* [H,W,3] world normals of the frame (image_input["normal_map"]) */
  /* The TRAINABLE rows [train_begin, train_end).  ONLY (0, 0) means every row; train_begin == train_end != 0 is an EMPTY
   * range (a fully frozen map: rendered, loss evaluated, nothing differentiated or stepped).  RTG-SLAM renders cat(unstable, stable) but only the
   * unstable Gaussians are parameters of the optimisation (mapper.py:143-156 parametrizes self.pointcloud only,
   * :1026-1108 concatenates the stable rows without requires_grad): rows outside the range are rendered, never
   * differentiated (no gradient slot, no SplatGrad record, row_state stays 0), never stepped.  With a range the Adam
   * state m_* / v_* / ever_*, attach->init_* and confidence cover ONLY those rows: element 0 belongs to row train_begin.
   * Everything else (parameters, activated arrays, gradient rows, row_state) is indexed by the row itself. */
  int32_t train_begin, train_end;
  /* The per-Gaussian tail.  tail_mode 0 (default): ONE fused kernel over the live rows - slot sums, chain rule,
   * activation backward, attach gradient, Adam and re-activation with the SplatGrad record and the gradient rows held in
   * registers (rtgs_map_fused_tail; the arena's gradient rows are then NOT produced: they stay zero, row_state 0) - whenever
   * the step can take it (rtgs_slam_map_step on its own, normal_weight == 0); 1: always the three-kernel form
   * (grad_reduce, preprocess_bwd, rtgs_map_tail_rows - what rtgs_slam_map_step_front + the multi-GPU exchange use).
   * live_counts (nullable, device uint32[2]): += {rows with gradient, rows stepped} of the fused tail. */
  int32_t tail_mode;
  uint32_t* live_counts;
} rtgs_map_step_args;
/* The normal term on its own: value added to loss4[0], gradient added to d_normal[owner] with the row marked live in
 * row_state (nullable: dense gradients) - a row that carried no gradient is all-zero by the arena's invariant, so marking
 * it keeps the invariant.  scratch: 2 floats.  skip_flag (nullable): non-zero = do nothing (speculative forward). */
int rtgs_slam_normal_loss(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                          const uint8_t* render_mask, int32_t H, int32_t W, float normal_weight, float* scratch2,
                          float* loss_out4, float* d_normal, uint8_t* row_state, const uint32_t* skip_flag, void* stream);
/* ... with a trainable range: owners outside [train_begin, train_end) count in the value, receive no gradient. */
int rtgs_slam_normal_loss_range(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                                const uint8_t* render_mask, int32_t H, int32_t W, float normal_weight, float* scratch2,
                                float* loss_out4, float* d_normal, uint8_t* row_state, const uint32_t* skip_flag,
                                int32_t train_begin, int32_t train_end, void* stream);
/* The two halves of rtgs_slam_normal_loss_range, for callers whose pixels are spread over several ranks (tile bands): the
 * sums {sum of 1 - cos, count} land in scratch2, the caller all-reduces the two floats (the reference's mean is over the
 * pixels of the WHOLE image, mapper.py:433-442), and the second half divides by the global count. */
int rtgs_slam_normal_loss_sums(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                               const uint8_t* render_mask, int32_t H, int32_t W, float* scratch2, const uint32_t* skip_flag,
                               void* stream);
int rtgs_slam_normal_loss_grads(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                                const uint8_t* render_mask, int32_t H, int32_t W, float normal_weight, const float* sums2,
                                float* loss_out4, float* d_normal, uint8_t* row_state, const uint32_t* skip_flag,
                                int32_t train_begin, int32_t train_end, void* stream);
int rtgs_slam_map_step(const rtgs_map_step_args* args, int64_t* num_rendered_host, void* stream);
/* A HIP stream whose CU mask leaves the last `reserve_cus` compute units of the current device unused - for the MAPPER of
 * a tracker || mapper pipeline: the tracker's short dependent kernels then always find free wave slots.
 * NULL on failure or when reserve_cus is not in (0, #CUs).  Destroy with rtgs_stream_destroy. */
void* rtgs_stream_create_reserving(int32_t reserve_cus);
void rtgs_stream_destroy(void* stream);
/* The fused tail on its own (after the WALK of a backward, rtgs_raster_backward_walk_ctx): geom_buffer / image_buffer of
 * that forward, spec_fail = rtgs_raster_spec_fail_ptr_ctx() or NULL. */
int rtgs_map_fused_tail(const rtgs_raster_settings* settings, const rtgs_map_step_args* args, void* geom_buffer,
                        const void* image_buffer, const uint32_t* spec_fail, uint32_t* live_counts2, void* stream);
/* ... with the number of Gaussians the forward listed for binning (rtgs_raster_last_listed_ctx; 0 = unknown): sizes the
 * kernel's row chunks - few live rows: few large chunks; many: many small ones.  RTGS_FUSED_CHUNK=512|1024|2048 forces. */
int rtgs_map_fused_tail_hint(const rtgs_raster_settings* settings, const rtgs_map_step_args* args, void* geom_buffer,
                             const void* image_buffer, const uint32_t* spec_fail, uint32_t* live_counts2,
                             uint32_t listed_hint, void* stream);
/* Gaussians the last VERIFIED speculative forward on the context listed for binning (near-slice work list, or the list of
 * the visible ones when the slice was declined); 0 when unknown. */
uint32_t rtgs_raster_last_listed_ctx(rtgs_ctx* ctx);
/* Byte offsets of what a backward's walk leaves for a fused consumer: [0] clamp flags (u8[P], geometry buffer), [1] first
 * gradient slot per Gaussian (u32[P], geometry), [2] slots taken per Gaussian (u32[P], geometry), [3] BwdInfo (image
 * buffer), [4] touched bytes (u8[P], inside grad_scratch, behind the P SplatGrad records). */
int rtgs_raster_backward_buffers(int32_t P, int32_t image_height, int32_t image_width, size_t* out8_host);
/* The geometry / binning / image buffers the most recent forward on the context obtained from its resize callbacks
 * (out3_host[0..2]) - for a native caller that enqueues forward and backward back to back and would otherwise have to
 * call back into its allocator (a Python callback costs ~5 us of GPU idle time at each of the step's two hand-overs). */
int rtgs_raster_last_buffers_ctx(rtgs_ctx* ctx, void** out3_host);
/* One-shot: eight 32-bit words the NEXT forward on the context clears inside its blend kernel (stream-ordered before
 * anything enqueued after the forward) - rtgs_slam_map_step clears its loss sums this way instead of with a memset. */
void rtgs_raster_set_aux_zero_ctx(rtgs_ctx* ctx, void* eight_words);
/* The same call without its last stage (rtgs_map_tail_rows): the gradient rows of this rank's view are in the arena,
 * nothing has been stepped.  Multi-GPU callers exchange the rows (below) before they run the tail. */
int rtgs_slam_map_step_front(const rtgs_map_step_args* args, int64_t* num_rendered_host, void* stream);
int rtgs_slam_map_step_ctx(rtgs_ctx* ctx, const rtgs_map_step_args* args, int64_t* num_rendered_host, void* stream);
int rtgs_slam_map_step_front_ctx(rtgs_ctx* ctx, const rtgs_map_step_args* args, int64_t* num_rendered_host, void* stream);

/* Sparse gradient exchange for multi-GPU optimisation: only rows that received gradient travel, in fixed-capacity lists
 * whose length rides in band - no host synchronisation before the collective.
 *   list = [1 + capacity] rows x 64 float words.  Row 0: [0] number of rows the sender had (uint32 bits; > capacity =
 *   overflow, the list is then incomplete), [1] capacity.  Data row: [0] Gaussian id (uint32 bits), 1..3 d_xyz, 4..51
 *   d_shs, 52 d_opacity, 53..55 d_scales, 56..59 d_rotations, 60..62 d_normal.
 * rtgs_rows_pack compacts the state-1 rows of a row-state arena into such a list (count_scratch: device uint32).
 * rtgs_rows_overflow reads the headers of `world` gathered lists (laid out back to back) and writes
 * flag_and_counts[0] = 1 if any sender overflowed, [1 + r] = rank r's count.  rtgs_rows_apply writes a list back into an
 * arena - mode 0 zeroes the listed rows, mode 1 adds them and marks the rows state 1 - and does nothing when *skip_flag
 * is non-zero (pass the overflow flag; NULL = never skip).  A rank zeroes its own rows, then adds the lists of ranks
 * 0..W-1 in that order - every replica sums in the same order - and runs rtgs_map_tail_rows with the same skip_flag.
 * After an overflow nothing was changed on any rank; the host enlarges the capacity and repeats the exchange. */
int rtgs_rows_pack(const uint8_t* row_state, int32_t P, float* d_xyz, float* d_shs, float* d_opacity, float* d_scales,
                   float* d_rotations, float* d_normal, float* out_list, int32_t capacity, uint32_t* count_scratch,
                   void* stream);
int rtgs_rows_overflow(const float* gathered_lists, int32_t world, int32_t capacity, uint32_t* flag_and_counts, void* stream);
int rtgs_rows_apply(const float* list, int32_t capacity, int32_t mode, float* d_xyz, float* d_shs, float* d_opacity,
                    float* d_scales, float* d_rotations, float* d_normal, uint8_t* row_state, const uint32_t* skip_flag,
                    void* stream);

/* sizeof of the two structs above, for bindings that mirror them (rtg_slam_amd/_lib.py checks both at load time). */
size_t rtgs_map_step_args_size(void);
size_t rtgs_raster_settings_size(void);

const char* rtgs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RTGS_RASTER_H */
