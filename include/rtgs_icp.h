/*
 * rtgs_icp.h - C ABI of the MI355X (gfx950) projective point-to-plane ICP tracker.
 *
 * The reference has no native boundary here: SLAM/icp.py is ~25 PyTorch launches per
 * Gauss-Newton iteration plus a GPU->CPU->GPU round trip for the 6x6 inverse
 * (/root/reference/SLAM/icp.py:313-325).  These entry points replace, function by function:
 *
 *   rtgs_icp_build_pyramids  <- ImagePyramids (icp.py:337-355) + build_vertex_pyramid /
 *                               compute_vertex_map (SLAM/utils.py:511-521, :65-75) +
 *                               build_normal_pyramid / compute_normal_map (SLAM/utils.py:523-527, :100-122)
 *   rtgs_icp_step            <- ICP.compute_residuals_jacobian + compute_jtj + compute_jtr
 *                               (icp.py:52-119), one evaluation, returns the normal equations
 *   rtgs_icp_track           <- the level loop of IcpTracker.predict_pose (icp.py:428-451):
 *                               per level `iters` x { residuals/Jacobian/6x6 reduction, damped
 *                               solve (lev_mar_H icp.py:248-256, least_square_solve :328-334),
 *                               SE(3) exp update (exp_se3 :271-310) } with the pose resident on
 *                               the device and ZERO host synchronisation, then point2plane_loss
 *                               (icp.py:7-13) at full resolution
 *   rtgs_icp_fill_model_depth<- IcpTracker.update_last_status (icp.py:397-415)
 *
 * Conventions as in rtgs_raster.h: device pointers unless named *_host, dense row-major
 * float32, enqueue on `stream`, 0 = success / negative = error, nothing throws.
 * Maps are [H,W,3] (vertex, normal) or [H,W] (depth), exactly the reference's layouts.
 */
#ifndef RTGS_ICP_H
#define RTGS_ICP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTGS_ICP_MAX_LEVELS 8

/* Level l of an L-level pyramid has size (H >> (L-1-l), W >> (L-1-l)) and intrinsics
 * K * 2^-(L-1-l) with K[2][2] = 1 (SLAM/utils.py:515-520): level 0 is the coarsest. */

/* depth[H,W] -> per-level vertex[l] / normal[l] maps (caller-allocated, [H_l,W_l,3]).
 * K: device float[9] row-major intrinsics of the full-resolution image.
 * scratch: device, >= rtgs_icp_scratch_bytes() bytes. */
int rtgs_icp_build_pyramids(const float* depth, int32_t H, int32_t W, const float* K, int32_t levels,
                            float* const* vertex_out_host, float* const* normal_out_host,
                            void* scratch, void* stream);

/* The tracker is a chain of ~20 short DEPENDENT launches; beside a mapper on another stream every launch costs
 * dispatch latency, so the small memsets of the plain calls are avoidable launches on the frame's critical path:
 * rtgs_icp_scratch_init arms a scratch ONCE (zero; both min / max sets to their neutral value); after that
 *   rtgs_icp_build_pyramids_ex(flags = RTGS_ICP_PYR_SCRATCH_READY [| RTGS_ICP_PYR_SECOND_SET]) issues no memset - the
 *     build uses one min / max set and re-arms the other for the NEXT build: the caller alternates SECOND_SET from call
 *     to call on the same scratch;
 *   rtgs_icp_track(flags | RTGS_ICP_FLAG_SCRATCH_READY) issues no memset either (launch-per-iteration form).
 * Results are bit-identical to the plain calls. */
int rtgs_icp_scratch_init(void* scratch, void* stream);
#define RTGS_ICP_PYR_SCRATCH_READY 1
#define RTGS_ICP_PYR_SECOND_SET 2
int rtgs_icp_build_pyramids_ex(const float* depth, int32_t H, int32_t W, const float* K, int32_t levels,
                               float* const* vertex_out_host, float* const* normal_out_host,
                               void* scratch, int32_t flags, void* stream);

/* One evaluation of the normal equations at `pose` (device float[16], row-major 4x4, maps
 * source-frame points into the target frame):
 *   JtJ_out[36], Jtr_out[6], nvalid_out[1] (float, number of valid correspondences) on device.
 * K: device float[9] of THIS level (already down-scaled). */
int rtgs_icp_step(const float* vertex_src, const float* normal_src, const float* vertex_tgt,
                  const float* normal_tgt, int32_t H, int32_t W, const float* K, const float* pose,
                  float distance_threshold, float cos_normal_threshold,
                  float* JtJ_out, float* Jtr_out, float* nvalid_out, void* scratch, void* stream);

typedef struct rtgs_icp_level {
  int32_t H, W;
  float downscale;            /* icp_downscales[l], multiplies the full-resolution K            */
  int32_t iters;              /* icp_downscale_iters[l]                                         */
  const float* vertex_src;    /* current frame t1 (icp.py:438-441 argument order)               */
  const float* normal_src;
  const float* vertex_tgt;    /* previous frame / model t0                                      */
  const float* normal_tgt;
} rtgs_icp_level;

/* Runs the whole multi-level Gauss-Newton loop on the device, pose resident there, no host synchronisation:
 * one launch per Gauss-Newton iteration by default, or - flags & RTGS_ICP_FLAG_PERSISTENT - ONE persistent kernel (a
 * workgroup per CU, a grid barrier per iteration, every workgroup takes the same step on its own copy of the pose).
 * Same arithmetic; the persistent form is ~8 % faster with the device to itself, the default form coexists better with
 * kernels of other streams.  Environment override: RTGS_ICP_PERSISTENT=0/1.
 *   pose_inout: device float[16], initial guess in, estimate out (pose_t1_t0)
 *   stats_out : device float[4] = { valid_ratio of the last iteration (icp.py:46-47),
 *               point2plane loss at the last level (icp.py:443-447), number of solves that
 *               hit a non-SPD system (pose left unchanged for that iteration),
 *               1 if the persistent kernel gave up waiting at a grid barrier (pose_inout is then unchanged; the
 *               wait is bounded so that a scheduling pathology cannot hang the device), else 0 } */
int rtgs_icp_track(const rtgs_icp_level* levels_host, int32_t n_levels, const float* K,
                   float distance_threshold, float cos_normal_threshold, float damping,
                   float* pose_inout, float* stats_out, void* scratch, int32_t flags, void* stream);
#define RTGS_ICP_FLAG_PERSISTENT 1
#define RTGS_ICP_FLAG_SCRATCH_READY 4   /* scratch armed by rtgs_icp_scratch_init and only ever used by these entry points */
#define RTGS_ICP_FLAG_FROM_IDENTITY 8   /* the initial guess is the identity: pose_inout need not be initialised (saves the
                                           caller two fill / copy launches) */
#define RTGS_ICP_FLAG_F32_SOLVE 16      /* the Gauss-Newton update in FLOAT32 in the reference's order of operations (icp.py:248-334:
                                           lev_mar_H, LU inverse as torch.inverse / LAPACK does it, -invH @ Rhs, exp_se3, exp @ pose)
                                           instead of the float64 Cholesky; launch-per-iteration chain only.  A measurement aid: it
                                           separates "the solve" from "gate-flip amplification" in the distance to the reference's
                                           float32 answer on noisy depth */
#define RTGS_ICP_FLAG_CLUSTER 2     /* every level but the finest in one launch on a cluster of workgroups of ONE XCD (an
                                       in-XCD barrier per Gauss-Newton iteration), the finest level one launch per iteration */

/* In-place model-depth hole filling (icp.py:397-415): render_depth[H,W] takes frame_depth where
 * |render - frame| > dist_thr, or render == 0, or 1 - cos(render_normal, frame_normal) >
 * normal_thr, and frame_depth > 0. */
int rtgs_icp_fill_model_depth(float* render_depth, const float* frame_depth, const float* render_normal,
                              const float* frame_normal, int32_t H, int32_t W, float dist_thr,
                              float normal_thr, void* stream);

size_t rtgs_icp_scratch_bytes(void);

#ifdef __cplusplus
}
#endif
#endif /* RTGS_ICP_H */
