/*
 * rtgs_slam.h - C ABI of the callers and data producers either side of RTG-SLAM's hot path (R9, f-1, f-3),
 * MI355X (gfx950) build.  Same conventions as rtgs_raster.h: device pointers, dense row-major float32 / int32 / uint8,
 * inputs borrowed, enqueued on `stream` (hipStream_t as void*), 0 = success / negative = error.
 *
 * What each entry point replaces in the reference:
 *   tile-mask producers      SLAM/utils.py:681-734 (pixelmask2tilemask, transmission2tilemask, colorerror2tilemask)
 *                            and the render-range step of mapper.py:471-508
 *   rtgs_knn3                simple_knn._C.distCUDA2 (un-vendored CUDA submodule; call site gaussian_pointcloud.py:376)
 *   rtgs_knn3_query          pytorch3d.ops.knn_points as Mapping.temp_points_filter uses it (mapper.py:803-827), and the
 *                            new-point rows of update_geometry's distCUDA2 (gaussian_pointcloud.py:366-381)
 *   rtgs_accumulate_error    cuda_utils._C.accumulate_gaussian_error (un-vendored; call site mapper.py:541-565)
 *   frame preprocessing      tracker.py:97-159 -> SLAM/utils.py:65-139 (vertex / normal / confidence maps),
 *                            SLAM/utils.py:550-589 (bilateral filter), SLAM/utils.py:141-183 (sample_pixels' mask)
 *   rtgs_gather_rows3        the normal-map gather of Renderer.render, SLAM/render.py:130-133
 */
#ifndef RTGS_SLAM_H
#define RTGS_SLAM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tile masks (16 x 16 tiles unless `stride` says otherwise; the grid is ceil(H/stride) x ceil(W/stride), pixels
 *      beyond the image count as 0 exactly as the reference's zero padding) -------------------------------------- */

/* Per-tile SUM of a pixel map: src_kind 0 = uint8 mask (non-zero = 1), 1 = float32.  tile_sum[gy*gx] float32. */
int rtgs_tile_sum(const void* src, int32_t src_kind, int32_t H, int32_t W, int32_t stride, float* tile_sum, void* stream);

/* transmission2tilemask: tile on iff mean(mask) > ratio, i.e. sum > ratio * stride^2 (mask sums are exact integers). */
int rtgs_transmission2tilemask(const uint8_t* pixelmask, int32_t H, int32_t W, int32_t stride, float ratio,
                               int32_t* tile_mask, float* tile_sum_scratch, void* stream);
/* pixelmask2tilemask: tile on iff any pixel is. */
int rtgs_pixelmask2tilemask(const uint8_t* pixelmask, int32_t H, int32_t W, int32_t stride, int32_t* tile_mask,
                            float* tile_sum_scratch, void* stream);
/* colorerror2tilemask: the k = (int)(tiles * top_ratio) tiles with the largest mean error are on (ties at the k-th value:
 * lower tile index first).  Any number of tiles.  The reference computes int(numel * top_ratio) in double
 * (SLAM/utils.py:708-734); top_ratio crosses this ABI as float32, where e.g. 0.7 is 0.69999999 and k can come out one
 * short - a binding that holds the ratio in double computes k itself and calls the _k form (the Python host does). */
int rtgs_colorerror2tilemask(const float* color_error, int32_t H, int32_t W, int32_t stride, float top_ratio,
                             int32_t* tile_mask, float* tile_sum_scratch, void* stream);
int rtgs_colorerror2tilemask_k(const float* color_error, int32_t H, int32_t W, int32_t stride, int32_t top_k,
                               int32_t* tile_mask, float* tile_sum_scratch, void* stream);
/* The render-range step of mapper.py:471-508 in one call, straight from the rasterizer's T_map:
 *   render_mask = (T_map != 1)  [uint8, H*W],  tile_mask = transmission2tilemask(render_mask, 16, ratio),
 *   *count_out = number of set pixels (device uint32; render_ratio = count / pixels). */
int rtgs_render_range(const float* T_map, int32_t H, int32_t W, float ratio, uint8_t* render_mask, int32_t* tile_mask,
                      uint32_t* count_out, float* tile_sum_scratch, void* stream);

/* ---- simple_knn.distCUDA2 ------------------------------------------------------------------------------------- */
/* For each of the N points (float32 [N,3]): its three nearest OTHER points.  mean_dist2[N] = mean of the three squared
 * distances (d^2 = dx*dx + dy*dy + dz*dz in float32), idx[N,3] int32 ascending by distance, dist2[N,3] (may be NULL).
 * Exact (Morton order + bounding-box pruning, no approximation).  Fewer than three other points: FLT_MAX / -1.
 * scratch: rtgs_knn3_scratch_bytes(N) bytes. */
size_t rtgs_knn3_scratch_bytes(int32_t N);
int rtgs_knn3(const float* points, int32_t N, float* mean_dist2, int32_t* idx, float* dist2, void* scratch, void* stream);
/* Cross-set form: for each of the Nq QUERY points its three nearest REFERENCE points (idx[Nq,3] into ref_points, ascending
 * by distance; dist2[Nq,3], may be NULL; fewer than three references: -1 / FLT_MAX).  Exact, same search structure (built
 * over the references; queries are seeded by a binary search of their Morton code).  What it replaces in the reference:
 * pytorch3d.ops.knn_points(temp_xyz, exist_xyz, K=3) in Mapping.temp_points_filter (mapper.py:803-827), and - with
 * self_offset >= 0: query i IS reference self_offset + i and does not find itself - the `knn_indices[:points_num]` rows
 * of distCUDA2(total_xyz) in GaussianPointCloud.update_geometry (gaussian_pointcloud.py:366-381), which needs the
 * neighbours of the NEW points only.  self_offset < 0: the sets are unrelated.  ref_box6 (device float[6] = lo xyz, hi xyz;
 * NULL = none): references outside the OPEN box are ignored - bbox_filter (SLAM/utils.py:737-744), which both call sites
 * apply to the existing points before the search, without a compaction.  The queries are visited in Morton order (sorted
 * inside the call) so that the lanes of a wave open the same boxes.  scratch: rtgs_knn3_query_scratch_bytes(Nr, Nq). */
size_t rtgs_knn3_query_scratch_bytes(int32_t Nr, int32_t Nq);
int rtgs_knn3_query(const float* ref_points, int32_t Nr, const float* query_points, int32_t Nq, int32_t self_offset,
                    const float* ref_box6, int32_t* idx, float* dist2, void* scratch, void* stream);
/* The same search split by what changes between frames (round 6; Mapping.temp_to_optimize, i.e. update_geometry's neighbour
 * search over cat(new points, every existing Gaussian), gaussian_pointcloud.py:366-381).  build_ref leaves the search structure
 * of Nr >= 1 reference points in `built` (rtgs_knn3_built_bytes(Nr) bytes; it holds a COPY of the points: rebuild when they
 * change); query_built answers rtgs_knn3_query(self_offset = -1) against it (query_scratch: rtgs_knn3_query_built_scratch_bytes(Nq)).
 * dynamic_merge: for every query i its three nearest among (a) the three stable neighbours handed in (dist2_stable / idx_stable
 * [Nq,3], idx = stable row or -1), (b) the OTHER queries (query i is not its own neighbour) and (c) the Nu unstable points, the
 * last two compared directly; references outside the open box ref_box6 are ignored as above.  Result indices address
 * cat(queries, existing rows): query j -> j, stable row r -> Nq + r, unstable point u -> Nq + n_stable + u.  Same float32
 * distance expression as rtgs_knn3_query: the result equals the one-structure search up to the order of equidistant points. */
size_t rtgs_knn3_built_bytes(int32_t Nr);
size_t rtgs_knn3_query_built_scratch_bytes(int32_t Nq);
int rtgs_knn3_build_ref(const float* ref_points, int32_t Nr, void* built, void* stream);
int rtgs_knn3_query_built(const void* built, int32_t Nr, const float* query_points, int32_t Nq, const float* ref_box6, int32_t* idx,
                          float* dist2, void* query_scratch, void* stream);
int rtgs_knn3_dynamic_merge(const float* query_points, int32_t Nq, const float* unstable_points, int32_t Nu, int32_t n_stable,
                            const float* dist2_stable, const int32_t* idx_stable, const float* ref_box6, int32_t* idx, float* dist2,
                            void* stream);

/* ---- cuda_utils.accumulate_gaussian_error --------------------------------------------------------------------- */
/* FROZEN semantics (the CUDA source is absent; mapper.py:541-571 is the evidence): colour error goes to the Gaussian in
 * color_index, depth and normal error to the Gaussian in depth_index (-1 = nobody).  Outputs [P]: mean (mean != 0) or
 * sum of each error over the attributed pixels (0 where none), and outlier_count = attributed pixels over threshold.
 * scratch: 2 * P floats (pixel counts). */
int rtgs_accumulate_error(int32_t H, int32_t W, int32_t P, const float* color_err, const float* depth_err,
                          const float* normal_err, const int32_t* color_index, const int32_t* depth_index, float thr_c,
                          float thr_d, float thr_n, int32_t mean, float* g_color, float* g_depth, float* g_normal,
                          int32_t* outlier_count, float* scratch, void* stream);

/* ---- frame preprocessing --------------------------------------------------------------------------------------- */
/* bilateralFilter_torch(depth, radius, sigma_color, sigma_space) - SLAM/utils.py:550-589. */
int rtgs_bilateral_filter(const float* depth, int32_t H, int32_t W, int32_t radius, float sigma_color, float sigma_space,
                          float* out, void* stream);
/* The map part of Tracker.map_preprocess (tracker.py:114-131) after the optional filter: range mask, vertex / normal /
 * confidence maps, invalid-confidence mask zeroing all four.  K: device float[9].  Outputs: depth_out[H,W],
 * vertex_out[H,W,3], normal_out[H,W,3], conf_out[H,W], bad_out[H,W] uint8.  scratch: H*W*3 floats + 16 bytes. */
size_t rtgs_frame_preprocess_scratch_bytes(int32_t H, int32_t W);
int rtgs_frame_preprocess(const float* depth_in, int32_t H, int32_t W, const float* K, float min_depth, float max_depth,
                          float invalid_confidence_thresh, float* depth_out, float* vertex_out, float* normal_out,
                          float* conf_out, uint8_t* bad_out, void* scratch, void* stream);
/* Candidate pixels of sample_pixels (SLAM/utils.py:141-183): select_mask (NULL = all) minus pixels whose normal sums to
 * exactly 0, compacted IN INDEX ORDER: indices_out[*count_out] int32 flat pixel indices.  scratch: rtgs_compact_scratch_bytes. */
size_t rtgs_compact_scratch_bytes(int32_t n);
int rtgs_sample_candidates(const float* normal_map, const uint8_t* select_mask, int32_t H, int32_t W, int32_t* indices_out,
                           int32_t* count_out, uint8_t* flags_scratch, void* scratch, void* stream);

/* ---- Renderer.render's normal map (SLAM/render.py:130-133) ----------------------------------------------------- */
/* ---- per-frame mask / error producers of the mapper ------------------------------------------------------------- */
/* Mapping.temp_points_init (mapper.py:728-775) in one pass: transmission_mask = T > thr_T & depth > 0; error_mask =
 * ((|depth - render_depth| > thr_depth & depth > 0 & depth_index > -1) | (mean_c |frame - render colour| > thr_color &
 * depth > 0 & T < thr_T)) & ~transmission_mask; counts2[0..1] = set pixels of the two (device uint32, zeroed here).  Maps
 * are [H,W] (colours [3,H,W]) float32 / int32; masks uint8. */
int rtgs_add_masks(const float* T_map, const float* depth, const float* render_depth, const float* render_color_chw,
                   const float* frame_color_chw, const int32_t* depth_index, int32_t H, int32_t W, float thr_transmission,
                   float thr_depth, float thr_color, uint8_t* transmission_mask, uint8_t* error_mask, uint32_t* counts2,
                   void* stream);
/* Mapping.error_gaussians_remove (mapper.py:527-540): depth_error = |depth - render_depth|, 0 where the render lies behind the
 * frame, the frame has no depth or the pixel has no depth owner; color_error = sum_c |frame - render colour|, 0 where the
 * frame has no depth.  The inputs of rtgs_accumulate_error. */
int rtgs_frame_errors(const float* depth, const float* render_depth, const float* render_color_chw, const float* frame_color_chw,
                      const int32_t* depth_index, int32_t H, int32_t W, float* color_error, float* depth_error, void* stream);
/* Two bookkeeping passes of the mapper as single kernels (round 6).
 * error_counters - Mapping.error_gaussians_remove's strikes (mapper.py:541-565): for the first nf (stable) rows,
 * depth_counter += (g_depth > depth_strike_thr), color_counter += (g_color > color_strike_thr) (the caller passes twice the add
 * thresholds); delete_mask = depth_counter >= limit, release_mask = color_counter >= limit and not deleted; counts2 = their sums.
 * delete_mask - Mapping.gaussians_delete (mapper.py:298-335) for a cloud of n rows with activated scales [n,3]: radius
 * (sum - min) / 2 > 10 x the cloud's mean radius, or (add_tick != NULL) time_now - add_tick > window; count1 = set entries.  One
 * workgroup: meant for the unstable cloud (a few thousand rows). */
int rtgs_error_counters(int32_t nf, const float* g_color, const float* g_depth, float color_strike_thr, float depth_strike_thr,
                        int32_t* depth_counter, int32_t* color_counter, int32_t limit, uint8_t* delete_mask, uint8_t* release_mask,
                        uint32_t* counts2, void* stream);
int rtgs_delete_mask(int32_t n, const float* scales, const int32_t* add_tick, int32_t time_now, int32_t window, uint8_t* mask,
                     uint32_t* count1, void* stream);

/* The new Gaussians of a frame (round 6): what Mapping._new_points and the tail of Mapping.temp_to_optimize (this package's
 * mapping.py; the reference: GaussianPointCloud.add_empty_points / update_geometry, gaussian_pointcloud.py:305-405, compute_rot
 * SLAM/utils.py:216-221, mapper.py:886-899) spell as ~80 tensor operations, as two kernels with the same float32 operations.
 * gather_new_points: pick int64[n] (pixel indices into the [H*W,3] maps) -> xyz, UNIT normal (v / (|v| + 1e-8)), colour, rotation
 * quaternion (w,x,y,z) turning z onto the normal (identity_rot != 0: (1,0,0,0), the reference's branch for xyz_factor = 1,1,1).
 * (A pass of exactly THREE points must not come here: the reference's torch.cross without a dim then crosses along the batch.)
 * new_rows: candidate i with its three nearest neighbours (dist2 / idx [n,3] of rtgs_knn3_query over cat(candidates, existing);
 * idx < n names a candidate - radius 1e-6 -, idx >= n the existing Gaussian idx - n with activated scales exist_scales[., 3],
 * -1 no neighbour) -> packed59[n,59] raw rows (xyz | SH dc from the colour | zeros | raw opacity | log(scale_factor * s * factor)
 * | rotation) of EVERY candidate and valid[n] = 0 where the candidate lies inside three radii of a neighbour. */
int rtgs_gather_new_points(const int64_t* pick, int32_t n, const float* vertex_map, const float* normal_map, const float* color_map,
                           int32_t identity_rot, float* xyz, float* normal, float* color, float* rots, void* stream);
/* draw_new_points: SLAM/utils.py:171's `randperm(n_cand)[:k]` and gather_new_points in one launch.  cand int32[n_cand] = the
 * pixels sample_pixels may draw from (rtgs_sample_candidates); output i takes cand[perm(i)], perm a keyed bijection of
 * [0, n_cand) (balanced Feistel network + cycle walking; `key` = a fresh 64-bit word per pass from the caller's seeded
 * generator), so the k outputs are distinct.  pick_out int32[k] (optional) receives the drawn pixel indices.  k <= n_cand.
 * Not for k == 3 either.  filter_keep: Mapping.temp_points_filter's decision (mapper.py:812-826) behind the neighbour query:
 * keep[i] = 0 iff for one of the three neighbours idx[i,.] >= 0: sqrt(dist2) < ratio * radius(scales[idx]) (radius = (sum -
 * min) / 2 of the activated scales, gaussian_pointcloud.py:515-519; ratio 0.6).  bbox_pad: out6 = [min - pad | max + pad] of
 * n >= 1 points, the neighbour query's box (SLAM/utils.py:737-744), one single-workgroup launch. */
int rtgs_draw_new_points(const int32_t* cand, int32_t n_cand, int32_t k, uint64_t key, const float* vertex_map, const float* normal_map,
                         const float* color_map, int32_t identity_rot, float* xyz, float* normal, float* color, float* rots,
                         int32_t* pick_out, void* stream);
int rtgs_filter_keep(int32_t n, const float* dist2, const int32_t* idx, const float* scales, float ratio, uint8_t* keep, void* stream);
int rtgs_bbox_pad(int32_t n, const float* xyz, float pad, float* out6, void* stream);
/* The two ordered compactions of Mapping.temp_to_optimize as one single-workgroup launch each (the tensor form: nonzero - five
 * launches - and a gather per array).  compact_points: the candidates with keep[i] != 0 (the filter's survivors, mapper.py:826),
 * in order, into out_* (capacity n rows each); count_out int32[1] = how many.  append_valid_rows: the rows of packed59 [n,59]
 * (rtgs_new_rows) with valid[i] != 0, in order, split into the map's parameter blocks at xyz_dst [.,3] / shs_dst [.,48] /
 * raw8_dst [.,8] (pointers to the first free row; the caller guarantees n free rows), and for each of the n_aux <= 8 side arrays
 * (4-byte elements, one per row; host array of device pointers to the first free row) the 32-bit pattern aux_fill_bits[k];
 * count_out = how many rows were appended. */
int rtgs_compact_points(int32_t n, const uint8_t* keep, const float* xyz, const float* color, const float* opacity_raw,
                        const float* rots, float* out_xyz, float* out_color, float* out_opacity_raw, float* out_rots,
                        int32_t* count_out, void* stream);
int rtgs_append_valid_rows(int32_t n, const uint8_t* valid, const float* rows59, float* xyz_dst, float* shs_dst, float* raw8_dst,
                           int32_t n_aux, void* const* aux_dst, const uint32_t* aux_fill_bits, int32_t* count_out, void* stream);
int rtgs_new_rows(int32_t n, const float* xyz, const float* color, const float* opacity_raw, const float* rots, const float* dist2,
                  const int32_t* idx, const float* exist_scales, float min_radius, float max_radius, float scale_factor,
                  float factor_x, float factor_y, float factor_z, float* packed59, uint8_t* valid, void* stream);

/* Mapping.temp_points_attach (mapper.py:830-883): attach_out[i] = 1 iff point i projects (w2c16 row-major 4x4, pinhole fx fy
 * cx cy, truncation like Camera.get_uv, scene/cameras.py:161-168) inside the image onto a pixel whose stable colour index
 * is >= 0 and lies within max_plane_dist of that Gaussian's plane (stable_xyz / stable_normal rows). */
int rtgs_attach_test(const float* points, int32_t n, const float* w2c16, float fx, float fy, float cx, float cy, int32_t H,
                     int32_t W, const int32_t* stable_color_index, const float* stable_xyz, const float* stable_normal,
                     float max_plane_dist, uint8_t* attach_out, void* stream);
/* transform_map (SLAM/utils.py:56-63; tracker.py:283-288 builds vertex_map_w / normal_map_w with it): out[i] = T[:3,:3] in[i] +
 * T[:3,3] for n 3-vectors; transform16 = device float[16], row-major 4x4 (pass get_rot(c2w) - zero translation - for normals). */
int rtgs_transform_map(const float* map3, int64_t n, const float* transform16, float* out3, void* stream);
/* out[3,n]: out[:, p] = rows[index[p]] (rows float32 [N,3]) where index[p] >= 0, zeros elsewhere - the reference's two
 * boolean-mask indexings (two device-to-host syncs per render) as one kernel.  scatter = its backward: grad_rows[index[p]] +=
 * g[:, p] (accumulates; caller zeroes). */
int rtgs_gather_rows3(const float* rows, const int32_t* index, int32_t n, float* out, void* stream);
int rtgs_scatter_rows3(const float* g, const int32_t* index, int32_t n, float* grad_rows, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RTGS_SLAM_H */
