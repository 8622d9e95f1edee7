// Shared declarations of the gfx950 rasterizer kernels (internal; the public ABI is
// include/rtgs_raster.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

namespace rtgs {

constexpr int TILE = 16;
constexpr int BLOCK = TILE * TILE;   // 256 threads = 4 wave64, each wave a 16x4 pixel strip
constexpr int WAVE = 64;
#ifndef RTGS_BWD_BATCH
#define RTGS_BWD_BATCH 128
#endif
constexpr int BATCH = RTGS_BWD_BATCH;   // tile-list entries blend_bwd stages through LDS per round (<= BLOCK)

// One 64-byte record per Gaussian: everything blend_fwd / blend_bwd need, so that a list
// entry costs exactly one aligned 64-B gather.
struct __attribute__((aligned(16))) Splat {
  float u, v;          // pixel centre
  float ca, cb, cc;    // conic (inverse 2D covariance)
  float o;             // opacity
  float r, g, b;       // view-dependent colour, clamped at 0
  float nx, ny, nz;    // camera-space plane normal
  float pd;            // plane offset n_c . p_c
  float z;             // camera-space depth of the centre
  float hx, hy;        // half extents of the alpha >= 1/255 region (pixels, with margin); 0 = unknown
};
static_assert(sizeof(Splat) == 64, "Splat must be one 64-byte line");

// Per-Gaussian gradient record produced by blend_bwd, consumed by preprocess_bwd.
struct __attribute__((aligned(16))) SplatGrad {
  float du, dv;
  float dca, dcb, dcc;
  float dop;
  float dr, dg, db;
  float dnx, dny, dnz;
  float dpd;
  float pad0, pad1, pad2;
};
static_assert(sizeof(SplatGrad) == 64, "SplatGrad must be one 64-byte line");

struct RasterParams {
  int H, W, gx, gy, P, M, deg;
  float fx, fy, cx, cy, tanfovx, tanfovy, scale_modifier;
  float opaque_thr, depth_thr, normal_thr, color_sigma, T_thr;
  const float* view;     // [16] W2C transposed
  const float* campos;   // [3]
  const float* bg;       // [3]
  // Speculative forward (raster_api.hip): device word that is non-zero when the sizes the host guessed for this call
  // did not hold; kernels that would overrun a buffer or change persistent state return at once.  nullptr = not
  // speculative.
  const uint32_t* spec_fail;
};
__device__ __forceinline__ bool spec_failed(const uint32_t* f) { return f != nullptr && *f != 0u; }
// What bin_tilescan checks for a speculative forward (all pointers nullptr = not speculative): the instance total, the
// longest tile list and the gradient-slot total against the capacities the host allocated / launched for, and - when
// the host assumed the near slice would be declined again - that the cut the kernels derived is negative.
struct SpecCaps {
  uint32_t* fail;
  uint32_t R, longest, slots;
  const int32_t* cut;
};

// the totals step of a one-pass binning (raster_bin.hip: bin_finish)
struct BinFinish {
  const uint32_t* count; int ntiles;
  uint32_t* info; uint32_t* info_host; const uint32_t* slot_a; const uint32_t* slot_b;
  const uint32_t* listed;            // length of the work list (published for the host's next launch geometry)
  uint32_t seq; SpecCaps caps;
};

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct GeomLayout {
  size_t splats, tiles_touched, offsets, radii, clamped, sat, tile_count, cursor, info, block_counts, scan_temp, total;
  size_t scan_temp_bytes;
  // near-slice pass (raster_api.hip): [zero_begin, zero_end) is cleared by preprocess_fwd on every call
  size_t zero_begin, zero_end;
  size_t tile_count1, ranges1_bwd, slice_hist, slice_cover, slice_ctr;   // inside the zeroed span (tile_count is too)
  size_t zbin, cursor1, ranges1, mask2, block_counts1, bucket1, list1, uv, slice_ids;
  size_t vis_ids, block_counts_vis;   // list of every visible Gaussian + its per-workgroup tile counts (declined single pass)
  size_t slot_count;                 // [P] u32: slots taken per Gaussian (backward; zeroed by it)
  size_t slice_seg;                                                   // entries per tile segment of list1 / bucket1 (0: none)
  size_t slice_cap, slice_max_list;                                  // capacity of bucket1 / list1; of the work list
};

// Near-slice selection: a Gaussian belongs to the slice iff its depth bin (monotone in the f32 depth bits, 32 bins
// per octave from 0.2) is <= the cut every workgroup derives from the area histogram: the largest bin whose
// cumulative 3-sigma-rect instance count still fits `cap`.
constexpr int SLICE_BINS = 256;
constexpr int SLICE_MAX_LIST = 3072;        // longest near-slice tile list that is sorted (longer: tile left to pass 2)
constexpr float SLICE_MIN_COVER = 24.f;     // automatic mode: sum(radius^2) of the slice per pixel of the image, at least
constexpr float SLICE_MIN_RATIO = 2.f;      // automatic mode: instances of the whole map / instances of the slice, at least
struct SliceSel {
  int mode;                      // 0 = every Gaussian; 1 = near slice only; 2 = every Gaussian, exit if no tile is unfinished
  const uint8_t* zbin;           // [P] depth bin, 255 = not visible
  const uint32_t* hist;          // [2 * SLICE_BINS]: rect-instance count per bin, then Gaussian count per bin
  uint32_t cap;                  // instance budget of the slice
  uint32_t max_list;             // Gaussian budget of the slice (length of the work list)
  const uint32_t* ctr;           // ctr[0] = tiles the near slice left unfinished
  // mode 2 only: Gaussian-level reject against the unfinished-tile mask.  sat = summed-area table of that mask
  // ((gx+1) x (gy+1), launch_mask_sat), uv = projected centres of the geometry pre-pass.  A Gaussian whose tile rect
  // holds no unfinished tile is neither shaded nor enumerated (its Splat record may be unwritten - never touch it).
  const int32_t* sat;
  const float2* uv;
  // automatic mode (decide != 0): the workgroups also decide, from the histograms alone, whether the slice is worth
  // running at all; if not, the cut is -1 (empty slice, every tile goes to the second pass).  cover[bin] = sum of
  // radius^2 (pixels^2) of the bin's Gaussians: ~ the optical depth x pixels they can deposit.
  const unsigned long long* cover;   // [SLICE_BINS]
  uint32_t npix;                     // pixels of the tile grid
  int decide;
};
// number of unfinished tiles inside the tile rect [x0,x1) x [y0,y1)
__device__ __forceinline__ int sat_count(const int32_t* __restrict__ sat, int gx, int x0, int y0, int x1, int y1) {
  const int sw = gx + 1;
  return sat[y1 * sw + x1] - sat[y0 * sw + x1] - sat[y1 * sw + x0] + sat[y0 * sw + x0];
}
// blend_fwd role in the two-pass forward
struct SlicePass {
  int mode;                      // 0 = single pass; 1 = near slice (records which tiles finished); 2 = remaining tiles only
  const int32_t* user_mask;      // mode 1
  int32_t* mask2;                // mode 1: written; mode 2: read
  uint2* ranges_bwd;             // mode 1: range of a finished tile, (0,0) otherwise - what blend_bwd walks
  uint2* ranges_main;            // mode 1: the main pass's ranges, written empty (pass 2 overwrites them if it runs)
};
// Shading work list of the two-pass forward (preprocess_fwd_kernel<2>, bin_count / bin_scatter in slice mode):
// the near slice's Gaussian ids, appended in arbitrary order by slice_compact_kernel.
struct SliceList {
  const uint32_t* ids;           // nullptr = no list (every Gaussian, by index)
  const uint32_t* count;         // device word: number of ids
};
__device__ __forceinline__ uint32_t slice_bin_of(float z) {
  const int b = (int)(__float_as_uint(z) >> 18) - (int)(0x3E4CCCCDu >> 18);     // 0.2f
  return (uint32_t)(b < 0 ? 0 : (b > SLICE_BINS - 2 ? SLICE_BINS - 2 : b));
}
struct BinLayout {
  size_t keys_a, keys_b, vals_a, vals_b, sort_temp, total;
  size_t sort_temp_bytes;
  size_t slot_grads;                 // backward partial slots (see BwdInfo); 0 slots = not laid out
};
struct ImgLayout {
  size_t ranges, n_contrib, bwd_info, tile_mode, depth_pos, tile_last, tile_recs, tile_masks, depth_aux, total;
};
// What blend_fwd leaves for the entry-per-lane backward at a place that depends on the TILE alone (round 6).  The backward's
// chain "tile range -> list ids -> record gather -> block test" was three dependent memory round trips and ~150 VALU
// instructions per entry before the first walk step (a quarter of a wave's life: profiles/r05_bwd_stamps_*); the forward
// has every one of these values in registers when it stages a batch, so it stores them: the record of list position
// `pos < TILE_RECS` of tile t at recs[(t * TILE_RECS + pos) * 3], the 16 block bits beside it, and per PIXEL the two words
// of its depth owner the plane partials need.  The backward then issues every load of its prologue at once - one round trip.
// Positions >= TILE_RECS (lists walked deeper than 256 entries) take the gather path as before.
constexpr int TILE_RECS = 256;
struct TileCache {
  float4* recs;        // [tiles][TILE_RECS][3]: u v ca cb | cc o r g | b id - -
  uint16_t* masks;     // [tiles][TILE_RECS]: blocks_reached() of the entry on this tile
  float2* depth_aux;   // [H * W]: 1 / (n_c . r) of the pixel's depth owner, and the depth D = pd / (n_c . r) itself (0, 0: none)
};
// blend_fwd leaves one word per tile for its backward: bits 0..1 = the walk - 0 tile-uniform strip walk (the tile's 4x4
// blocks share its list), 1 row-granular walk (each block needs only a fraction of it), 2 entry-per-lane MFMA walk
// (raster_bwd_mfma.hip); bits 8.. hold the measured share in 1/1000.  See raster_bwd.hip.
#ifndef RTGS_ROWS_MAX_SHARE
#define RTGS_ROWS_MAX_SHARE 0.6f
#endif
constexpr float ROWS_MAX_SHARE = RTGS_ROWS_MAX_SHARE;

// What the backward needs to know about the forward that produced its buffers, left in the image buffer by the
// forward (device memory: the backward never reads it on the host).
//
// Gradient partials without global float atomics: every Gaussian owns a run of 64-byte slots, one per tile of its
// rect (gbase = exclusive scan of the rect areas); a tile that has gradient for the Gaussian takes the next free slot of
// the run (ONE integer atomic on the Gaussian's counter instead of nine float atomics on its record) and stores its
// partial there with plain stores; grad_reduce then sums the first `count` slots of every touched Gaussian.  The slot
// space is allocated with the binning buffer (its size is known at the forward's host sync); if it would be too large
// the backward falls back to float atomics on the SplatGrad records.
struct BwdInfo {
  SplatGrad* slot_grads;       // [slots]
  uint32_t slots;              // capacity
  uint32_t use_slots;          // 0 = accumulate into SplatGrad records with global atomics
};
struct BwdInfoInit { BwdInfo* dst; SplatGrad* slot_grads; uint32_t slots, use_slots; };   // written by a kernel that runs anyway
constexpr uint32_t SLOTS_MAX = 12u << 20;   // 12 Mi slots = 768 MiB of partials; above that: atomics

// Tile rectangle of a Gaussian: C (int) truncation then clamp - identical to
// floor-and-clamp on the clamped range.  ONE definition for preprocess, binning and the backward's slot index.
__device__ __forceinline__ void tile_rect_of(float u, float v, int radius, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
  const float r = (float)radius;
  x0 = min(gx, max(0, (int)((u - r) / (float)TILE)));
  y0 = min(gy, max(0, (int)((v - r) / (float)TILE)));
  x1 = min(gx, max(0, (int)((u + r + (float)(TILE - 1)) / (float)TILE)));
  y1 = min(gy, max(0, (int)((v + r + (float)(TILE - 1)) / (float)TILE)));
}

// Row-granular tile walks (blend_fwd / blend_bwd): which of the 16 4x4 pixel blocks of a tile (bit b = block
// (b & 3, b >> 2)) a splat can reach with alpha >= 1/255.  Pixel centres of block column i are tx0 + 4 i .. tx0 + 4 i + 3.
// Two conservative tests, both must pass:
//  * the bounding box of the alpha >= 1/255 ellipse (hx, hy of the Splat record: 1 % + 0.5 px margin);
//  * the ellipse itself: q(d) = ca dx^2 + 2 cb dx dy + cc dy^2 <= tau = 2 ln(255 o) somewhere on the block.  q is convex,
//    so its minimum over a rectangle that does not contain the centre sits on one of the two edges facing the centre:
//    with (ex, ey) the rectangle's point nearest the centre per axis, the candidates are the minimum along the line
//    dx = ex (dy = -cb ex / cc clamped to the rectangle) and along dy = ey.  1 % + 0.02 margin on tau covers the
//    rounding of the blend's own power / exp evaluation.  On both bench scenes the ellipse test removes a quarter of
//    the (block, entry) pairs the box test lets through (the blocks at the corners of the box).
// ONE definition: the backward must evaluate every (entry, pixel) pair the forward blended.
__device__ __forceinline__ uint32_t blocks_reached(float u, float v, float hx, float hy, float ca, float cb, float cc,
                                                   float o, float tx0, float ty0) {
  const float tau = 1.01f * fmaxf(2.f * __logf(255.f * fmaxf(o, 1e-12f)), 0.f) + 0.02f;
  const float kc = -cb / cc, ka = -cb / ca, twob = 2.f * cb;
  float ex[4], lox[4];
  uint32_t xm = 0, ym = 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float lo_x = tx0 + 4.f * (float)i, lo_y = ty0 + 4.f * (float)i;
    xm |= (!((u + hx < lo_x) | (u - hx > lo_x + 3.f))) ? (1u << i) : 0u;
    ym |= (!((v + hy < lo_y) | (v - hy > lo_y + 3.f))) ? (0x000fu << (4 * i)) : 0u;
    lox[i] = lo_x - u;
    ex[i] = __builtin_amdgcn_fmed3f(0.f, lox[i], lox[i] + 3.f);      // nearest offset to the centre inside the block, per axis
  }
  uint32_t em = 0;
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {                                       // not unrolled: the staging threads are short of registers
    const float loy = ty0 + 4.f * (float)j - v;
    const float ey = __builtin_amdgcn_fmed3f(0.f, loy, loy + 3.f);
    const float dxu = ka * ey, cey2 = cc * ey * ey, tbey = twob * ey;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float dy1 = __builtin_amdgcn_fmed3f(kc * ex[i], loy, loy + 3.f);
      const float q1 = fmaf(dy1, fmaf(cc, dy1, twob * ex[i]), ca * ex[i] * ex[i]);
      const float dx2 = __builtin_amdgcn_fmed3f(dxu, lox[i], lox[i] + 3.f);
      const float q2 = fmaf(dx2, fmaf(ca, dx2, tbey), cey2);
      em |= (fminf(q1, q2) <= tau) ? ((1u << i) << (4 * j)) : 0u;
    }
  }
  return (xm * 0x1111u) & ym & em;
}

// The same two tests for the four 8x8 QUADRANTS of a tile (bit q = quadrant (q & 1, q >> 1); pixel centres of quadrant
// column i are tx0 + 8 i .. tx0 + 8 i + 7): what the backward's MFMA walk compacts its per-wave sub-lists from.  A
// quadrant is the union of four blocks, so this mask covers every block blocks_reached() reports.
__device__ __forceinline__ uint32_t quads_reached(float u, float v, float hx, float hy, float ca, float cb, float cc,
                                                  float o, float tx0, float ty0) {
  const float tau = 1.01f * fmaxf(2.f * __logf(255.f * fmaxf(o, 1e-12f)), 0.f) + 0.02f;
  const float kc = -cb / cc, ka = -cb / ca, twob = 2.f * cb;
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const float loy = ty0 + 8.f * (float)j - v;
    const bool yin = !((hy < loy) | (-hy > loy + 7.f));
    const float ey = __builtin_amdgcn_fmed3f(0.f, loy, loy + 7.f);      // nearest offset to the centre inside the quadrant
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float lox = tx0 + 8.f * (float)i - u;
      const bool xin = !((hx < lox) | (-hx > lox + 7.f));
      const float ex = __builtin_amdgcn_fmed3f(0.f, lox, lox + 7.f);
      const float dy1 = __builtin_amdgcn_fmed3f(kc * ex, loy, loy + 7.f);
      const float q1 = fmaf(dy1, fmaf(cc, dy1, twob * ex), ca * ex * ex);
      const float dx2 = __builtin_amdgcn_fmed3f(ka * ey, lox, lox + 7.f);
      const float q2 = fmaf(dx2, fmaf(ca, dx2, twob * ey), cc * ey * ey);
      m |= (xin & yin & (fminf(q1, q2) <= tau)) ? (1u << (2 * j + i)) : 0u;
    }
  }
  return m;
}

// Pinned SH constants (utils/sh_utils.py:26-45 of the reference).
#define RTGS_SH_C0 0.28209479177387814f
#define RTGS_SH_C1 0.4886025119029199f
#define RTGS_SH_C2_0 1.0925484305920792f
#define RTGS_SH_C2_1 -1.0925484305920792f
#define RTGS_SH_C2_2 0.31539156525252005f
#define RTGS_SH_C2_3 -1.0925484305920792f
#define RTGS_SH_C2_4 0.5462742152960396f
#define RTGS_SH_C3_0 -0.5900435899266435f
#define RTGS_SH_C3_1 2.890611442640554f
#define RTGS_SH_C3_2 -0.4570457994644658f
#define RTGS_SH_C3_3 0.3731763325901154f
#define RTGS_SH_C3_4 -0.4570457994644658f
#define RTGS_SH_C3_5 1.445305721320277f
#define RTGS_SH_C3_6 -0.5900435899266435f

// max over the wave's 64 lanes, in every lane - row swaps and row rotations only: no ds_bpermute, hence none of its six
// per-lane address registers (which, loop-invariant, are hoisted out of the tile loop and then spilled)
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t x) {
  const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  x = max(r[0], r[1]);
  const auto q = __builtin_amdgcn_permlane16_swap(x, x, false, false);
  x = max(q[0], q[1]);
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x128, 0xf, 0xf, false));    // row_ror:8
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x124, 0xf, 0xf, false));    // row_ror:4
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x122, 0xf, 0xf, false));    // row_ror:2
  x = max(x, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x121, 0xf, 0xf, false));    // row_ror:1
  return x;
}

// The alpha of one (Gaussian, pixel) pair.  Written with explicit _rn intrinsics so the forward
// and the backward kernel evaluate bit-identical skip / stop decisions regardless of how the
// compiler contracts the surrounding code.
__device__ __forceinline__ float splat_power(float ca, float cb, float cc, float dx, float dy) {
  float q = __fmaf_rn(ca * dx, dx, (cc * dy) * dy);   // ca*dx*dx + cc*dy*dy
  return __fmaf_rn(-0.5f, q, -(cb * dx) * dy);
}

// exp(x) for x <= 0 through the hardware exp2 (v_exp_f32, ~1 ulp): |rel err| <~ 4e-7 for the
// powers that survive the alpha >= 1/255 test (x > -5.6).  Shared by forward and backward so
// both take identical skip decisions.
__device__ __forceinline__ float splat_exp(float x) {
  return __builtin_amdgcn_exp2f(__fmul_rn(x, 1.4426950408889634f));
}

// The forward's host sync: a kernel publishes up to seven payload words into pinned host memory, the host spins on the
// sequence word.  host[0..6] = payload, host[7] = sequence, host[8] = checksum of (sequence, payload).
//  * fence -> explicit drain -> relaxed store: ROCm 7.2 can drop the s_waitcnt after buffer_wbl2 of a release STORE when
//    it believes the wave's vmcnt scoreboard is empty (MI355X_MICROARCH.md, compiler hazard) - the flag would overtake
//    the payload;
//  * the checksum makes the host independent of the order in which the writes become visible to it (the words travel
//    as separate PCIe writes; relaxed ordering on that path is a platform setting): it accepts the payload only when
//    sequence AND checksum match what it reads, and re-reads otherwise (raster_api.hip: wait_published).
constexpr int HOST_WORDS = 16;
__host__ __device__ __forceinline__ uint32_t host_checksum(uint32_t seq, const uint32_t (&w)[7]) {
  uint32_t h = seq * 2654435761u;
#pragma unroll
  for (int k = 0; k < 7; ++k) h = (h ^ w[k]) * 16777619u;
  return h;
}
__device__ __forceinline__ void publish_to_host(uint32_t* host, const uint32_t (&w)[7], uint32_t seq) {
#pragma unroll
  for (int k = 0; k < 7; ++k) host[k] = w[k];
  host[8] = host_checksum(seq, w);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(&host[7], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Cut bin of the near slice, recomputed by every workgroup from the 256-bin histogram (needs BLOCK = 256 threads, one
// bin each, all of them calling): the largest bin whose cumulative instance count still fits sel.cap.
__device__ __forceinline__ int slice_cut(const SliceSel& sel) {
  __shared__ uint32_t s_wsum[2][BLOCK / 64];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  uint32_t incl = sel.hist[tid], incl_n = sel.hist[SLICE_BINS + tid];
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)incl, off), on = (uint32_t)__shfl_up((int)incl_n, off);
    if (lane >= off) { incl += o; incl_n += on; }
  }
  if (lane == 63) { s_wsum[0][w] = incl; s_wsum[1][w] = incl_n; }
  __syncthreads();
  unsigned long long cum = incl, cum_n = incl_n;
  for (int k = 0; k < w; ++k) { cum += s_wsum[0][k]; cum_n += s_wsum[1][k]; }
  // both sums are monotone in the bin index: a prefix of bins fits the two budgets
  const int n_ok = __syncthreads_count(cum <= (unsigned long long)sel.cap && cum_n <= (unsigned long long)sel.max_list);
  const int cut = n_ok - 1;
  if (!sel.decide || cut < 0) return cut;
  // Is the slice worth its fixed cost?  (a) it must be able to saturate the image: every pixel needs an optical depth
  // of -ln(T_threshold) ~ 9 before its walk stops, so the slice's Gaussians must carry a multiple of that per pixel
  // (a single-layer surface map fails here: the nearest bins cover only the nearest part of the surface);  (b) it
  // must leave most of the map behind it, or the second pass saves nothing.  Same inputs, same reduction order in
  // every workgroup and kernel that calls this -> the same decision everywhere.
  __shared__ float s_dec[3][BLOCK / 64];
  float cov = tid <= cut ? (float)sel.cover[tid] : 0.f;
  float ins = tid <= cut ? (float)sel.hist[tid] : 0.f;
  float tot = (float)sel.hist[tid];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { cov += __shfl_xor(cov, off); ins += __shfl_xor(ins, off); tot += __shfl_xor(tot, off); }
  if (lane == 0) { s_dec[0][w] = cov; s_dec[1][w] = ins; s_dec[2][w] = tot; }
  __syncthreads();
  cov = (s_dec[0][0] + s_dec[0][1]) + (s_dec[0][2] + s_dec[0][3]);
  ins = (s_dec[1][0] + s_dec[1][1]) + (s_dec[1][2] + s_dec[1][3]);
  tot = (s_dec[2][0] + s_dec[2][1]) + (s_dec[2][2] + s_dec[2][3]);
  const bool use = cov >= SLICE_MIN_COVER * (float)sel.npix && tot >= SLICE_MIN_RATIO * ins;
  return use ? cut : -1;
}

}  // namespace rtgs
