// C-ABI entry points of the rasterizer (include/rtgs_raster.h): argument checks, scratch-buffer
// carving, the rocPRIM scan / radix sort between the hand-written kernels, and launch order.
#include "../../include/rtgs_raster.h"
#include "../../include/rtgs_debug.h"
#include "raster_common.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <math.h>
#include <new>
#include <stdlib.h>
#include <string.h>

namespace rtgs {
void launch_mask_sat(const int32_t*, int, int, int32_t*, hipStream_t);
void launch_preprocess_fwd(const RasterParams&, const float*, const float*, const float*, const float*, const float*,
                           const float*, const int32_t*, Splat*, uint32_t*, int32_t*, uint8_t*, int32_t*, uint32_t*, int,
                           uint8_t*, hipStream_t);
void launch_emit_keys(const RasterParams&, const Splat*, const int32_t*, const uint32_t*, const int32_t*, uint64_t*,
                      uint32_t*, hipStream_t);
void launch_tile_ranges(int64_t, const uint64_t*, uint2*, hipStream_t);
void set_fwd_stamps(void* dev);
void set_bwd_debug(int bits);
void set_bwd_stamps(void* dev);
void launch_blend_fwd(const RasterParams&, const uint2*, const uint32_t*, const Splat*, float*, float*, int32_t*,
                      int32_t*, float*, float*, float*, uint32_t*, unsigned long long*, SlicePass, uint32_t*, uint32_t*, uint32_t*, int, uint32_t*,
                      TileCache, uint32_t, hipStream_t);
void launch_blend_bwd(const RasterParams&, const uint2*, const uint32_t*, const Splat*, const float*, const float*,
                      const uint32_t*, const int32_t*, const float*, const float*, const uint32_t*, uint32_t*,
                      const BwdInfo*, SplatGrad*, uint8_t*, const uint32_t*, int, uint32_t, uint32_t, hipStream_t);
void launch_blend_bwd_entry(const RasterParams&, const uint2*, const uint32_t*, const Splat*, const float*, const uint32_t*,
                            const int32_t*, const uint32_t*, const uint32_t*, const float*, const float*, const uint32_t*, uint32_t*,
                            const BwdInfo*, SplatGrad*, uint8_t*, const uint32_t*, uint32_t, uint32_t, TileCache, uint32_t, hipStream_t);
void launch_grad_reduce(int, const uint8_t*, const uint32_t*, uint32_t*, const BwdInfo*, SplatGrad*, const uint32_t*, hipStream_t);
void launch_preprocess_bwd(const RasterParams&, const float*, const float*, const float*, const float*, const float*,
                           const float*, const int32_t*, const uint8_t*, SplatGrad*, uint8_t*, uint8_t*, float*, float*,
                           float*, float*, float*, float*, hipStream_t);

size_t bin_lds_limit_tiles();
int bin_sort_capacity();
size_t bin_block_counts_bytes(int, int);
int launch_bin_count(const RasterParams&, const Splat*, const int32_t*, const int32_t*, uint32_t*, uint16_t*, SliceSel,
                     SliceList, size_t, hipStream_t);
size_t bin_slice_block_counts_bytes(int, size_t, int);
size_t bin_list_block_counts_bytes(int, int);
void launch_visible_compact(int, const uint8_t*, uint32_t*, uint32_t*, const uint32_t*, uint32_t*, uint32_t*, const uint32_t*, hipStream_t,
                            const SliceSel* = nullptr, int32_t* = nullptr, uint32_t* = nullptr);
void launch_slice_compact(int, SliceSel, uint32_t*, uint32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t*, hipStream_t);
void launch_slice_publish(int, const int32_t*, const int32_t*, const uint32_t*, uint32_t*, uint32_t*, uint32_t, uint32_t*, hipStream_t,
                          const uint32_t* seg_count = nullptr);
void launch_preprocess_cull(const RasterParams&, const float*, const float*, const float*, uint32_t*, int32_t*, int32_t*,
                            uint32_t*, int, uint8_t*, float2*, hipStream_t);
void launch_preprocess_shade(const RasterParams&, const float*, const float*, const float*, const float*, const float*,
                             const float*, Splat*, int32_t*, uint8_t*, float2*, SliceList, SliceSel, size_t, hipStream_t);
void launch_bin_tilescan(int, const uint32_t*, uint2*, uint32_t*, uint32_t*, uint32_t*, const uint32_t*, const uint32_t*,
                         const uint32_t*, const uint32_t*, uint32_t, SpecCaps, hipStream_t);
void launch_bin_scatter(const RasterParams&, const Splat*, const int32_t*, const int32_t*, const uint16_t*, uint32_t*,
                        unsigned long long*, SliceSel, SliceList, size_t, hipStream_t);
void launch_slice_hist(int, const uint8_t*, const uint32_t*, const int32_t*, uint32_t*, unsigned long long*, hipStream_t,
                       BwdInfoInit bi = BwdInfoInit{nullptr, nullptr, 0u, 0u});
void launch_bin_tilesort(int, uint32_t, const uint2*, const unsigned long long*, uint32_t*, const uint32_t*, hipStream_t,
                         const uint32_t* seg_count = nullptr, uint32_t seg = 0, uint2* ranges_out = nullptr,
                         const BinFinish* finish = nullptr);
void launch_bin_place(const RasterParams&, const Splat*, const int32_t*, const int32_t*, uint32_t*, unsigned long long*, uint32_t,
                      uint32_t*, SliceSel, SliceList, size_t, hipStream_t);

}  // namespace rtgs

// Everything a forward / backward remembers between calls lives in a context (include/rtgs_raster.h: rtgs_ctx).  The
// plain entry points use one process-wide default context; callers that render from several threads create one
// context per thread.
enum { EV_F0 = 0, EV_PRE, EV_SL_BIN, EV_SL_BLEND, EV_SCAN, EV_BIN0, EV_EMIT, EV_SORT, EV_RANGES, EV_BLEND0, EV_BLEND, EV_B0,
       EV_BWALK, EV_BBLEND, EV_BPRE, EV_N };
struct rtgs_ctx {
  int64_t stats[8] = {0};
  // Near-slice (occlusion) pass: 0 = off, 1 = always, 2 = automatic (large maps; the kernels themselves decide from
  // the depth histograms whether the slice is worth running - raster_common.h: slice_cut).
  int slice_mode = 2;
  int slice_budget = 384;
  int64_t slice_stats[4] = {0};     // used, near-slice instances, tiles finished, tiles left to pass 2
  unsigned long long* counters = nullptr;
  uint32_t last_listed = 0;         // Gaussians the last verified speculative forward listed for binning
  bool onepass = true;              // one-pass placement into per-tile segments where the layout provides them
  bool prof = false;                // optional per-stage HIP-event timing (bench.py's roofline leg)
  bool force_sort_path = false;     // testing aid: take the global radix-sort binning path
  int bwd_walk = 0;                 // 0 / 3 = MFMA walk (default); 1 = strip walk everywhere; 2 = row-granular walk everywhere; 4 = per-tile choice between those two
  bool ev_init = false;
  hipEvent_t ev[EV_N];
  bool ev_set[EV_N] = {false};
  uint32_t* info_host = nullptr;    // pinned words the kernels publish totals into (the forward's host sync)
  uint32_t seq = 0;
  // what the most recent forward on this context left for its backward (see backward_impl)
  const void* hint_geom = nullptr;
  bool hint_slice_lists = true, hint_main_lists = true;
  int hint_walk = -2;               // what that forward wrote into tile_mode (-1 per-tile choice, 0 strip, 1 rows, 2 MFMA)
  void *last_geom = nullptr, *last_bin = nullptr, *last_img = nullptr;   // buffers of the most recent forward (rtgs_raster_last_buffers_ctx)
  uint32_t* aux_zero = nullptr;     // eight words the NEXT forward's blend clears (one-shot; rtgs_raster_set_aux_zero_ctx)
  // Automatic near-slice mode: whether a call takes the slice is decided on the device from that call's histograms and
  // never depends on history.  Only HOW the host learns it does: after a call that declined, the next one asks for the
  // decision (one extra pinned-flag sync, ~10 us) before it launches the slice's kernels instead of launching them
  // blind over an empty work list (~50 us of empty launches on a surface map).
  bool ask_first = false;
  // ---- speculative forward (RTGS_FWD_SPECULATE; see rtgs_raster.h) ----
  // What the last verified forward on this context looked like: the shape the next speculative one assumes.
  struct Plan {
    bool valid = false;
    int32_t P = 0, H = 0, W = 0;
    int kind = -1;                  // 0 = single pass, 1 = near slice finished every tile, 2 = slice declined, single pass over the visible list
    int slice_mode = 0, slice_budget = 0;
    uint32_t R = 0, R1 = 0, longest = 0, slots = 0, n_fin = 0;
    // slowly decaying maxima of the three totals: the capacities of the next speculative call come from these, so that
    // an optimisation that alternates between the views of a window (mapper.py:176-183 picks a random frame per
    // iteration) does not fail its guess every time it returns to the larger view
    uint32_t R_hi = 0, longest_hi = 0, slots_hi = 0;
    void note(uint32_t r, uint32_t l, uint32_t s) {
      R = r; longest = l; slots = s;
      R_hi = r > R_hi - R_hi / 16 ? r : R_hi - R_hi / 16;
      longest_hi = l > longest_hi - longest_hi / 16 ? l : longest_hi - longest_hi / 16;
      slots_hi = s > slots_hi - slots_hi / 16 ? s : slots_hi - slots_hi / 16;
    }
  } plan;
  struct Spec {
    bool pending = false;           // a speculative forward awaits rtgs_raster_forward_verify
    int kind = -1;
    uint32_t seq = 0, capR = 0, capL = 0, capS = 0;
    const void* geom = nullptr;     // geometry buffer of that forward (the backward guards itself only for this one)
    const uint32_t* fail_dev = nullptr;
    void* stream = nullptr;
    int32_t ntiles = 0;
    int64_t G_total = 0, B_total = 0, I_total = 0;
  } spec;
  bool speculation = true;
  int64_t spec_stats[3] = {0, 0, 0};   // speculative forwards, failed ones, forwards that could not speculate
  // Forwards WITHOUT a backward (the plain renders of a SLAM frame: RTGS_FWD_NO_BACKWARD) keep a history of their own - the
  // Gaussian count changes from one to the next, only the image and the sort class of the longest tile list are assumed -
  // and check their guess before they return (forward_impl: `immediate`), so nothing is ever left pending.
  Plan plan_plain;                     // ... on large maps (the near slice is considered; kind 2 once it was declined)
  Plan plan_plain0;                    // ... on maps below the large-map size (kind 0: one pass over every Gaussian)
  bool plain_onepass = true;           // RTGS_PLAIN_ONEPASS=0: count / scan / scatter with the two host waits, as before round 6
  int64_t plain_stats[2] = {0, 0};     // immediate one-pass forwards, those that had to be redone
};

namespace rtgs {

static rtgs_ctx* default_ctx() {
  static rtgs_ctx* c = [] {
    rtgs_ctx* n = new rtgs_ctx();
    if (const char* e = getenv("RTGS_NEAR_SLICE")) n->slice_mode = atoi(e);
    if (const char* e = getenv("RTGS_NEAR_SLICE_BUDGET")) { const int b = atoi(e); if (b > 0) n->slice_budget = b; }
    if (const char* e = getenv("RTGS_SPECULATE")) n->speculation = atoi(e) != 0;
    if (const char* e = getenv("RTGS_BIN_ONEPASS")) n->onepass = atoi(e) != 0;
    if (const char* e = getenv("RTGS_PLAIN_ONEPASS")) n->plain_onepass = atoi(e) != 0;
    if (const char* e = getenv("RTGS_BWD_WALK")) { const int m = atoi(e); n->bwd_walk = (m >= 1 && m <= 4) ? m : 0; }
    return n;
  }();
  return c;
}
static inline rtgs_ctx* use(rtgs_ctx* c) { return c ? c : default_ctx(); }

static void prof_mark(rtgs_ctx* c, int which, hipStream_t st) {
  if (!c->prof) return;
  if (!c->ev_init) {
    for (int i = 0; i < EV_N; ++i) (void)hipEventCreate(&c->ev[i]);
    c->ev_init = true;
  }
  (void)hipEventRecord(c->ev[which], st);
  c->ev_set[which] = true;
}

static int bits_for(uint32_t n) {   // bits needed to represent values in [0, n)
  int b = 0;
  while ((1ull << b) < (unsigned long long)n) ++b;
  return b < 1 ? 1 : b;
}

// The forward's host sync: spin on the pinned words a kernel publishes (raster_common.h: publish_to_host).  The payload is
// accepted only when the sequence word AND the checksum over (sequence, payload) agree with what was read: whatever order
// the device's writes become visible in, a torn view is re-read.  `pub` receives a consistent snapshot of the payload.
static int wait_published_words(uint32_t* info_host, uint32_t seq, hipStream_t st, uint32_t (&pub)[7]) {
  volatile uint32_t* h = info_host;
  auto snapshot_ok = [&]() -> bool {
    if (h[7] != seq) return false;
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    uint32_t w[7];
    for (int k = 0; k < 7; ++k) w[k] = h[k];
    if (h[8] != host_checksum(seq, w)) return false;
    for (int k = 0; k < 7; ++k) pub[k] = w[k];
    return true;
  };
  for (long spin = 0; spin < 4000000L; ++spin) {
    if (snapshot_ok()) return RTGS_OK;
    __builtin_ia32_pause();
  }
  if (hipStreamSynchronize(st) != hipSuccess) return RTGS_E_HIP;   // the kernel has certainly finished: everything is visible now
  return snapshot_ok() ? RTGS_OK : RTGS_E_HIP;
}

// sort-class ceiling (launch_bin_tilesort) that covers lists of `longest` entries with a quarter of headroom
static uint32_t sort_class_cap(uint32_t longest) {
  const uint64_t want = (uint64_t)longest + longest / 4 + 16;
  const uint32_t classes[5] = {256u, 1024u, 3072u, 8192u, 16384u};
  for (uint32_t c : classes) if (want <= c) return c;
  return 16384u;
}

__global__ void bwd_info_kernel(BwdInfo* dst, SplatGrad* slot_grads, uint32_t slots, uint32_t use_slots) {
  dst->slot_grads = slot_grads; dst->slots = slots; dst->use_slots = use_slots;
}

// Offsets the BACKWARD reads (splats, radii, clamped, ranges1_bwd, list1) do not depend on `budget`: everything sized
// by the near-slice budget sits behind them, so a backward never needs to know the budget of its forward.
// Maps of at least this many Gaussians (on >= 256 tiles) take the "large map" forward: the near slice is CONSIDERED (its
// kernels decide per call whether it runs) and tiles own segments for one-pass placement.  RTGS_LARGE_MAP_MIN at load time.
static int large_map_min() {
  static const int v = [] { const char* e = getenv("RTGS_LARGE_MAP_MIN"); const int x = e ? atoi(e) : 100000; return x > 0 ? x : 100000; }();
  return v;
}

static GeomLayout geom_layout(int32_t P, int gx, int gy, int budget) {
  GeomLayout L{};
  size_t off = 0;
  const size_t Pn = (size_t)(P > 0 ? P : 1);
  L.splats = off; off = align_up(off + Pn * sizeof(Splat));
  L.tiles_touched = off; off = align_up(off + Pn * sizeof(uint32_t));
  L.offsets = off; off = align_up(off + Pn * sizeof(uint32_t));
  L.radii = off; off = align_up(off + Pn * sizeof(int32_t));
  L.clamped = off; off = align_up(off + Pn);
  L.sat = off; off = align_up(off + (size_t)(gx + 1) * (gy + 1) * sizeof(int32_t));
  L.tile_count = off; off = align_up(off + (size_t)gx * gy * sizeof(uint32_t));
  L.cursor = off; off = align_up(off + (size_t)gx * gy * sizeof(uint32_t));
  L.info = off; off = align_up(off + 4 * sizeof(uint32_t));
  L.block_counts = off; off = align_up(off + bin_block_counts_bytes((int)Pn, gx * gy));
  {
    const size_t nt = (size_t)gx * gy;
    L.zero_begin = L.tile_count;                 // tile_count .. slice_ctr are contiguous and cleared by preprocess_fwd
    off = L.tile_count + nt * sizeof(uint32_t);  // (re-lays cursor / info / block_counts behind the zeroed span)
    L.tile_count1 = off; off += nt * sizeof(uint32_t);
    L.ranges1_bwd = off; off += nt * sizeof(uint2);
    L.slice_hist = off; off += 2 * SLICE_BINS * sizeof(uint32_t);
    L.slice_cover = off; off += SLICE_BINS * sizeof(unsigned long long);
    L.slice_ctr = off; off += 8 * sizeof(uint32_t);   // [0] unfinished tiles [2] slice list length [3] cut [4] slot cursor of the slice, or visible-list length with [5] its slot cursor
    L.slot_count = off; off += Pn * sizeof(uint32_t);   // zero between calls: cleared here, and again by grad_reduce
    L.zero_end = off; off = align_up(off);
    L.cursor = off; off = align_up(off + nt * sizeof(uint32_t));
    L.info = off; off = align_up(off + 4 * sizeof(uint32_t));
    L.block_counts = off; off = align_up(off + bin_block_counts_bytes((int)Pn, gx * gy));
    L.zbin = off; off = align_up(off + Pn);
    L.cursor1 = off; off = align_up(off + nt * sizeof(uint32_t));
    L.ranges1 = off; off = align_up(off + nt * sizeof(uint2));
    L.mask2 = off; off = align_up(off + nt * sizeof(int32_t));
    L.uv = off; off = align_up(off + Pn * sizeof(float2));
    L.slice_cap = nt * (size_t)(budget > 0 ? budget : 1);
    L.slice_max_list = L.slice_cap < 65536 ? L.slice_cap : 65536;   // every listed Gaussian covers >= 1 tile
    if (L.slice_max_list > Pn) L.slice_max_list = Pn;
    L.vis_ids = off; off = align_up(off + Pn * sizeof(uint32_t));             // work list of every visible Gaussian
    L.block_counts_vis = off; off = align_up(off + bin_list_block_counts_bytes((int)Pn, gx * gy));
    // one-pass placement (bin_place_kernel): on maps where the automatic mode considers the slice, every tile owns a
    // SEGMENT of SLICE_MAX_LIST entries in list1 / bucket1 (sparse use; 37 KB per tile, 119 MB at 1200x680 of 288 GB)
    L.slice_seg = (Pn >= (size_t)large_map_min() && nt >= 256) ? (size_t)SLICE_MAX_LIST : 0;
    const size_t cap1 = L.slice_seg ? (nt * L.slice_seg > L.slice_cap ? nt * L.slice_seg : L.slice_cap) : L.slice_cap;
    L.list1 = off; off = align_up(off + cap1 * sizeof(uint32_t));             // last budget-independent OFFSET
    L.block_counts1 = off; off = align_up(off + bin_slice_block_counts_bytes((int)Pn, L.slice_max_list, gx * gy));
    L.slice_ids = off; off = align_up(off + L.slice_max_list * sizeof(uint32_t));
    L.bucket1 = off; off = align_up(off + cap1 * sizeof(uint64_t));
  }
  size_t tb = 0, tb2 = 0;
  (void)rocprim::inclusive_scan(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, Pn, rocprim::plus<uint32_t>());
  (void)rocprim::exclusive_scan(nullptr, tb2, (uint32_t*)nullptr, (uint32_t*)nullptr, 0u, Pn, rocprim::plus<uint32_t>());
  if (tb2 > tb) tb = tb2;
  L.scan_temp_bytes = tb;
  L.scan_temp = off; off = align_up(off + tb);
  L.total = off;
  return L;
}

static BinLayout bin_layout(int64_t R, int ntiles, bool sort_path, size_t slots) {
  BinLayout L{};
  size_t off = 0;
  const size_t Rn = (size_t)(R > 0 ? R : 1);
  L.vals_b = off; off = align_up(off + Rn * sizeof(uint32_t));     // point_list: offset 0 in BOTH layouts
  L.keys_a = off; off = align_up(off + Rn * sizeof(uint64_t));     // tile buckets / unsorted keys
  if (!sort_path) {
    L.keys_b = L.vals_a = L.sort_temp = off; L.sort_temp_bytes = 0;
    if (slots > 0) { L.slot_grads = off; off = align_up(off + slots * sizeof(SplatGrad)); }   // BwdInfo
    L.total = off;
    return L;
  }
  L.keys_b = off; off = align_up(off + Rn * sizeof(uint64_t));
  L.vals_a = off; off = align_up(off + Rn * sizeof(uint32_t));
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (uint64_t*)nullptr, (uint64_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, Rn, 0u, (unsigned)(32 + bits_for((uint32_t)ntiles)));
  L.sort_temp_bytes = tb;
  L.sort_temp = off; off = align_up(off + tb);
  L.total = off;
  return L;
}

static ImgLayout img_layout(int H, int W, int ntiles) {
  ImgLayout L{};
  size_t off = 0;
  L.ranges = off; off = align_up(off + (size_t)ntiles * sizeof(uint2));
  L.n_contrib = off; off = align_up(off + (size_t)H * W * sizeof(uint32_t));
  L.bwd_info = off; off = align_up(off + sizeof(BwdInfo));
  L.tile_mode = off; off = align_up(off + (size_t)ntiles * sizeof(uint32_t));
  L.depth_pos = off; off = align_up(off + (size_t)H * W * sizeof(uint32_t));
  L.tile_last = off; off = align_up(off + (size_t)ntiles * sizeof(uint32_t));
  // the TileCache (raster_common.h): 48 B of record + 2 B of block mask for the first TILE_RECS list positions of every tile
  // (39.6 + 1.7 MB at 1200x680), two plane words per pixel
  L.tile_recs = off; off = align_up(off + (size_t)ntiles * TILE_RECS * 3 * sizeof(float4));
  L.tile_masks = off; off = align_up(off + (size_t)ntiles * TILE_RECS * sizeof(uint16_t));
  L.depth_aux = off; off = align_up(off + (size_t)H * W * sizeof(float2));
  L.total = off;
  return L;
}

static int make_params(const rtgs_raster_settings* s, int32_t P, int32_t M, RasterParams& p) {
  if (!s || P < 0 || M < 1 || M > 16) return RTGS_E_INVALID;
  if (s->image_height <= 0 || s->image_width <= 0) return RTGS_E_INVALID;
  if (s->sh_degree < 0 || s->sh_degree > 3 || (s->sh_degree + 1) * (s->sh_degree + 1) > M) return RTGS_E_INVALID;
  if (!s->bg || !s->viewmatrix || !s->campos) return RTGS_E_INVALID;
  p.H = s->image_height; p.W = s->image_width;
  p.gx = (p.W + TILE - 1) / TILE; p.gy = (p.H + TILE - 1) / TILE;
  p.P = P; p.M = M; p.deg = s->sh_degree;
  p.tanfovx = s->tanfovx; p.tanfovy = s->tanfovy;
  p.fx = (float)(p.W / (2.0 * (double)s->tanfovx));      // utils/graphics_utils.py:89-90
  p.fy = (float)(p.H / (2.0 * (double)s->tanfovy));
  p.cx = s->cx > 0.f ? s->cx : 0.5f * (float)(p.W - 1);
  p.cy = s->cy > 0.f ? s->cy : 0.5f * (float)(p.H - 1);
  p.scale_modifier = s->scale_modifier;
  p.opaque_thr = s->opaque_threshold; p.depth_thr = s->depth_threshold; p.normal_thr = s->normal_threshold;
  p.color_sigma = s->color_sigma; p.T_thr = s->T_threshold;
  p.view = s->viewmatrix; p.campos = s->campos; p.bg = s->bg;
  p.spec_fail = nullptr;
  return RTGS_OK;
}

#define HIP_TRY(expr)                         \
  do {                                        \
    hipError_t e_ = (expr);                   \
    if (e_ != hipSuccess) return RTGS_E_HIP;  \
  } while (0)

static int dbg_sync(const rtgs_raster_settings* s, hipStream_t st) {
  if (s->debug) {
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipGetLastError());
  }
  return RTGS_OK;
}
#define DBG(s, st)                              \
  do {                                          \
    int r_ = dbg_sync(s, st);                   \
    if (r_ != RTGS_OK) return r_;               \
  } while (0)

}  // namespace rtgs

using namespace rtgs;

extern "C" {

const char* rtgs_version(void) { return "rtgs-hip 0.2.0 (gfx950)"; }

rtgs_ctx* rtgs_ctx_create(void) {
  rtgs_ctx* c = new (std::nothrow) rtgs_ctx();
  if (c) { c->slice_mode = default_ctx()->slice_mode; c->slice_budget = default_ctx()->slice_budget; c->bwd_walk = default_ctx()->bwd_walk; c->onepass = default_ctx()->onepass; c->speculation = default_ctx()->speculation; c->plain_onepass = default_ctx()->plain_onepass; }
  return c;
}
void rtgs_ctx_destroy(rtgs_ctx* c) {
  if (!c || c == default_ctx()) return;
  if (c->ev_init) for (int i = 0; i < EV_N; ++i) (void)hipEventDestroy(c->ev[i]);
  if (c->info_host) (void)hipHostFree(c->info_host);
  delete c;
}

size_t rtgs_raster_geom_bytes_ctx(rtgs_ctx* c, int32_t P, int32_t H, int32_t W) {
  return geom_layout(P, (W + TILE - 1) / TILE, (H + TILE - 1) / TILE, use(c)->slice_budget).total;
}
size_t rtgs_raster_geom_bytes(int32_t P, int32_t H, int32_t W) { return rtgs_raster_geom_bytes_ctx(nullptr, P, H, W); }
size_t rtgs_raster_binning_bytes(int64_t R, int32_t H, int32_t W) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  return bin_layout(R, gx * gy, true, 0).total;
}
size_t rtgs_raster_image_bytes(int32_t H, int32_t W) {
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  return img_layout(H, W, gx * gy).total;
}
// [P x SplatGrad (64 B)] [P x touched byte]
size_t rtgs_raster_backward_scratch_bytes(int32_t P) {
  const size_t n = (size_t)(P > 0 ? P : 1);
  return align_up(n * sizeof(SplatGrad)) + align_up(n);
}

int rtgs_raster_last_stats_ctx(rtgs_ctx* c, int64_t* out) {
  if (!out) return RTGS_E_INVALID;
  memcpy(out, use(c)->stats, sizeof(use(c)->stats));
  return RTGS_OK;
}
void rtgs_raster_set_counters_ctx(rtgs_ctx* c, void* counters) { use(c)->counters = (unsigned long long*)counters; }

int rtgs_raster_forward_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* s, int32_t P, int32_t M, const float* means3D,
                            const float* opacities, const float* shs, const float* scales, const float* rotations,
                            const float* normal_w, const int32_t* tile_mask, float* out_color, float* out_depth,
                            int32_t* out_cidx, int32_t* out_didx, float* out_cw, float* out_dw, float* out_T,
                            int32_t* out_radii, rtgs_resize_fn geom_resize, void* geom_user,
                            rtgs_resize_fn binning_resize, void* binning_user, rtgs_resize_fn image_resize,
                            void* image_user, int64_t* num_rendered_host, int32_t flags, void* stream) {
  rtgs_ctx* c = use(ctx);
  // one-shot: consumed HERE, before any early return - a refused forward must not leave the pointer armed for the next,
  // unrelated forward on this context
  uint32_t* const aux_zero = c->aux_zero;
  c->aux_zero = nullptr;
  RasterParams p;
  int rc = make_params(s, P, M, p);
  if (rc != RTGS_OK) return rc;
  if (P > 0 && (!means3D || !opacities || !shs || !scales || !rotations || !normal_w)) return RTGS_E_INVALID;
  if (!tile_mask || !out_color || !out_depth || !out_cidx || !out_didx || !out_cw || !out_dw || !out_T)
    return RTGS_E_INVALID;
  if (!geom_resize || !binning_resize || !image_resize || !num_rendered_host) return RTGS_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = p.gx * p.gy;

  const GeomLayout G = geom_layout(P, p.gx, p.gy, c->slice_budget);
  char* geom = (char*)geom_resize(geom_user, G.total);
  const ImgLayout I = img_layout(p.H, p.W, ntiles);
  char* img = (char*)image_resize(image_user, I.total);
  if (!geom || !img) return RTGS_E_ALLOC;
  Splat* splats = (Splat*)(geom + G.splats);
  uint32_t* tiles_touched = (uint32_t*)(geom + G.tiles_touched);
  uint32_t* offsets = (uint32_t*)(geom + G.offsets);
  int32_t* radii = (int32_t*)(geom + G.radii);
  uint8_t* clamped = (uint8_t*)(geom + G.clamped);
  int32_t* sat = (int32_t*)(geom + G.sat);
  uint2* ranges = (uint2*)(img + I.ranges);
  uint32_t* n_contrib = (uint32_t*)(img + I.n_contrib);
  uint32_t* depth_pos = (uint32_t*)(img + I.depth_pos);
  uint32_t* tile_last = (uint32_t*)(img + I.tile_last);
  // always written, also under RTGS_FWD_NO_BACKWARD: a backward that is called anyway (slower, atomics) must not meet stale
  // walk choices of an earlier forward in a recycled image buffer
  uint32_t* const tile_mode = (uint32_t*)(img + I.tile_mode);
  // what blend_fwd writes there: the MFMA walk (2) by default, a forced pixel-per-lane walk (0 strip, 1 rows), or their
  // per-tile choice from the measured list share (-1: bwd_walk 4)
  const int fwd_walk = c->bwd_walk == 0 || c->bwd_walk == 3 ? 2 : (c->bwd_walk == 4 ? -1 : c->bwd_walk - 1);
  c->last_geom = geom; c->last_img = img; c->last_bin = nullptr;
  // what blend_fwd leaves for the entry-per-lane backward (nothing when the caller promises there is no backward)
  const TileCache tcache = (flags & RTGS_FWD_NO_BACKWARD) ? TileCache{nullptr, nullptr, nullptr}
                                                          : TileCache{(float4*)(img + I.tile_recs), (uint16_t*)(img + I.tile_masks), (float2*)(img + I.depth_aux)};

  int64_t R = 0, R1 = 0;
  for (int i = 0; i < EV_B0; ++i) c->ev_set[i] = false;
  prof_mark(c, EV_F0, st);
  uint32_t* tile_count = (uint32_t*)(geom + G.tile_count);
  uint32_t* cursor = (uint32_t*)(geom + G.cursor);
  uint32_t* info = (uint32_t*)(geom + G.info);
  uint16_t* block_counts = (uint16_t*)(geom + G.block_counts);
  uint32_t* zero_words = (uint32_t*)(geom + G.zero_begin);
  const int zero_n = (int)((G.zero_end - G.zero_begin) / sizeof(uint32_t));
  uint32_t* tile_count1 = (uint32_t*)(geom + G.tile_count1);
  uint2* ranges1_bwd = (uint2*)(geom + G.ranges1_bwd);
  uint32_t* slice_hist = (uint32_t*)(geom + G.slice_hist);
  unsigned long long* slice_cover = (unsigned long long*)(geom + G.slice_cover);
  uint32_t* slice_ctr = (uint32_t*)(geom + G.slice_ctr);
  uint8_t* zbin = (uint8_t*)(geom + G.zbin);
  uint2* ranges1 = (uint2*)(geom + G.ranges1);
  int32_t* mask2 = (int32_t*)(geom + G.mask2);
  uint32_t* list1 = (uint32_t*)(geom + G.list1);
  uint32_t longest = 0;
  // one-pass placement of the near slice's instances into per-tile segments (RTGS_BIN_ONEPASS=0: count / scan / scatter)
  const uint32_t seg1 = c->onepass ? (uint32_t)G.slice_seg : 0u;
  // gradient slots of the backward (BwdInfo): gbase = exclusive scan of the rect areas, into `offsets`
  const bool want_bwd = !(flags & RTGS_FWD_NO_BACKWARD);
  uint32_t slots = 0;
  auto scan_all = [&]() -> int {
    size_t tb = G.scan_temp_bytes;
    HIP_TRY(rocprim::exclusive_scan(geom + G.scan_temp, tb, tiles_touched, offsets, 0u, (size_t)P, rocprim::plus<uint32_t>(), st));
    return RTGS_OK;
  };
  bool sort_path = (size_t)ntiles > bin_lds_limit_tiles() || c->force_sort_path;
  // near-slice pass: mode 1 forces it (tests); automatic mode considers it on large maps only, and there the kernels
  // decide from the depth histograms whether it runs (an empty slice sends every tile to the second pass)
  const bool slice_auto = c->slice_mode == 2;
  bool sliced = !sort_path && P > 0 && (c->slice_mode == 1 || (slice_auto && P >= large_map_min() && ntiles >= 256));
  const bool sliced0 = sliced;                 // what this call started as (the variable follows the passes below)
  SlicePass pass{0, nullptr, nullptr, nullptr, nullptr};
  bool declined = false, considered = false;
  SliceList vis{nullptr, nullptr};             // every visible Gaussian, when the declined single pass built the list
  // pinned host words the kernels publish totals into, and the spin that waits for them: a few microseconds instead
  // of the ~25 us a blocking hipStreamSynchronize takes to wake up (bounded; falls back)
  if (!c->info_host) {
    HIP_TRY(hipHostMalloc((void**)&c->info_host, HOST_WORDS * sizeof(uint32_t), hipHostMallocCoherent));
    memset(c->info_host, 0, HOST_WORDS * sizeof(uint32_t));
  }
  uint32_t* const info_host = c->info_host;
  uint32_t pub[7] = {0, 0, 0, 0, 0, 0, 0};
  auto wait_published = [&](uint32_t seq) -> int { return wait_published_words(info_host, seq, st, pub); };
  const int32_t* mask_main = tile_mask;        // tile mask of the pass that ends in the host sync
  uint32_t n_left = 0, n_fin = 0;

  // ------------------------------------------------------------------------------------------------------------------
  // Speculative forward (RTGS_FWD_SPECULATE): no host wait inside the call.  The host assumes the call looks like the last
  // verified one on this context - same kind of pass structure, instance / list / slot totals within a margin of the last
  // ones - sizes the binning buffer and picks the sort classes from that, and enqueues EVERYTHING; the kernel that learns
  // the real numbers (bin_tilescan, or slice_publish) checks them against the capacities and raises a device word if they
  // do not hold, on which every later kernel that could overrun a buffer or touch persistent state returns at once.
  // rtgs_raster_forward_verify reads the published numbers afterwards (the GPU is busy with the rest of the step by
  // then) and tells the caller whether to redo the step without speculation.  Results are those of the plain call.
  // ------------------------------------------------------------------------------------------------------------------
  if (c->spec.pending) return RTGS_E_INVALID;          // the previous speculative forward was never verified
  c->spec.geom = nullptr; c->spec.fail_dev = nullptr;
  {
    // `immediate`: a forward without a backward whose last such forward on this image declined the near slice runs the same
    // one-pass flow (no count / scan / scatter, no host wait before the blend) and looks at the published totals itself,
    // while the blend runs; a wrong guess (a list outgrew the assumed sort class, the slice was taken after all) falls
    // through to the classic path below.  Nothing but the image size and that class is assumed: P may differ.
    rtgs_ctx::Plan& pp = sliced ? c->plan_plain : c->plan_plain0;
    const bool immediate = !(flags & RTGS_FWD_SPECULATE) && !want_bwd && c->speculation && c->plain_onepass && c->onepass &&
                           P > 0 && !sort_path && !s->debug && pp.valid && (sliced ? (pp.kind == 2 && G.slice_seg) : pp.kind == 0) &&
                           pp.H == p.H && pp.W == p.W && pp.slice_mode == c->slice_mode && pp.slice_budget == c->slice_budget;
    const rtgs_ctx::Plan& pl = immediate ? pp : c->plan;
    const bool eligible = immediate ||
                          ((flags & RTGS_FWD_SPECULATE) && c->speculation && want_bwd && P > 0 && !sort_path && !s->debug &&
                           pl.valid && pl.P == P && pl.H == p.H && pl.W == p.W && pl.slice_mode == c->slice_mode &&
                           pl.slice_budget == c->slice_budget && (pl.kind == 0 ? !sliced : (pl.kind == 1 || pl.kind == 2) && sliced));
    const uint32_t capR = pl.R_hi + pl.R_hi / 8u + 8192u, capL = sort_class_cap(pl.longest_hi),
                   capS = immediate ? 0u : pl.slots_hi + pl.slots_hi / 8u + 8192u;
    if ((flags & RTGS_FWD_SPECULATE) && !(eligible && (pl.kind == 1 || capS <= SLOTS_MAX))) ++c->spec_stats[2];
    if (eligible && (pl.kind == 1 || capS <= SLOTS_MAX)) {
      uint32_t* const fail = slice_ctr + 6;            // inside the span every forward clears
      const SliceSel sel1{1, zbin, slice_hist, (uint32_t)G.slice_cap, (uint32_t)G.slice_max_list, slice_ctr, nullptr, nullptr,
                          slice_cover, (uint32_t)ntiles * (uint32_t)(TILE * TILE), slice_auto ? 1 : 0};
      const SpecCaps nocaps{nullptr, 0u, 0u, 0u, nullptr};
      if (++c->seq == 0u) c->seq = 1u;
      size_t b_total = 0;
      bool seg_main = false;                     // main pass placed into per-tile segments: no bound on the total
      if (pl.kind == 1) {
        // the near slice finished every tile last time: pass 1 only, the slot space is the slice's instance budget
        launch_preprocess_cull(p, means3D, scales, rotations, tiles_touched, radii, out_radii, zero_words, zero_n, zbin,
                               (float2*)(geom + G.uv), st);
        prof_mark(c, EV_PRE, st);
        const uint32_t sl = (uint32_t)G.slice_cap;
        const bool use_sl = sl > 0 && sl <= SLOTS_MAX;
        const BinLayout B = bin_layout(0, ntiles, false, use_sl ? (size_t)sl : 0);
        char* bin = (char*)binning_resize(binning_user, B.total);
        if (!bin) return RTGS_E_ALLOC;
        c->last_bin = bin;
        b_total = B.total;
        launch_slice_hist(P, zbin, tiles_touched, radii, slice_hist, slice_cover, st,
                          BwdInfoInit{(BwdInfo*)(img + I.bwd_info), (SplatGrad*)(bin + B.slot_grads), use_sl ? sl : 0u, use_sl ? 1u : 0u});
        launch_slice_compact(P, sel1, (uint32_t*)(geom + G.slice_ids), slice_ctr + 2, tiles_touched, offsets, slice_ctr + 4,
                             nullptr, 0u, nullptr, st);
        const SliceList work{(const uint32_t*)(geom + G.slice_ids), slice_ctr + 2};
        launch_preprocess_shade(p, means3D, opacities, shs, scales, rotations, normal_w, splats, radii, clamped,
                                (float2*)(geom + G.uv), work, sel1, G.slice_max_list, st);
        if (seg1) {       // one pass: no count, no scan (tile_count1 was cleared by the cull pass; the sort writes ranges1)
          launch_bin_place(p, splats, radii, tile_mask, tile_count1, (unsigned long long*)(geom + G.bucket1), seg1, nullptr,
                           sel1, work, G.slice_max_list, st);
          launch_bin_tilesort(ntiles, (uint32_t)SLICE_MAX_LIST, ranges1, (const unsigned long long*)(geom + G.bucket1), list1,
                              nullptr, st, tile_count1, seg1, ranges1);
        } else {
        if (launch_bin_count(p, splats, radii, tile_mask, tile_count1, (uint16_t*)(geom + G.block_counts1), sel1, work,
                             G.slice_max_list, st) != 0)
          return RTGS_E_HIP;
        launch_bin_tilescan(ntiles, tile_count1, ranges1, (uint32_t*)(geom + G.cursor1), info + 2, nullptr, nullptr,
                            nullptr, nullptr, nullptr, 0u, nocaps, st);
        launch_bin_scatter(p, splats, radii, tile_mask, (const uint16_t*)(geom + G.block_counts1),
                           (uint32_t*)(geom + G.cursor1), (unsigned long long*)(geom + G.bucket1), sel1, work,
                           G.slice_max_list, st);
        launch_bin_tilesort(ntiles, (uint32_t)SLICE_MAX_LIST, ranges1, (const unsigned long long*)(geom + G.bucket1), list1,
                            nullptr, st);
        }
        prof_mark(c, EV_SL_BIN, st);
        const SlicePass pass1{1, tile_mask, mask2, ranges1_bwd, ranges};
        launch_blend_fwd(p, ranges1, list1, splats, out_color, out_depth, out_cidx, out_didx, out_cw, out_dw, out_T,
                         n_contrib, c->counters, pass1, tile_mode, depth_pos, tile_last, fwd_walk, aux_zero, tcache, seg1, st);
        prof_mark(c, EV_SL_BLEND, st);      // the bracket holds the blend alone (bench.py's roofline divides by it)
        launch_slice_publish(ntiles, tile_mask, mask2, info + 2, slice_ctr, info_host, c->seq, fail, st, seg1 ? tile_count1 : nullptr);
        prof_mark(c, EV_BLEND0, st); prof_mark(c, EV_BLEND, st);
        c->hint_slice_lists = true; c->hint_main_lists = false;
        c->slice_stats[0] = 1; c->slice_stats[1] = pl.R1; c->slice_stats[2] = pl.n_fin; c->slice_stats[3] = 0;
      } else {
        // one pass (kind 2, through the visible list): every tile owns a segment of the sort class that covered the last
        // verified longest list; a list that outgrows it raises `fail` like any other wrong guess
        // (a forward without a backward on a small map - `immediate`, kind 0 - takes segments too: no count, no scan)
        const uint32_t seg2 = (c->onepass && ((pl.kind == 2 && G.slice_seg) || (immediate && pl.kind == 0))) ? capL : 0u;
        const BinLayout B = bin_layout(seg2 ? (int64_t)ntiles * (int64_t)seg2 : (int64_t)capR, ntiles, false, (size_t)capS);
        seg_main = seg2 != 0u;
        char* bin = (char*)binning_resize(binning_user, B.total);
        if (!bin) return RTGS_E_ALLOC;
        c->last_bin = bin;
        b_total = B.total;
        RasterParams pg = p;                       // the guarded kernels' view of the parameters
        pg.spec_fail = fail;
        SliceList vl{nullptr, nullptr};
        if (pl.kind == 2) {
          // the kernels declined the slice last time: assume they do again, single pass over the visible list
          launch_preprocess_cull(p, means3D, scales, rotations, tiles_touched, radii, out_radii, zero_words, zero_n, zbin,
                                 (float2*)(geom + G.uv), st);
          prof_mark(c, EV_PRE, st);
          launch_slice_hist(P, zbin, tiles_touched, radii, slice_hist, slice_cover, st,
                            BwdInfoInit{(BwdInfo*)(img + I.bwd_info), (SplatGrad*)(bin + B.slot_grads), capS, immediate ? 0u : 1u});
          // (the decision - is the slice still declined? - is re-derived inside visible_compact: no slice_compact launch)
          vl = SliceList{(const uint32_t*)(geom + G.vis_ids), slice_ctr + 4};
          launch_visible_compact(P, zbin, (uint32_t*)(geom + G.vis_ids), slice_ctr + 4, tiles_touched, offsets, slice_ctr + 5, fail, st,
                                 &sel1, (int32_t*)(slice_ctr + 3), fail);
          launch_preprocess_shade(pg, means3D, opacities, shs, scales, rotations, normal_w, splats, radii, clamped,
                                  (float2*)(geom + G.uv), vl, sel1, (size_t)P, st);
        } else {
          launch_preprocess_fwd(p, means3D, opacities, shs, scales, rotations, normal_w, nullptr, splats, tiles_touched, radii,
                                clamped, out_radii, zero_words, zero_n, nullptr, st);
          if (!immediate && (rc = scan_all()) != RTGS_OK) return rc;       // gradient-slot bases: only a backward reads them
          prof_mark(c, EV_PRE, st);
        }
        const SliceSel sel2{0, nullptr, nullptr, 0u, 0u, slice_ctr, nullptr, (const float2*)(geom + G.uv), nullptr, 0u, 0};
        if (seg2) {
          if (pl.kind != 2)
            hipLaunchKernelGGL(bwd_info_kernel, dim3(1), dim3(1), 0, st, (BwdInfo*)(img + I.bwd_info),
                               (SplatGrad*)(bin + B.slot_grads), capS, immediate ? 0u : 1u);
          launch_bin_place(pg, splats, radii, tile_mask, tile_count, (unsigned long long*)(bin + B.keys_a), seg2, fail, sel2, vl,
                           (size_t)P, st);
          prof_mark(c, EV_SCAN, st); prof_mark(c, EV_BIN0, st); prof_mark(c, EV_EMIT, st);
          const BinFinish fin{tile_count, ntiles, info, info_host, slice_ctr + 5, nullptr, slice_ctr + 4, c->seq,
                              SpecCaps{fail, 0xffffffffu, capL, immediate ? 0xffffffffu : capS,
                                       pl.kind == 2 ? (const int32_t*)(slice_ctr + 3) : nullptr}};
          launch_bin_tilesort(ntiles, capL, ranges, (const unsigned long long*)(bin + B.keys_a), (uint32_t*)(bin + B.vals_b), fail, st,
                              tile_count, seg2, ranges, &fin);
        } else {
        if (launch_bin_count(pg, splats, radii, tile_mask, tile_count, vl.ids ? (uint16_t*)(geom + G.block_counts_vis) : block_counts,
                             sel2, vl, (size_t)P, st) != 0)
          return RTGS_E_HIP;
        const SpecCaps caps{fail, capR, capL, immediate ? 0xffffffffu : capS, pl.kind == 2 ? (const int32_t*)(slice_ctr + 3) : nullptr};
        launch_bin_tilescan(ntiles, tile_count, ranges, cursor, info, info_host, nullptr, nullptr,
                            vl.ids ? slice_ctr + 5 : offsets + (P - 1), vl.ids ? nullptr : tiles_touched + (P - 1), c->seq, caps, st);
        prof_mark(c, EV_SCAN, st);
        if (pl.kind != 2)
          hipLaunchKernelGGL(bwd_info_kernel, dim3(1), dim3(1), 0, st, (BwdInfo*)(img + I.bwd_info),
                             (SplatGrad*)(bin + B.slot_grads), capS, 1u);
        prof_mark(c, EV_BIN0, st);
        launch_bin_scatter(pg, splats, radii, tile_mask, vl.ids ? (const uint16_t*)(geom + G.block_counts_vis) : block_counts,
                           cursor, (unsigned long long*)(bin + B.keys_a), sel2, vl, (size_t)P, st);
        prof_mark(c, EV_EMIT, st);
        launch_bin_tilesort(ntiles, capL, ranges, (const unsigned long long*)(bin + B.keys_a), (uint32_t*)(bin + B.vals_b), fail, st);
        }
        prof_mark(c, EV_SORT, st);
        prof_mark(c, EV_BLEND0, st);
        launch_blend_fwd(pg, ranges, (uint32_t*)(bin + B.vals_b), splats, out_color, out_depth, out_cidx, out_didx, out_cw,
                         out_dw, out_T, n_contrib, c->counters, SlicePass{0, nullptr, nullptr, nullptr, nullptr}, tile_mode, depth_pos, tile_last, fwd_walk, aux_zero, tcache, seg2, st);
        prof_mark(c, EV_BLEND, st);
        c->hint_slice_lists = false; c->hint_main_lists = true;
        c->slice_stats[0] = pl.kind == 2 ? 1 : 0; c->slice_stats[1] = 0; c->slice_stats[2] = 0;
        c->slice_stats[3] = pl.kind == 2 ? (int64_t)ntiles : 0;
      }
      HIP_TRY(hipGetLastError());
      c->hint_geom = geom; c->hint_walk = fwd_walk;
      if (immediate) {
        // the blend is on its way; the totals were published before it started
        if ((rc = wait_published(c->seq)) != RTGS_OK) return rc;
        const bool ok = (seg_main || pub[0] <= capR) && pub[1] <= capL && (pl.kind != 2 || (int32_t)pub[6] < 0);
        ++c->plain_stats[0];
        if (ok) {
          pp.note(pub[0], pub[1], 0u);
          c->last_listed = pub[2];
          c->plan.valid = false;             // as after every forward without a backward (see the end of this function)
          c->stats[0] = (int64_t)pub[0]; c->stats[1] = 32 + bits_for((uint32_t)ntiles); c->stats[2] = ntiles;
          c->stats[3] = (int64_t)G.total; c->stats[4] = (int64_t)b_total; c->stats[5] = (int64_t)I.total; c->stats[6] = 1;
          c->stats[7] = (int64_t)pub[1];
          *num_rendered_host = (int64_t)pub[0];
          return RTGS_OK;
        }
        // wrong guess: every guarded kernel after the one that noticed returned at once; start over on the classic path
        pp.valid = false;
        ++c->plain_stats[1];
      } else {
      c->spec.pending = true; c->spec.kind = pl.kind; c->spec.seq = c->seq; c->spec.capR = seg_main ? 0xffffffffu : capR; c->spec.capL = capL;
      c->spec.capS = capS; c->spec.geom = geom; c->spec.fail_dev = fail; c->spec.stream = stream; c->spec.ntiles = ntiles;
      c->spec.G_total = (int64_t)G.total; c->spec.B_total = (int64_t)b_total; c->spec.I_total = (int64_t)I.total;
      ++c->spec_stats[0];
      *num_rendered_host = (int64_t)pl.R + (int64_t)pl.R1;      // the last verified call's; the exact number comes with verify
      if (*num_rendered_host < 1) *num_rendered_host = 1;
      return RTGS_OK;
      }
    }
  }
  if (P == 0) HIP_TRY(hipMemsetAsync(zero_words, 0, (size_t)zero_n * sizeof(uint32_t), st));   // ranges1_bwd for the backward
  if (P > 0) {
    if (sort_path) {   // only the fallback path needs the 3-sigma-rect tile counts
      launch_mask_sat(tile_mask, p.gx, p.gy, sat, st);
      DBG(s, st);
    }
    const SliceSel sel1{1, zbin, slice_hist, (uint32_t)G.slice_cap, (uint32_t)G.slice_max_list, slice_ctr, nullptr, nullptr,
                        slice_cover, (uint32_t)ntiles * (uint32_t)(TILE * TILE), slice_auto ? 1 : 0};
    if (!sliced) {
      launch_preprocess_fwd(p, means3D, opacities, shs, scales, rotations, normal_w, sort_path ? sat : nullptr, splats,
                            tiles_touched, radii, clamped, out_radii, zero_words, zero_n, nullptr, st);
      if (!sort_path && want_bwd && (rc = scan_all()) != RTGS_OK) return rc;
    } else {
      // geometry of every Gaussian now, Splat records only where a list will read them
      launch_preprocess_cull(p, means3D, scales, rotations, tiles_touched, radii, out_radii, zero_words, zero_n, zbin,
                             (float2*)(geom + G.uv), st);
    }
    DBG(s, st);
    prof_mark(c, EV_PRE, st);
    if (sliced) {
      // Pass 1: bin, sort and blend only the nearest Gaussians (as many depth bins as fit the instance budget).  Tiles
      // whose every pixel saturates inside the slice are final; blend_fwd leaves a tile mask of the others.  No host
      // sync: the arrays are sized by the budget, the sort classes are launched blind.
      launch_slice_hist(P, zbin, tiles_touched, radii, slice_hist, slice_cover, st);
      const bool ask = slice_auto && c->ask_first;
      if (ask && ++c->seq == 0u) c->seq = 1u;
      launch_slice_compact(P, sel1, (uint32_t*)(geom + G.slice_ids), slice_ctr + 2, tiles_touched, offsets, slice_ctr + 4,
                           ask ? info_host : nullptr, c->seq, nullptr, st);
      if (ask) {
        if ((rc = wait_published(c->seq)) != RTGS_OK) return rc;
        declined = (int32_t)pub[6] < 0;
      }
    }
    if (sliced && declined) {
      // the kernels declined the slice and the host knows: shade every visible Gaussian and go on as a single pass
      // ... through a compact list of the visible ones (dense lanes in the shade / count / scatter kernels)
      vis = SliceList{(const uint32_t*)(geom + G.vis_ids), slice_ctr + 4};
      launch_visible_compact(P, zbin, (uint32_t*)(geom + G.vis_ids), slice_ctr + 4, tiles_touched, offsets, slice_ctr + 5, nullptr, st);
      launch_preprocess_shade(p, means3D, opacities, shs, scales, rotations, normal_w, splats, radii, clamped,
                              (float2*)(geom + G.uv), vis, sel1, (size_t)P, st);
      sliced = false;
      considered = true;
    }
    if (sliced) {
      const SliceList work{(const uint32_t*)(geom + G.slice_ids), slice_ctr + 2};
      launch_preprocess_shade(p, means3D, opacities, shs, scales, rotations, normal_w, splats, radii, clamped,
                              (float2*)(geom + G.uv), work, sel1, G.slice_max_list, st);
      if (seg1) {
        launch_bin_place(p, splats, radii, tile_mask, tile_count1, (unsigned long long*)(geom + G.bucket1), seg1, nullptr,
                         sel1, work, G.slice_max_list, st);
        launch_bin_tilesort(ntiles, (uint32_t)SLICE_MAX_LIST, ranges1, (const unsigned long long*)(geom + G.bucket1), list1,
                            nullptr, st, tile_count1, seg1, ranges1);
      } else {
      if (launch_bin_count(p, splats, radii, tile_mask, tile_count1, (uint16_t*)(geom + G.block_counts1), sel1, work,
                           G.slice_max_list, st) != 0)
        return RTGS_E_HIP;
      launch_bin_tilescan(ntiles, tile_count1, ranges1, (uint32_t*)(geom + G.cursor1), info + 2, nullptr, nullptr,
                          nullptr, nullptr, nullptr, 0u, SpecCaps{nullptr, 0u, 0u, 0u, nullptr}, st);
      launch_bin_scatter(p, splats, radii, tile_mask, (const uint16_t*)(geom + G.block_counts1),
                         (uint32_t*)(geom + G.cursor1), (unsigned long long*)(geom + G.bucket1), sel1, work,
                         G.slice_max_list, st);
      launch_bin_tilesort(ntiles, (uint32_t)SLICE_MAX_LIST, ranges1, (const unsigned long long*)(geom + G.bucket1), list1,
                          nullptr, st);
      }
      DBG(s, st);
      prof_mark(c, EV_SL_BIN, st);
      if (++c->seq == 0u) c->seq = 1u;
      const SlicePass pass1{1, tile_mask, mask2, ranges1_bwd, ranges};
      launch_blend_fwd(p, ranges1, list1, splats, out_color, out_depth, out_cidx, out_didx, out_cw, out_dw, out_T,
                       n_contrib, c->counters, pass1, tile_mode, depth_pos, tile_last, fwd_walk, aux_zero, tcache, seg1, st);
      prof_mark(c, EV_SL_BLEND, st);
      launch_slice_publish(ntiles, tile_mask, mask2, info + 2, slice_ctr, info_host, c->seq, nullptr, st, seg1 ? tile_count1 : nullptr);
      DBG(s, st);
      // the forward's host sync: the last tile of the slice publishes how many tiles are left
      if ((rc = wait_published(c->seq)) != RTGS_OK) return rc;
      n_left = pub[2]; n_fin = pub[3];
      R1 = (int64_t)pub[4];          // total of the slice lists, finished or not (accounting only)
      if (n_left > 0) {
        mask_main = mask2;
        pass = SlicePass{2, nullptr, mask2, nullptr, nullptr};
        // pass 2 bins against the whole map, but only Gaussians whose tile rect holds an unfinished tile (summed-area
        // table of the pass-2 mask) are shaded and enumerated
        launch_mask_sat(mask2, p.gx, p.gy, sat, st);
        SliceSel sel_rest = sel1;
        sel_rest.sat = sat; sel_rest.uv = (const float2*)(geom + G.uv);
        launch_preprocess_shade(p, means3D, opacities, shs, scales, rotations, normal_w, splats, radii, clamped,
                                (float2*)(geom + G.uv), SliceList{nullptr, nullptr}, sel_rest, 0, st);
        if (want_bwd && (rc = scan_all()) != RTGS_OK) return rc;
      } else if (want_bwd) {
        // every tile is final: only the slice's Gaussians can receive gradient; slice_compact already laid their slot
        // runs out (gbase in `offsets`), and the sum of their rect areas is within the slice's instance budget by
        // construction of the cut - no scan over the map, no host round trip for the size
        slots = (uint32_t)G.slice_cap;
      }
    }
    if (!sort_path && !(sliced && n_left == 0)) {
      // exact per-tile counts -> ranges; the one host sync of the forward sizes the instance arrays
      const SliceSel sel2{sliced ? 2 : 0, nullptr, nullptr, 0u, 0u, slice_ctr, sliced ? sat : nullptr,
                          (const float2*)(geom + G.uv), nullptr, 0u, 0};
      if (launch_bin_count(p, splats, radii, mask_main, tile_count,
                           vis.ids ? (uint16_t*)(geom + G.block_counts_vis) : block_counts, sel2, vis, (size_t)P, st) != 0)
        return RTGS_E_HIP;
      if (++c->seq == 0u) c->seq = 1u;
      launch_bin_tilescan(ntiles, tile_count, ranges, cursor, info, info_host, nullptr, nullptr,
                          // size of the gradient-slot space: the cursor visible_compact advanced, or the end of the full scan
                          want_bwd ? (vis.ids ? slice_ctr + 5 : offsets + (P - 1)) : nullptr,
                          want_bwd && !vis.ids ? tiles_touched + (P - 1) : nullptr, c->seq,
                          SpecCaps{nullptr, 0u, 0u, 0u, nullptr}, st);
      DBG(s, st);
      prof_mark(c, EV_SCAN, st);
      if ((rc = wait_published(c->seq)) != RTGS_OK) return rc;
      R = (int64_t)pub[0];
      longest = pub[1];
      slots = pub[5];
      if ((int)longest > bin_sort_capacity()) {   // a tile list too long for the LDS sort: redo with rect counts
        sort_path = true;
        if (sliced) {         // the global-sort path renders every tile itself: drop the slice's results
          HIP_TRY(hipMemsetAsync(ranges1_bwd, 0, (size_t)ntiles * sizeof(uint2), st));
          sliced = false; R1 = 0; n_left = n_fin = 0;
          pass = SlicePass{0, nullptr, nullptr, nullptr, nullptr};
          mask_main = tile_mask;
        }
        launch_mask_sat(tile_mask, p.gx, p.gy, sat, st);
        launch_preprocess_fwd(p, means3D, opacities, shs, scales, rotations, normal_w, sat, splats, tiles_touched,
                              radii, clamped, out_radii, nullptr, 0, nullptr, st);
        DBG(s, st);
      }
    }
    if (sort_path) {
      size_t tb = G.scan_temp_bytes;
      HIP_TRY(rocprim::inclusive_scan(geom + G.scan_temp, tb, tiles_touched, offsets, (size_t)P, rocprim::plus<uint32_t>(), st));
      uint32_t total = 0;
      HIP_TRY(hipMemcpyAsync(&total, offsets + (P - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      prof_mark(c, EV_SCAN, st);
      HIP_TRY(hipStreamSynchronize(st));
      R = (int64_t)total;
    }
  }
  *num_rendered_host = R + R1;

  const bool use_slots = want_bwd && !sort_path && slots > 0 && slots <= SLOTS_MAX;
  const BinLayout B = bin_layout(R, ntiles, sort_path, use_slots ? (size_t)slots : 0);
  char* bin = (char*)binning_resize(binning_user, B.total);
  if (!bin) return RTGS_E_ALLOC;
  c->last_bin = bin;
  hipLaunchKernelGGL(bwd_info_kernel, dim3(1), dim3(1), 0, st, (BwdInfo*)(img + I.bwd_info),
                     (SplatGrad*)(bin + B.slot_grads), use_slots ? slots : 0u, use_slots ? 1u : 0u);
  uint64_t* keys_a = (uint64_t*)(bin + B.keys_a);
  uint64_t* keys_b = (uint64_t*)(bin + B.keys_b);
  uint32_t* vals_a = (uint32_t*)(bin + B.vals_a);
  uint32_t* vals_b = (uint32_t*)(bin + B.vals_b);

  const int sort_bits = 32 + bits_for((uint32_t)ntiles);
  if (P == 0 || sort_path) HIP_TRY(hipMemsetAsync(ranges, 0, (size_t)ntiles * sizeof(uint2), st));
  if (R > 0 && !sort_path) {
    prof_mark(c, EV_BIN0, st);
    launch_bin_scatter(p, splats, radii, mask_main, vis.ids ? (const uint16_t*)(geom + G.block_counts_vis) : block_counts, cursor,
                       (unsigned long long*)keys_a,
                       SliceSel{sliced ? 2 : 0, nullptr, nullptr, 0u, 0u, slice_ctr, sliced ? sat : nullptr,
                                (const float2*)(geom + G.uv), nullptr, 0u, 0},
                       vis, (size_t)P, st);
    DBG(s, st);
    prof_mark(c, EV_EMIT, st);
    launch_bin_tilesort(ntiles, longest, ranges, (const unsigned long long*)keys_a, vals_b, nullptr, st);
    DBG(s, st);
    prof_mark(c, EV_SORT, st);
  } else if (R > 0) {
    prof_mark(c, EV_BIN0, st);
    launch_emit_keys(p, splats, radii, offsets, tile_mask, keys_a, vals_a, st);
    DBG(s, st);
    prof_mark(c, EV_EMIT, st);
    size_t tb = B.sort_temp_bytes;
    HIP_TRY(rocprim::radix_sort_pairs(bin + B.sort_temp, tb, keys_a, keys_b, vals_a, vals_b, (size_t)R, 0u,
                                      (unsigned)sort_bits, st));
    DBG(s, st);
    prof_mark(c, EV_SORT, st);
    launch_tile_ranges(R, keys_b, ranges, st);
    DBG(s, st);
    prof_mark(c, EV_RANGES, st);
  }
  prof_mark(c, EV_BLEND0, st);
  if (!sliced || n_left > 0)     // pass 2 (or the only pass); with every tile finished by the slice there is nothing to draw
    launch_blend_fwd(p, ranges, vals_b, splats, out_color, out_depth, out_cidx, out_didx, out_cw, out_dw, out_T,
                     n_contrib, c->counters, pass, tile_mode, depth_pos, tile_last, fwd_walk, aux_zero, tcache, 0u, st);
  prof_mark(c, EV_BLEND, st);
  DBG(s, st);
  HIP_TRY(hipGetLastError());
  c->stats[0] = R + R1; c->stats[1] = sort_bits; c->stats[2] = ntiles; c->stats[3] = (int64_t)G.total;
  c->stats[4] = (int64_t)B.total; c->stats[5] = (int64_t)I.total;
  c->stats[6] = sort_path ? 0 : 1; c->stats[7] = (int64_t)longest;
  c->slice_stats[0] = (sliced || considered) ? 1 : 0; c->slice_stats[1] = R1; c->slice_stats[2] = n_fin;
  c->slice_stats[3] = considered ? (int64_t)ntiles : (int64_t)n_left;      // declined: every tile goes to the single pass
  if (slice_auto && (sliced || considered)) c->ask_first = considered || (R1 == 0 && n_fin == 0);
  // which of the two list sets the backward of THIS forward has to walk (host-side hint, keyed by the geometry buffer)
  c->hint_geom = geom; c->hint_slice_lists = sliced && n_fin > 0; c->hint_main_lists = R > 0; c->hint_walk = fwd_walk;
  {
    // what a speculative forward on this context may assume next time (rtgs_raster_forward_verify keeps it current)
    rtgs_ctx::Plan& pl = c->plan;
    const bool same_shape = pl.valid && pl.P == P && pl.H == p.H && pl.W == p.W;
    // from the FINAL state of this call: a forward that fell back to the global sort (a list longer than the LDS sort
    // holds) leaves no plan - speculating on it would fail its guard word on every step and run every step twice
    pl.kind = sort_path ? -1 : ((sliced && n_left == 0) ? 1 : (considered ? 2 : (!sliced && P > 0 ? 0 : -1)));
    pl.valid = pl.kind >= 0 && want_bwd && R <= 0xffffffffll;
    pl.P = P; pl.H = p.H; pl.W = p.W; pl.slice_mode = c->slice_mode; pl.slice_budget = c->slice_budget;
    if (!same_shape || !pl.valid) { pl.R_hi = pl.longest_hi = pl.slots_hi = 0; }     // another map / image: start the maxima over
    pl.note((uint32_t)R, longest, slots);
    pl.R1 = (uint32_t)R1; pl.n_fin = n_fin;
    if (!want_bwd) {
      // ... and what the next forward WITHOUT a backward on this image may assume (any Gaussian count)
      rtgs_ctx::Plan& pp = sliced0 ? c->plan_plain : c->plan_plain0;
      const bool same_image = pp.valid && pp.H == p.H && pp.W == p.W;
      pp.kind = sort_path ? -1 : (sliced0 ? (considered ? 2 : -1) : (P > 0 ? 0 : -1));
      pp.valid = pp.kind >= 0 && R <= 0xffffffffll;
      pp.P = P; pp.H = p.H; pp.W = p.W; pp.slice_mode = c->slice_mode; pp.slice_budget = c->slice_budget;
      if (!same_image || !pp.valid) { pp.R_hi = pp.longest_hi = pp.slots_hi = 0; }
      pp.note((uint32_t)R, longest, 0u);
    }
  }
  return RTGS_OK;
}

static int backward_impl(rtgs_ctx* ctx, const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R, const float* means3D,
                         const float* opacities, const float* shs, const float* scales, const float* rotations,
                         const float* normal_w, void* geom_buffer, void* binning_buffer,
                         const void* image_buffer, const float* out_color, const float* out_T,
                         const int32_t* out_didx,
                         const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans3D, float* dL_dopacities,
                         float* dL_dshs, float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                         void* grad_scratch, uint8_t* row_state, int32_t train_begin, int32_t train_end, void* stream,
                         bool walk_only = false) {
  rtgs_ctx* c = use(ctx);   // NOTE: the geometry buffer is written here (slot counters): it is scratch of the pair
  RasterParams p;
  int rc = make_params(s, P, M, p);
  if (rc != RTGS_OK) return rc;
  if (P == 0) return RTGS_OK;
  // the backward of a speculative forward guards itself with that forward's device word (see rtgs_raster_forward_verify)
  if (c->spec.geom == geom_buffer && c->spec.fail_dev) p.spec_fail = c->spec.fail_dev;
  if (!means3D || !opacities || !shs || !scales || !rotations || !normal_w || !geom_buffer || !binning_buffer ||
      !image_buffer || !out_color || !out_T || !out_didx || !dL_dcolor || !dL_ddepth || !grad_scratch || R < 0)
    return RTGS_E_INVALID;
  if (!walk_only && (!dL_dmeans3D || !dL_dopacities || !dL_dshs || !dL_dscales || !dL_drotations || !dL_dnormal_w)) return RTGS_E_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const int ntiles = p.gx * p.gy;
  const GeomLayout G = geom_layout(P, p.gx, p.gy, 1);   // only budget-independent offsets are read here
  const BinLayout B = bin_layout(R, ntiles, false, 0);  // point_list sits at offset 0 in every layout
  const ImgLayout I = img_layout(p.H, p.W, ntiles);
  const char* geom = (const char*)geom_buffer;
  const char* bin = (const char*)binning_buffer;
  const char* img = (const char*)image_buffer;
  SplatGrad* grads = (SplatGrad*)grad_scratch;
  for (int i = EV_B0; i < EV_N; ++i) c->ev_set[i] = false;
  prof_mark(c, EV_B0, st);
  uint8_t* touched = (uint8_t*)grad_scratch + align_up((size_t)P * sizeof(SplatGrad));
  // row-state mode: the caller zeroed the scratch once and preprocess_bwd re-zeroes every line it consumes
  if (!row_state) HIP_TRY(hipMemsetAsync(grads, 0, rtgs_raster_backward_scratch_bytes(P), st));
  const BwdInfo* binfo = (const BwdInfo*)(img + I.bwd_info);
  const uint32_t* gbase = (const uint32_t*)(geom + G.offsets);
  const int32_t* radii = (const int32_t*)(geom + G.radii);
  uint32_t* slot_count = (uint32_t*)(geom + G.slot_count);      // scratch of the backward inside the geometry buffer
  const uint32_t* tile_mode = (const uint32_t*)(img + I.tile_mode);   // per tile: which of the two walks (blend_fwd decided)
  if (R > 0) {
    // two-pass forward: tiles the near slice finished walk its lists (ranges1_bwd is all-empty otherwise, and a
    // workgroup with an empty range returns at once); every other tile walks the main lists
    // (a launch whose every workgroup would find an empty range is skipped when this context still remembers the forward)
    const bool hinted = c->hint_geom == geom_buffer;
    // bit 0 strip, bit 1 row-granular, bit 2 MFMA walk: a forced walk launches only its kernel
    // the kernels to launch follow what the FORWARD wrote into tile_mode (remembered with its geometry buffer), not the
    // context's setting at backward time; a backward this context does not remember launches all three - each kernel
    // takes only the tiles that carry its walk
    const int which = !hinted ? 7 : (c->hint_walk == 2 ? 4 : (c->hint_walk == -1 ? 3 : (c->hint_walk == 0 ? 1 : 2)));
    if (train_begin < 0 || train_end > P || train_end < train_begin) return RTGS_E_INVALID;
    const uint32_t t0 = (uint32_t)train_begin, tn = (uint32_t)(train_end - train_begin);   // trainable rows (the rest: rendered only)
    const uint32_t* depth_pos = (const uint32_t*)(img + I.depth_pos);
    for (int set = 0; set < 2; ++set) {
      if (hinted && !(set == 0 ? c->hint_slice_lists : c->hint_main_lists)) continue;
      const uint2* rg = set == 0 ? (const uint2*)(geom + G.ranges1_bwd) : (const uint2*)(img + I.ranges);
      const uint32_t* pl = set == 0 ? (const uint32_t*)(geom + G.list1) : (const uint32_t*)(bin + B.vals_b);
      if (which & 3)
        launch_blend_bwd(p, rg, pl, (const Splat*)(geom + G.splats), out_color, out_T, (const uint32_t*)(img + I.n_contrib),
                         out_didx, dL_dcolor, dL_ddepth, gbase, slot_count, binfo, grads, touched, tile_mode, which & 3, t0, tn, st);
      if (which & 4)
        launch_blend_bwd_entry(p, rg, pl, (const Splat*)(geom + G.splats), out_color, (const uint32_t*)(img + I.n_contrib),
                               out_didx, depth_pos, (const uint32_t*)(img + I.tile_last), dL_dcolor, dL_ddepth, gbase, slot_count, binfo, grads,
                               touched, tile_mode, t0, tn,
                               TileCache{(float4*)(img + I.tile_recs), (uint16_t*)(img + I.tile_masks), (float2*)(img + I.depth_aux)},
                               set == 0 ? (uint32_t)(BLOCK / 4) : (uint32_t)BLOCK,      // the near slice's first batch is a quarter batch
                               st);
    }
    prof_mark(c, EV_BWALK, st);
    // sum each touched Gaussian's slots into its SplatGrad record (no-op on the atomic fallback)
    if (!walk_only) launch_grad_reduce(P, touched, gbase, slot_count, binfo, grads, p.spec_fail, st);
    DBG(s, st);
  }
  prof_mark(c, EV_BBLEND, st);
  if (walk_only) { prof_mark(c, EV_BPRE, st); HIP_TRY(hipGetLastError()); return RTGS_OK; }   // a fused consumer takes over
  launch_preprocess_bwd(p, means3D, opacities, shs, scales, rotations, normal_w, radii,
                        (const uint8_t*)(geom + G.clamped), grads, touched, row_state, dL_dmeans3D, dL_dopacities,
                        dL_dshs, dL_dscales, dL_drotations, dL_dnormal_w, st);
  prof_mark(c, EV_BPRE, st);
  DBG(s, st);
  HIP_TRY(hipGetLastError());
  return RTGS_OK;
}

int rtgs_raster_backward_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R,
                             const float* means3D, const float* opacities, const float* shs, const float* scales,
                             const float* rotations, const float* normal_w, void* geom_buffer,
                             void* binning_buffer, const void* image_buffer, const float* out_color,
                             const float* out_T, const int32_t* out_didx, const float* dL_dcolor, const float* dL_ddepth,
                             float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs, float* dL_dscales,
                             float* dL_drotations, float* dL_dnormal_w, void* grad_scratch, void* stream) {
  return backward_impl(ctx, s, P, M, R, means3D, opacities, shs, scales, rotations, normal_w, geom_buffer, binning_buffer,
                       image_buffer, out_color, out_T, out_didx, dL_dcolor, dL_ddepth, dL_dmeans3D, dL_dopacities,
                       dL_dshs, dL_dscales, dL_drotations, dL_dnormal_w, grad_scratch, nullptr, 0, P, stream);
}

int rtgs_raster_backward_rows_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R,
                                  const float* means3D, const float* opacities, const float* shs, const float* scales,
                                  const float* rotations, const float* normal_w, void* geom_buffer,
                                  void* binning_buffer, const void* image_buffer, const float* out_color,
                                  const float* out_T, const int32_t* out_didx, const float* dL_dcolor,
                                  const float* dL_ddepth, float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                                  float* dL_dscales, float* dL_drotations, float* dL_dnormal_w, void* grad_scratch,
                                  uint8_t* row_state, void* stream) {
  return rtgs_raster_backward_range_ctx(ctx, s, P, M, R, means3D, opacities, shs, scales, rotations, normal_w, geom_buffer,
                                        binning_buffer, image_buffer, out_color, out_T, out_didx, dL_dcolor, dL_ddepth,
                                        dL_dmeans3D, dL_dopacities, dL_dshs, dL_dscales, dL_drotations, dL_dnormal_w,
                                        grad_scratch, row_state, 0, P, stream);
}

int rtgs_raster_backward_range_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R,
                                   const float* means3D, const float* opacities, const float* shs, const float* scales,
                                   const float* rotations, const float* normal_w, void* geom_buffer,
                                   void* binning_buffer, const void* image_buffer, const float* out_color,
                                   const float* out_T, const int32_t* out_didx, const float* dL_dcolor,
                                   const float* dL_ddepth, float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                                   float* dL_dscales, float* dL_drotations, float* dL_dnormal_w, void* grad_scratch,
                                   uint8_t* row_state, int32_t train_begin, int32_t train_end, void* stream) {
  if (P > 0 && !row_state) return RTGS_E_INVALID;
  return backward_impl(ctx, s, P, M, R, means3D, opacities, shs, scales, rotations, normal_w, geom_buffer, binning_buffer,
                       image_buffer, out_color, out_T, out_didx, dL_dcolor, dL_ddepth, dL_dmeans3D, dL_dopacities,
                       dL_dshs, dL_dscales, dL_drotations, dL_dnormal_w, grad_scratch, row_state, train_begin, train_end, stream);
}

int rtgs_raster_backward_walk_ctx(rtgs_ctx* ctx, const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R,
                                   const float* means3D, const float* opacities, const float* shs, const float* scales,
                                   const float* rotations, const float* normal_w, void* geom_buffer,
                                   void* binning_buffer, const void* image_buffer, const float* out_color,
                                   const float* out_T, const int32_t* out_didx, const float* dL_dcolor,
                                   const float* dL_ddepth, float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs,
                                   float* dL_dscales, float* dL_drotations, float* dL_dnormal_w, void* grad_scratch,
                                   uint8_t* row_state, int32_t train_begin, int32_t train_end, void* stream) {
  return backward_impl(ctx, s, P, M, R, means3D, opacities, shs, scales, rotations, normal_w, geom_buffer, binning_buffer,
                       image_buffer, out_color, out_T, out_didx, dL_dcolor, dL_ddepth, dL_dmeans3D, dL_dopacities,
                       dL_dshs, dL_dscales, dL_drotations, dL_dnormal_w, grad_scratch, row_state, train_begin, train_end, stream, true);
}

void rtgs_raster_set_profiling_ctx(rtgs_ctx* c, int enable) { use(c)->prof = enable != 0; }
void rtgs_raster_set_near_slice_ctx(rtgs_ctx* c, int mode, int budget_per_tile) {
  use(c)->slice_mode = mode;
  if (budget_per_tile > 0) use(c)->slice_budget = budget_per_tile;
}
int rtgs_raster_last_slice_stats_ctx(rtgs_ctx* c, int64_t* out) {
  if (!out) return RTGS_E_INVALID;
  memcpy(out, use(c)->slice_stats, sizeof(use(c)->slice_stats));
  return RTGS_OK;
}
void rtgs_raster_force_sort_path_ctx(rtgs_ctx* c, int enable) { use(c)->force_sort_path = enable != 0; }
int rtgs_raster_forward_verify_ctx(rtgs_ctx* ctx, int64_t* num_rendered_host) {
  rtgs_ctx* c = use(ctx);
  if (!c->spec.pending) return 0;
  c->spec.pending = false;
  uint32_t pub[7] = {0, 0, 0, 0, 0, 0, 0};
  const int rc = wait_published_words(c->info_host, c->spec.seq, (hipStream_t)c->spec.stream, pub);
  if (rc != RTGS_OK) { c->plan.valid = false; return rc; }
  rtgs_ctx::Plan& pl = c->plan;
  bool ok;
  if (c->spec.kind == 1) {
    const uint32_t n_left = pub[2];
    ok = n_left == 0u;
    pl.R1 = pub[4]; pl.n_fin = pub[3]; pl.R = 0; pl.longest = 0;
    c->last_listed = pub[0];
    c->slice_stats[1] = pub[4]; c->slice_stats[2] = pub[3]; c->slice_stats[3] = n_left;
  } else {
    ok = pub[0] <= c->spec.capR && pub[1] <= c->spec.capL && pub[5] <= c->spec.capS && (c->spec.kind != 2 || (int32_t)pub[6] < 0);
    pl.note(pub[0], pub[1], pub[5]); pl.R1 = 0;
    c->last_listed = pub[2];
  }
  c->stats[0] = (int64_t)pl.R + (int64_t)pl.R1; c->stats[1] = 32 + bits_for((uint32_t)c->spec.ntiles); c->stats[2] = c->spec.ntiles;
  c->stats[3] = c->spec.G_total; c->stats[4] = c->spec.B_total; c->stats[5] = c->spec.I_total; c->stats[6] = 1;
  c->stats[7] = (int64_t)pl.longest;
  if (num_rendered_host) *num_rendered_host = c->stats[0] > 0 ? c->stats[0] : 1;
  if (!ok) { pl.valid = false; ++c->spec_stats[1]; }
  return ok ? 0 : 1;
}
const uint32_t* rtgs_raster_spec_fail_ptr_ctx(rtgs_ctx* ctx) {
  rtgs_ctx* c = use(ctx);
  return c->spec.pending ? c->spec.fail_dev : nullptr;
}
void rtgs_raster_set_speculation_ctx(rtgs_ctx* ctx, int enable) {
  rtgs_ctx* c = use(ctx);
  c->speculation = enable != 0;
  c->plan = rtgs_ctx::Plan();          // forget the history: the next forward runs plainly and starts it over
}
int rtgs_raster_speculation_stats_ctx(rtgs_ctx* ctx, int64_t* out3) {
  if (!out3) return RTGS_E_INVALID;
  memcpy(out3, use(ctx)->spec_stats, sizeof(use(ctx)->spec_stats));
  return RTGS_OK;
}
int rtgs_raster_plain_stats_ctx(rtgs_ctx* ctx, int64_t* out2) {
  if (!out2) return RTGS_E_INVALID;
  memcpy(out2, use(ctx)->plain_stats, sizeof(use(ctx)->plain_stats));
  return RTGS_OK;
}
void rtgs_raster_set_plain_onepass_ctx(rtgs_ctx* ctx, int enable) {
  rtgs_ctx* c = use(ctx);
  c->plain_onepass = enable != 0;
  c->plan_plain = rtgs_ctx::Plan();
  c->plan_plain0 = rtgs_ctx::Plan();
}
uint32_t rtgs_raster_last_listed_ctx(rtgs_ctx* c) { return use(c)->last_listed; }
void rtgs_raster_set_bwd_debug(int bits) { rtgs::set_bwd_debug(bits); }
void rtgs_raster_set_bwd_stamps(void* dev) { rtgs::set_bwd_stamps(dev); }
void rtgs_raster_set_fwd_stamps(void* dev) { rtgs::set_fwd_stamps(dev); }
void rtgs_raster_set_onepass_ctx(rtgs_ctx* c, int on) { use(c)->onepass = on != 0; use(c)->plan.valid = false; }
void rtgs_raster_set_bwd_walk_ctx(rtgs_ctx* c, int mode) { use(c)->bwd_walk = (mode >= 1 && mode <= 4) ? mode : 0; }
void rtgs_raster_set_aux_zero_ctx(rtgs_ctx* ctx, void* eight_words) { use(ctx)->aux_zero = (uint32_t*)eight_words; }
int rtgs_raster_last_buffers_ctx(rtgs_ctx* ctx, void** out3) {
  rtgs_ctx* c = use(ctx);
  if (!out3 || !c->last_geom || !c->last_bin || !c->last_img) return RTGS_E_INVALID;
  out3[0] = c->last_geom; out3[1] = c->last_bin; out3[2] = c->last_img;
  return RTGS_OK;
}
int rtgs_raster_backward_buffers(int32_t P, int32_t H, int32_t W, size_t* out) {
  if (!out || P < 0 || H <= 0 || W <= 0) return RTGS_E_INVALID;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const GeomLayout G = geom_layout(P, gx, gy, 1);        // budget-independent offsets only
  const ImgLayout I = img_layout(H, W, gx * gy);
  out[0] = G.clamped; out[1] = G.offsets; out[2] = G.slot_count; out[3] = I.bwd_info;
  out[4] = align_up((size_t)(P > 0 ? P : 1) * sizeof(SplatGrad)); out[5] = G.radii; out[6] = 0; out[7] = 0;
  return RTGS_OK;
}
int rtgs_raster_image_offsets(int32_t H, int32_t W, size_t* out) {
  if (!out || H <= 0 || W <= 0) return RTGS_E_INVALID;
  const int gx = (W + TILE - 1) / TILE, gy = (H + TILE - 1) / TILE;
  const ImgLayout I = img_layout(H, W, gx * gy);
  out[0] = I.ranges; out[1] = I.n_contrib; out[2] = I.bwd_info; out[3] = I.tile_mode; out[4] = I.total; out[5] = I.depth_pos;
  return RTGS_OK;
}

int rtgs_raster_last_timings_ctx(rtgs_ctx* ctx, float* ms) {
  rtgs_ctx* c = use(ctx);
  if (!ms) return RTGS_E_INVALID;
  for (int i = 0; i < 12; ++i) ms[i] = -1.f;
  if (!c->ev_init) return RTGS_OK;
  // [0] preprocess_fwd(+sat) [1] count+scan of the main pass [2] scatter / emit_keys [3] sort [4] tile_ranges
  // [5] blend_fwd of the main pass [6] blend_bwd (the launches that have lists to walk) [7] preprocess_bwd
  // [8] near slice: hist+count+scan+scatter+sort [9] near slice: blend_fwd [10] grad_reduce [11] unused
  const int pre_end = c->ev_set[EV_SL_BLEND] ? EV_SL_BLEND : EV_PRE;
  const int pairs[11][2] = {{EV_F0, EV_PRE}, {pre_end, EV_SCAN}, {EV_BIN0, EV_EMIT}, {EV_EMIT, EV_SORT},
                            {EV_SORT, EV_RANGES}, {EV_BLEND0, EV_BLEND}, {EV_B0, EV_BWALK}, {EV_BBLEND, EV_BPRE},
                            {EV_PRE, EV_SL_BIN}, {EV_SL_BIN, EV_SL_BLEND}, {EV_BWALK, EV_BBLEND}};
  for (int i = 0; i < 11; ++i) {
    const int a = pairs[i][0], b = pairs[i][1];
    if (!c->ev_set[a] || !c->ev_set[b]) continue;
    if (hipEventSynchronize(c->ev[b]) != hipSuccess) return RTGS_E_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, c->ev[a], c->ev[b]) == hipSuccess) ms[i] = t;
  }
  return RTGS_OK;
}

// ---- the plain entry points: the same calls on the process-wide default context ----------------------------------
int rtgs_raster_forward(const rtgs_raster_settings* s, int32_t P, int32_t M, const float* means3D,
                        const float* opacities, const float* shs, const float* scales, const float* rotations,
                        const float* normal_w, const int32_t* tile_mask, float* out_color, float* out_depth,
                        int32_t* out_cidx, int32_t* out_didx, float* out_cw, float* out_dw, float* out_T,
                        int32_t* out_radii, rtgs_resize_fn geom_resize, void* geom_user,
                        rtgs_resize_fn binning_resize, void* binning_user, rtgs_resize_fn image_resize,
                        void* image_user, int64_t* num_rendered_host, void* stream) {
  return rtgs_raster_forward_ctx(nullptr, s, P, M, means3D, opacities, shs, scales, rotations, normal_w, tile_mask,
                                 out_color, out_depth, out_cidx, out_didx, out_cw, out_dw, out_T, out_radii, geom_resize,
                                 geom_user, binning_resize, binning_user, image_resize, image_user, num_rendered_host, 0, stream);
}
int rtgs_raster_backward(const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R, const float* means3D,
                         const float* opacities, const float* shs, const float* scales, const float* rotations,
                         const float* normal_w, void* geom_buffer, void* binning_buffer,
                         const void* image_buffer, const float* out_color, const float* out_T, const int32_t* out_didx,
                         const float* dL_dcolor, const float* dL_ddepth, float* dL_dmeans3D, float* dL_dopacities,
                         float* dL_dshs, float* dL_dscales, float* dL_drotations, float* dL_dnormal_w,
                         void* grad_scratch, void* stream) {
  return backward_impl(nullptr, s, P, M, R, means3D, opacities, shs, scales, rotations, normal_w, geom_buffer,
                       binning_buffer, image_buffer, out_color, out_T, out_didx, dL_dcolor, dL_ddepth, dL_dmeans3D,
                       dL_dopacities, dL_dshs, dL_dscales, dL_drotations, dL_dnormal_w, grad_scratch, nullptr, 0, P, stream);
}
int rtgs_raster_backward_rows(const rtgs_raster_settings* s, int32_t P, int32_t M, int64_t R, const float* means3D,
                              const float* opacities, const float* shs, const float* scales, const float* rotations,
                              const float* normal_w, void* geom_buffer, void* binning_buffer,
                              const void* image_buffer, const float* out_color, const float* out_T,
                              const int32_t* out_didx, const float* dL_dcolor, const float* dL_ddepth,
                              float* dL_dmeans3D, float* dL_dopacities, float* dL_dshs, float* dL_dscales,
                              float* dL_drotations, float* dL_dnormal_w, void* grad_scratch, uint8_t* row_state,
                              void* stream) {
  return rtgs_raster_backward_rows_ctx(nullptr, s, P, M, R, means3D, opacities, shs, scales, rotations, normal_w,
                                       geom_buffer, binning_buffer, image_buffer, out_color, out_T, out_didx, dL_dcolor,
                                       dL_ddepth, dL_dmeans3D, dL_dopacities, dL_dshs, dL_dscales, dL_drotations,
                                       dL_dnormal_w, grad_scratch, row_state, stream);
}
int rtgs_raster_last_stats(int64_t* out) { return rtgs_raster_last_stats_ctx(nullptr, out); }
void rtgs_raster_set_counters(void* counters) { rtgs_raster_set_counters_ctx(nullptr, counters); }
void rtgs_raster_set_profiling(int enable) { rtgs_raster_set_profiling_ctx(nullptr, enable); }
void rtgs_raster_set_near_slice(int mode, int budget) { rtgs_raster_set_near_slice_ctx(nullptr, mode, budget); }
int rtgs_raster_last_slice_stats(int64_t* out) { return rtgs_raster_last_slice_stats_ctx(nullptr, out); }
void rtgs_raster_force_sort_path(int enable) { rtgs_raster_force_sort_path_ctx(nullptr, enable); }
int rtgs_raster_last_timings(float* ms) { return rtgs_raster_last_timings_ctx(nullptr, ms); }

}  // extern "C"
