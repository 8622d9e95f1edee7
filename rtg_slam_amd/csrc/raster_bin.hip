// Tile binning of the gfx950 rasterizer without a global sort.
//
//   bin_count    per-tile instance counts   (work-balanced enumeration, LDS-aggregated atomics, one
//                global add per block x tile)
//   bin_tilescan exclusive scan over tiles  -> ranges, cursors, total R, longest list
//   bin_scatter  (depth bits << 32 | id) into the tile's bucket (LDS-aggregated slot reservation)
//   bin_tilesort one workgroup per tile: in-LDS sort of the bucket by (depth, id) - bitonic or LSD radix by
//                size class -> point_list (ids, front to back)
//   slice_hist / slice_compact   the near-slice pass of the two-pass forward (raster_api.hip): area / count
//                histograms over monotone depth bins, and the work list of the Gaussians in front of the cut;
//                bin_count / bin_scatter then walk that list instead of all Gaussians
//
// Replaces the upstream design's duplicateWithKeys + 64-bit global radix sort + identifyTileRanges
// (6+ passes over 12 B x instances) with one 8-B write, one 8-B read and one 4-B write per
// instance.  The resulting per-tile order is (depth, Gaussian id) ascending - exactly the stable
// order of the sort-based path and of oracle/raster_oracle.py::bin_tiles.
//
// Tiles of the 3-sigma rect that provably contain no pixel with
// alpha >= 1/255 are dropped here: such an entry is skipped by every pixel of the tile in
// blend_fwd, so removing it changes no output.  The test is the minimum of the conic form over
// the tile's pixel-centre box against 2 ln(255 o), with a safety margin far above float rounding.
#include "raster_common.h"
#include <mutex>

namespace rtgs {

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: remember, per device, the largest
// dynamic-LDS size already granted, so the attribute is raised once per size and device instead of on every launch
// (several devices or threads in one process are safe).
struct LdsGrant {
  std::mutex m;
  size_t got[64];
  explicit LdsGrant(size_t base) { for (size_t& g : got) g = base; }
  void ensure(const void* fn, size_t lds) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) { (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); return; }
    std::lock_guard<std::mutex> lk(m);
    if (lds > got[dev]) {
      (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      got[dev] = lds;
    }
  }
};

constexpr int GPB = 1024;       // Gaussians per workgroup in bin_count / bin_scatter
#ifndef RTGS_SLICE_GPB
#define RTGS_SLICE_GPB 32
#endif
constexpr int SLICE_GPB = RTGS_SLICE_GPB;   // ... when they walk the near slice's short work list: 8 Gaussians per wave (each covers
                                // tens of tiles, the candidates are spread over all 64 lanes), many workgroups

struct BinG {
  float u, v, ca, cb, cc, thr;  // thr = 2 ln(255 o) with margin; < 0 -> never visible
  float ica, icc;               // 1/ca, 1/cc
  int x0, y0, x1, y1;
  uint32_t zbits;
};

__device__ __forceinline__ bool load_bing(const RasterParams& p, const Splat* __restrict__ splats,
                                          const int32_t* __restrict__ radii, const uint8_t* __restrict__ zbin, int cut,
                                          const int32_t* __restrict__ sat, const float2* __restrict__ uv, int i, BinG& g) {
  g.x0 = g.y0 = g.x1 = g.y1 = 0;
  g.u = g.v = g.ca = g.cb = g.cc = 0.f; g.thr = -1.f; g.zbits = 0; g.ica = g.icc = 0.f;
  if (i >= p.P) return false;
  if (zbin && (int)zbin[i] > cut) return false;      // not in the near slice (255 = invisible)
  const int radius = radii[i];
  if (radius <= 0) return false;
  if (sat) {      // second pass: does the rect hold an unfinished tile at all?  (before the Splat is touched)
    const float2 c = uv[i];
    int x0, y0, x1, y1;
    tile_rect_of(c.x, c.y, radius, p.gx, p.gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) <= 0 || sat_count(sat, p.gx, x0, y0, x1, y1) == 0) return false;
  }
  const float4 r0 = reinterpret_cast<const float4*>(splats + i)[0];
  const float4 r1 = reinterpret_cast<const float4*>(splats + i)[1];
  g.u = r0.x; g.v = r0.y; g.ca = r0.z; g.cb = r0.w; g.cc = r1.x;
  const float o = r1.y;
  if (!(o >= 1.f / 255.f)) return false;            // alpha = min(.99, o G) <= o < 1/255 everywhere
  g.zbits = __float_as_uint(reinterpret_cast<const float*>(splats + i)[13]);
  const bool pd = (g.ca > 0.f) && (g.cc > 0.f) && (g.ca * g.cc - g.cb * g.cb > 0.f);
  // q <= 2 ln(255 o) <=> alpha >= 1/255; margin: 1e-3 relative + 1e-2 absolute on q (float rounding
  // of power in blend is ~1e-6 relative).  Non-PD conics (never seen; det guard) keep every tile.
  g.thr = pd ? (2.f * __logf(255.f * o)) * 1.001f + 1e-2f : 3.0e38f;
  g.ica = __builtin_amdgcn_rcpf(g.ca); g.icc = __builtin_amdgcn_rcpf(g.cc);   // only place the clamp points
  tile_rect_of(g.u, g.v, radius, p.gx, p.gy, g.x0, g.y0, g.x1, g.y1);
  return (g.x1 - g.x0) * (g.y1 - g.y0) > 0;
}

// min over the tile's pixel-centre box of q(d) = ca dx^2 + 2 cb dx dy + cc dy^2, d = centre - pixel
__device__ __forceinline__ bool tile_visible(const BinG& g, int tx, int ty) {
  const float dx0 = g.u - (float)(tx * TILE + TILE - 1), dx1 = g.u - (float)(tx * TILE);
  const float dy0 = g.v - (float)(ty * TILE + TILE - 1), dy1 = g.v - (float)(ty * TILE);
  if (dx0 <= 0.f && dx1 >= 0.f && dy0 <= 0.f && dy1 >= 0.f) return true;
  float qmin = 3.4e38f;
  const float ica = g.ica, icc = g.icc;
#pragma unroll
  for (int e = 0; e < 2; ++e) {                      // edges dx = const
    const float dx = e ? dx1 : dx0;
    const float dy = fminf(dy1, fmaxf(dy0, -g.cb * dx * icc));
    qmin = fminf(qmin, g.ca * dx * dx + 2.f * g.cb * dx * dy + g.cc * dy * dy);
  }
#pragma unroll
  for (int e = 0; e < 2; ++e) {                      // edges dy = const
    const float dy = e ? dy1 : dy0;
    const float dx = fminf(dx1, fmaxf(dx0, -g.cb * dy * ica));
    qmin = fminf(qmin, g.ca * dx * dx + 2.f * g.cb * dx * dy + g.cc * dy * dy);
  }
  return qmin <= g.thr;
}

// ---------------------------------------------------------------------------------------------
// Work-balanced enumeration: the (Gaussian, tile) candidates of the 64 Gaussians a wave holds are
// laid end to end (wave prefix sum of the rect areas) and processed 64 candidates per round, one per
// lane, whatever the mix of 1-tile and 700-tile rects - instead of one lane looping over its own rect
// while 63 wait.  Ownership of a candidate is recovered without a search: the starts that fall into
// the round are OR-ed into a 64-bit head mask in LDS and a lane's owner is
//   (#Gaussians started before the round) + popcount(head mask up to the lane) - 1
// over the compacted, rank-ordered record table the wave keeps in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int GREC = 12;                         // words per compacted record
struct WaveBin {
  uint32_t rec[64 * GREC];
  uint32_t start[64];
  unsigned long long head;
  unsigned long long pad;
};

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v, int lane) {
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)v, off);
    if (lane >= off) v += o;
  }
  return v;
}

template <class F>
__device__ __forceinline__ void enumerate_balanced(const RasterParams& p, const Splat* __restrict__ splats,
                                                   const int32_t* __restrict__ radii, const int32_t* __restrict__ mask,
                                                   const uint8_t* __restrict__ zbin, int cut, SliceList list, int gpb,
                                                   const int32_t* __restrict__ sat, const float2* __restrict__ uv,
                                                   WaveBin* wb, F f) {
  const int lane = threadIdx.x & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const unsigned long long le_mask = lt_mask | (1ull << lane);
  const int n_list = list.ids ? (int)*list.count : 0;
  const int iters = list.ids ? 1 : gpb / BLOCK;
  for (int k = 0; k < iters; ++k) {
    int i = blockIdx.x * gpb + k * BLOCK + (int)threadIdx.x;
    if (list.ids) {               // slice work list (already filtered by depth bin): gpb / 4 consecutive ids per wave
      const int per_wave = gpb / (BLOCK / 64);
      const int j = blockIdx.x * gpb + (int)(threadIdx.x >> 6) * per_wave + lane;
      i = (lane < per_wave && j < n_list) ? (int)list.ids[j] : p.P;
    }
    BinG g;
    const bool live = load_bing(p, splats, radii, zbin, cut, sat, uv, i, g);
    const int w = g.x1 - g.x0;
    const uint32_t area = live ? (uint32_t)(w * (g.y1 - g.y0)) : 0u;
    const unsigned long long nz = __builtin_amdgcn_ballot_w64(area > 0u);
    if (nz == 0ull) continue;
    const uint32_t incl = wave_incl_scan_u32(area, lane);
    const uint32_t excl = incl - area;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    if (area > 0u) {
      const int rank = __popcll(nz & lt_mask);
      uint32_t* r = wb->rec + rank * GREC;
      r[0] = __float_as_uint(g.u); r[1] = __float_as_uint(g.v); r[2] = __float_as_uint(g.ca); r[3] = __float_as_uint(g.cb);
      r[4] = __float_as_uint(g.cc); r[5] = __float_as_uint(g.thr); r[6] = __float_as_uint(g.ica); r[7] = __float_as_uint(g.icc);
      r[8] = (uint32_t)g.x0 | ((uint32_t)g.y0 << 16); r[9] = (uint32_t)w; r[10] = (uint32_t)i; r[11] = g.zbits;
      wb->start[rank] = excl;
    }
    uint32_t started = 0;                          // Gaussians (dense ranks) whose first candidate lies before the round
    for (uint32_t it0 = 0; it0 < total; it0 += 64) {
      // three relaxed atomics on ONE LDS word: per-location coherence keeps them in program order
      if (lane == 0) __hip_atomic_store(&wb->head, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (area > 0u && excl >= it0 && excl < it0 + 64u)
        __hip_atomic_fetch_or(&wb->head, 1ull << (excl - it0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const unsigned long long hm = __hip_atomic_load(&wb->head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const uint32_t item = it0 + (uint32_t)lane;
      bool hit = false;
      int t = 0, owner = 0;
      if (item < total) {
        owner = (int)started + __popcll(hm & le_mask) - 1;
        const uint32_t* r = wb->rec + owner * GREC;
        BinG b;
        b.u = __uint_as_float(r[0]); b.v = __uint_as_float(r[1]); b.ca = __uint_as_float(r[2]); b.cb = __uint_as_float(r[3]);
        b.cc = __uint_as_float(r[4]); b.thr = __uint_as_float(r[5]); b.ica = __uint_as_float(r[6]); b.icc = __uint_as_float(r[7]);
        const uint32_t xy = r[8], bw = r[9];
        const uint32_t local = item - wb->start[owner];
        const int ry = (int)(((float)local + 0.5f) * __builtin_amdgcn_rcpf((float)bw));   // exact: margin 0.5/bw >> rcp error
        const int rx = (int)local - ry * (int)bw;
        const int tx = (int)(xy & 0xffffu) + rx, ty = (int)(xy >> 16) + ry;
        t = ty * p.gx + tx;
        hit = mask[t] != 0 && tile_visible(b, tx, ty);
      }
      f(hit, t, owner);                              // every lane of the wave calls: f may ballot
      started += (uint32_t)__popcll(hm);
    }
  }
}

__global__ void __launch_bounds__(256) bin_count_kernel(RasterParams p, const Splat* __restrict__ splats,
                                                        const int32_t* __restrict__ radii,
                                                        const int32_t* __restrict__ mask,
                                                        uint32_t* __restrict__ tile_count,
                                                        uint16_t* __restrict__ block_counts, SliceSel sel, SliceList list,
                                                        int gpb) {
  extern __shared__ uint32_t s_cnt[];
  __shared__ WaveBin s_wb[BLOCK / 64];
  const int ntiles = p.gx * p.gy;
  if (spec_failed(p.spec_fail)) return;              // speculative forward already known to be wrong: its lists are not valid
  if (sel.mode == 2 && sel.ctr[0] == 0u) return;     // the near slice finished every tile: nothing left to bin
  if (list.ids && blockIdx.x * gpb >= (int)*list.count) return;     // past the end of the slice work list
  for (int t = threadIdx.x; t < ntiles; t += BLOCK) s_cnt[t] = 0;
  __syncthreads();
  enumerate_balanced(p, splats, radii, mask, nullptr, 0, list, gpb, sel.mode == 2 ? sel.sat : nullptr, sel.uv,
                     &s_wb[threadIdx.x >> 6], [&](bool hit, int t, int) { if (hit) atomicAdd(&s_cnt[t], 1u); });
  __syncthreads();
  // the workgroup's row of per-tile counts is kept for bin_scatter (same Gaussian -> workgroup mapping),
  // which therefore needs only ONE enumeration sweep; a workgroup holds GPB <= 65535 Gaussians, so u16 fits
  uint16_t* row = block_counts + (size_t)blockIdx.x * ntiles;
  for (int t = threadIdx.x; t < ntiles; t += BLOCK) {
    const uint32_t c = s_cnt[t];                     // masked tiles were never counted
    row[t] = (uint16_t)c;
    if (c) atomicAdd(&tile_count[t], c);
  }
}

// one workgroup: ranges[t] = [start, end), cursor[t] = start, info = {R, longest list}
__global__ void __launch_bounds__(1024) bin_tilescan_kernel(int ntiles, const uint32_t* __restrict__ tile_count,
                                                            uint2* __restrict__ ranges, uint32_t* __restrict__ cursor,
                                                            uint32_t* __restrict__ info, uint32_t* __restrict__ info_host,
                                                            const uint32_t* __restrict__ extra_src,
                                                            const uint32_t* __restrict__ extra_src2,
                                                            const uint32_t* __restrict__ slot_a,
                                                            const uint32_t* __restrict__ slot_b, uint32_t seq,
                                                            SpecCaps caps) {
  __shared__ uint32_t s_sum[1024];
  __shared__ uint32_t s_max[1024];
  const int tid = threadIdx.x;
  const int per = (ntiles + 1023) / 1024;
  const int lo = min(ntiles, tid * per), hi = min(ntiles, lo + per);
  uint32_t sum = 0, mx = 0;
  for (int t = lo; t < hi; ++t) { const uint32_t c = tile_count[t]; sum += c; mx = max(mx, c); }
  s_sum[tid] = sum; s_max[tid] = mx;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {          // Hillis-Steele inclusive scan (10 steps, once per frame)
    const uint32_t a = (tid >= off) ? s_sum[tid - off] : 0u;
    const uint32_t m = (tid >= off) ? s_max[tid - off] : 0u;
    __syncthreads();
    s_sum[tid] += a; s_max[tid] = max(s_max[tid], m);
    __syncthreads();
  }
  uint32_t start = s_sum[tid] - sum;
  for (int t = lo; t < hi; ++t) {
    const uint32_t c = tile_count[t];
    ranges[t] = make_uint2(start, start + c);
    cursor[t] = start;
    start += c;
  }
  if (tid == 1023) {
    info[0] = s_sum[1023]; info[1] = s_max[1023];
    const uint32_t slots_total = (slot_a ? slot_a[0] : 0u) + (slot_b ? slot_b[0] : 0u);
    const int32_t cut = caps.cut ? *caps.cut : 0;
    if (caps.fail)      // speculative forward: do the sizes the host guessed hold?  (and, if it assumed so, was the slice declined?)
      *caps.fail = (*caps.fail != 0u || s_sum[1023] > caps.R || s_max[1023] > caps.longest || slots_total > caps.slots ||
                    (caps.cut && cut >= 0)) ? 1u : 0u;
    if (info_host) {   // pinned host words the forward's one host sync reads: no device-to-host copy launch in between
      const uint32_t w[7] = {s_sum[1023], s_max[1023], extra_src ? extra_src[0] : 0u, extra_src ? extra_src[1] : 0u,   // near-slice tile counters
                             extra_src2 ? extra_src2[0] : 0u,                                                  // near-slice instance total
                             // size of the backward's gradient-slot space: last exclusive-scan value + last rect area
                             slots_total, (uint32_t)cut};
      // publish: the host spins on the sequence word instead of paying a blocking stream sync's wake-up latency
      publish_to_host(info_host, w, seq);
    }
  }
}

// The workgroup reads the per-tile counts bin_count left for it, reserves its slots with ONE global atomic
// per touched tile, then places its instances in a single enumeration sweep (LDS atomics for the local rank).
// Measured alternatives on the 1.2 M scene: re-counting with a second sweep 1.8x slower; one returning global
// atomic per instance 2.9x slower (7.9 M atomics on 3 225 cursors).
__global__ void __launch_bounds__(256) bin_scatter_kernel(RasterParams p, const Splat* __restrict__ splats,
                                                          const int32_t* __restrict__ radii,
                                                          const int32_t* __restrict__ mask,
                                                          const uint16_t* __restrict__ block_counts,
                                                          uint32_t* __restrict__ cursor,
                                                          unsigned long long* __restrict__ bucket, SliceSel sel,
                                                          SliceList list, int gpb) {
  extern __shared__ uint32_t s_mem[];
  __shared__ WaveBin s_wb[BLOCK / 64];
  const int ntiles = p.gx * p.gy;
  if (spec_failed(p.spec_fail)) return;            // the bucket array was sized by a guess that did not hold
  if (list.ids && blockIdx.x * gpb >= (int)*list.count) return;
  uint32_t* s_cnt = s_mem;
  uint32_t* s_base = s_mem + ntiles;
  const uint16_t* row = block_counts + (size_t)blockIdx.x * ntiles;
  for (int t = threadIdx.x; t < ntiles; t += BLOCK) {
    const uint32_t c = row[t];                       // 0 for masked tiles
    s_base[t] = c ? atomicAdd(&cursor[t], c) : 0xffffffffu;
    s_cnt[t] = 0;
  }
  __syncthreads();
  enumerate_balanced(p, splats, radii, mask, nullptr, 0, list, gpb, sel.mode == 2 ? sel.sat : nullptr, sel.uv,
                     &s_wb[threadIdx.x >> 6], [&](bool hit, int t, int owner) {
    const uint32_t base = hit ? s_base[t] : 0xffffffffu;
    if (base != 0xffffffffu) {
      const uint32_t* r = s_wb[threadIdx.x >> 6].rec + owner * GREC;
      const uint32_t slot = base + atomicAdd(&s_cnt[t], 1u);
      bucket[slot] = ((unsigned long long)r[11] << 32) | r[10];
    }
  });
}

// ---------------------------------------------------------------------------------------------
// ONE-PASS placement into per-tile SEGMENTS (round 4): tile t owns bucket[t * seg, (t + 1) * seg) - the near slice's
// lists are capped at SLICE_MAX_LIST anyway (a longer one leaves its tile to pass 2), the main pass of a speculative
// forward takes the sort-class capacity of the last verified longest list - so nothing has to be counted and scanned
// before a key can be placed: bin_count, bin_tilescan and the per-workgroup count rows are not launched.
// A workgroup enumerates its Gaussians ONCE: every hit takes its rank inside the workgroup from an LDS counter of its
// tile and is parked in the wave's staging area as (owner, tile, rank) - 4 bytes; then one global atomic per touched
// tile reserves the workgroup's run of the segment, and the parked hits are written out.  A wave whose staging area
// is full places the rest with one global atomic per hit (both forms reserve on the same counter).
// count[t] ends as the tile's TRUE number of hits; keys beyond the segment are dropped (near slice: the tile is left
// to pass 2, as before; main pass: `fail` is raised and the host redoes the call with exact sizes).
// ---------------------------------------------------------------------------------------------
struct SegBins { uint32_t* count; unsigned long long* bucket; uint32_t seg; uint32_t* fail; };
constexpr int STAGE_W = 1024;                     // parked hits per wave

__global__ void __launch_bounds__(256) bin_place_kernel(RasterParams p, const Splat* __restrict__ splats,
                                                        const int32_t* __restrict__ radii,
                                                        const int32_t* __restrict__ mask, SegBins sb, SliceSel sel,
                                                        SliceList list, int gpb, int stage_cap) {
  extern __shared__ uint32_t s_mem[];
  __shared__ WaveBin s_wb[BLOCK / 64];
  __shared__ uint32_t s_stage[BLOCK / 64][STAGE_W];
  const int ntiles = p.gx * p.gy;
  if (spec_failed(p.spec_fail)) return;
  if (sel.mode == 2 && sel.ctr[0] == 0u) return;
  if (list.count && blockIdx.x * gpb >= (int)*list.count) return;       // no list: every Gaussian (a forward without a backward on a small map)
  uint32_t* s_cnt = s_mem;
  for (int t = threadIdx.x; t < ntiles; t += BLOCK) s_cnt[t] = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  uint32_t* stage = s_stage[w];
  const uint32_t* rec = s_wb[w].rec;
  uint32_t parked = 0;                            // wave-uniform
  enumerate_balanced(p, splats, radii, mask, nullptr, 0, list, gpb, sel.mode == 2 ? sel.sat : nullptr, sel.uv,
                     &s_wb[w], [&](bool hit, int t, int owner) {
    const unsigned long long m = __builtin_amdgcn_ballot_w64(hit);
    if (m == 0ull) return;
    const uint32_t pos = parked + (uint32_t)__popcll(m & lt_mask);
    if (hit) {
      if (pos < (uint32_t)stage_cap) {
        const uint32_t rank = atomicAdd(&s_cnt[t], 1u);              // < gpb <= 256: a Gaussian meets a tile once
        stage[pos] = (uint32_t)owner | ((uint32_t)t << 6) | (rank << 20);
      } else {
        const uint32_t slot = atomicAdd(&sb.count[t], 1u);
        const uint32_t* r = rec + owner * GREC;
        if (slot < sb.seg) sb.bucket[(size_t)t * sb.seg + slot] = ((unsigned long long)r[11] << 32) | r[10];
        else if (sb.fail) *sb.fail = 1u;
      }
    }
    parked += (uint32_t)__popcll(m);
  });
  __syncthreads();
  // reservations: eight returning atomics in flight per thread (issued one at a time each costs a round trip to the
  // memory side: 13 of them per thread were most of this kernel)
  for (int t0 = threadIdx.x; t0 < ntiles; t0 += BLOCK * 8) {
    uint32_t c[8], b[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int t = t0 + k * BLOCK; c[k] = t < ntiles ? s_cnt[t] : 0u; }
#pragma unroll
    for (int k = 0; k < 8; ++k) b[k] = c[k] ? atomicAdd(&sb.count[t0 + k * BLOCK], c[k]) : 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (c[k]) {
        s_cnt[t0 + k * BLOCK] = b[k];
        if (b[k] + c[k] > sb.seg && sb.fail) *sb.fail = 1u;
      }
  }
  __syncthreads();
  const uint32_t n = min(parked, (uint32_t)stage_cap);
  for (uint32_t k = lane; k < n; k += 64) {
    const uint32_t e = stage[k];
    const uint32_t t = (e >> 6) & 0x3fffu;
    const uint32_t slot = s_cnt[t] + (e >> 20);
    const uint32_t* r = rec + (e & 63u) * GREC;
    if (slot < sb.seg) sb.bucket[(size_t)t * sb.seg + slot] = ((unsigned long long)r[11] << 32) | r[10];
  }
}

// Segment mode (one-pass placement): the tile's range is [tile * seg, + count[tile]); the first sort launch - it
// visits every tile - also writes it where the blends and the backward read it.  count may exceed seg (keys beyond were
// dropped): no size class takes such a tile, and the consumers treat it as they did before (slice: left to pass 2).
struct TileSeg { const uint32_t* count; uint32_t seg; uint2* ranges_out; };
__device__ __forceinline__ uint2 tile_range(const uint2* __restrict__ ranges, const TileSeg& ts, bool writer) {
  if (!ts.count) return ranges[blockIdx.x];
  const uint32_t x = blockIdx.x * ts.seg;
  const uint2 r = make_uint2(x, x + ts.count[blockIdx.x]);
  if (writer && ts.ranges_out) ts.ranges_out[blockIdx.x] = r;
  return r;
}

// What bin_tilescan's last thread did, for a forward that placed its instances in one pass and never scanned: totals of
// the per-tile counts, the speculative forward's capacity check, the words the host's verify waits for.  Run by ONE EXTRA
// workgroup of the first sort launch (128 threads, ~25 counts each) beside the tiles' sorts - no launch of its own.
__device__ __forceinline__ void bin_finish(const BinFinish& f) {
  __shared__ uint32_t s_sum[2], s_max[2];
  uint32_t sum = 0, mx = 0;
  for (int t = threadIdx.x; t < f.ntiles; t += 128) { const uint32_t c = f.count[t]; sum += c; mx = max(mx, c); }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { sum += (uint32_t)__shfl_xor((int)sum, off); mx = max(mx, (uint32_t)__shfl_xor((int)mx, off)); }
  if ((threadIdx.x & 63) == 0) { s_sum[threadIdx.x >> 6] = sum; s_max[threadIdx.x >> 6] = mx; }
  __syncthreads();
  if (threadIdx.x != 0) return;
  const uint32_t R = s_sum[0] + s_sum[1], longest = max(s_max[0], s_max[1]);
  f.info[0] = R; f.info[1] = longest;
  const uint32_t slots_total = (f.slot_a ? f.slot_a[0] : 0u) + (f.slot_b ? f.slot_b[0] : 0u);
  const int32_t cut = f.caps.cut ? *f.caps.cut : 0;
  if (f.caps.fail && (R > f.caps.R || longest > f.caps.longest || slots_total > f.caps.slots || (f.caps.cut && cut >= 0)))
    *f.caps.fail = 1u;                           // only ever raised here: bin_place may have raised it already
  if (f.info_host) {
    const uint32_t w[7] = {R, longest, f.listed ? f.listed[0] : 0u, 0u, 0u, slots_total, (uint32_t)cut};
    publish_to_host(f.info_host, w, f.seq);
  }
}

// one workgroup per tile; keys (depth bits << 32 | id) are unique, so the order is total
// Launched once per size class [lo, hi): a workgroup whose tile is outside its class exits at
// once, so every class runs with the LDS footprint (and occupancy) its lists need.
template <int THREADS>
__global__ void __launch_bounds__(THREADS) bin_tilesort_kernel(const uint2* __restrict__ ranges,
                                                               const unsigned long long* __restrict__ bucket,
                                                               uint32_t* __restrict__ point_list, int lo, int hi,
                                                               const uint32_t* __restrict__ spec_fail, TileSeg ts,
                                                               BinFinish fin) {
  extern __shared__ unsigned long long s_key[];
  if (fin.count && (int)blockIdx.x == fin.ntiles) { bin_finish(fin); return; }     // also after a failed guess: the host waits for it
  if (spec_failed(spec_fail)) return;
  const uint2 r = tile_range(ranges, ts, threadIdx.x == 0);
  const int n = (int)(r.y - r.x);
  if (n < lo || n >= hi) return;
  const int tid = threadIdx.x;
  if (n == 1) { if (tid == 0) point_list[r.x] = (uint32_t)bucket[r.x]; return; }
  int n2 = 2;
  while (n2 < n) n2 <<= 1;
  for (int i = tid; i < n2; i += THREADS) s_key[i] = (i < n) ? bucket[r.x + i] : ~0ull;
  __syncthreads();
  for (int k = 2; k <= n2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < (n2 >> 1); i += THREADS) {
        const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));     // index with bit j clear
        const int b = a | j;
        const unsigned long long ka = s_key[a], kb = s_key[b];
        const bool up = (a & k) == 0;
        if ((ka > kb) == up) { s_key[a] = kb; s_key[b] = ka; }
      }
      __syncthreads();
    }
  }
  for (int i = tid; i < n; i += THREADS) point_list[r.x + i] = (uint32_t)s_key[i];
}

// ---------------------------------------------------------------------------------------------
// LDS radix sort of one tile bucket: LSD over the bytes of the depth bits that actually vary inside
// the tile (usually 3 of 4), stable within a pass (wave-ordered segments, ballot ranks), then a
// parallel fix-up that orders runs of bit-equal depths by Gaussian id.  ~6 LDS operations per key and pass instead of the
// ~1.5 x 78 stages of the bitonic network.
// ---------------------------------------------------------------------------------------------
template <int THREADS>
__global__ void __launch_bounds__(THREADS) bin_tilesort_radix_kernel(const uint2* __restrict__ ranges,
                                                                     const unsigned long long* __restrict__ bucket,
                                                                     uint32_t* __restrict__ point_list, int lo, int hi,
                                                                     int cap, const uint32_t* __restrict__ spec_fail,
                                                                     TileSeg ts) {
  constexpr int NW = THREADS / 64;
  extern __shared__ unsigned long long s_dyn[];
  unsigned long long* kA = s_dyn;
  unsigned long long* kB = s_dyn + cap;
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(s_dyn + 2 * cap);      // [NW][256]
  uint32_t* s_tot = s_cnt + NW * 256;                                   // [256]
  __shared__ uint32_t s_or;
  if (spec_failed(spec_fail)) return;
  const uint2 r = tile_range(ranges, ts, false);
  const int n = (int)(r.y - r.x);
  if (n < lo || n >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) s_or = 0;
  __syncthreads();
  const uint32_t z0 = (uint32_t)(bucket[r.x] >> 32);
  uint32_t vary = 0;
  for (int i = tid; i < n; i += THREADS) {
    const unsigned long long k = bucket[r.x + i];
    kA[i] = k;
    vary |= (uint32_t)(k >> 32) ^ z0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) vary |= (uint32_t)__shfl_xor((int)vary, off);
  if (lane == 0 && vary) atomicOr(&s_or, vary);
  __syncthreads();
  const uint32_t vbits = s_or;
  const int seg = ((n + NW - 1) / NW + 63) & ~63;        // keys per wave, multiple of 64
  const int wlo = min(n, w * seg), whi = min(n, wlo + seg);
  for (int pass = 0; pass < 4; ++pass) {
    if (((vbits >> (8 * pass)) & 0xffu) == 0) continue;   // this byte is constant inside the tile
    const int shift = 32 + 8 * pass;
    for (int i = tid; i < NW * 256; i += THREADS) s_cnt[i] = 0;
    __syncthreads();
    for (int i = wlo + lane; i < whi; i += 64) atomicAdd(&s_cnt[w * 256 + (int)((kA[i] >> shift) & 0xff)], 1u);
    __syncthreads();
    if (tid < 256) {                                      // per digit: exclusive prefix over waves + digit total
      uint32_t run = 0;
#pragma unroll
      for (int ww = 0; ww < NW; ++ww) { const uint32_t c = s_cnt[ww * 256 + tid]; s_cnt[ww * 256 + tid] = run; run += c; }
      s_tot[tid] = run;
    }
    __syncthreads();
    if (tid < 64) {                                       // exclusive scan of the 256 digit totals by one wave
      uint32_t v0 = s_tot[4 * tid], v1 = s_tot[4 * tid + 1], v2 = s_tot[4 * tid + 2], v3 = s_tot[4 * tid + 3];
      const uint32_t mine = v0 + v1 + v2 + v3;
      uint32_t inc = mine;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (tid >= off) inc += o;
      }
      const uint32_t ex = inc - mine;
      s_tot[4 * tid] = ex; s_tot[4 * tid + 1] = ex + v0; s_tot[4 * tid + 2] = ex + v0 + v1; s_tot[4 * tid + 3] = ex + v0 + v1 + v2;
    }
    __syncthreads();
    for (int i0 = wlo; i0 < whi; i0 += 64) {
      const int i = i0 + lane;
      const bool act = i < whi;
      const unsigned long long k = act ? kA[i] : 0ull;
      const int d = (int)((k >> shift) & 0xff);
      unsigned long long same = __builtin_amdgcn_ballot_w64(act);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64((d >> bit) & 1);
        same &= ((d >> bit) & 1) ? m : ~m;
      }
      if (act) {
        const unsigned long long below = same & ((1ull << lane) - 1ull);
        const uint32_t base = s_cnt[w * 256 + d] + s_tot[d];
        kB[base + (uint32_t)__popcll(below)] = k;
        if ((same >> lane) == 1ull) s_cnt[w * 256 + d] += (uint32_t)__popcll(same);   // highest lane of the group
      }
    }
    __syncthreads();
    unsigned long long* tmp = kA; kA = kB; kB = tmp;
  }
  // Runs of bit-equal depth (the bucket was filled by atomics, so equal depths arrive in arbitrary order): order them
  // by Gaussian id.  Usually there is none.  Otherwise every key finds its run with two binary searches over the
  // depth-sorted array and its rank inside the run by counting the smaller ids (broadcast LDS reads) - O(run) per key
  // and fully parallel, whatever the run length (a wall seen head-on puts hundreds of bit-equal depths into one tile).
  bool tie = false;
  for (int i = tid + 1; i < n; i += THREADS) tie |= (uint32_t)(kA[i] >> 32) == (uint32_t)(kA[i - 1] >> 32);
  if (__syncthreads_or(tie)) {
    for (int i = tid; i < n; i += THREADS) {
      const unsigned long long key = kA[i];
      const uint32_t z = (uint32_t)(key >> 32);
      int lo = 0, hi = i;                       // first index with depth == z
      while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)(kA[mid] >> 32) < z) lo = mid + 1; else hi = mid; }
      const int first = lo;
      lo = i; hi = n;                           // one past the last index with depth == z
      while (lo < hi) { const int mid = (lo + hi) >> 1; if ((uint32_t)(kA[mid] >> 32) <= z) lo = mid + 1; else hi = mid; }
      int rank = first;
      if (lo - first > 1)
        for (int j = first; j < lo; ++j) rank += (kA[j] < key) ? 1 : 0;
      else
        rank = i;
      kB[rank] = key;
    }
    __syncthreads();
    unsigned long long* tmp = kA; kA = kB; kB = tmp;
  }
  for (int i = tid; i < n; i += THREADS) point_list[r.x + i] = (uint32_t)kA[i];
}

// ------------------------------------------------------------------------------ launchers
size_t bin_lds_limit_tiles() { return 16000; }       // 2 x 4 B x tiles must fit the 160 KiB LDS
int bin_sort_capacity() { return 16384; }            // 16384 x 8 B = 128 KiB

// Gaussians per workgroup: the whole map by index -> GPB; the near slice's short list (<= 65 536 ids, each covering tens
// of tiles) -> small work items, many workgroups; the list of every visible Gaussian -> one wave's worth per wave
#ifndef RTGS_LIST_GPB
#define RTGS_LIST_GPB 256
#endif
constexpr int LIST_GPB = RTGS_LIST_GPB;
static inline int list_gpb(const SliceList& list, size_t max_items) {
  return !list.ids ? GPB : (max_items > 65536 ? LIST_GPB : SLICE_GPB);
}
size_t bin_list_block_counts_bytes(int P, int ntiles) {          // rows of the visible-list mode
  const size_t rows = P > 65536 ? ((size_t)P + LIST_GPB - 1) / LIST_GPB : ((size_t)P + SLICE_GPB - 1) / SLICE_GPB;
  return rows * (size_t)ntiles * sizeof(uint16_t);
}
size_t bin_block_counts_bytes(int P, int ntiles) {
  return (size_t)((P + GPB - 1) / GPB) * (size_t)ntiles * sizeof(uint16_t);
}
// rows of the slice pass: the work list holds at most slice_cap / 1 Gaussians, but never more than P
size_t bin_slice_block_counts_bytes(int P, size_t max_list, int ntiles) {
  const size_t n = max_list < (size_t)P ? max_list : (size_t)P;
  return ((n + SLICE_GPB - 1) / SLICE_GPB) * (size_t)ntiles * sizeof(uint16_t);
}

// depth histograms of the visible Gaussians (input of slice_cut): 3-sigma-rect instances, Gaussians, and
// sum(radius^2) per bin (the "cover" the automatic mode's decision uses)
__global__ void __launch_bounds__(256) slice_hist_kernel(int P, const uint8_t* __restrict__ zbin,
                                                         const uint32_t* __restrict__ rect_area,
                                                         const int32_t* __restrict__ radii,
                                                         uint32_t* __restrict__ hist,
                                                         unsigned long long* __restrict__ cover, BwdInfoInit bi) {
  __shared__ uint32_t s_h[SLICE_BINS], s_c[SLICE_BINS], s_r[SLICE_BINS];
  if (bi.dst && blockIdx.x == 0 && threadIdx.x == 0) {     // host-known words of the backward (was a kernel of its own)
    bi.dst->slot_grads = bi.slot_grads; bi.dst->slots = bi.slots; bi.dst->use_slots = bi.use_slots;
  }
  s_h[threadIdx.x] = 0; s_c[threadIdx.x] = 0; s_r[threadIdx.x] = 0;
  __syncthreads();
  // 8 consecutive Gaussians per thread, all loads in flight at once (the grid-stride form was a chain of
  // dependent 1-byte loads: 19 us for 6 MB).  radius^2 is capped at 2^16 per Gaussian: a workgroup sums at most
  // ~10^4 of them into a 32-bit LDS word.
  for (int i0 = (blockIdx.x * 256 + threadIdx.x) * 8; i0 < P; i0 += gridDim.x * 2048)
  if (i0 + 8 <= P) {
    const uint2 zb = *reinterpret_cast<const uint2*>(zbin + i0);
    const uint4 a0 = *reinterpret_cast<const uint4*>(rect_area + i0);
    const uint4 a1 = *reinterpret_cast<const uint4*>(rect_area + i0 + 4);
    const int4 r0 = *reinterpret_cast<const int4*>(radii + i0);
    const int4 r1 = *reinterpret_cast<const int4*>(radii + i0 + 4);
    const uint32_t ar[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
    const int rr[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t bsel = ((k < 4 ? zb.x : zb.y) >> (8 * (k & 3))) & 0xffu;
      if (bsel != 255u) {
        const uint32_t r = (uint32_t)min(rr[k], 256);
        atomicAdd(&s_h[bsel], ar[k]); atomicAdd(&s_c[bsel], 1u); atomicAdd(&s_r[bsel], r * r);
      }
    }
  } else {
    for (int i = i0; i < P; ++i) {
      const uint32_t bsel = zbin[i];
      if (bsel != 255u) {
        const uint32_t r = (uint32_t)min(radii[i], 256);
        atomicAdd(&s_h[bsel], rect_area[i]); atomicAdd(&s_c[bsel], 1u); atomicAdd(&s_r[bsel], r * r);
      }
    }
  }
  __syncthreads();
  const uint32_t v = s_h[threadIdx.x], n = s_c[threadIdx.x];
  if (n) {
    atomicAdd(&hist[threadIdx.x], v); atomicAdd(&hist[SLICE_BINS + threadIdx.x], n);
    atomicAdd(&cover[threadIdx.x], (unsigned long long)s_r[threadIdx.x]);
  }
}
void launch_slice_hist(int P, const uint8_t* zbin, const uint32_t* rect_area, const int32_t* radii, uint32_t* hist,
                       unsigned long long* cover, hipStream_t st, BwdInfoInit bi) {
  if (P == 0) return;
  int blocks = (P + 2047) / 2048;       // few workgroups: each ends with up to 256 same-address global atomics
  if (blocks > 128) blocks = 128;
  hipLaunchKernelGGL(slice_hist_kernel, dim3(blocks), dim3(256), 0, st, P, zbin, rect_area, radii, hist, cover, bi);
}

// ids of the Gaussians in the near slice (depth bin <= cut), in arbitrary order (the tile sort orders by (depth, id));
// count in *n_list.  Each workgroup stages the ids of its contiguous chunk in LDS and reserves its output range with
// ONE global atomic (one atomic per wave serialised on a single address: 225 us for 1.2 M Gaussians).
constexpr int COMPACT_CHUNK = 8192;
__global__ void __launch_bounds__(256) slice_compact_kernel(int P, SliceSel sel, uint32_t* __restrict__ ids,
                                                            uint32_t* __restrict__ n_list,
                                                            const uint32_t* __restrict__ rect_area,
                                                            uint32_t* __restrict__ gbase, uint32_t* __restrict__ slot_cursor,
                                                            uint32_t* __restrict__ host, uint32_t seq,
                                                            uint32_t* __restrict__ fail_if_taken) {
  __shared__ uint32_t s_ids[COMPACT_CHUNK];
  __shared__ uint32_t s_n, s_base, s_tot, s_gb, s_run, s_w[4];
  const int cut = slice_cut(sel);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    reinterpret_cast<int32_t*>(n_list)[1] = cut;   // slice_ctr[3]
    // speculative forward that assumed "declined": raise the word NOW - the kernels the host already queued for the
    // declined single pass (visible_compact, shade, bin_count) share counter words with the slice and must not run
    if (fail_if_taken && cut >= 0) *fail_if_taken = 1u;
    if (host) {   // the host asked to hear the decision before it launches the slice's kernels (raster_api.hip)
      const uint32_t w[7] = {0u, 0u, 0u, 0u, 0u, 0u, (uint32_t)cut};
      publish_to_host(host, w, seq);
    }
  }
  if (threadIdx.x == 0) { s_n = 0; s_tot = 0; s_run = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int begin = blockIdx.x * COMPACT_CHUNK;
  uint32_t zall[COMPACT_CHUNK / 1024];             // the chunk's depth bins, all loads in flight before the first ballot
#pragma unroll
  for (int j = 0; j < COMPACT_CHUNK / 1024; ++j) {
    const int i = begin + j * 1024 + (int)threadIdx.x * 4;
    uint32_t zb4 = 0xffffffffu;
    if (i + 4 <= P) zb4 = *reinterpret_cast<const uint32_t*>(sel.zbin + i);
    else for (int k = 0; k < 4; ++k) if (i + k < P) zb4 = (zb4 & ~(0xffu << (8 * k))) | ((uint32_t)sel.zbin[i + k] << (8 * k));
    zall[j] = zb4;
  }
#pragma unroll
  for (int j = 0; j < COMPACT_CHUNK / 1024; ++j) {
    const int i = begin + j * 1024 + (int)threadIdx.x * 4;
    const uint32_t zb4 = zall[j];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int zb = (int)((zb4 >> (8 * k)) & 0xffu);
      const bool in = zb != 255 && zb <= cut;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
      if (m == 0ull) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (in) s_ids[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(i + k);
    }
  }
  __syncthreads();
  const uint32_t n = s_n;
  if (n == 0u) return;
  // The slice's Gaussians also get their runs of gradient slots here (BwdInfo: one slot per tile of the rect): the
  // workgroup reserves the sum of its rect areas with one atomic and lays its Gaussians out inside - if the slice
  // finishes every tile these are the only Gaussians in play and no scan over the map is needed (the total is within
  // the slice's instance budget by construction of the cut); otherwise the full scan overwrites gbase later.
  uint32_t mine = 0;
  for (uint32_t k = threadIdx.x; k < n; k += 256) mine += rect_area[s_ids[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
  if (lane == 0) atomicAdd(&s_tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) { s_base = atomicAdd(n_list, n); s_gb = atomicAdd(slot_cursor, s_tot); }
  __syncthreads();
  for (uint32_t k0 = 0; k0 < n; k0 += 256) {
    const uint32_t k = k0 + threadIdx.x;
    const uint32_t id = k < n ? s_ids[k] : 0u;
    const uint32_t a = k < n ? rect_area[id] : 0u;
    uint32_t incl = a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t pre = s_run;
    for (int q = 0; q < w; ++q) pre += s_w[q];
    if (k < n) { gbase[id] = s_gb + pre + incl - a; ids[s_base + k] = id; }
    __syncthreads();
    if (threadIdx.x == 0) s_run += (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
  }
}
void launch_slice_compact(int P, SliceSel sel, uint32_t* ids, uint32_t* n_list, const uint32_t* rect_area, uint32_t* gbase,
                          uint32_t* slot_cursor, uint32_t* host, uint32_t seq, uint32_t* fail_if_taken, hipStream_t st) {
  if (P == 0) return;
  hipLaunchKernelGGL(slice_compact_kernel, dim3((P + COMPACT_CHUNK - 1) / COMPACT_CHUNK), dim3(256), 0, st, P, sel, ids,
                     n_list, rect_area, gbase, slot_cursor, host, seq, fail_if_taken);
}

// Every visible Gaussian (zbin != 255), compacted into a work list: when the near slice is declined and the host knows
// it, the single pass that follows shades / counts / scatters through this list with dense lanes instead of sweeping
// all P Gaussians with one lane in six active (a surface map shows ~18 % of its Gaussians to a view).
constexpr int VIS_CHUNK = 8192;      // ids per workgroup at most: few workgroups - each ends in ONE same-address atomic (~25 ns, serialised)
// `chunk` (a multiple of 1024, <= VIS_CHUNK) is chosen per launch: at SLAM sizes (100-300 k Gaussians) 8192 ids per workgroup
// left 12-36 workgroups on 256 CUs, each a serial chain of 8 ballot rounds and 32 scan rounds - 29.5 us per call, 22 000 calls in
// the 2 000-frame sequence (profiles/r06_sequence_*): the third-largest kernel of a SLAM frame
__global__ void __launch_bounds__(256) visible_compact_kernel(int P, int chunk, const uint8_t* __restrict__ zbin, uint32_t* __restrict__ ids,
                                                              uint32_t* __restrict__ n_list,
                                                              const uint32_t* __restrict__ rect_area,
                                                              uint32_t* __restrict__ gbase, uint32_t* __restrict__ slot_cursor,
                                                              const uint32_t* __restrict__ spec_fail, SliceSel sel,
                                                              int32_t* __restrict__ cut_out, uint32_t* __restrict__ fail_if_taken) {
  __shared__ uint32_t s_ids[VIS_CHUNK];
  if (spec_failed(spec_fail)) return;
  if (cut_out) {
    // Speculative forward that assumed "the kernels decline the near slice" (raster_api.hip, plan kind 2): the decision is
    // re-derived HERE from this call's histograms (every workgroup, as slice_compact does) instead of by a slice_compact
    // launch in front of this kernel - 6 us + a kernel boundary per map iteration on a SLAM-sized map, where the slice is
    // declined every time.  If the slice WOULD run, the guess was wrong: raise the word and touch nothing.
    const int cut = slice_cut(sel);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      *cut_out = cut;
      if (cut >= 0) *fail_if_taken = 1u;
    }
    if (cut >= 0) return;
  }
  __shared__ uint32_t s_n, s_base, s_tot, s_gb, s_run, s_w[4];
  if (threadIdx.x == 0) { s_n = 0; s_tot = 0; s_run = 0; }
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int begin = blockIdx.x * chunk;
  for (int i0 = begin; i0 < begin + chunk && i0 < P; i0 += 256 * 4) {
    const int i = i0 + (int)threadIdx.x * 4;
    uint32_t zb4 = 0xffffffffu;
    if (i + 4 <= P) zb4 = *reinterpret_cast<const uint32_t*>(zbin + i);
    else for (int k = 0; k < 4; ++k) if (i + k < P) zb4 = (zb4 & ~(0xffu << (8 * k))) | ((uint32_t)zbin[i + k] << (8 * k));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool in = ((zb4 >> (8 * k)) & 0xffu) != 255u;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(in);
      if (m == 0ull) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (in) s_ids[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)(i + k);
    }
  }
  __syncthreads();
  const uint32_t n = s_n;
  if (n == 0u) return;
  // gradient-slot runs of the listed Gaussians (one slot per tile of the rect), as slice_compact lays them out for
  // the slice: one atomic per workgroup reserves the chunk's total, a scan inside places the runs - replaces a
  // scan over the whole map (gbase is only ever read for Gaussians some tile lists, i.e. visible ones)
  uint32_t mine = 0;
  for (uint32_t k = threadIdx.x; k < n; k += 256) mine += rect_area[s_ids[k]];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mine += (uint32_t)__shfl_xor((int)mine, off);
  if (lane == 0) atomicAdd(&s_tot, mine);
  __syncthreads();
  if (threadIdx.x == 0) {
    // list length and slot cursor are neighbours (slot_cursor = n_list + 1, n_list 8-byte aligned): ONE 64-bit atomic for
    // both; the cursor sits in the high word, so a (pathological) overflow of the slot total cannot reach the count
    const unsigned long long old = atomicAdd(reinterpret_cast<unsigned long long*>(n_list),
                                             ((unsigned long long)s_tot << 32) | (unsigned long long)n);
    s_base = (uint32_t)old; s_gb = (uint32_t)(old >> 32);
  }
  __syncthreads();
  for (uint32_t k0 = 0; k0 < n; k0 += 256) {
    const uint32_t k = k0 + threadIdx.x;
    const uint32_t id = k < n ? s_ids[k] : 0u;
    const uint32_t a = k < n ? rect_area[id] : 0u;
    uint32_t incl = a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t t = (uint32_t)__shfl_up((int)incl, off);
      if (lane >= off) incl += t;
    }
    if (lane == 63) s_w[w] = incl;
    __syncthreads();
    uint32_t pre = s_run;
    for (int q = 0; q < w; ++q) pre += s_w[q];
    if (k < n) { gbase[id] = s_gb + pre + incl - a; ids[s_base + k] = id; }
    __syncthreads();
    if (threadIdx.x == 0) s_run += (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
    __syncthreads();
  }
}
void launch_visible_compact(int P, const uint8_t* zbin, uint32_t* ids, uint32_t* n_list, const uint32_t* rect_area,
                            uint32_t* gbase, uint32_t* slot_cursor, const uint32_t* spec_fail, hipStream_t st,
                            const SliceSel* decide, int32_t* cut_out, uint32_t* fail_if_taken) {
  if (P == 0) return;
  const SliceSel nosel{};
  const SliceSel sel = decide ? *decide : nosel;
  if (!decide) { cut_out = nullptr; fail_if_taken = nullptr; }
  // ~192+ workgroups below a million Gaussians, VIS_CHUNK ids each above
  int chunk = VIS_CHUNK;
  if (P < 1000000) { chunk = ((P / 192 + 1023) / 1024) * 1024; chunk = chunk < 1024 ? 1024 : (chunk > VIS_CHUNK ? VIS_CHUNK : chunk); }
  hipLaunchKernelGGL(visible_compact_kernel, dim3((P + chunk - 1) / chunk), dim3(256), 0, st, P, chunk, zbin, ids, n_list,
                     rect_area, gbase, slot_cursor, spec_fail, sel, cut_out, fail_if_taken);
}

int launch_bin_count(const RasterParams& p, const Splat* splats, const int32_t* radii, const int32_t* mask,
                     uint32_t* tile_count, uint16_t* block_counts, SliceSel sel, SliceList list, size_t max_items,
                     hipStream_t st) {
  const int ntiles = p.gx * p.gy;      // tile_count was cleared by preprocess_fwd
  if (p.P == 0) return 0;
  const size_t lds = (size_t)ntiles * sizeof(uint32_t);
  static LdsGrant grant(48 * 1024);
  grant.ensure((const void*)bin_count_kernel, lds);
  const int gpb = list_gpb(list, max_items);
  const size_t n = list.ids && max_items < (size_t)p.P ? max_items : (size_t)p.P;
  hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned)((n + gpb - 1) / gpb)), dim3(BLOCK), lds, st, p, splats, radii, mask,
                     tile_count, block_counts, sel, list, gpb);
  return 0;
}
void launch_bin_tilescan(int ntiles, const uint32_t* tile_count, uint2* ranges, uint32_t* cursor, uint32_t* info,
                         uint32_t* info_host, const uint32_t* extra_src, const uint32_t* extra_src2,
                         const uint32_t* slot_a, const uint32_t* slot_b, uint32_t seq, SpecCaps caps, hipStream_t st) {
  hipLaunchKernelGGL(bin_tilescan_kernel, dim3(1), dim3(1024), 0, st, ntiles, tile_count, ranges, cursor, info,
                     info_host, extra_src, extra_src2, slot_a, slot_b, seq, caps);
}
void launch_bin_scatter(const RasterParams& p, const Splat* splats, const int32_t* radii, const int32_t* mask,
                        const uint16_t* block_counts, uint32_t* cursor, unsigned long long* bucket, SliceSel sel,
                        SliceList list, size_t max_items, hipStream_t st) {
  if (p.P == 0) return;
  const int ntiles = p.gx * p.gy;
  const size_t lds = 2 * (size_t)ntiles * sizeof(uint32_t);
  static LdsGrant grant(32 * 1024);
  grant.ensure((const void*)bin_scatter_kernel, lds);
  const int gpb = list_gpb(list, max_items);
  const size_t n = list.ids && max_items < (size_t)p.P ? max_items : (size_t)p.P;
  hipLaunchKernelGGL(bin_scatter_kernel, dim3((unsigned)((n + gpb - 1) / gpb)), dim3(BLOCK), lds, st, p, splats, radii,
                     mask, block_counts, cursor, bucket, sel, list, gpb);
}
void launch_bin_place(const RasterParams& p, const Splat* splats, const int32_t* radii, const int32_t* mask,
                      uint32_t* seg_count, unsigned long long* bucket, uint32_t seg, uint32_t* fail, SliceSel sel,
                      SliceList list, size_t max_items, hipStream_t st) {
  if (p.P == 0) return;
  const int ntiles = p.gx * p.gy;
  const size_t lds = (size_t)ntiles * sizeof(uint32_t);
  static LdsGrant grant(24 * 1024);
  grant.ensure((const void*)bin_place_kernel, lds);
  static const int stage_cap = [] { const char* e = getenv("RTGS_BIN_STAGE"); const int v = e ? atoi(e) : STAGE_W;
                                    return v < 0 ? 0 : (v > STAGE_W ? STAGE_W : v); }();
  const int gpb = list_gpb(list, max_items);
  const size_t n = max_items < (size_t)p.P ? max_items : (size_t)p.P;
  // parked hits name their owner's record in the wave's area (6 bits), the tile (14 bits) and the rank: valid only while
  // a wave handles ONE 64-Gaussian batch (a work list with gpb <= BLOCK) and tiles fit 14 bits - otherwise place every
  // hit with its own atomic (always correct)
  const int stage = (list.ids != nullptr && gpb <= BLOCK && ntiles < 16384) ? stage_cap : 0;
  hipLaunchKernelGGL(bin_place_kernel, dim3((unsigned)((n + gpb - 1) / gpb)), dim3(BLOCK), lds, st, p, splats, radii, mask,
                     SegBins{seg_count, bucket, seg, fail}, sel, list, gpb, stage);
}

template <int THREADS>
static void launch_radix(int ntiles, const uint2* ranges, const unsigned long long* bucket, uint32_t* point_list, int lo,
                         int hi, int cap, const uint32_t* spec_fail, TileSeg ts, hipStream_t st) {
  const size_t lds = (size_t)cap * 16 + (size_t)(THREADS / 64 + 1) * 256 * sizeof(uint32_t);
  static LdsGrant grant(48 * 1024);            // one static per THREADS instantiation
  grant.ensure((const void*)bin_tilesort_radix_kernel<THREADS>, lds);
  hipLaunchKernelGGL(bin_tilesort_radix_kernel<THREADS>, dim3(ntiles), dim3(THREADS), lds, st, ranges, bucket, point_list,
                     lo, hi, cap, spec_fail, ts);
}

void launch_bin_tilesort(int ntiles, uint32_t longest, const uint2* ranges, const unsigned long long* bucket,
                         uint32_t* point_list, const uint32_t* spec_fail, hipStream_t st, const uint32_t* seg_count,
                         uint32_t seg, uint2* ranges_out, const BinFinish* finish) {
  const TileSeg ts{seg_count, seg, nullptr}, ts0{seg_count, seg, ranges_out};
  const BinFinish nofin{nullptr, 0, nullptr, nullptr, nullptr, nullptr, nullptr, 0u, SpecCaps{nullptr, 0u, 0u, 0u, nullptr}};
  const BinFinish fin = finish ? *finish : nofin;
  // size classes by list length; every class runs with the LDS footprint its lists need:
  //   (0,256]  bitonic, 128 threads      (256,1024] radix, 256 threads   (1024,3072] radix, 512 threads
  //   (3072,8192] radix, 1024 threads    (8192,16384] bitonic in place, 1024 threads
  hipLaunchKernelGGL(bin_tilesort_kernel<128>, dim3(ntiles + (fin.count ? 1 : 0)), dim3(128), 256 * 8, st, ranges, bucket, point_list,
                     1, 257, spec_fail, ts0, fin);
  if (longest > 256) launch_radix<256>(ntiles, ranges, bucket, point_list, 257, 1025, 1024, spec_fail, ts, st);
  if (longest > 1024) launch_radix<512>(ntiles, ranges, bucket, point_list, 1025, 3073, 3072, spec_fail, ts, st);
  if (longest > 3072) launch_radix<1024>(ntiles, ranges, bucket, point_list, 3073, 8193, 8192, spec_fail, ts, st);
  if (longest > 8192) {
    static LdsGrant grant(48 * 1024);
    grant.ensure((const void*)bin_tilesort_kernel<1024>, 16384 * 8);
    hipLaunchKernelGGL(bin_tilesort_kernel<1024>, dim3(ntiles), dim3(1024), 16384 * 8, st, ranges, bucket, point_list,
                       8193, 16385, spec_fail, ts, nofin);
  }
}

}  // namespace rtgs
