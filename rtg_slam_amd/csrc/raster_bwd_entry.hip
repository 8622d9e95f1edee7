// K7 blend_bwd, third walk: ENTRY-PER-LANE.
//
// The strip and row-granular walks (raster_bwd.hip) give a lane one PIXEL and an instruction one ENTRY: every entry
// then costs a cross-lane reduction of nine partials (a third of the instructions of kernels that are VALU-issue
// bound, profiles/r03_pmc_sq_*).  Here a wave step covers 16 ENTRIES x 4 PIXELS:
//
//     lane l = (k = l >> 4, n = l & 15)  <->  pixel k of a 2x2 quad, n-th entry of a group of 16 list entries
//
// What is sequential along a ray - T_k = prod (1 - a_m), the colour behind - becomes two in-row prefix scans over the 16
// entries (DPP row_shr:1/2/4/8, four instructions each) with a per-pixel carry between groups:
//     T_k  = Tc * exclusive_prod(1 - a)           Q_k = Qc - inclusive_sum((c . g) a T),   Qc(0) = C_pixel . g
//     dL/da_k = T_k (c_k . g) - Q_k / (1 - a_k)                                   (front to back)
// Per-pixel values (g, last contributor, carries) live in the lane whose n equals the quad's index and reach the 16
// entry lanes of their row through DPP row_newbcast as an OPERAND of the instruction that uses them.
//
// The sums over pixels
//     sum_p gda[p][e] * {1, x_p, y_p, x_p^2, x_p y_p, y_p^2}      (dL/dopacity and the five moments behind du dv dconic)
//     sum_p  w [p][e] * {g_r, g_g, g_b}[p]                         (dL/dcolour)
// stay in the lane: every lane keeps the nine sums of ITS pixel k over the 16 quads of the wave's 8x8 quadrant (8 plain VALU +
// 3 FMAs with a DPP operand per step); at the end of a group the four k rows meet in registers (v_permlane32_swap /
// v_permlane16_swap) and row 0 adds 9 x 16 sums to the entries' LDS accumulators.  (Round 4 ran these sums as two
// v_mfma_f32_16x16x4_f32 per step - the lane map above is that instruction's B-operand map.  Elegant, and slower: the MFMA
// holds the SIMD for most of its 32 cycles.  Measured A-B in round 5, 98.6 -> 89.0 us / 158.3 -> 145.2 us,
// profiles/r05_bwd_walk_ab.txt; the MFMA form left the tree in round 6.)
//
// A wave walks only the entries that reach its quadrant, 16 at a time, and skips the quads whose four pixels are past their
// last contributor.  The opaque-depth partials do not ride the walk at all: blend_fwd leaves the list position of every
// pixel's depth owner and the owners' four partials go straight to the entry accumulators once per batch (depth_adds).
//
// alpha, the skip tests and the contributor set are evaluated exactly as the forward does (same dx, splat_power,
// splat_exp, list positions against n_contrib); T differs from the forward's in rounding only (product order).
//
// ROUND 6 - the prologue is ONE memory round trip.  Round 5's per-wave stamps (profiles/r05_bwd_stamps_*.txt) put a quarter
// of a wave's life into three DEPENDENT round trips before the first step (tile range -> list ids -> record gather), a tenth
// into the barriers of a cross-wave compaction that waited for the tile's slowest gather, and another tenth into a second
// dependent chain for the depth owners' planes (depth_index -> Splat record).  All of these values pass through blend_fwd's
// registers, so blend_fwd now leaves them where the backward finds them from the TILE INDEX alone (TileCache,
// raster_common.h): records and block masks of the first TILE_RECS list positions at tile * TILE_RECS, the owner's plane
// words per pixel.  Every load of the prologue is issued before the first wait; the sub-list of a quadrant is compacted by
// its own wave from the block masks (no cross-wave step); one barrier, then the walk.  List positions >= TILE_RECS (a tile
// walked deeper than 256 entries) and forwards that left no cache (RTGS_FWD_NO_BACKWARD) take the gather path.
#include "raster_common.h"
#include <stdlib.h>

namespace rtgs {

constexpr int MB = 256;            // list entries staged per batch: one per thread
constexpr int MACC = 13;           // floats of an entry's LDS accumulator (odd stride: conflict-free; 16 would cost the fifth workgroup per CU)
// accumulator columns: 0 m0 = sum gda | 1 mx 2 my 3 mxx | 4 mxy 5 myy | 6 7 8 colour | 9..12 depth plane

template <int CTRL>
__device__ __forceinline__ float dppf(float old, float v) {      // lanes without a source keep `old`
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ float bcast(float v) {                // lane N of the own row of 16 (every lane has a source)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + N, 0xf, 0xf, false));
}
template <int N>
__device__ __forceinline__ uint32_t bcast_u(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x150 + N, 0xf, 0xf, false);
}
__device__ __forceinline__ float row_scan_mul(float v) {         // inclusive prefix product over the row's 16 lanes
  v *= dppf<0x111>(1.f, v);
  v *= dppf<0x112>(1.f, v);
  v *= dppf<0x114>(1.f, v);
  v *= dppf<0x118>(1.f, v);
  return v;
}
__device__ __forceinline__ float row_scan_add(float v) {         // inclusive prefix sum
  v += dppf<0x111>(0.f, v);
  v += dppf<0x112>(0.f, v);
  v += dppf<0x114>(0.f, v);
  v += dppf<0x118>(0.f, v);
  return v;
}

// Everything a lane holds during the walk of one group of 16 entries.
struct EntryWalk {
  // entry n of the group (constant over the 16 steps)
  float u, v, ca, cb, cc, o, cr, cg_, cbl;
  uint32_t pos;                    // list position (0x7fffffff: no entry in this lane)
  // pixel (quad = n, k) held by this lane: broadcast to the row when the walk is at quad n
  float G0, G1, G2, Tc, Qc;
  uint32_t last;
  // pixel coordinates of the lane's k for the four quad columns / rows
  float pxc[4], pyc[4];
  int n;                           // lane & 15
  // lane-accumulate form (round 5): the lane's own sums over the quads of its pixel k, reduced over k at the flush
  float xq[4], yq[4];              // pixel coordinates about the tile centre, per quad column / row
  float s0, sx, sy, sxx, sxy, syy, sr, sg, sb;

  // The same step WITHOUT the matrix cores (round 5).  Measured (tools/probe/valu_rate.hip, tools/bwd_stamps.py): a
  // v_mfma_f32_16x16x4_f32 holds the SIMD for its 32 cycles - VALU instructions of the other waves do not issue under it -
  // so the two MFMAs of a step cost 64 of its ~156 SIMD cycles, for 9 x 16 x 4 useful multiply-adds.  The lane keeps the
  // nine sums of ITS pixel k over the quads (8 plain VALU for the moments, 3 FMAs with the colour gradient as a DPP
  // operand: ~20 cycles) and the four k rows meet in the LDS accumulator at the flush.
  template <int S>
  __device__ __forceinline__ void step_lane(uint32_t stepmask) {
    if (!((stepmask >> S) & 1u)) return;
    float power, G, alpha;
    {
      float dx, dy, t1, t2;
      asm volatile(
          "v_sub_f32 %[dx], %[u], %[px]\n\t"
          "v_sub_f32 %[dy], %[v], %[py]\n\t"
          "v_mul_f32 %[t1], %[ca], %[dx]\n\t"
          "v_mul_f32 %[t2], %[cc], %[dy]\n\t"
          "v_mul_f32 %[t2], %[t2], %[dy]\n\t"
          "v_fma_f32 %[t1], %[t1], %[dx], %[t2]\n\t"
          "v_mul_f32 %[t2], %[cb], %[dx]\n\t"
          "v_mul_f32 %[t2], %[t2], %[dy]\n\t"
          "v_fma_f32 %[pw], -0.5, %[t1], -%[t2]\n\t"
          "v_min_f32 %[t1], 0, %[pw]\n\t"
          "v_mul_f32 %[t1], 0x3fb8aa3b, %[t1]\n\t"
          "v_exp_f32 %[G], %[t1]\n\t"
          "s_nop 0\n\t"
          "v_mul_f32 %[al], %[o], %[G]\n\t"
          "v_min_f32 %[al], 0x3f7d70a4, %[al]\n\t"
          : [dx] "=&v"(dx), [dy] "=&v"(dy), [t1] "=&v"(t1), [t2] "=&v"(t2), [pw] "=&v"(power), [G] "=&v"(G), [al] "=&v"(alpha)
          : [u] "v"(u), [v] "v"(v), [ca] "v"(ca), [cb] "v"(cb), [cc] "v"(cc), [o] "v"(o), [px] "v"(pxc[S & 3]), [py] "v"(pyc[S >> 2]));
    }
    const uint32_t lastp = bcast_u<S>(last);
    const bool valid = (pos < lastp) & !(power > 0.f) & !(alpha < 1.f / 255.f);
    if (__builtin_amdgcn_ballot_w64(valid) == 0ull) return;
    const float a = valid ? alpha : 0.f;
    const float Gv = valid ? G : 0.f;
    // lane n == S of every row owns quad S's carries.  The lane mask is a compare INSIDE the statement that uses it (one
    // VALU per step, vcc): as sixteen 64-bit "s" operands the masks are loop invariants the compiler holds in 32 SGPRs for
    // the whole kernel - which a persistent kernel with twenty pointers does not have (they went to VGPR lanes and scratch).
    float incl, sinc, gda, w, cg, ia, t0, gx, gy;
    asm volatile(
        "v_sub_f32 %[incl], 1.0, %[a]\n\t"
        "v_mul_f32_dpp %[cg], %[G0], %[cr] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"
        "v_rcp_f32 %[ia], %[incl]\n\t"
        "v_fmac_f32_dpp %[cg], %[G1], %[cgn] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %[incl], %[incl], %[incl] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[cg], %[G2], %[cb] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_mul_f32_dpp %[incl], %[incl], %[incl] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_dpp %[incl], %[incl], %[incl] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_mul_f32_dpp %[incl], %[incl], %[incl] row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_mul_f32_dpp %[t0], %[Tc], %[ia] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"      // Tc[quad] / (1 - a)
        "v_mul_f32 %[t0], %[t0], %[incl]\n\t"                                                       // Tk = Tc incl / (1 - a): the exclusive product without a shift
        "v_mul_f32 %[w], %[a], %[t0]\n\t"
        "v_mul_f32 %[sinc], %[cg], %[w]\n\t"
        "v_mul_f32 %[gda], %[t0], %[cg]\n\t"                                                        // Tk cg
        "v_fmac_f32_dpp %[sr], %[G0], %[w] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"      // colour sums of pixel k
        "v_add_f32_dpp %[sinc], %[sinc], %[sinc] row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[sg], %[G1], %[w] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"
        "v_fmac_f32_dpp %[sb], %[G2], %[w] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %[sinc], %[sinc], %[sinc] row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[sinc], %[sinc], %[sinc] row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %[sinc], %[sinc], %[sinc] row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "v_sub_f32_dpp %[t0], %[Qc], %[sinc] row_newbcast:%[S] row_mask:0xf bank_mask:0xf\n\t"    // Qk
        "v_fma_f32 %[gda], -%[t0], %[ia], %[gda]\n\t"                                               // Tk cg - Qk / (1 - a)
        "v_mul_f32 %[gda], %[Gv], %[gda]\n\t"
        : [incl] "=&v"(incl), [sinc] "=&v"(sinc), [gda] "=&v"(gda), [w] "=&v"(w), [cg] "=&v"(cg), [ia] "=&v"(ia), [t0] "=&v"(t0),
          [sr] "+v"(sr), [sg] "+v"(sg), [sb] "+v"(sb)
        : [a] "v"(a), [Gv] "v"(Gv), [cr] "v"(cr), [cgn] "v"(cg_), [cb] "v"(cbl), [G0] "v"(G0), [G1] "v"(G1), [G2] "v"(G2),
          [Tc] "v"(Tc), [Qc] "v"(Qc), [S] "n"(S));
    float tn, qn;
    asm volatile(
        "v_cmp_eq_u32 vcc, %[S], %[n]\n\t"
        // moments of pixel k about the tile centre (three plain instructions ahead of the DPP reads of incl / sinc)
        "v_add_f32 %[s0], %[s0], %[gda]\n\t"
        "v_mul_f32 %[gx], %[gda], %[x]\n\t"
        "v_mul_f32 %[gy], %[gda], %[y]\n\t"
        // the quad's carries move past this group: lane n == S of every row owns them
        "v_mul_f32_dpp %[tn], %[incl], %[Tc] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_subrev_f32_dpp %[qn], %[sinc], %[Qc] row_newbcast:15 row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32 %[sx], %[sx], %[gx]\n\t"
        "v_add_f32 %[sy], %[sy], %[gy]\n\t"
        "v_cndmask_b32 %[Tc], %[Tc], %[tn], vcc\n\t"
        "v_cndmask_b32 %[Qc], %[Qc], %[qn], vcc\n\t"
        "v_fmac_f32 %[sxx], %[gx], %[x]\n\t"
        "v_fmac_f32 %[sxy], %[gx], %[y]\n\t"
        "v_fmac_f32 %[syy], %[gy], %[y]\n\t"
        : [tn] "=&v"(tn), [qn] "=&v"(qn), [gx] "=&v"(gx), [gy] "=&v"(gy), [Tc] "+v"(Tc), [Qc] "+v"(Qc), [s0] "+v"(s0), [sx] "+v"(sx),
          [sy] "+v"(sy), [sxx] "+v"(sxx), [sxy] "+v"(sxy), [syy] "+v"(syy)
        : [incl] "v"(incl), [sinc] "v"(sinc), [gda] "v"(gda), [x] "v"(xq[S & 3]), [y] "v"(yq[S >> 2]), [n] "v"(n), [S] "n"(S)
        : "vcc");
  }
  __device__ __forceinline__ void run16_lane(uint32_t sm) {
    step_lane<0>(sm); step_lane<1>(sm); step_lane<2>(sm); step_lane<3>(sm); step_lane<4>(sm); step_lane<5>(sm); step_lane<6>(sm);
    step_lane<7>(sm); step_lane<8>(sm); step_lane<9>(sm); step_lane<10>(sm); step_lane<11>(sm); step_lane<12>(sm); step_lane<13>(sm);
    step_lane<14>(sm); step_lane<15>(sm);
  }
};

// x[n] + x[n + 16] + x[n + 32] + x[n + 48] in every lane (gfx950's row swaps: upper half <-> lower half, odd rows <-> even rows)
__device__ __forceinline__ float sum_rows(float x) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  const float y = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  const auto q = __builtin_amdgcn_permlane16_swap(__float_as_uint(y), __float_as_uint(y), false, false);
  return __uint_as_float(q[0]) + __uint_as_float(q[1]);
}
// The opaque-depth partials of the wave's pixels whose owner entry is staged (rel = its index in the batch) on their way to
// the owners' LDS accumulators.  Neighbouring pixels share their owner (a near Gaussian owns a whole quadrant) and same-address
// LDS float adds serialise (21 us of the headline launch when every lane added for itself, round 4), so equal keys are merged
// in registers first.  Round 4 did that with a loop over the wave's DISTINCT owners, one full wave reduction per owner - fine
// for one owner, 12-17 % of a wave's lifetime where a quadrant has dozens (tools/bwd_stamps.py).  Round 5: a fixed binary
// tree over the lane index - rows 0|1 and 2|3 (v_permlane16_swap), rows 0|2 (v_permlane32_swap), then lanes n | n+1, n+2,
// n+4, n+8 of row 0 (DPP row_shl / row_shr: the quads of the quadrant, neighbours first) - where the representative of the lower half absorbs the representative of the
// upper half IF their owners are equal; whoever was not absorbed adds for itself.  One owner per wave: one lane adds.  All
// different: 64 lanes add to 64 addresses.  ~100 instructions either way, no loop.
template <int CTRL>
__device__ __forceinline__ uint32_t dppu(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, false);
}
__device__ __forceinline__ void depth_adds(bool pend, uint32_t rel, float d0, float d1, float d2, float d3, float* s_acc, int lane) {
  if (__builtin_amdgcn_ballot_w64(pend) == 0ull) return;
  constexpr uint32_t NONE = 0xffffffffu;
  uint32_t key = pend ? rel : NONE;
  const int k = lane >> 4, n = lane & 15;
  {   // rows 0|1, 2|3
    const auto q = __builtin_amdgcn_permlane16_swap(key, key, false, false);
    const uint32_t other = (k & 1) ? q[0] : q[1];
    const bool same = key != NONE && other == key;
    const auto e0 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
    const auto e1 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
    const auto e2 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d2), __float_as_uint(d2), false, false);
    const auto e3 = __builtin_amdgcn_permlane16_swap(__float_as_uint(d3), __float_as_uint(d3), false, false);
    const bool take = same && !(k & 1);
    d0 += take ? __uint_as_float(e0[1]) : 0.f; d1 += take ? __uint_as_float(e1[1]) : 0.f;
    d2 += take ? __uint_as_float(e2[1]) : 0.f; d3 += take ? __uint_as_float(e3[1]) : 0.f;
    if (same && (k & 1)) key = NONE;
  }
  {   // rows 0|2
    const auto q = __builtin_amdgcn_permlane32_swap(key, key, false, false);
    const uint32_t other = (k & 2) ? q[0] : q[1];
    const bool same = key != NONE && other == key && !(k & 1);
    const auto e0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d0), __float_as_uint(d0), false, false);
    const auto e1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d1), __float_as_uint(d1), false, false);
    const auto e2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d2), __float_as_uint(d2), false, false);
    const auto e3 = __builtin_amdgcn_permlane32_swap(__float_as_uint(d3), __float_as_uint(d3), false, false);
    const bool take = same && k == 0;
    d0 += take ? __uint_as_float(e0[1]) : 0.f; d1 += take ? __uint_as_float(e1[1]) : 0.f;
    d2 += take ? __uint_as_float(e2[1]) : 0.f; d3 += take ? __uint_as_float(e3[1]) : 0.f;
    if (same && k == 2) key = NONE;
  }
#define RTGS_DEPTH_LEVEL(SH)                                                                                          \
  {                                                                                                                    \
    const uint32_t up = dppu<0x100 + SH>(NONE, key), down = dppu<0x110 + SH>(NONE, key); /* lane n + SH, lane n - SH */ \
    const float u0 = dppf<0x100 + SH>(0.f, d0), u1 = dppf<0x100 + SH>(0.f, d1), u2 = dppf<0x100 + SH>(0.f, d2),        \
                u3 = dppf<0x100 + SH>(0.f, d3);                                                                        \
    const bool lower = k == 0 && (n & (2 * SH - 1)) == 0, upper = k == 0 && (n & (2 * SH - 1)) == SH;                 \
    const bool take = lower && key != NONE && up == key;                                                               \
    d0 += take ? u0 : 0.f; d1 += take ? u1 : 0.f; d2 += take ? u2 : 0.f; d3 += take ? u3 : 0.f;                        \
    if (upper && key != NONE && down == key) key = NONE;                                                               \
  }
  RTGS_DEPTH_LEVEL(1) RTGS_DEPTH_LEVEL(2) RTGS_DEPTH_LEVEL(4) RTGS_DEPTH_LEVEL(8)
#undef RTGS_DEPTH_LEVEL
  if (key != NONE) {
    float* const acc = &s_acc[key * MACC + 9];
    atomicAdd(acc + 0, d0); atomicAdd(acc + 1, d1); atomicAdd(acc + 2, d2); atomicAdd(acc + 3, d3);
  }
}

// The sub-list of a wave's quadrant, compacted by the wave itself: lane l looks at the block masks of list entries 4 l .. 4 l + 3
// of the batch (`mk`: four 16-bit masks), keeps those that reach quadrant `qm` and lie below `m`, and writes their indices to
// `sub` in list order (four ballots, no cross-wave step).  Returns the length of the sub-list.
__device__ __forceinline__ int compact_quadrant(uint2 mk, uint32_t qm, int m, int lane, uint8_t* __restrict__ sub) {
  uint32_t bits = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t mask = ((j < 2 ? mk.x : mk.y) >> (16 * (j & 1))) & 0xffffu;
    bits |= (((mask & qm) != 0u) & (4 * lane + j < m)) ? (1u << j) : 0u;
  }
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint32_t pre = 0, total = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned long long bal = __builtin_amdgcn_ballot_w64(((bits >> j) & 1u) != 0u);
    pre += (uint32_t)__popcll(bal & lt);
    total += (uint32_t)__popcll(bal);
  }
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if ((bits >> j) & 1u) sub[pre + (uint32_t)__popc(bits & ((1u << j) - 1u))] = (uint8_t)(4 * lane + j);
  return (int)total;
}

// ---------------------------------------------------------------------------------------------------------------------
// One workgroup per tile, in tile order (the hardware's dispatcher is the queue).  Round 6 also built the kernel PERSISTENT
// and queue-fed - one workgroup per resident slot taking tiles from a device queue, longest walk first, launch constants read
// once, the next tile's loads issued behind the previous tile's deferred tail - in three forms; all three are parity-green and
// all three are SLOWER than this (profiles/r06_persistent_bwd_ab.txt: 94-107 us against 85 on the headline scene; git
// e6aa13c has the code).  Why, by the stamps: (1) longest-first co-schedules the heavy tiles, which then share their SIMDs'
// issue slots and ALL take 60-80 us, while the light tiles that follow leave the SIMDs idle - the mix of a launch in tile
// order is the better schedule; (2) neighbouring tiles share records and gradient slots in L2: a scrambled order costs 8 us;
// (3) a workgroup that ends hands its slot to a fresh one while its stores are still in flight, a persistent one waits for
// them at its next barrier; (4) thirty loaded values cannot stay live across any other code at 96 VGPRs - the allocator
// answers with scratch - and at 128 VGPRs (four workgroups per CU) the hidden round trip does not pay for the lost wave.
// What the experiment left in the product kernel: the tile's words as vector loads beside the per-pixel values (as scalar
// loads they were five serialised waits), the carry masks from a compare instead of sixteen SGPR pairs, the wave maximum
// without ds_bpermute, the slot stores as GLOBAL stores (they were FLAT ones, counted on lgkmcnt: every LDS wait behind them
// waited for their acknowledgement), the accumulators cleared by the tail that reads them.
// ---------------------------------------------------------------------------------------------------------------------
// Everything the prologue of one tile loads: issued in one go, consumed after one wait.
struct TileLoads {
  uint32_t tmode, r0, r1, tlast;        // the tile's words (same address in every lane)
  uint32_t fail;                        // the speculation word of the forward (0: lists are what the host assumed)
  uint4 info;                           // BwdInfo: slot_grads (two words), slots, use_slots
  uint32_t last;                        // per pixel
  float g0, g1, g2, oc0, oc1, oc2, gD;
  int owner;
  uint32_t dpos;
  float2 aux;
  float4 c0, c1, c2;                    // the record of list position tid (TileCache)
  uint2 mk;                             // block masks of list positions 4 lane .. 4 lane + 3
};

// STAMP (measurement only, rtgs_raster_set_bwd_stamps): every wave leaves, per TILE, fourteen 64-bit words - wall clock (100 MHz)
// at the start and end of the tile's turn, shader cycles spent in the group loop and in the whole turn, groups walked, quad
// steps entered, the cycles of the other phases (prologue: until every first load has landed | accumulator zeroing + depth
// partials | staging: records -> LDS (gather path: the gather and the quadrant test) | compaction + the barrier before the
// walk | barrier behind the loop: the tile's slowest quadrant | per-entry tail: moments -> slot store, incl. the barrier
// behind it) and two marks inside the prologue (unused | every load of the prologue there).
template <bool STAMP>
__device__ __forceinline__ void blend_bwd_entry_body(
    const RasterParams& p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const Splat* __restrict__ splats, const float* __restrict__ out_color, const uint32_t* __restrict__ n_contrib,
    const int32_t* __restrict__ depth_index, const uint32_t* __restrict__ depth_pos, const uint32_t* __restrict__ tile_last,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const uint32_t* __restrict__ gbase, uint32_t* __restrict__ slot_count, const BwdInfo* __restrict__ info,
    SplatGrad* __restrict__ grads, uint8_t* __restrict__ touched, const uint32_t* __restrict__ tile_mode,
    uint32_t t0, uint32_t tn, uint32_t dbg, unsigned long long* __restrict__ stamps, const TileCache& tc, uint32_t pre) {
  __shared__ float4 s_rec[MB * 3];              // u v ca cb | cc o r g | b id blockmask -
  __shared__ float s_acc[MB * MACC];            // per-entry sums of the tile (LDS float adds: one flush per wave and group)
  __shared__ uint8_t s_sub[4][MB];              // per quadrant: the staged entries that reach it, in list order

  const int tid = threadIdx.x;
  const uint32_t HW = (uint32_t)(p.H * p.W);

  // every load of a tile's prologue; the addresses need the tile index only (the TileCache is allocated for every tile, so
  // its loads are in bounds whether or not the forward filled it)
  auto issue = [&](int packed, TileLoads& L) {
    // per-lane address parts are formed HERE, per tile: hoisted out of the tile loop they live in scratch, and a reload from
    // scratch at the top of a turn is a wait for every store of the previous tile's tail
    int tl = tid;
    asm volatile("" : "+v"(tl));
    const int ln = tl & 15, lk = (tl >> 4) & 3, lw = tl >> 6;
    const int lx = (lw & 1) * 8 + 2 * (ln & 3) + (lk & 1), ly = (lw >> 1) * 8 + 2 * (ln >> 2) + (lk >> 1);      // the lane's pixel inside the tile
    const int bx = packed & 0xffff, by = packed >> 16;
    const int tile = by * p.gx + bx;
    const int px = bx * TILE + lx, py = by * TILE + ly;
    const uint32_t pix = (px < p.W && py < p.H) ? (uint32_t)(py * p.W + px) : 0u;       // clamped: the loads carry no branch
    // the tile's words as VECTOR loads (every lane the same address; `tz` is a zero the compiler cannot see): as scalar loads
    // they must be waited for with lgkmcnt(0) - in front of the loads that follow
    int tz = 0;
    asm volatile("" : "+v"(tz));
    tz += tile;
    L.tmode = tile_mode[tz];
    const uint2 rg = ranges[tz];
    L.r0 = rg.x; L.r1 = rg.y;
    L.tlast = tile_last[tz];
    // the launch's words that live in memory travel the same way: as scalar loads each is a dependent round trip in front
    // of the first vector load.  (Everything this prologue reads exists whether or not the speculation held; the word is
    // tested before anything is read THROUGH a list.)
    L.fail = p.spec_fail ? p.spec_fail[tz - tile] : 0u;
    L.info = reinterpret_cast<const uint4*>(info)[tz - tile];
    L.last = n_contrib[pix];
    L.g0 = dL_dcolor[pix]; L.g1 = dL_dcolor[HW + pix]; L.g2 = dL_dcolor[2 * HW + pix];
    L.oc0 = out_color[pix]; L.oc1 = out_color[HW + pix]; L.oc2 = out_color[2 * HW + pix];
    L.owner = depth_index[pix];
    L.gD = dL_ddepth[pix];
    L.dpos = depth_pos[pix];
    {
      // the records of the first `pre` list positions travel with the prologue (64 behind a near-slice pass - its first batch,
      // where most tiles end: a mean of 53 entries on the headline scene - else the whole cache line-up of 256: all 256 for
      // every tile were 40 MB of reads per launch that 78 % of the headline's tiles never used, PMC traffic 3.3x algorithmic);
      // a tile that walks deeper fetches the rest once its length is known - one more round trip, still no dependent chain
      const float4* const rsrc = tc.recs + ((size_t)tile * TILE_RECS + (size_t)tl) * 3;
      if ((uint32_t)tl < pre) { L.c0 = rsrc[0]; L.c1 = rsrc[1]; L.c2 = rsrc[2]; }
      L.mk = reinterpret_cast<const uint2*>(tc.masks + (size_t)tile * TILE_RECS)[tl & 63];      // entries 4 lane .. 4 lane + 3
      L.aux = tc.depth_aux[pix];
    }
  };

  const int cur = (int)((blockIdx.y << 16) | blockIdx.x);
  unsigned long long st_wall = 0, st_cyc = 0, st_walk = 0, st_groups = 0, st_steps = 0;
  unsigned long long st_seg[6] = {0, 0, 0, 0, 0, 0}, st_t = 0, st_p2 = 0, st_p1 = 0;
  if constexpr (STAMP) { st_wall = wall_clock64(); st_cyc = __builtin_readcyclecounter(); }
  TileLoads L;
  issue(cur, L);
  if constexpr (STAMP) st_p1 = __builtin_readcyclecounter() - st_cyc;      // every load of the tile is issued
  // under the loads' flight: the accumulators start at zero (later batches: the per-entry tail clears what it has read)
  for (int q = tid; q < MB * MACC; q += BLOCK) s_acc[q] = 0.f;
  __syncthreads();

  bool use_slots = false;                       // BwdInfo, once its words are here
  SplatGrad* slot_grads = nullptr;
  // one thread per staged entry: moments -> d(u, v, conic), slot store; the thread clears the sums it has read
  auto entry_tail = [&](int m, float cxT, float cyT) {
    int tt = tid;
    asm volatile("" : "+v"(tt));
    float t[13];
    bool any = false;
    uint32_t gid = 0u;
    if (tt < m) {
#pragma unroll
      for (int q = 0; q < 13; ++q) { t[q] = s_acc[tt * MACC + q]; any |= (t[q] != 0.f); }
#pragma unroll
      for (int q = 0; q < 13; ++q) s_acc[tt * MACC + q] = 0.f;
      gid = __float_as_uint(s_rec[tt * 3 + 2].y);
    }
    const bool store = any && gid - t0 < tn && !(dbg & 4u);         // a frozen row (outside [t0, t0 + tn)) takes no slot
    if (!store) return;
    // the slot run of the Gaussian and its next free slot: two independent round trips, issued together
    uint32_t slot = 0u;
    if (use_slots) slot = gbase[gid] + atomicAdd(&slot_count[gid], 1u);
    const float4 q0 = s_rec[tt * 3 + 0], q1 = s_rec[tt * 3 + 1];       // u v ca cb | cc o r g
    // sums over the pixels of gdl = o gda times powers of d = centre - pixel, from the moments about the tile centre
    const float ut = q0.x - cxT, vt = q0.y - cyT, o = q1.y;
    const float m0 = t[0], mx = t[1], my = t[2], mxx = t[3], mxy = t[4], myy = t[5];
    const float sx = o * (ut * m0 - mx), sy = o * (vt * m0 - my);
    const float sxx = o * (ut * (ut * m0 - 2.f * mx) + mxx);
    const float sxy = o * (ut * (vt * m0 - my) - vt * mx + mxy);
    const float syy = o * (vt * (vt * m0 - 2.f * my) + myy);
    const float du = -(q0.z * sx + q0.w * sy), dv = -(q1.x * sy + q0.w * sx);
    const float dca = -0.5f * sxx, dcb = -sxy, dcc = -0.5f * syy, dop = m0;
    touched[gid] = 1;
    if (use_slots) {
      // SplatGrad order: du dv dca dcb | dcc dop dr dg | db dnx dny dnz | dpd - - -
      // (slot_grads comes out of BwdInfo, i.e. out of memory: without the address space the stores are FLAT ones, which
      // count on lgkmcnt as well - every LDS wait and barrier behind them would wait for their acknowledgement)
      typedef float vf4 __attribute__((ext_vector_type(4)));
      typedef __attribute__((address_space(1))) vf4 global_vf4;
      global_vf4* dst = (global_vf4*)(slot_grads + slot);
      dst[0] = vf4{du, dv, dca, dcb};
      dst[1] = vf4{dcc, dop, t[6], t[7]};
      dst[2] = vf4{t[8], t[9], t[10], t[11]};
      dst[3] = vf4{t[12], 0.f, 0.f, 0.f};
    } else {
      float* dst = reinterpret_cast<float*>(grads + gid);
      const float vals[13] = {du, dv, dca, dcb, dcc, dop, t[6], t[7], t[8], t[9], t[10], t[11], t[12]};
#pragma unroll
      for (int q = 0; q < 13; ++q)
        if (vals[q] != 0.f) unsafeAtomicAdd(dst + q, vals[q]);
    }
  };
  {
    // wave = 8x8 quadrant, step = 2x2 quad s of it, DPP row k = pixel of the quad; THIS lane's own pixel is (quad n, k).
    // (From a laundered thread index, per tile: see issue().)
    int tq = tid;
    asm volatile("" : "+v"(tq));
    const int lane = tq & 63, wv = tq >> 6;
    const int n = lane & 15, k = lane >> 4;
    const int qx0 = (wv & 1) * 8, qy0 = (wv >> 1) * 8;
    const int lx = qx0 + 2 * (n & 3) + (k & 1), ly = qy0 + 2 * (n >> 2) + (k >> 1);      // the lane's pixel inside the tile
    const uint32_t qmask = 0x0033u << (2 * (wv & 1) + 8 * (wv >> 1));    // the 4x4 blocks of this wave's quadrant (blocks_reached numbering)
    const int bx = cur & 0xffff, by = cur >> 16;
    const int tile = by * p.gx + bx;
    const int px = bx * TILE + lx, py = by * TILE + ly;
    const bool inside = px < p.W && py < p.H;
    const uint32_t pix = inside ? (uint32_t)(py * p.W + px) : 0u;
    // ---- the first wait of the turn: every load of the prologue lands
    const uint32_t tmode = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tmode);
    if constexpr (STAMP) { asm volatile("" ::"s"(tmode)); st_p1 |= (__builtin_readcyclecounter() - st_cyc) << 32; }   // the FIRST load is back
    const uint32_t range_x = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.r0), range_y = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.r1);
    const uint32_t tlast = (uint32_t)__builtin_amdgcn_readfirstlane((int)L.tlast);
    if (__builtin_amdgcn_readfirstlane((int)L.fail) != 0) return;        // the speculative forward's lists were not built: the host redoes the step
    use_slots = __builtin_amdgcn_readfirstlane((int)L.info.w) != 0;
    slot_grads = reinterpret_cast<SplatGrad*>(((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.info.y) << 32) |
                                                               (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)L.info.x));
    // the tile stages no further than its last contributor (0: another walk has this tile)
    const int nuse = ((tmode & 3u) == 2u) ? min((int)(range_y - range_x), (int)tlast) : 0;
    const bool cached = (tmode & 4u) != 0u && !(dbg & 8u);   // this forward filled the cache for this tile (bit 3: ignore it, A-B / tests)
    if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t = __builtin_readcyclecounter(); st_p2 = t - st_cyc; st_seg[0] = st_p2; st_t = t; }

    const float tx0 = (float)(bx * TILE), ty0 = (float)(by * TILE);
    const float cxT = tx0 + 7.5f, cyT = ty0 + 7.5f;     // moments are taken about the tile centre

    EntryWalk W;
    W.n = n;
    W.last = inside ? L.last : 0u;
    W.G0 = inside ? L.g0 : 0.f; W.G1 = inside ? L.g1 : 0.f; W.G2 = inside ? L.g2 : 0.f;
    W.Tc = 1.f;
    // (colour behind the walk) . g + T_final (bg . g) before the first entry = the pixel's output colour . g
    W.Qc = L.oc0 * W.G0 + L.oc1 * W.G1 + L.oc2 * W.G2;
    // Opaque-surface depth: D = pd / (n_c . r); only the pixel's owner receives it.  A pixel owns at
    // most one entry of the whole list: its four partials go to that entry's accumulator in the batch that stages it.  The
    // two words they need beyond the pixel's own ray - 1 / (n_c . r) and D - come from the forward (TileCache::depth_aux);
    // without a cache, from the owner's Splat record (a dependent gather).
    const bool has_owner = inside && L.owner >= 0 && L.gD != 0.f && !(dbg & 2u);
    const uint32_t dpos = has_owner ? L.dpos : 0xffffffffu;
    // a wave walks no further than its own last contributor
    const uint32_t wave_last = wave_max_u32(W.last);

    // the depth owners of the batch [base, base + m): kk = -gD D / (n_c . r).  The accumulators are zero and visible (the
    // barrier behind the previous tail), so this runs BEFORE the barrier that precedes the walk.
    auto depth_batch = [&](int base, int m, float iden, float Dd, float gD, bool reload) {
      const bool pend = dpos >= (uint32_t)base && dpos < (uint32_t)(base + m);
      if (__builtin_amdgcn_ballot_w64(pend) == 0ull) return;
      float d0 = 0.f, d1 = 0.f, d2 = 0.f, d3 = 0.f;
      if (pend) {
        const float rx = ((float)px - p.cx) / p.fx, ry = ((float)py - p.cy) / p.fy;
        uint32_t pl = pix;
        asm volatile("" : "+v"(pl));              // the addresses of these rare loads are formed here, not held from the prologue on
        if (reload) { gD = dL_ddepth[pl]; const float2 a2 = cached ? tc.depth_aux[pl] : make_float2(0.f, 0.f); iden = a2.x; Dd = a2.y; }
        if (!cached) {
          const int owner = depth_index[pl];
          const float4 r2 = reinterpret_cast<const float4*>(splats + owner)[2];   // b nx ny nz
          const float pd = reinterpret_cast<const float*>(splats + owner)[12];
          iden = 1.f / (r2.y * rx + r2.z * ry + r2.w);
          Dd = pd * iden;
        }
        const float kk = -gD * Dd * iden;
        d0 = kk * rx; d1 = kk * ry; d2 = kk; d3 = gD * iden;
      }
      depth_adds(pend, dpos - (uint32_t)base, d0, d1, d2, d3, s_acc, lane);
    };
    // gather path of one batch: a record per thread through the list, tested against the four quadrants here; returns the
    // length of this wave's sub-list.  One barrier inside (the masks of all staged entries must be in LDS).
    auto stage_gather = [&](int base, int m) -> int {
      float txl = tx0, tyl = ty0;                        // laundered: the test's per-tile constants must not be hoisted
      asm volatile("" : "+v"(txl), "+v"(tyl));           // out of the batch loop (they would be live through the walk)
      int tl = tid, ll = lane;
      asm volatile("" : "+v"(tl), "+v"(ll));            // this path's addresses are formed here (held from the prologue on they spill)
      if (tid < m) {
        const uint32_t id = point_list[range_x + base + tl];
        const float4* src = reinterpret_cast<const float4*>(splats + id);
        const float4 q0 = src[0];
        s_rec[tid * 3 + 0] = q0;
        const float4 q1 = src[1];
        s_rec[tid * 3 + 1] = q1;
        const float b = reinterpret_cast<const float*>(splats + id)[8];
        const float2 hxy = reinterpret_cast<const float2*>(splats + id)[7];
        const uint32_t reach = quads_reached(q0.x, q0.y, hxy.x, hxy.y, q0.z, q0.w, q1.x, q1.y, txl, tyl);
        const uint32_t bm = ((reach & 1u) ? 0x0033u : 0u) | ((reach & 2u) ? 0x00ccu : 0u) | ((reach & 4u) ? 0x3300u : 0u) | ((reach & 8u) ? 0xcc00u : 0u);
        s_rec[tid * 3 + 2] = make_float4(b, __uint_as_float(id), __uint_as_float(bm), 0.f);
      }
      __syncthreads();
      const int e0 = 4 * ll;
      const uint32_t m0 = e0 + 0 < m ? __float_as_uint(s_rec[(e0 + 0) * 3 + 2].z) : 0u, m1 = e0 + 1 < m ? __float_as_uint(s_rec[(e0 + 1) * 3 + 2].z) : 0u;
      const uint32_t m2 = e0 + 2 < m ? __float_as_uint(s_rec[(e0 + 2) * 3 + 2].z) : 0u, m3 = e0 + 3 < m ? __float_as_uint(s_rec[(e0 + 3) * 3 + 2].z) : 0u;
      return compact_quadrant(make_uint2(m0 | (m1 << 16), m2 | (m3 << 16)), qmask, m, lane, s_sub[wv]);
    };

    // ---- the first batch: it alone consumes the prologue's loads
    int base = 0, m = min(MB, nuse), cnt = 0;
    if (nuse > 0) {
      if (cached) {
        // the forward's records: already in registers, one LDS write; the sub-list from the block masks, by this wave alone
        if (tid < m) {
          if ((uint32_t)tid >= pre) {
            int tl = tid;
            asm volatile("" : "+v"(tl));
            const float4* const rsrc = tc.recs + ((size_t)tile * TILE_RECS + (size_t)tl) * 3;
            L.c0 = rsrc[0]; L.c1 = rsrc[1]; L.c2 = rsrc[2];
          }
          s_rec[tid * 3 + 0] = L.c0; s_rec[tid * 3 + 1] = L.c1; s_rec[tid * 3 + 2] = L.c2;
        }
        cnt = compact_quadrant(L.mk, qmask, m, lane, s_sub[wv]);
      } else {
        cnt = stage_gather(0, m);
      }
      depth_batch(0, m, L.aux.x, L.aux.y, L.gD, false);
    }
    if constexpr (STAMP) { const unsigned long long t = __builtin_readcyclecounter(); st_seg[2] += t - st_t; st_t = t; }
    for (;;) {
      const bool lastb = base + MB >= nuse;
      __syncthreads();                                   // records, sub-lists and depth partials of the batch are in LDS
      if constexpr (STAMP) { const unsigned long long t = __builtin_readcyclecounter(); st_seg[3] += t - st_t; st_t = t; }

      // ---- the wave walks its quadrant's sub-list, 16 entries at a time
      {
        // Per-lane constants of the walk, (re)built per batch from the laundered lane coordinates: held across the batch
        // loop they would be live through the staging and cost a wave per SIMD.
        int kq = k;
        asm volatile("" : "+v"(kq));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          W.pxc[j] = (float)(bx * TILE + qx0 + 2 * j + (kq & 1));
          W.pyc[j] = (float)(by * TILE + qy0 + 2 * j + (kq >> 1));
          W.xq[j] = W.pxc[j] - cxT; W.yq[j] = W.pyc[j] - cyT;
        }
      }
      unsigned long long w_in = 0;
      if constexpr (STAMP) w_in = __builtin_readcyclecounter();
      for (int g0i = 0; g0i < cnt && !(dbg & 1u); g0i += 16) {
        const bool have = g0i + n < cnt;
        const int e = have ? (int)s_sub[wv][g0i + n] : 0;
        W.pos = have ? (uint32_t)(base + e) : 0x7fffffffu;
        const uint32_t first_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)W.pos);
        if (first_pos >= wave_last) break;                 // positions increase: the wave is through
        // quads with a pixel that still blends at or behind the group's first entry
        const unsigned long long lb = __builtin_amdgcn_ballot_w64(W.last > first_pos);
        const uint32_t stepmask = (uint32_t)((lb | (lb >> 16) | (lb >> 32) | (lb >> 48)) & 0xffffull);
        if (stepmask == 0u) continue;
        if constexpr (STAMP) { st_groups += 1; st_steps += (unsigned long long)__popc(stepmask); }
        const float4 r0 = s_rec[e * 3 + 0], r1 = s_rec[e * 3 + 1];
        const float rb = s_rec[e * 3 + 2].x;
        W.u = r0.x; W.v = r0.y; W.ca = r0.z; W.cb = r0.w; W.cc = r1.x; W.o = r1.y; W.cr = r1.z; W.cg_ = r1.w; W.cbl = rb;
        W.s0 = W.sx = W.sy = W.sxx = W.sxy = W.syy = W.sr = W.sg = W.sb = 0.f;
        W.run16_lane(stepmask);
        // the four pixel rows k of an entry meet in registers (v_permlane32_swap / v_permlane16_swap: two instructions per
        // sum) and row 0 adds them to the entry's accumulator.  64 lanes adding for themselves cost 4x the LDS float adds,
        // and those run at about a lane per cycle per CU: measured 153 / 277 us instead of 99 / 165.
        const float f0 = sum_rows(W.s0), f1 = sum_rows(W.sx), f2 = sum_rows(W.sy), f3 = sum_rows(W.sxx), f4 = sum_rows(W.sxy);
        const float f5 = sum_rows(W.syy), f6 = sum_rows(W.sr), f7 = sum_rows(W.sg), f8 = sum_rows(W.sb);
        if (have && k == 0) {
          float* const acc = &s_acc[e * MACC];
          atomicAdd(acc + 0, f0); atomicAdd(acc + 1, f1); atomicAdd(acc + 2, f2); atomicAdd(acc + 3, f3); atomicAdd(acc + 4, f4);
          atomicAdd(acc + 5, f5); atomicAdd(acc + 6, f6); atomicAdd(acc + 7, f7); atomicAdd(acc + 8, f8);
        }
      }
      if constexpr (STAMP) { st_t = __builtin_readcyclecounter(); st_walk += st_t - w_in; }
      __syncthreads();
      if constexpr (STAMP) { const unsigned long long t = __builtin_readcyclecounter(); st_seg[4] += t - st_t; st_t = t; }

      if (lastb) {
        entry_tail(m, cxT, cyT);
        if constexpr (STAMP) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long t2 = __builtin_readcyclecounter(); st_seg[5] += t2 - st_t; st_t = t2; }
        break;
      }
      entry_tail(m, cxT, cyT);
      __syncthreads();                                   // the tail has read (and cleared) its LDS: the next batch may write it
      if constexpr (STAMP) { const unsigned long long t2 = __builtin_readcyclecounter(); st_seg[5] += t2 - st_t; st_t = t2; }
      base += MB;
      m = min(MB, nuse - base);
      cnt = stage_gather(base, m);
      depth_batch(base, m, 0.f, 0.f, 0.f, true);
      if constexpr (STAMP) { const unsigned long long t2 = __builtin_readcyclecounter(); st_seg[2] += t2 - st_t; st_t = t2; }
    }
    if constexpr (STAMP) {
      if (lane == 0) {
        unsigned long long* o = stamps + (size_t)(tile * 4 + wv) * 14;
        o[0] = st_wall; o[1] = wall_clock64(); o[2] = st_walk; o[3] = __builtin_readcyclecounter() - st_cyc; o[4] = st_groups; o[5] = st_steps;
#pragma unroll
        for (int q = 0; q < 6; ++q) o[6 + q] = st_seg[q];
        o[12] = st_p1; o[13] = st_p2;
      }
    }
  }
}

#define RTGS_BWD_ARGS                                                                                                             \
  RasterParams p, const uint2 *__restrict__ ranges, const uint32_t *__restrict__ point_list, const Splat *__restrict__ splats,       \
      const float *__restrict__ out_color, const uint32_t *__restrict__ n_contrib, const int32_t *__restrict__ depth_index,          \
      const uint32_t *__restrict__ depth_pos, const uint32_t *__restrict__ tile_last, const float *__restrict__ dL_dcolor,           \
      const float *__restrict__ dL_ddepth, const uint32_t *__restrict__ gbase, uint32_t *__restrict__ slot_count,                    \
      const BwdInfo *__restrict__ info, SplatGrad *__restrict__ grads, uint8_t *__restrict__ touched,                               \
      const uint32_t *__restrict__ tile_mode, uint32_t t0, uint32_t tn, uint32_t dbg, unsigned long long *__restrict__ stamps,       \
      TileCache tc, uint32_t pre
#define RTGS_BWD_PASS                                                                                                             \
  p, ranges, point_list, splats, out_color, n_contrib, depth_index, depth_pos, tile_last, dL_dcolor, dL_ddepth, gbase, slot_count, \
      info, grads, touched, tile_mode, t0, tn, dbg, stamps, tc, pre
// the product kernel
__global__ void __launch_bounds__(256, 5) blend_bwd_entry_kernel(RTGS_BWD_ARGS) { blend_bwd_entry_body<false>(RTGS_BWD_PASS); }
// the same, leaving per-wave time stamps (tools/bwd_stamps.py)
__global__ void __launch_bounds__(256, 5) blend_bwd_entry_stamped_kernel(RTGS_BWD_ARGS) { blend_bwd_entry_body<true>(RTGS_BWD_PASS); }
#undef RTGS_BWD_ARGS
#undef RTGS_BWD_PASS

// timing decompositions (tools only; rtgs_raster_set_bwd_debug): bit 0 walk off, bit 1 depth partials off, bit 2 stores off -
// results are then wrong by construction, so the bits are NOT read from the environment and the tools that set
// them clear them again.  Bit 3 (results unchanged): ignore the TileCache, take the gather path (tests, A-B)
static int g_bwd_dbg = 0;
static unsigned long long* g_bwd_stamps = nullptr;
void set_bwd_debug(int bits) { g_bwd_dbg = bits & 15; }
void set_bwd_stamps(void* dev) { g_bwd_stamps = (unsigned long long*)dev; }

void launch_blend_bwd_entry(const RasterParams& p, const uint2* ranges, const uint32_t* point_list, const Splat* splats,
                            const float* out_color, const uint32_t* n_contrib, const int32_t* depth_index,
                            const uint32_t* depth_pos, const uint32_t* tile_last, const float* dL_dcolor, const float* dL_ddepth,
                            const uint32_t* gbase, uint32_t* slot_count, const BwdInfo* info, SplatGrad* grads, uint8_t* touched,
                            const uint32_t* tile_mode, uint32_t t0, uint32_t tn, TileCache tc, uint32_t pre, hipStream_t st) {
  const uint32_t dbg = (uint32_t)g_bwd_dbg;
#define RTGS_BWD_LAUNCH(KERNEL)                                                                                                      \
  hipLaunchKernelGGL(KERNEL, dim3(p.gx, p.gy), dim3(BLOCK), 0, st, p, ranges, point_list, splats, out_color, n_contrib, depth_index, \
                     depth_pos, tile_last, dL_dcolor, dL_ddepth, gbase, slot_count, info, grads, touched, tile_mode, t0, tn, dbg,    \
                     g_bwd_stamps, tc, pre)
  if (g_bwd_stamps) RTGS_BWD_LAUNCH(blend_bwd_entry_stamped_kernel); else RTGS_BWD_LAUNCH(blend_bwd_entry_kernel);
#undef RTGS_BWD_LAUNCH
}

}  // namespace rtgs
