// gfx950 kernels + C ABI of the SLAM-side ops around the hot path (include/rtgs_slam.h): tile-mask producers, the
// 3-nearest-neighbour search behind simple_knn.distCUDA2, the per-Gaussian error accumulation behind
// cuda_utils.accumulate_gaussian_error, and the frame preprocessing of Tracker.map_preprocess.
// Compiled with -ffp-contract=off: the float32 op order of distances / stencils follows the reference's torch code
// (oracle/slam_ops_oracle.py restates it; tests compare bit for bit where torch's rounding is reproducible).
#include "../../include/rtgs_slam.h"
#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>
#include <float.h>
#include <math.h>

namespace rtgs_slam {

__device__ __forceinline__ uint32_t enc_f(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);        // monotone: a < b <=> enc(a) < enc(b)
}
__device__ __forceinline__ float dec_f(uint32_t e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// ------------------------------------------------------------------------------------------------ tile masks
// one workgroup per tile; KIND 0 = uint8 mask, 1 = float32, 2 = float32 T_map (pixel on iff T != 1; also writes the mask)
template <int KIND>
__global__ void __launch_bounds__(256) tile_sum_kernel(const void* __restrict__ src, int H, int W, int stride, int gx,
                                                       float* __restrict__ tile_sum, uint8_t* __restrict__ mask_out) {
  const int tx = blockIdx.x % gx, ty = blockIdx.x / gx;
  float acc = 0.f;
  for (int q = threadIdx.x; q < stride * stride; q += 256) {
    const int x = tx * stride + q % stride, y = ty * stride + q / stride;
    if (x < W && y < H) {
      const size_t i = (size_t)y * W + x;
      if constexpr (KIND == 0) acc += reinterpret_cast<const uint8_t*>(src)[i] ? 1.f : 0.f;
      if constexpr (KIND == 1) acc += reinterpret_cast<const float*>(src)[i];
      if constexpr (KIND == 2) {
        const bool on = reinterpret_cast<const float*>(src)[i] != 1.f;
        mask_out[i] = on ? 1 : 0;
        acc += on ? 1.f : 0.f;
      }
    }
  }
  __shared__ float s[4];
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) tile_sum[blockIdx.x] = (s[0] + s[1]) + (s[2] + s[3]);
}

// mode 0: mean > ratio (transmission2tilemask); mode 1: any (pixelmask2tilemask).  Also the total of the sums.
__global__ void __launch_bounds__(1024) tile_threshold_kernel(const float* __restrict__ tile_sum, int ntiles, float area,
                                                              float ratio, int mode, int32_t* __restrict__ mask,
                                                              uint32_t* __restrict__ count_out) {
  float tot = 0.f;
  for (int t = threadIdx.x; t < ntiles; t += 1024) {
    const float s = tile_sum[t];
    mask[t] = mode == 0 ? ((s / area) > ratio ? 1 : 0) : (s > 0.f ? 1 : 0);
    tot += s;
  }
  if (!count_out) return;
  __shared__ float sh[16];
  tot = wave_sum(tot);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = tot;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f;
    for (int i = 0; i < 16; ++i) t += sh[i];
    count_out[0] = (uint32_t)(t + 0.5f);                    // sums of 0/1 pixels: exact integers in float32 up to 2^24
  }
}

// top-k tiles by mean value: 4-pass MSB radix select of the k-th largest key, then ties in index order.  The keys are
// recomputed from the tile sums on every pass (a few thousand L2-resident floats) instead of being cached in LDS: no
// 64-KB static LDS block (which only gfx950 could hold) and no limit on the number of tiles.
__global__ void __launch_bounds__(1024) tile_topk_kernel(const float* __restrict__ tile_sum, int ntiles, float area, int k,
                                                         int32_t* __restrict__ mask) {
  __shared__ uint32_t s_hist[256];
  __shared__ uint32_t s_scan[1024];
  __shared__ uint32_t s_prefix, s_k;
  const int tid = threadIdx.x;
  auto key_of = [&](int t) { return enc_f(tile_sum[t] / area); };
  if (tid == 0) { s_prefix = 0u; s_k = (uint32_t)k; }
  __syncthreads();
  if (k <= 0) { for (int t = tid; t < ntiles; t += 1024) mask[t] = 0; return; }
  if (k >= ntiles) { for (int t = tid; t < ntiles; t += 1024) mask[t] = 1; return; }
  for (int pass = 3; pass >= 0; --pass) {
    const int shift = 8 * pass;
    if (tid < 256) s_hist[tid] = 0;
    __syncthreads();
    const uint32_t prefix = s_prefix, himask = pass == 3 ? 0u : (0xffffffffu << (shift + 8));
    for (int t = tid; t < ntiles; t += 1024) {
      const uint32_t key = key_of(t);
      if ((key & himask) == prefix) atomicAdd(&s_hist[(key >> shift) & 0xffu], 1u);
    }
    __syncthreads();
    if (tid == 0) {                                          // walk the digits from the top until k keys are covered
      uint32_t need = s_k, d = 255;
      for (;; --d) {
        const uint32_t c = s_hist[d];
        if (c >= need) break;
        need -= c;
        if (d == 0) break;
      }
      s_k = need;
      s_prefix = prefix | (d << shift);
    }
    __syncthreads();
  }
  const uint32_t kth = s_prefix;                             // key of the k-th largest value; s_k = how many ties to take
  const uint32_t take = s_k;
  const int per = (ntiles + 1023) / 1024;
  const int lo = min(ntiles, tid * per), hi = min(ntiles, lo + per);
  uint32_t ties = 0;
  for (int t = lo; t < hi; ++t) ties += key_of(t) == kth ? 1u : 0u;
  s_scan[tid] = ties;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {
    const uint32_t a = tid >= off ? s_scan[tid - off] : 0u;
    __syncthreads();
    s_scan[tid] += a;
    __syncthreads();
  }
  uint32_t rank = s_scan[tid] - ties;
  for (int t = lo; t < hi; ++t) {
    const uint32_t key = key_of(t);
    int on = key > kth ? 1 : 0;
    if (key == kth) { on = rank < take ? 1 : 0; ++rank; }
    mask[t] = on;
  }
}

// ------------------------------------------------------------------------------------------------ 3-NN search
constexpr int KNN_BOX = 1024;      // sorted points per AABB: the all-against-all search (rtgs_knn3)
constexpr int KNN_QBOX = 256;      // ... the cross-set query (rtgs_knn3_query): boxes are pre-tested 64 at a time, so small ones cost little to skip

__global__ void __launch_bounds__(256) fill_f32_kernel(float* __restrict__ out, int n, float v) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = v;
}

__global__ void __launch_bounds__(256) knn_bbox_kernel(const float* __restrict__ pts, int N, uint32_t* __restrict__ bbox) {
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N; i += gridDim.x * 256)
#pragma unroll
    for (int c = 0; c < 3; ++c) { const uint32_t e = enc_f(pts[(size_t)i * 3 + c]); mn[c] = min(mn[c], e); mx[c] = max(mx[c], e); }
  // one set of six global atomics per WORKGROUP (per wave, on a grid of 1024 workgroups, the 24 576 atomics on six words
  // took 48 us - the longest step of the structure's build)
  __shared__ uint32_t s_mn[4][3], s_mx[4][3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      mn[c] = min(mn[c], (uint32_t)__shfl_xor((int)mn[c], off));
      mx[c] = max(mx[c], (uint32_t)__shfl_xor((int)mx[c], off));
    }
    if ((threadIdx.x & 63) == 0) { s_mn[threadIdx.x >> 6][c] = mn[c]; s_mx[threadIdx.x >> 6][c] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    const int c = threadIdx.x;
    atomicMin(&bbox[c], min(min(s_mn[0][c], s_mn[1][c]), min(s_mn[2][c], s_mn[3][c])));
    atomicMax(&bbox[3 + c], max(max(s_mx[0][c], s_mx[1][c]), max(s_mx[2][c], s_mx[3][c])));
  }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) {   // 10 bits -> every third bit
  x = (x | (x << 16)) & 0x030000FFu;
  x = (x | (x << 8)) & 0x0300F00Fu;
  x = (x | (x << 4)) & 0x030C30C3u;
  x = (x | (x << 2)) & 0x09249249u;
  return x;
}
__global__ void __launch_bounds__(256) knn_morton_kernel(const float* __restrict__ pts, int N, const uint32_t* __restrict__ bbox,
                                                         uint32_t* __restrict__ codes, uint32_t* __restrict__ order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  uint32_t q[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float lo = dec_f(bbox[c]), hi = dec_f(bbox[3 + c]);
    const float ext = hi - lo;
    const float t = ext > 0.f ? (pts[(size_t)i * 3 + c] - lo) / ext : 0.f;
    q[c] = (uint32_t)fminf(1023.f, fmaxf(0.f, t * 1023.f));
  }
  codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  order[i] = (uint32_t)i;
}

// points gathered into Morton order as float4 (w unused) + the AABB of every run of KNN_BOX sorted points
__global__ void __launch_bounds__(256) knn_gather_kernel(const float* __restrict__ pts, int N, const uint32_t* __restrict__ order,
                                                         float4* __restrict__ sorted, float* __restrict__ boxes, int box) {
  const int b = blockIdx.x;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int q = threadIdx.x; q < box; q += 256) {
    const int i = b * box + q;
    if (i < N) {
      const uint32_t o = order[i];
      const float x = pts[(size_t)o * 3], y = pts[(size_t)o * 3 + 1], z = pts[(size_t)o * 3 + 2];
      sorted[i] = make_float4(x, y, z, __int_as_float((int)o));     // .w: the point's index (rtgs_knn3_query's self test)
      mn[0] = fminf(mn[0], x); mn[1] = fminf(mn[1], y); mn[2] = fminf(mn[2], z);
      mx[0] = fmaxf(mx[0], x); mx[1] = fmaxf(mx[1], y); mx[2] = fmaxf(mx[2], z);
    }
  }
  __shared__ float s[4][6];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { mn[c] = fminf(mn[c], __shfl_xor(mn[c], off)); mx[c] = fmaxf(mx[c], __shfl_xor(mx[c], off)); }
    if ((threadIdx.x & 63) == 0) { s[threadIdx.x >> 6][c] = mn[c]; s[threadIdx.x >> 6][3 + c] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x;
    float v = s[0][c];
    for (int w = 1; w < 4; ++w) v = c < 3 ? fminf(v, s[w][c]) : fmaxf(v, s[w][c]);
    boxes[b * 6 + c] = v;
  }
}

__device__ __forceinline__ void knn_insert(float d, int j, float (&bd)[3], int (&bj)[3]) {
  if (d < bd[2]) {
    if (d < bd[1]) {
      bd[2] = bd[1]; bj[2] = bj[1];
      if (d < bd[0]) { bd[1] = bd[0]; bj[1] = bj[0]; bd[0] = d; bj[0] = j; }
      else { bd[1] = d; bj[1] = j; }
    } else { bd[2] = d; bj[2] = j; }
  }
}
__device__ __forceinline__ float dist2(const float4 a, const float4 b) {
  const float dx = a.x - b.x, dy = a.y - b.y, dz = a.z - b.z;
  return (dx * dx + dy * dy) + dz * dz;
}

// one lane per point, in Morton order (the lanes of a wave are neighbours in space, so they prune the same boxes)
__global__ void __launch_bounds__(256) knn_search_kernel(const float4* __restrict__ sorted, int N, const uint32_t* __restrict__ order,
                                                         const float* __restrict__ boxes, int nboxes,
                                                         float* __restrict__ mean_dist2, int32_t* __restrict__ idx,
                                                         float* __restrict__ dist_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const bool live = i < N;
  const float4 p = live ? sorted[i] : make_float4(0.f, 0.f, 0.f, 0.f);
  float bd[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  int bj[3] = {-1, -1, -1};
  if (live) {
    // start from the neighbours in Morton order: a tight bound before any box is opened
    for (int j = max(0, i - 3); j <= min(N - 1, i + 3); ++j)
      if (j != i) knn_insert(dist2(p, sorted[j]), j, bd, bj);
  }
  for (int b = 0; b < nboxes; ++b) {
    const float* bx = boxes + b * 6;                            // wave-uniform: scalar loads
    const float ex = fmaxf(0.f, fmaxf(bx[0] - p.x, p.x - bx[3]));
    const float ey = fmaxf(0.f, fmaxf(bx[1] - p.y, p.y - bx[4]));
    const float ez = fmaxf(0.f, fmaxf(bx[2] - p.z, p.z - bx[5]));
    // a lower bound of the distance to anything in the box, with slack for its rounding: never prunes a true neighbour
    const float lower = ((ex * ex + ey * ey) + ez * ez) * 0.9999f;
    const bool open = live && lower < bd[2];
    if (__builtin_amdgcn_ballot_w64(open) == 0ull) continue;
    if (open) {
      const int j0 = b * KNN_BOX, j1 = min(N, j0 + KNN_BOX);
      for (int j = j0; j < j1; ++j) {
        if (j == i) continue;
        const int already = (j == bj[0]) | (j == bj[1]) | (j == bj[2]);    // the +-3 seeds
        if (!already) knn_insert(dist2(p, sorted[j]), j, bd, bj);
      }
    }
  }
  if (!live) return;
  const uint32_t o = order[i];
  mean_dist2[o] = ((bd[0] + bd[1]) + bd[2]) / 3.f;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    idx[(size_t)o * 3 + k] = bj[k] >= 0 ? (int32_t)order[bj[k]] : -1;
    if (dist_out) dist_out[(size_t)o * 3 + k] = bd[k];
  }
}

// Cross-set query (rtgs_knn3_query): the three nearest REFERENCE points of every query point.  Queries come in caller order
// (no sort): a lane finds where its point would sit in the references' Morton order by a binary search of its code among
// the sorted codes and seeds its bound with the eight references around that position - close in space as a rule - so that
// the box sweep opens only boxes near the query.  `self_offset` >= 0: query i IS reference self_offset + i (the new points
// of GaussianPointCloud.update_geometry, gaussian_pointcloud.py:366-405, sit at the front of `total_xyz`) and is skipped.
__global__ void __launch_bounds__(256) knn_query_codes_kernel(const float* __restrict__ query, int Nq, const uint32_t* __restrict__ bbox,
                                                              uint32_t* __restrict__ codes, uint32_t* __restrict__ order) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nq) return;
  uint32_t q[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {                                   // the REFERENCES' box: a query outside clamps to its faces
    const float lo = dec_f(bbox[c]), hi = dec_f(bbox[3 + c]);
    const float ext = hi - lo;
    const float t = ext > 0.f ? (query[(size_t)i * 3 + c] - lo) / ext : 0.f;
    q[c] = (uint32_t)fminf(1023.f, fmaxf(0.f, t * 1023.f));
  }
  codes[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
  order[i] = (uint32_t)i;
}

// Queries are taken in MORTON order (q_order: the lanes of a wave are neighbours in space and open the same few boxes) and a
// query is PARTS lanes (lane = part * QW + query-of-the-wave, QW = 64 / PARTS): part p owns the references at sorted
// positions j with j % PARTS == p, keeps its own three best, and the lists are merged at the end.  An opened box (256
// references) comes in one round trip - four coalesced loads per lane - and is compared from LDS.  History: one lane per query,
// caller order, a global load per reference inside the compare loop: 2.8 ms per call on 100 000 references (1.9 ms in Morton
// order), 40 % of the SLAM sequence's mapping time in round 5; four lanes per query, boxes of 1 024 walked one by one with a
// scalar load each and streamed in tiles of 64: 0.31-0.42 ms; round 6 (this form): boxes pre-tested 64 at a time, the home box
// first, a box per round trip, and SIXTEEN lanes per query when the call is small (a SLAM frame asks for a few hundred new
// points: with sixteen queries per wave the launch was a handful of waves, each opening the union of its queries' boxes).
template <int PARTS>
__global__ void __launch_bounds__(256) knn_query_kernel(const float4* __restrict__ sorted, int N, const uint32_t* __restrict__ codes_sorted,
                                                        const float* __restrict__ boxes, int nboxes,
                                                        const float* __restrict__ query, int Nq, const uint32_t* __restrict__ q_order,
                                                        const uint32_t* __restrict__ q_codes_sorted, int self_offset,
                                                        const float* __restrict__ ref_box, int32_t* __restrict__ idx,
                                                        float* __restrict__ dist_out) {
  constexpr int QW = 64 / PARTS;
  __shared__ float4 s_tile[4][KNN_QBOX];
  __shared__ float s_md[4][QW][PARTS][3];
  __shared__ int s_mi[4][QW][PARTS][3];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int qw = lane % QW, part = lane / QW;
  const int slot = (blockIdx.x * 4 + wv) * QW + qw;
  const bool live = slot < Nq;
  const int i = live ? (int)q_order[slot] : 0;
  // optional open box (lo, hi): references outside it do not exist for the search (bbox_filter, SLAM/utils.py:737-744)
  float blo[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, bhi[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  if (ref_box) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { blo[c] = ref_box[c]; bhi[c] = ref_box[3 + c]; }
  }
  auto inbox = [&](const float4 s) {
    return s.x > blo[0] && s.y > blo[1] && s.z > blo[2] && s.x < bhi[0] && s.y < bhi[1] && s.z < bhi[2];
  };
  float4 p = make_float4(0.f, 0.f, 0.f, 0.f);
  float bd[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  int bj[3] = {-1, -1, -1};
  const int self = (live && self_offset >= 0) ? self_offset + i : -1;
  int home = 0;
  if (live) {
    p = make_float4(query[(size_t)i * 3], query[(size_t)i * 3 + 1], query[(size_t)i * 3 + 2], 0.f);
    const uint32_t code = q_codes_sorted[slot];
    int lo = 0, hi = N;                                         // lower bound of `code`
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (codes_sorted[mid] < code) lo = mid + 1; else hi = mid; }
    home = min(lo, N - 1) / KNN_QBOX;
  }
  // one box, for the lanes that ask for it (`open`; wave-uniform call): ONE round trip - four coalesced loads per lane, issued
  // together - then compared from LDS (tile by tile with the next tile's load behind the compares it was four dependent
  // round trips per box, and a launch of a few waves IS its chain of round trips)
  auto scan_box = [&](int b, bool open) {
    const int j0 = b * KNN_QBOX, j1 = min(N, j0 + KNN_QBOX);
    float4 in4[KNN_QBOX / 64];
#pragma unroll
    for (int q = 0; q < KNN_QBOX / 64; ++q) in4[q] = (j0 + q * 64 + lane < j1) ? sorted[j0 + q * 64 + lane] : make_float4(0.f, 0.f, 0.f, 0.f);
    __builtin_amdgcn_wave_barrier();                            // the previous box is read
#pragma unroll
    for (int q = 0; q < KNN_QBOX / 64; ++q) s_tile[wv][q * 64 + lane] = in4[q];
    __builtin_amdgcn_wave_barrier();
    if (open) {
      const int n = j1 - j0;
#pragma unroll 4
      for (int t = part; t < n; t += PARTS) {                   // positions j0 + t with (j0 + t) % PARTS == part (j0 is a multiple of 256)
        const float4 s = s_tile[wv][t];
        if (__float_as_int(s.w) != self && inbox(s)) knn_insert(dist2(p, s), j0 + t, bd, bj);
      }
    }
  };
  // Boxes: the HOME box of the wave's first query first (it holds the query's Morton neighbours: the bound is tight from
  // the start), then all others outward from it, SIXTY-FOUR AT A TIME: lane l tests box c * 64 + l against the bounding box
  // of the wave's queries and the loosest bound any of its lanes holds - a conservative form of every lane's own test - and
  // only the boxes that pass are looked at by the lanes themselves.  (Round 5 walked all boxes one by one with a scalar load
  // and its wait per box: 286 dependent round trips per wave on the 290 k references of a grown SLAM map.)
  // The bound a box is tested against is the QUERY's: each of its PARTS lanes keeps the three nearest of ITS share of the
  // references, and the third-best of one share (what a lane alone knows) is a loose bound - after the home box a lane of
  // sixteen has seen sixteen references.  The exact third-nearest of the union falls out of three steps of a PARTS-way merge
  // of the lanes' sorted triples (min over the group, the first lane that holds it advances): log2(PARTS) shuffles and a
  // ballot per step, after every box that was read.  (Round 6, last session: boxes opened per query fall several-fold.)
  unsigned long long gmask = 0ull;
#pragma unroll
  for (int k = 0; k < PARTS; ++k) gmask |= 1ull << (k * QW);
  gmask <<= qw;
  auto query_bound = [&]() -> float {
    float h0 = bd[0], h1 = bd[1], h2 = bd[2];                  // this lane's sorted triple; h0 is its head
    float m = FLT_MAX;
#pragma unroll
    for (int step = 0; step < 3; ++step) {
      m = h0;
#pragma unroll
      for (int off = QW; off < 64; off <<= 1) m = fminf(m, __shfl_xor(m, off));
      const unsigned long long holders = __builtin_amdgcn_ballot_w64(h0 == m) & gmask;
      if (holders && lane == (int)__builtin_ctzll(holders)) { h0 = h1; h1 = h2; h2 = FLT_MAX; }
    }
    return m;
  };
  const unsigned long long lv = __builtin_amdgcn_ballot_w64(live);
  const int b0 = lv ? min(nboxes - 1, __builtin_amdgcn_readlane(home, __builtin_ctzll(lv))) : 0;
  scan_box(b0, live);
  float qb = query_bound();
  float wlo[3], whi[3];
  {
    const float q3[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      float a = live ? q3[c] : FLT_MAX, b = live ? q3[c] : -FLT_MAX;
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) { a = fminf(a, __shfl_xor(a, off)); b = fmaxf(b, __shfl_xor(b, off)); }
      wlo[c] = a; whi[c] = b;
    }
  }
  const int nchunks = (nboxes + 63) >> 6, c0 = b0 >> 6;
  for (int step = 0; step < 2 * nchunks; ++step) {
    const int d = (step + 1) >> 1;
    const int ck = (step & 1) ? c0 - d : c0 + d;                // c0, c0 - 1, c0 + 1, c0 - 2, ...
    if (ck < 0 || ck >= nchunks) continue;
    float wb = live ? qb : 0.f;                                 // the loosest bound of the wave
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wb = fmaxf(wb, __shfl_xor(wb, off));
    const int bl = ck * 64 + lane;
    float b6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool cand = false;
    if (bl < nboxes && bl != b0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) b6[c] = boxes[bl * 6 + c];
      const float ex = fmaxf(0.f, fmaxf(b6[0] - whi[0], wlo[0] - b6[3]));
      const float ey = fmaxf(0.f, fmaxf(b6[1] - whi[1], wlo[1] - b6[4]));
      const float ez = fmaxf(0.f, fmaxf(b6[2] - whi[2], wlo[2] - b6[5]));
      const float lower = ((ex * ex + ey * ey) + ez * ez) * 0.9999f;
      // a run of references entirely outside the filter box holds nothing to find
      const bool outside = b6[3] <= blo[0] || b6[4] <= blo[1] || b6[5] <= blo[2] || b6[0] >= bhi[0] || b6[1] >= bhi[1] || b6[2] >= bhi[2];
      cand = !outside && lower < wb;
    }
    unsigned long long todo = __builtin_amdgcn_ballot_w64(cand);
    while (todo) {
      const int l = __builtin_ctzll(todo);
      todo &= todo - 1ull;
      float bx[6];
#pragma unroll
      for (int c = 0; c < 6; ++c) bx[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b6[c]), l));
      const float ex = fmaxf(0.f, fmaxf(bx[0] - p.x, p.x - bx[3]));
      const float ey = fmaxf(0.f, fmaxf(bx[1] - p.y, p.y - bx[4]));
      const float ez = fmaxf(0.f, fmaxf(bx[2] - p.z, p.z - bx[5]));
      // a lower bound of the distance to anything in the box, with slack for its rounding: never prunes a true neighbour
      const float lower = ((ex * ex + ey * ey) + ez * ez) * 0.9999f;
      const bool open = live && lower < qb;
      if (__builtin_amdgcn_ballot_w64(open) == 0ull) continue;
      scan_box(ck * 64 + l, open);
      qb = query_bound();
    }
  }
  // ---- merge the parts of every query (disjoint reference sets: no duplicates)
  int bid[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) bid[k] = bj[k] >= 0 ? __float_as_int(sorted[bj[k]].w) : -1;
#pragma unroll
  for (int k = 0; k < 3; ++k) { s_md[wv][qw][part][k] = bd[k]; s_mi[wv][qw][part][k] = bid[k]; }
  __builtin_amdgcn_wave_barrier();
  if (!live || part != 0) return;
  for (int q = 1; q < PARTS; ++q)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int id = s_mi[wv][qw][q][k];
      if (id >= 0) knn_insert(s_md[wv][qw][q][k], id, bd, bid);
    }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    idx[(size_t)i * 3 + k] = bid[k];
    if (dist_out) dist_out[(size_t)i * 3 + k] = bd[k];
  }
}

// knn_dynamic_merge: the neighbour search of Mapping.temp_to_optimize split by what changes between frames.  The three nearest
// STABLE Gaussians of every query come from a structure that is rebuilt only when the stable rows change (rtgs_knn3_build_ref /
// rtgs_knn3_query_built: the Morton sort of ~300 000 references was two thirds of the frame's search); the few thousand references
// that do change - the queries themselves (a new point is not its own neighbour) and the unstable Gaussians - are compared
// directly, one wave per query, and merged in.  Index space of the result = cat(queries, all existing rows): a query i is i, the
// stable row r is Nq + r, the unstable row u (u-th row behind the n_stable stable ones) is Nq + n_stable + u.  Same distance
// expression and the same open box as knn_query_kernel, so the result equals the one-structure search (up to the order of
// equidistant neighbours).
__global__ void __launch_bounds__(256) knn_dynamic_merge_kernel(const float* __restrict__ query, int Nq, const float* __restrict__ unstable,
                                                                int Nu, int n_stable, const float* __restrict__ d2_stable,
                                                                const int32_t* __restrict__ idx_stable, const float* __restrict__ ref_box,
                                                                int32_t* __restrict__ idx, float* __restrict__ dist_out) {
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= Nq) return;
  float blo[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, bhi[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  if (ref_box) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { blo[c] = ref_box[c]; bhi[c] = ref_box[3 + c]; }
  }
  const float4 p = make_float4(query[(size_t)i * 3], query[(size_t)i * 3 + 1], query[(size_t)i * 3 + 2], 0.f);
  float bd[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  int bj[3] = {-1, -1, -1};
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int r = idx_stable[(size_t)i * 3 + k];
      if (r >= 0) knn_insert(d2_stable[(size_t)i * 3 + k], Nq + r, bd, bj);
    }
  }
  for (int j = lane; j < Nq + Nu; j += 64) {
    if (j == i) continue;
    const float* src = j < Nq ? query + (size_t)j * 3 : unstable + (size_t)(j - Nq) * 3;
    const float4 s = make_float4(src[0], src[1], src[2], 0.f);
    if (s.x > blo[0] && s.y > blo[1] && s.z > blo[2] && s.x < bhi[0] && s.y < bhi[1] && s.z < bhi[2])
      knn_insert(dist2(p, s), j < Nq ? j : j + n_stable, bd, bj);
  }
  // 64 sorted triples -> the three smallest: butterfly of triple merges
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    float od[3]; int oj[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { od[k] = __shfl_xor(bd[k], off); oj[k] = __shfl_xor(bj[k], off); }
#pragma unroll
    for (int k = 0; k < 3; ++k) if (oj[k] >= 0) knn_insert(od[k], oj[k], bd, bj);
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k) { idx[(size_t)i * 3 + k] = bj[k]; if (dist_out) dist_out[(size_t)i * 3 + k] = bd[k]; }
  }
}

struct KnnLayout {
  size_t bbox, codes, codes_sorted, order_in, order, sorted, boxes, cub, total, cub_bytes;
};
static KnnLayout knn_layout(int N) {
  KnnLayout L{};
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t n = (size_t)(N > 0 ? N : 1);
  size_t off = 0;
  L.bbox = off; off = al(off + 6 * sizeof(uint32_t));
  L.codes = off; off = al(off + n * 4);
  L.codes_sorted = off; off = al(off + n * 4);
  L.order_in = off; off = al(off + n * 4);
  L.order = off; off = al(off + n * 4);
  L.sorted = off; off = al(off + n * sizeof(float4));
  L.boxes = off; off = al(off + ((n + KNN_QBOX - 1) / KNN_QBOX) * 6 * sizeof(float));      // sized for the smaller of the two box sizes
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, n, 0u, 30u);
  L.cub_bytes = tb;
  L.cub = off; off = al(off + tb);
  L.total = off;
  return L;
}

// ------------------------------------------------------------------------------------------------ error accumulation
// Lanes of a wave that point at the same Gaussian are summed first (an opaque disc owns runs of neighbouring pixels), so
// a group costs one set of atomics instead of one per pixel.
__global__ void __launch_bounds__(256) accumulate_error_kernel(int n, const float* __restrict__ ce, const float* __restrict__ de,
                                                               const float* __restrict__ ne, const int32_t* __restrict__ ci,
                                                               const int32_t* __restrict__ di, float thr_c, float thr_d,
                                                               float thr_n, float* __restrict__ g_c, float* __restrict__ g_d,
                                                               float* __restrict__ g_n, int32_t* __restrict__ outl,
                                                               float* __restrict__ cnt_c, float* __restrict__ cnt_d, int P) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int lane = threadIdx.x & 63;
  const bool live = i < n;
  // indices outside [0, P) (stale index maps after a delete, or P passed as a subset count) are ignored like -1
  int ic = live ? ci[i] : -1, id = live ? di[i] : -1;
  if ((unsigned)ic >= (unsigned)P) ic = -1;
  if ((unsigned)id >= (unsigned)P) id = -1;
  const float ec = live ? ce[i] : 0.f, ed = live ? de[i] : 0.f, en = live ? ne[i] : 0.f;
  // Runs of equal owners along the wave (an opaque disc owns runs of neighbouring pixels) are summed by a SEGMENTED scan - six
  // shuffle rounds whatever the number of runs - and the last lane of a run adds its totals.  (Round 5 looped over the wave's
  // distinct owners with three to four full wave reductions each: 5-10 owners per wave on a SLAM map, 109 us per frame.)
  auto seg_scan = [&](int key, float (&v)[4], int nv) -> bool {
    const int prev = __shfl_up(key, 1);
    int head = (lane == 0 || prev != key) ? 1 : 0;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      float o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = q < nv ? __shfl_up(v[q], off) : 0.f;
      const int oh = __shfl_up(head, off);
      if (lane >= off && !head) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += o[q];
        head |= oh;
      }
    }
    const int next = __shfl_down(key, 1);
    return lane == 63 || next != key;            // the run's last lane holds its totals
  };
  {
    float v[4] = {ec, 1.f, ec > thr_c ? 1.f : 0.f, 0.f};
    const bool tail = seg_scan(ic, v, 3);
    if (tail && ic >= 0) {
      unsafeAtomicAdd(&g_c[ic], v[0]); unsafeAtomicAdd(&cnt_c[ic], v[1]);
      if (v[2] > 0.f) atomicAdd(&outl[ic], (int)v[2]);
    }
  }
  {
    float v[4] = {ed, en, 1.f, (ed > thr_d ? 1.f : 0.f) + (en > thr_n ? 1.f : 0.f)};
    const bool tail = seg_scan(id, v, 4);
    if (tail && id >= 0) {
      unsafeAtomicAdd(&g_d[id], v[0]); unsafeAtomicAdd(&g_n[id], v[1]); unsafeAtomicAdd(&cnt_d[id], v[2]);
      if (v[3] > 0.f) atomicAdd(&outl[id], (int)v[3]);
    }
  }
}
__global__ void __launch_bounds__(256) error_mean_kernel(int P, float* __restrict__ g_c, float* __restrict__ g_d,
                                                         float* __restrict__ g_n, const float* __restrict__ cnt_c,
                                                         const float* __restrict__ cnt_d) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float cc = fmaxf(cnt_c[i], 1.f), cd = fmaxf(cnt_d[i], 1.f);
  g_c[i] = g_c[i] / cc; g_d[i] = g_d[i] / cd; g_n[i] = g_n[i] / cd;
}

// ------------------------------------------------------------------------------------------------ frame preprocessing
constexpr int BF_MAXR = 8;
__global__ void __launch_bounds__(256) bilateral_kernel(const float* __restrict__ depth, int H, int W, int radius,
                                                        float inv2ss, float two_sc2, float* __restrict__ out) {
  __shared__ float s_t[(16 + 2 * BF_MAXR) * (16 + 2 * BF_MAXR)];
  const int tw = 16 + 2 * radius;
  const int x0 = blockIdx.x * 16 - radius, y0 = blockIdx.y * 16 - radius;
  for (int q = threadIdx.x; q < tw * tw; q += 256) {
    const int x = x0 + q % tw, y = y0 + q / tw;
    s_t[q] = (x >= 0 && x < W && y >= 0 && y < H) ? depth[(size_t)y * W + x] : 0.f;      // zero padding
  }
  __syncthreads();
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int x = blockIdx.x * 16 + lx, y = blockIdx.y * 16 + ly;
  if (x >= W || y >= H) return;
  const float d = s_t[(ly + radius) * tw + lx + radius];
  float wsum = 0.f, psum = 0.f;
  for (int i = -radius; i <= radius; ++i)
    for (int j = -radius; j <= radius; ++j) {
      if (i * i + j * j > radius * radius) continue;
      const float tap = s_t[(ly + radius + i) * tw + lx + radius + j];
      const float sw = -(float)(i * i + j * j) * inv2ss;
      const float df = d - tap;
      const float cw = -(df * df) / two_sc2;
      const float w = tap != 0.f ? expf(sw + cw) : 0.f;
      wsum += w;
      psum += w * tap;
    }
  out[(size_t)y * W + x] = wsum == 0.f ? 0.f : psum / wsum;
}

// range mask + back-projection (SLAM/utils.py:65-75) + min / max of the masked depth (for compute_normal_map's mask)
__global__ void __launch_bounds__(256) frame_vertex_kernel(const float* __restrict__ depth, int H, int W, const float* __restrict__ K,
                                                           float dmin, float dmax, float* __restrict__ vertex,
                                                           uint32_t* __restrict__ mm) {
  // grid-stride over at most 512 workgroups: one pair of same-address atomics per WORKGROUP below - with a workgroup per 256
  // pixels (3 188 of them at 1200x680) those atomics were most of the kernel's 76 us
  uint32_t emin = 0xffffffffu, emax = 0u;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < H * W; i += gridDim.x * 256) {
    float d = depth[i];
    if (!((d > dmin) && (d < dmax))) d = 0.f;
    const int x = i % W, y = i / W;
    vertex[(size_t)i * 3] = (((float)x - K[2]) / K[0]) * d;
    vertex[(size_t)i * 3 + 1] = (((float)y - K[5]) / K[4]) * d;
    vertex[(size_t)i * 3 + 2] = d;
    const uint32_t e = enc_f(d);
    emin = min(emin, e); emax = max(emax, e);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    emin = min(emin, (uint32_t)__shfl_xor((int)emin, off));
    emax = max(emax, (uint32_t)__shfl_xor((int)emax, off));
  }
  __shared__ uint32_t s[2][4];
  if ((threadIdx.x & 63) == 0) { s[0][threadIdx.x >> 6] = emin; s[1][threadIdx.x >> 6] = emax; }
  __syncthreads();
  if (threadIdx.x == 0) {
    atomicMin(&mm[0], min(min(s[0][0], s[0][1]), min(s[0][2], s[0][3])));
    atomicMax(&mm[1], max(max(s[1][0], s[1][1]), max(s[1][2], s[1][3])));
  }
}

// Sobel normal (replicate pad, conv2d tap order, torch.cross / torch.norm rounding - see icp.hip), |cos| to the viewing
// ray (SLAM/utils.py:124-139), invalid-confidence mask, and the four zeroed output maps (tracker.py:121-131)
__global__ void __launch_bounds__(256) frame_normal_kernel(const float* __restrict__ V, int H, int W, const float* __restrict__ K,
                                                           const uint32_t* __restrict__ mm, float conf_thr,
                                                           float* __restrict__ depth_out, float* __restrict__ vertex_out,
                                                           float* __restrict__ normal_out, float* __restrict__ conf_out,
                                                           uint8_t* __restrict__ bad_out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= H * W) return;
  const int y = idx / W, x = idx % W;
  const int ym = max(y - 1, 0), yp = min(y + 1, H - 1), xm = max(x - 1, 0), xp = min(x + 1, W - 1);
  float gx[3], gy[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a00 = V[((size_t)ym * W + xm) * 3 + c], a01 = V[((size_t)ym * W + x) * 3 + c], a02 = V[((size_t)ym * W + xp) * 3 + c];
    const float a10 = V[((size_t)y * W + xm) * 3 + c], a12 = V[((size_t)y * W + xp) * 3 + c];
    const float a20 = V[((size_t)yp * W + xm) * 3 + c], a21 = V[((size_t)yp * W + x) * 3 + c], a22 = V[((size_t)yp * W + xp) * 3 + c];
    gx[c] = -a00 + a02 - 2.f * a10 + 2.f * a12 - a20 + a22;
    gy[c] = -a00 - 2.f * a01 - a02 + a20 + 2.f * a21 + a22;
  }
  float nx = __fmaf_rn(gy[1], gx[2], -(gy[2] * gx[1]));
  float ny = __fmaf_rn(gy[2], gx[0], -(gy[0] * gx[2]));
  float nz = __fmaf_rn(gy[0], gx[1], -(gy[1] * gx[0]));
  const float mag = sqrtf(__fmaf_rn(nz, nz, __fmaf_rn(ny, ny, nx * nx))) + 1e-8f;
  nx /= mag; ny /= mag; nz /= mag;
  const float vx = V[(size_t)idx * 3], vy = V[(size_t)idx * 3 + 1], d = V[(size_t)idx * 3 + 2];
  if (d <= dec_f(mm[0]) || d >= dec_f(mm[1])) { nx = 0.f; ny = 0.f; nz = 0.f; }
  // viewing ray, normalised with +1e-8 (compute_confidence_map), then F.cosine_similarity(eps = 1e-8)
  float rx = ((float)x - K[2]) / K[0], ry = ((float)y - K[5]) / K[4], rz = 1.f;
  const float rm = sqrtf(__fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx))) + 1e-8f;
  rx /= rm; ry /= rm; rz /= rm;
  const float na = fmaxf(sqrtf(__fmaf_rn(nz, nz, __fmaf_rn(ny, ny, nx * nx))), 1e-8f);
  const float nb = fmaxf(sqrtf(__fmaf_rn(rz, rz, __fmaf_rn(ry, ry, rx * rx))), 1e-8f);
  const float cs = ((nx / na) * (rx / nb) + (ny / na) * (ry / nb)) + (nz / na) * (rz / nb);
  const float conf = fabsf(cs);
  const bool bad = (nx == 0.f && ny == 0.f && nz == 0.f) || (conf < conf_thr);
  depth_out[idx] = bad ? 0.f : d;
  vertex_out[(size_t)idx * 3] = bad ? 0.f : vx; vertex_out[(size_t)idx * 3 + 1] = bad ? 0.f : vy; vertex_out[(size_t)idx * 3 + 2] = bad ? 0.f : d;
  normal_out[(size_t)idx * 3] = bad ? 0.f : nx; normal_out[(size_t)idx * 3 + 1] = bad ? 0.f : ny; normal_out[(size_t)idx * 3 + 2] = bad ? 0.f : nz;
  conf_out[idx] = bad ? 0.f : conf;
  bad_out[idx] = bad ? 1 : 0;
}

__global__ void __launch_bounds__(256) sample_flags_kernel(const float* __restrict__ normal, const uint8_t* __restrict__ select,
                                                           int n, uint8_t* __restrict__ flags) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = (normal[(size_t)i * 3] + normal[(size_t)i * 3 + 1]) + normal[(size_t)i * 3 + 2];
  flags[i] = ((!select || select[i] != 0) && s != 0.f) ? 1 : 0;
}

static inline int grid1(long long n) { return (int)((n + 255) / 256); }

// ---- the new Gaussians of a frame: what Mapping._new_points and the tail of Mapping.temp_to_optimize spell as ~80 torch
// operations (gaussian_pointcloud.py:305-405, SLAM/utils.py:216-221, general_utils.py:185-191), as two kernels.  Same float32
// operations in the same order (this file is built without FMA contraction), so results equal the torch form bit for bit
// wherever torch's own kernels evaluate the same expression tree (tests/test_slam_ops_gpu.py).
// gather_new_points: the sampled pixels `pick` -> position, UNIT normal, colour, rotation z -> normal.
__global__ void __launch_bounds__(256) gather_new_points_kernel(const int64_t* __restrict__ pick, int n, const float* __restrict__ vertex,
                                                                const float* __restrict__ normal, const float* __restrict__ color,
                                                                int identity_rot, float* __restrict__ xyz, float* __restrict__ nrm,
                                                                float* __restrict__ col, float* __restrict__ rot) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const int64_t q = pick[i];
  const float vx = vertex[q * 3], vy = vertex[q * 3 + 1], vz = vertex[q * 3 + 2];
  const float nx = normal[q * 3], ny = normal[q * 3 + 1], nz = normal[q * 3 + 2];
  xyz[i * 3] = vx; xyz[i * 3 + 1] = vy; xyz[i * 3 + 2] = vz;
  col[i * 3] = color[q * 3]; col[i * 3 + 1] = color[q * 3 + 1]; col[i * 3 + 2] = color[q * 3 + 2];
  const float len = sqrtf((nx * nx + ny * ny) + nz * nz) + 1e-8f;         // normals / (norm + 1e-8), gaussian_pointcloud.py:315-318
  const float tx = nx / len, ty = ny / len, tz = nz / len;
  nrm[i * 3] = tx; nrm[i * 3 + 1] = ty; nrm[i * 3 + 2] = tz;
  if (identity_rot) { rot[i * 4] = 1.f; rot[i * 4 + 1] = 0.f; rot[i * 4 + 2] = 0.f; rot[i * 4 + 3] = 0.f; return; }
  // compute_rot((0, 0, 1), t): axis = z x t = (-t_y, t_x, 0), normalised TWICE as the reference does, angle = acos(t_z)
  float ax = 0.f * tz - 1.f * ty, ay = 1.f * tx - 0.f * tz, az = 0.f * ty - 0.f * tx;
  float al = sqrtf((ax * ax + ay * ay) + az * az) + 1e-8f;
  ax = ax / al; ay = ay / al; az = az / al;
  const float angle = acosf(tz);
  al = sqrtf((ax * ax + ay * ay) + az * az) + 1e-8f;
  ax = ax / al; ay = ay / al; az = az / al;
  const float h = angle / 2.f, sh = sinf(h);
  rot[i * 4] = cosf(h); rot[i * 4 + 1] = ax * sh; rot[i * 4 + 2] = ay * sh; rot[i * 4 + 3] = az * sh;
}

// draw_new_points: the uniform draw without replacement of SLAM/utils.py:171 (randperm(n_cand)[:k]) and the gather above in
// one kernel.  Output i takes candidate perm(i), perm = a keyed bijection of [0, n_cand): a balanced Feistel network over the
// next even number of bits, walked until it lands inside the range (cycle walking keeps it a bijection; the domain is < 4 n_cand,
// so a walk takes under four steps on average).  A prefix of a permutation never repeats an element, which is all "without
// replacement" asks; the key is a fresh 64-bit word per pass from the host's seeded generator.
__device__ __forceinline__ uint32_t draw_mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t draw_perm(uint32_t i, uint32_t n, uint32_t half_bits, uint64_t key) {
  const uint32_t mask = (1u << half_bits) - 1u;
  const uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
  uint32_t x = i;
  do {
    uint32_t l = x >> half_bits, r = x & mask;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      const uint32_t f = draw_mix(r ^ ((round & 1) ? k1 : k0) ^ (0x9e3779b9u * (uint32_t)(round + 1))) & mask;
      const uint32_t t = l ^ f;
      l = r; r = t;
    }
    x = (l << half_bits) | r;
  } while (x >= n);
  return x;
}
__global__ void __launch_bounds__(256) draw_new_points_kernel(const int32_t* __restrict__ cand, uint32_t n_cand, int k, uint32_t half_bits,
                                                              uint64_t key, const float* __restrict__ vertex,
                                                              const float* __restrict__ normal, const float* __restrict__ color,
                                                              int identity_rot, float* __restrict__ xyz, float* __restrict__ nrm,
                                                              float* __restrict__ col, float* __restrict__ rot, int32_t* __restrict__ pick_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= k) return;
  const int64_t q = cand[draw_perm((uint32_t)i, n_cand, half_bits, key)];
  if (pick_out) pick_out[i] = (int32_t)q;
  const float vx = vertex[q * 3], vy = vertex[q * 3 + 1], vz = vertex[q * 3 + 2];
  const float nx = normal[q * 3], ny = normal[q * 3 + 1], nz = normal[q * 3 + 2];
  xyz[i * 3] = vx; xyz[i * 3 + 1] = vy; xyz[i * 3 + 2] = vz;
  col[i * 3] = color[q * 3]; col[i * 3 + 1] = color[q * 3 + 1]; col[i * 3 + 2] = color[q * 3 + 2];
  const float len = sqrtf((nx * nx + ny * ny) + nz * nz) + 1e-8f;
  const float tx = nx / len, ty = ny / len, tz = nz / len;
  nrm[i * 3] = tx; nrm[i * 3 + 1] = ty; nrm[i * 3 + 2] = tz;
  if (identity_rot) { rot[i * 4] = 1.f; rot[i * 4 + 1] = 0.f; rot[i * 4 + 2] = 0.f; rot[i * 4 + 3] = 0.f; return; }
  float ax = 0.f * tz - 1.f * ty, ay = 1.f * tx - 0.f * tz, az = 0.f * ty - 0.f * tx;      // as gather_new_points_kernel
  float al = sqrtf((ax * ax + ay * ay) + az * az) + 1e-8f;
  ax = ax / al; ay = ay / al; az = az / al;
  const float angle = acosf(tz);
  al = sqrtf((ax * ax + ay * ay) + az * az) + 1e-8f;
  ax = ax / al; ay = ay / al; az = az / al;
  const float h = angle / 2.f, sh = sinf(h);
  rot[i * 4] = cosf(h); rot[i * 4 + 1] = ax * sh; rot[i * 4 + 2] = ay * sh; rot[i * 4 + 3] = az * sh;
}

// filter_keep: the decision of Mapping.temp_points_filter (mapper.py:803-827) behind the neighbour query: a new point is dropped
// when one of its (up to three) nearest unstable Gaussians is closer than 0.6 of that Gaussian's radius ((sum - min) / 2 of its
// activated scales, gaussian_pointcloud.py:515-519).  The same float32 operations as the tensor form (sqrt, *, <).
__global__ void __launch_bounds__(256) filter_keep_kernel(int n, const float* __restrict__ d2, const int32_t* __restrict__ idx,
                                                          const float* __restrict__ scales, float ratio, uint8_t* __restrict__ keep) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  bool inside = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int32_t j = idx[i * 3 + k];
    if (j < 0) continue;
    const float* sc = scales + (size_t)j * 3;
    const float s0 = sc[0], s1 = sc[1], s2 = sc[2];
    const float radius = (((s0 + s1) + s2) - fminf(fminf(s0, s1), s2)) / 2.f;
    inside = inside || (sqrtf(d2[i * 3 + k]) < radius * ratio);
  }
  keep[i] = inside ? 0 : 1;
}

// bbox_pad: [min - pad | max + pad] of n points (the box Mapping hands the neighbour query, SLAM/utils.py:737-744's bounds) as
// ONE single-workgroup launch instead of min, max, two offsets and a concatenation.  n is a frame's new points (<= ~41 000).
__global__ void __launch_bounds__(1024) bbox_pad_kernel(int n, const float* __restrict__ xyz, float pad, float* __restrict__ out6) {
  __shared__ float s_lo[16][3], s_hi[16][3];
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (int i = threadIdx.x; i < n; i += 1024) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float v = xyz[i * 3 + c]; lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v); }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
  }
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) { s_lo[w][c] = lo[c]; s_hi[w][c] = hi[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 3) {
    float a = s_lo[0][threadIdx.x], b = s_hi[0][threadIdx.x];
    for (int k = 1; k < 16; ++k) { a = fminf(a, s_lo[k][threadIdx.x]); b = fmaxf(b, s_hi[k][threadIdx.x]); }
    out6[threadIdx.x] = a - pad;
    out6[3 + threadIdx.x] = b + pad;
  }
}

// Ordered compactions of a frame's new points, one single-workgroup launch each (n <= ~41 000 candidates; the tensor form is
// nonzero - five launches and the host wait - plus one gather per array).  block_slot: this thread's output slot within a
// chunk of 1024 flags and the chunk's total, by ballot + sixteen wave counts.
__device__ __forceinline__ int block_slot(bool flag, int* s_cnt, int& total) {
  const unsigned long long b = __ballot(flag);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) s_cnt[w] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { const int c = s_cnt[k]; off += k < w ? c : 0; tot += c; }
  __syncthreads();
  total = tot;
  return off + __popcll(b & ((1ull << lane) - 1ull));
}
// compact_points: the candidates the filter kept (Mapping.temp_to_optimize's first compaction, mapper.py:826), in order.
__global__ void __launch_bounds__(1024) compact_points_kernel(int n, const uint8_t* __restrict__ keep, const float* __restrict__ xyz,
                                                              const float* __restrict__ color, const float* __restrict__ opac,
                                                              const float* __restrict__ rots, float* __restrict__ o_xyz,
                                                              float* __restrict__ o_color, float* __restrict__ o_opac,
                                                              float* __restrict__ o_rots, int32_t* __restrict__ count_out) {
  __shared__ int s_cnt[16];
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + (int)threadIdx.x;
    const bool flag = i < n && keep[i] != 0;
    int tot;
    const int o = base + block_slot(flag, s_cnt, tot);
    if (flag) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { o_xyz[o * 3 + c] = xyz[i * 3 + c]; o_color[o * 3 + c] = color[i * 3 + c]; }
      o_opac[o] = opac[i];
#pragma unroll
      for (int c = 0; c < 4; ++c) o_rots[o * 4 + c] = rots[i * 4 + c];
    }
    base += tot;
  }
  if (threadIdx.x == 0) *count_out = base;
}
// append_valid_rows: the second compaction (the candidates update_geometry accepts) straight into the map's own arrays behind
// its last row - packed rows split into the three parameter blocks, every side array set to its value for a new row.
struct AuxFill { uint32_t* dst[8]; uint32_t bits[8]; int n; };
__global__ void __launch_bounds__(1024) append_valid_rows_kernel(int n, const uint8_t* __restrict__ valid, const float* __restrict__ rows59,
                                                                 float* __restrict__ d_xyz, float* __restrict__ d_shs,
                                                                 float* __restrict__ d_raw8, AuxFill aux, int32_t* __restrict__ count_out) {
  __shared__ int s_cnt[16];
  int base = 0;
  for (int c0 = 0; c0 < n; c0 += 1024) {
    const int i = c0 + (int)threadIdx.x;
    const bool flag = i < n && valid[i] != 0;
    int tot;
    const int o = base + block_slot(flag, s_cnt, tot);
    if (flag) {
      const float* r = rows59 + (size_t)i * 59;
#pragma unroll
      for (int c = 0; c < 3; ++c) d_xyz[(size_t)o * 3 + c] = r[c];
#pragma unroll 4
      for (int c = 0; c < 48; ++c) d_shs[(size_t)o * 48 + c] = r[3 + c];
#pragma unroll
      for (int c = 0; c < 8; ++c) d_raw8[(size_t)o * 8 + c] = r[51 + c];
      for (int k = 0; k < aux.n; ++k) aux.dst[k][o] = aux.bits[k];
    }
    base += tot;
  }
  if (threadIdx.x == 0) *count_out = base;
}

// new_rows: update_geometry (gaussian_pointcloud.py:366-405) + the packing of the new rows (mapper.py:886-899).  Candidate i
// has its three nearest neighbours among (the n candidates, then the existing Gaussians): in-plane scale = rms of
// (distance - 3 radius) over the three, clamped; a candidate INSIDE three radii of a neighbour is invalid.  Writes the packed
// 59-column row of EVERY candidate (the caller keeps the valid ones) and the validity byte.
__global__ void __launch_bounds__(256) new_rows_kernel(int n, const float* __restrict__ xyz, const float* __restrict__ color,
                                                       const float* __restrict__ opacity_raw, const float* __restrict__ rots,
                                                       const float* __restrict__ d2, const int32_t* __restrict__ idx,
                                                       const float* __restrict__ exist_scales, float min_radius, float max_radius,
                                                       float scale_factor, float fx, float fy, float fz,
                                                       float* __restrict__ packed, uint8_t* __restrict__ valid) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int32_t j = idx[i * 3 + k];
    float dist = 1e30f;
    if (j >= 0) {
      float radius = 1e-6f;                                          // get_radius of a still unscaled candidate
      if (j >= n) {
        const float* sc = exist_scales + (size_t)(j - n) * 3;
        const float s0 = sc[0], s1 = sc[1], s2 = sc[2];
        radius = (((s0 + s1) + s2) - fminf(fminf(s0, s1), s2)) / 2.f; // (sum - min) / 2, gaussian_pointcloud.py:515-519
      }
      dist = sqrtf(d2[i * 3 + k]) - 3.f * radius;
    }
    bad = bad || (dist < 0.f);
    acc = k == 0 ? dist * dist : acc + dist * dist;
  }
  float sc = sqrtf(acc / 3.f);
  sc = fminf(fmaxf(sc, min_radius), max_radius);
  float* row = packed + (size_t)i * 59;
#pragma unroll 1
  for (int c = 0; c < 59; ++c) row[c] = 0.f;
  row[0] = xyz[i * 3]; row[1] = xyz[i * 3 + 1]; row[2] = xyz[i * 3 + 2];
#pragma unroll
  for (int c = 0; c < 3; ++c) row[3 + c] = (color[i * 3 + c] - 0.5f) / 0.28209479177387814f;      // RGB2SH
  row[51] = opacity_raw[i];
  const float base = scale_factor * sc;
  row[52] = logf(base * fx); row[53] = logf(base * fy); row[54] = logf(base * fz);
#pragma unroll
  for (int c = 0; c < 4; ++c) row[55 + c] = rots[i * 4 + c];
  valid[i] = bad ? 0 : 1;
}

// ---- two per-frame bookkeeping passes of the mapper as single kernels (round 6; each was 12-14 tensor operations and a
// host synchronisation of its own).
// error_counters (mapper.py:541-565): strikes of the stable Gaussians - depth / colour error above twice the add threshold -
// the delete and release decisions at `limit` strikes, and their counts.
__global__ void __launch_bounds__(256) error_counters_kernel(int nf, const float* __restrict__ g_color, const float* __restrict__ g_depth,
                                                             float thr_c, float thr_d, int32_t* __restrict__ dcnt,
                                                             int32_t* __restrict__ ccnt, int limit, uint8_t* __restrict__ ddel,
                                                             uint8_t* __restrict__ crel, uint32_t* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool del = false, rel = false;
  if (i < nf) {
    const int d = dcnt[i] + (g_depth[i] > thr_d ? 1 : 0), c = ccnt[i] + (g_color[i] > thr_c ? 1 : 0);
    dcnt[i] = d; ccnt[i] = c;
    del = d >= limit;
    rel = c >= limit && !del;
    ddel[i] = del ? 1 : 0; crel[i] = rel ? 1 : 0;
  }
  const unsigned long long md = __builtin_amdgcn_ballot_w64(del), mr = __builtin_amdgcn_ballot_w64(rel);
  if ((threadIdx.x & 63) == 0) {
    if (md) atomicAdd(&counts[0], (uint32_t)__popcll(md));
    if (mr) atomicAdd(&counts[1], (uint32_t)__popcll(mr));
  }
}

// delete_mask (mapper.py:298-335): radius = (sum of scales - smallest) / 2 > 10 x the cloud's mean radius, or - unstable cloud -
// older than the time window.  ONE workgroup (the unstable cloud of a SLAM map is a few thousand rows): mean first, then the
// mask and its count.
__global__ void __launch_bounds__(1024) delete_mask_kernel(int n, const float* __restrict__ scales, const int32_t* __restrict__ add_tick,
                                                           int time_now, int window, uint8_t* __restrict__ mask,
                                                           uint32_t* __restrict__ count) {
  __shared__ double s_sum[16];
  __shared__ uint32_t s_cnt[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  double acc = 0.0;
  for (int i = tid; i < n; i += 1024) {
    const float s0 = scales[i * 3], s1 = scales[i * 3 + 1], s2 = scales[i * 3 + 2];
    acc += (double)((((s0 + s1) + s2) - fminf(fminf(s0, s1), s2)) / 2.f);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
  if (lane == 0) s_sum[w] = acc;
  __syncthreads();
  double tot = 0.0;
  for (int q = 0; q < 16; ++q) tot += s_sum[q];
  const float thr = (float)(tot / (double)n) * 10.f;
  uint32_t c = 0;
  for (int i = tid; i < n; i += 1024) {
    const float s0 = scales[i * 3], s1 = scales[i * 3 + 1], s2 = scales[i * 3 + 2];
    const float r = (((s0 + s1) + s2) - fminf(fminf(s0, s1), s2)) / 2.f;
    const bool d = r > thr || (add_tick != nullptr && (time_now - add_tick[i]) > window);
    mask[i] = d ? 1 : 0;
    c += d ? 1u : 0u;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off);
  if (lane == 0) s_cnt[w] = c;
  __syncthreads();
  if (tid == 0) { uint32_t t = 0; for (int q = 0; q < 16; ++q) t += s_cnt[q]; count[0] = t; }
}

}  // namespace rtgs_slam

using namespace rtgs_slam;

#define SLAM_TRY(expr)                    \
  do {                                    \
    if ((expr) != hipSuccess) return -2;  \
  } while (0)


namespace {
// Renderer.render's normal map (SLAM/render.py:130-133): out[:, p] = rows[index[p]] where index[p] >= 0, else 0.
__global__ void __launch_bounds__(256) gather_rows3_kernel(const float* __restrict__ rows, const int32_t* __restrict__ index, int n,
                                                           float* __restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int id = index[p];
  float x = 0.f, y = 0.f, z = 0.f;
  if (id >= 0) { x = rows[3 * (size_t)id]; y = rows[3 * (size_t)id + 1]; z = rows[3 * (size_t)id + 2]; }
  out[p] = x; out[(size_t)n + p] = y; out[2 * (size_t)n + p] = z;
}
// its backward: grad_rows[index[p]] += g[:, p]
__global__ void __launch_bounds__(256) scatter_rows3_kernel(const float* __restrict__ g, const int32_t* __restrict__ index, int n,
                                                            float* __restrict__ grad_rows) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int id = index[p];
  if (id < 0) return;
  const float x = g[p], y = g[(size_t)n + p], z = g[2 * (size_t)n + p];
  if (x != 0.f) atomicAdd(grad_rows + 3 * (size_t)id, x);
  if (y != 0.f) atomicAdd(grad_rows + 3 * (size_t)id + 1, y);
  if (z != 0.f) atomicAdd(grad_rows + 3 * (size_t)id + 2, z);
}
}  // namespace

extern "C" {

int rtgs_tile_sum(const void* src, int32_t src_kind, int32_t H, int32_t W, int32_t stride, float* tile_sum, void* stream) {
  if (!src || !tile_sum || H <= 0 || W <= 0 || stride <= 0 || (src_kind != 0 && src_kind != 1)) return -1;
  const int gx = (W + stride - 1) / stride, gy = (H + stride - 1) / stride;
  hipStream_t st = (hipStream_t)stream;
  if (src_kind == 0)
    hipLaunchKernelGGL(tile_sum_kernel<0>, dim3(gx * gy), dim3(256), 0, st, src, H, W, stride, gx, tile_sum, (uint8_t*)nullptr);
  else
    hipLaunchKernelGGL(tile_sum_kernel<1>, dim3(gx * gy), dim3(256), 0, st, src, H, W, stride, gx, tile_sum, (uint8_t*)nullptr);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_transmission2tilemask(const uint8_t* pixelmask, int32_t H, int32_t W, int32_t stride, float ratio,
                               int32_t* tile_mask, float* tile_sum_scratch, void* stream) {
  if (!tile_mask || !tile_sum_scratch) return -1;
  int rc = rtgs_tile_sum(pixelmask, 0, H, W, stride, tile_sum_scratch, stream);
  if (rc) return rc;
  const int nt = ((W + stride - 1) / stride) * ((H + stride - 1) / stride);
  hipLaunchKernelGGL(tile_threshold_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)tile_sum_scratch, nt,
                     (float)(stride * stride), ratio, 0, tile_mask, (uint32_t*)nullptr);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_pixelmask2tilemask(const uint8_t* pixelmask, int32_t H, int32_t W, int32_t stride, int32_t* tile_mask,
                            float* tile_sum_scratch, void* stream) {
  if (!tile_mask || !tile_sum_scratch) return -1;
  int rc = rtgs_tile_sum(pixelmask, 0, H, W, stride, tile_sum_scratch, stream);
  if (rc) return rc;
  const int nt = ((W + stride - 1) / stride) * ((H + stride - 1) / stride);
  hipLaunchKernelGGL(tile_threshold_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)tile_sum_scratch, nt,
                     (float)(stride * stride), 0.f, 1, tile_mask, (uint32_t*)nullptr);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_colorerror2tilemask_k(const float* color_error, int32_t H, int32_t W, int32_t stride, int32_t top_k,
                               int32_t* tile_mask, float* tile_sum_scratch, void* stream) {
  if (!tile_mask || !tile_sum_scratch || stride <= 0 || H <= 0 || W <= 0) return -1;
  const int nt = ((W + stride - 1) / stride) * ((H + stride - 1) / stride);
  int rc = rtgs_tile_sum(color_error, 1, H, W, stride, tile_sum_scratch, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(tile_topk_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (const float*)tile_sum_scratch, nt,
                     (float)(stride * stride), (int)top_k, tile_mask);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_colorerror2tilemask(const float* color_error, int32_t H, int32_t W, int32_t stride, float top_ratio,
                             int32_t* tile_mask, float* tile_sum_scratch, void* stream) {
  if (stride <= 0 || H <= 0 || W <= 0) return -1;
  const int nt = ((W + stride - 1) / stride) * ((H + stride - 1) / stride);
  // NOTE: top_ratio crossed the ABI as float32; a caller that holds the ratio in double (the reference computes
  // int(numel * top_ratio) in double: 0.7 -> 0.69999999f can lose one tile) should compute k itself and call _k
  return rtgs_colorerror2tilemask_k(color_error, H, W, stride, (int32_t)((double)nt * (double)top_ratio), tile_mask,
                                    tile_sum_scratch, stream);
}

int rtgs_render_range(const float* T_map, int32_t H, int32_t W, float ratio, uint8_t* render_mask, int32_t* tile_mask,
                      uint32_t* count_out, float* tile_sum_scratch, void* stream) {
  if (!T_map || !render_mask || !tile_mask || !count_out || !tile_sum_scratch || H <= 0 || W <= 0) return -1;
  const int gx = (W + 15) / 16, gy = (H + 15) / 16;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(tile_sum_kernel<2>, dim3(gx * gy), dim3(256), 0, st, (const void*)T_map, H, W, 16, gx, tile_sum_scratch,
                     render_mask);
  hipLaunchKernelGGL(tile_threshold_kernel, dim3(1), dim3(1024), 0, st, (const float*)tile_sum_scratch, gx * gy, 256.f, ratio,
                     0, tile_mask, count_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

size_t rtgs_knn3_scratch_bytes(int32_t N) { return knn_layout(N).total; }

// bounding box -> Morton codes -> sorted order -> points gathered in that order + one AABB per run of KNN_BOX points
static int knn_build(const float* points, int32_t N, void* scratch, hipStream_t st, int box = KNN_BOX) {
  const KnnLayout L = knn_layout(N);
  char* s = (char*)scratch;
  uint32_t* bbox = (uint32_t*)(s + L.bbox);
  uint32_t* codes = (uint32_t*)(s + L.codes);
  uint32_t* codes_sorted = (uint32_t*)(s + L.codes_sorted);
  uint32_t* order_in = (uint32_t*)(s + L.order_in);
  uint32_t* order = (uint32_t*)(s + L.order);
  float4* sorted = (float4*)(s + L.sorted);
  float* boxes = (float*)(s + L.boxes);
  SLAM_TRY(hipMemsetAsync(bbox, 0xff, 3 * sizeof(uint32_t), st));
  SLAM_TRY(hipMemsetAsync(bbox + 3, 0, 3 * sizeof(uint32_t), st));
  int g = grid1(N);
  hipLaunchKernelGGL(knn_bbox_kernel, dim3(g > 256 ? 256 : g), dim3(256), 0, st, points, N, bbox);
  hipLaunchKernelGGL(knn_morton_kernel, dim3(g), dim3(256), 0, st, points, N, (const uint32_t*)bbox, codes, order_in);
  size_t tb = L.cub_bytes;
  SLAM_TRY(rocprim::radix_sort_pairs(s + L.cub, tb, codes, codes_sorted, order_in, order, (size_t)N, 0u, 30u, st));
  const int nboxes = (N + box - 1) / box;
  hipLaunchKernelGGL(knn_gather_kernel, dim3(nboxes), dim3(256), 0, st, points, N, (const uint32_t*)order, sorted, boxes, box);
  return 0;
}

int rtgs_knn3(const float* points, int32_t N, float* mean_dist2, int32_t* idx, float* dist2_out, void* scratch, void* stream) {
  if (N < 0 || (N > 0 && (!points || !mean_dist2 || !idx || !scratch))) return -1;
  if (N == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const KnnLayout L = knn_layout(N);
  char* s = (char*)scratch;
  const int rc = knn_build(points, N, scratch, st);
  if (rc != 0) return rc;
  const int nboxes = (N + KNN_BOX - 1) / KNN_BOX;
  hipLaunchKernelGGL(knn_search_kernel, dim3(grid1(N)), dim3(256), 0, st, (const float4*)(s + L.sorted), N,
                     (const uint32_t*)(s + L.order), (const float*)(s + L.boxes), nboxes, mean_dist2, idx, dist2_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

struct KnnQueryLayout { size_t codes, codes_sorted, order_in, order, cub, total, cub_bytes; };
static KnnQueryLayout knn_query_layout(int Nr, int Nq) {
  KnnQueryLayout Q{};
  auto al = [](size_t x) { return (x + 255) / 256 * 256; };
  const size_t n = (size_t)(Nq > 0 ? Nq : 1);
  size_t off = knn_layout(Nr).total;
  Q.codes = off; off = al(off + n * 4);
  Q.codes_sorted = off; off = al(off + n * 4);
  Q.order_in = off; off = al(off + n * 4);
  Q.order = off; off = al(off + n * 4);
  size_t tb = 0;
  (void)rocprim::radix_sort_pairs(nullptr, tb, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr,
                                  (uint32_t*)nullptr, n, 0u, 30u);
  Q.cub_bytes = tb;
  Q.cub = off; off = al(off + tb);
  Q.total = off;
  return Q;
}
size_t rtgs_knn3_query_scratch_bytes(int32_t Nr, int32_t Nq) { return knn_query_layout(Nr, Nq).total; }

// the query half of rtgs_knn3_query against a structure knn_build left in `built` (query-side scratch at `qs`)
static int knn_query_built(const void* built, int32_t Nr, const float* query_points, int32_t Nq, int32_t self_offset,
                           const float* ref_box6, int32_t* idx, float* dist2_out, void* qs, hipStream_t st) {
  const KnnLayout L = knn_layout(Nr);
  const KnnQueryLayout Q0 = knn_query_layout(Nr, Nq);
  const size_t base = knn_layout(Nr).total;
  KnnQueryLayout Q = Q0;                                     // the same layout, relative to qs
  Q.codes -= base; Q.codes_sorted -= base; Q.order_in -= base; Q.order -= base; Q.cub -= base;
  const char* s = (const char*)built;
  char* q = (char*)qs;
  uint32_t* q_codes = (uint32_t*)(q + Q.codes);
  uint32_t* q_codes_sorted = (uint32_t*)(q + Q.codes_sorted);
  uint32_t* q_order_in = (uint32_t*)(q + Q.order_in);
  uint32_t* q_order = (uint32_t*)(q + Q.order);
  hipLaunchKernelGGL(knn_query_codes_kernel, dim3(grid1(Nq)), dim3(256), 0, st, query_points, Nq, (const uint32_t*)(s + L.bbox),
                     q_codes, q_order_in);
  size_t tb = Q.cub_bytes;
  SLAM_TRY(rocprim::radix_sort_pairs(q + Q.cub, tb, q_codes, q_codes_sorted, q_order_in, q_order, (size_t)Nq, 0u, 30u, st));
  const int nboxes = (Nr + KNN_QBOX - 1) / KNN_QBOX;
  // a query is 16 lanes in a small call (<= 4 096 queries: 4 queries per wave, >= 4x the waves), 4 lanes in a large one
  if (Nq <= 1024)      // a few hundred queries spread over the scene (a frame's new points): ONE query per wave - the wave's bounding box is the query
    hipLaunchKernelGGL(knn_query_kernel<64>, dim3((Nq + 3) / 4), dim3(256), 0, st, (const float4*)(s + L.sorted), Nr,
                       (const uint32_t*)(s + L.codes_sorted), (const float*)(s + L.boxes), nboxes, query_points, Nq,
                       (const uint32_t*)q_order, (const uint32_t*)q_codes_sorted, self_offset, ref_box6, idx, dist2_out);
  else if (Nq <= 4096)
    hipLaunchKernelGGL(knn_query_kernel<16>, dim3((Nq + 15) / 16), dim3(256), 0, st, (const float4*)(s + L.sorted), Nr,
                       (const uint32_t*)(s + L.codes_sorted), (const float*)(s + L.boxes), nboxes, query_points, Nq,
                       (const uint32_t*)q_order, (const uint32_t*)q_codes_sorted, self_offset, ref_box6, idx, dist2_out);
  else
    hipLaunchKernelGGL(knn_query_kernel<4>, dim3((Nq + 63) / 64), dim3(256), 0, st, (const float4*)(s + L.sorted), Nr,
                       (const uint32_t*)(s + L.codes_sorted), (const float*)(s + L.boxes), nboxes, query_points, Nq,
                       (const uint32_t*)q_order, (const uint32_t*)q_codes_sorted, self_offset, ref_box6, idx, dist2_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

size_t rtgs_knn3_built_bytes(int32_t Nr) { return knn_layout(Nr).total; }
size_t rtgs_knn3_query_built_scratch_bytes(int32_t Nq) { return knn_query_layout(1, Nq).total - knn_layout(1).total; }

int rtgs_knn3_build_ref(const float* ref_points, int32_t Nr, void* built, void* stream) {
  if (Nr <= 0 || !ref_points || !built) return -1;
  return knn_build(ref_points, Nr, built, (hipStream_t)stream, KNN_QBOX);
}

int rtgs_knn3_query_built(const void* built, int32_t Nr, const float* query_points, int32_t Nq, const float* ref_box6, int32_t* idx,
                          float* dist2_out, void* query_scratch, void* stream) {
  if (Nr <= 0 || Nq < 0 || !built) return -1;
  if (Nq == 0) return 0;
  if (!query_points || !idx || !query_scratch) return -1;
  return knn_query_built(built, Nr, query_points, Nq, -1, ref_box6, idx, dist2_out, query_scratch, (hipStream_t)stream);
}

int rtgs_knn3_dynamic_merge(const float* query_points, int32_t Nq, const float* unstable_points, int32_t Nu, int32_t n_stable,
                            const float* dist2_stable, const int32_t* idx_stable, const float* ref_box6, int32_t* idx, float* dist2_out,
                            void* stream) {
  if (Nq < 0 || Nu < 0 || n_stable < 0) return -1;
  if (Nq == 0) return 0;
  if (!query_points || !dist2_stable || !idx_stable || !idx || (Nu > 0 && !unstable_points)) return -1;
  hipLaunchKernelGGL(knn_dynamic_merge_kernel, dim3((Nq + 3) / 4), dim3(256), 0, (hipStream_t)stream, query_points, (int)Nq,
                     unstable_points, (int)Nu, (int)n_stable, dist2_stable, idx_stable, ref_box6, idx, dist2_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_knn3_query(const float* ref_points, int32_t Nr, const float* query_points, int32_t Nq, int32_t self_offset,
                    const float* ref_box6, int32_t* idx, float* dist2_out, void* scratch, void* stream) {
  if (Nr < 0 || Nq < 0 || (Nq > 0 && (!query_points || !idx))) return -1;
  if (Nr > 0 && (!ref_points || !scratch)) return -1;
  if (self_offset >= 0 && (int64_t)self_offset + Nq > Nr) return -1;
  if (Nq == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  if (Nr == 0) {                                              // nobody to find: -1 / FLT_MAX, as rtgs_knn3 with < 4 points
    SLAM_TRY(hipMemsetAsync(idx, 0xff, (size_t)Nq * 3 * sizeof(int32_t), st));
    if (dist2_out) hipLaunchKernelGGL(fill_f32_kernel, dim3(grid1(Nq * 3)), dim3(256), 0, st, dist2_out, Nq * 3, FLT_MAX);
    SLAM_TRY(hipGetLastError());
    return 0;
  }
  const int rc = knn_build(ref_points, Nr, scratch, st, KNN_QBOX);
  if (rc != 0) return rc;
  return knn_query_built(scratch, Nr, query_points, Nq, self_offset, ref_box6, idx, dist2_out,
                         (char*)scratch + knn_layout(Nr).total, st);
}

int rtgs_accumulate_error(int32_t H, int32_t W, int32_t P, const float* color_err, const float* depth_err,
                          const float* normal_err, const int32_t* color_index, const int32_t* depth_index, float thr_c,
                          float thr_d, float thr_n, int32_t mean, float* g_color, float* g_depth, float* g_normal,
                          int32_t* outlier_count, float* scratch, void* stream) {
  if (H <= 0 || W <= 0 || P < 0) return -1;
  if (P == 0) return 0;
  if (!color_err || !depth_err || !normal_err || !color_index || !depth_index || !g_color || !g_depth || !g_normal ||
      !outlier_count || !scratch)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  if (g_depth == g_color + P && g_normal == g_depth + P && (float*)outlier_count == g_normal + P && scratch == g_normal + 2 * (size_t)P) {
    SLAM_TRY(hipMemsetAsync(g_color, 0, (size_t)P * 24, st));        // one allocation behind the five arrays: one clear
  } else {
    SLAM_TRY(hipMemsetAsync(g_color, 0, (size_t)P * 4, st));
    SLAM_TRY(hipMemsetAsync(g_depth, 0, (size_t)P * 4, st));
    SLAM_TRY(hipMemsetAsync(g_normal, 0, (size_t)P * 4, st));
    SLAM_TRY(hipMemsetAsync(outlier_count, 0, (size_t)P * 4, st));
    SLAM_TRY(hipMemsetAsync(scratch, 0, (size_t)P * 8, st));
  }
  const int n = H * W;
  hipLaunchKernelGGL(accumulate_error_kernel, dim3(grid1(n)), dim3(256), 0, st, n, color_err, depth_err, normal_err,
                     color_index, depth_index, thr_c, thr_d, thr_n, g_color, g_depth, g_normal, outlier_count, scratch,
                     scratch + P, (int)P);
  if (mean)
    hipLaunchKernelGGL(error_mean_kernel, dim3(grid1(P)), dim3(256), 0, st, P, g_color, g_depth, g_normal,
                       (const float*)scratch, (const float*)(scratch + P));
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_bilateral_filter(const float* depth, int32_t H, int32_t W, int32_t radius, float sigma_color, float sigma_space,
                          float* out, void* stream) {
  if (!depth || !out || H <= 0 || W <= 0 || radius < 0 || radius > BF_MAXR || !(sigma_color > 0.f) || !(sigma_space > 0.f))
    return -1;
  const float inv2ss = (float)(1.0 / (2.0 * (double)sigma_space * (double)sigma_space));
  const float two_sc2 = (float)(2.0 * (double)sigma_color * (double)sigma_color);
  hipLaunchKernelGGL(bilateral_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, (hipStream_t)stream, depth, H, W,
                     radius, inv2ss, two_sc2, out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

size_t rtgs_frame_preprocess_scratch_bytes(int32_t H, int32_t W) { return (size_t)H * W * 3 * sizeof(float) + 256; }

int rtgs_frame_preprocess(const float* depth_in, int32_t H, int32_t W, const float* K, float min_depth, float max_depth,
                          float invalid_confidence_thresh, float* depth_out, float* vertex_out, float* normal_out,
                          float* conf_out, uint8_t* bad_out, void* scratch, void* stream) {
  if (!depth_in || !K || !depth_out || !vertex_out || !normal_out || !conf_out || !bad_out || !scratch || H <= 0 || W <= 0)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  uint32_t* mm = (uint32_t*)scratch;
  float* V = (float*)((char*)scratch + 256);
  SLAM_TRY(hipMemsetAsync(mm, 0xff, sizeof(uint32_t), st));
  SLAM_TRY(hipMemsetAsync(mm + 1, 0, sizeof(uint32_t), st));
  const int n = H * W;
  hipLaunchKernelGGL(frame_vertex_kernel, dim3(grid1(n) > 512 ? 512 : grid1(n)), dim3(256), 0, st, depth_in, H, W, K, min_depth, max_depth, V, mm);
  hipLaunchKernelGGL(frame_normal_kernel, dim3(grid1(n)), dim3(256), 0, st, (const float*)V, H, W, K, (const uint32_t*)mm,
                     invalid_confidence_thresh, depth_out, vertex_out, normal_out, conf_out, bad_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

size_t rtgs_compact_scratch_bytes(int32_t n) {
  size_t tb = 0;
  (void)rocprim::select(nullptr, tb, rocprim::counting_iterator<int32_t>(0), (const uint8_t*)nullptr, (int32_t*)nullptr,
                        (int32_t*)nullptr, (size_t)(n > 0 ? n : 1));
  return tb + 256;
}

int rtgs_sample_candidates(const float* normal_map, const uint8_t* select_mask, int32_t H, int32_t W, int32_t* indices_out,
                           int32_t* count_out, uint8_t* flags_scratch, void* scratch, void* stream) {
  if (!normal_map || !indices_out || !count_out || !flags_scratch || !scratch || H <= 0 || W <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int n = H * W;
  hipLaunchKernelGGL(sample_flags_kernel, dim3(grid1(n)), dim3(256), 0, st, normal_map, select_mask, n, flags_scratch);
  size_t tb = rtgs_compact_scratch_bytes(n) - 256;
  SLAM_TRY(rocprim::select(scratch, tb, rocprim::counting_iterator<int32_t>(0), (const uint8_t*)flags_scratch, indices_out,
                           count_out, (size_t)n, st));
  SLAM_TRY(hipGetLastError());
  return 0;
}


// ---- per-frame mask / error producers of the mapper (one streaming pass each instead of ~25 torch launches) --------------
// Mapping.temp_points_init, mapper.py:728-775: where does the map not explain the frame?
//   transmission mask = T > thr_T & depth > 0                                  (:729-731)
//   error mask        = ((|depth - render depth| > thr_d & depth > 0 & depth index > -1)        (:757-761)
//                        | (mean_c |colour - render colour| > thr_c & depth > 0 & T < thr_T))   (:762-766)
//                       & ~transmission mask                                                     (:767-768)
// counts[0..1] += set pixels of the two masks (the caller sizes its two sample_pixels draws from them).
__global__ void __launch_bounds__(256) add_masks_kernel(const float* __restrict__ T, const float* __restrict__ depth,
                                                        const float* __restrict__ rdepth, const float* __restrict__ rcolor,
                                                        const float* __restrict__ fcolor, const int32_t* __restrict__ didx, int hw,
                                                        float thr_T, float thr_d, float thr_c, uint8_t* __restrict__ tmask,
                                                        uint8_t* __restrict__ emask, uint32_t* __restrict__ counts) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  bool tm = false, em = false;
  if (i < hw) {
    const float d = depth[i], t = T[i];
    const bool ok = d > 0.f;
    tm = t > thr_T && ok;
    const float de = fabsf(d - rdepth[i]);
    // torch: abs(frame - render).mean(dim=-1) over the three channels = ((a + b) + c) / 3
    const float ce = ((fabsf(fcolor[i] - rcolor[i]) + fabsf(fcolor[hw + i] - rcolor[hw + i])) + fabsf(fcolor[2 * hw + i] - rcolor[2 * hw + i])) / 3.f;
    const bool dm = de > thr_d && ok && didx[i] > -1;
    const bool cm = ce > thr_c && ok && t < thr_T;
    em = (cm || dm) && !tm;
    tmask[i] = tm ? 1 : 0;
    emask[i] = em ? 1 : 0;
  }
  const unsigned long long bt = __builtin_amdgcn_ballot_w64(tm), be = __builtin_amdgcn_ballot_w64(em);
  if ((threadIdx.x & 63) == 0) {
    if (bt) atomicAdd(&counts[0], (uint32_t)__popcll(bt));
    if (be) atomicAdd(&counts[1], (uint32_t)__popcll(be));
  }
}
int rtgs_add_masks(const float* T_map, const float* depth, const float* render_depth, const float* render_color_chw,
                   const float* frame_color_chw, const int32_t* depth_index, int32_t H, int32_t W, float thr_transmission,
                   float thr_depth, float thr_color, uint8_t* transmission_mask, uint8_t* error_mask, uint32_t* counts2,
                   void* stream) {
  if (H <= 0 || W <= 0 || !T_map || !depth || !render_depth || !render_color_chw || !frame_color_chw || !depth_index ||
      !transmission_mask || !error_mask || !counts2)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  SLAM_TRY(hipMemsetAsync(counts2, 0, 2 * sizeof(uint32_t), st));
  hipLaunchKernelGGL(add_masks_kernel, dim3(grid1(H * W)), dim3(256), 0, st, T_map, depth, render_depth, render_color_chw,
                     frame_color_chw, depth_index, H * W, thr_transmission, thr_depth, thr_color, transmission_mask, error_mask, counts2);
  SLAM_TRY(hipGetLastError());
  return 0;
}

// Mapping.error_gaussians_remove, mapper.py:527-540: the error maps that are back-projected onto the Gaussians.
//   depth error  = |depth - render depth|, 0 where the render is BEHIND the frame (depth - render < 0), where the frame has
//                  no depth or the pixel has no depth owner
//   colour error = sum_c |colour - render colour|, 0 where the frame has no depth
__global__ void __launch_bounds__(256) frame_errors_kernel(const float* __restrict__ depth, const float* __restrict__ rdepth,
                                                           const float* __restrict__ rcolor, const float* __restrict__ fcolor,
                                                           const int32_t* __restrict__ didx, int hw, float* __restrict__ color_err,
                                                           float* __restrict__ depth_err) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hw) return;
  const float d = depth[i], diff = d - rdepth[i];
  const bool invalid = d == 0.f || didx[i] == -1;
  depth_err[i] = (diff < 0.f || invalid) ? 0.f : fabsf(diff);
  const float ce = (fabsf(fcolor[i] - rcolor[i]) + fabsf(fcolor[hw + i] - rcolor[hw + i])) + fabsf(fcolor[2 * hw + i] - rcolor[2 * hw + i]);
  color_err[i] = d == 0.f ? 0.f : ce;
}
int rtgs_frame_errors(const float* depth, const float* render_depth, const float* render_color_chw, const float* frame_color_chw,
                      const int32_t* depth_index, int32_t H, int32_t W, float* color_error, float* depth_error, void* stream) {
  if (H <= 0 || W <= 0 || !depth || !render_depth || !render_color_chw || !frame_color_chw || !depth_index || !color_error || !depth_error)
    return -1;
  hipLaunchKernelGGL(frame_errors_kernel, dim3(grid1(H * W)), dim3(256), 0, (hipStream_t)stream, depth, render_depth,
                     render_color_chw, frame_color_chw, depth_index, H * W, color_error, depth_error);
  SLAM_TRY(hipGetLastError());
  return 0;
}

// Mapping.temp_points_attach, mapper.py:830-883: a new point whose projection lands on a pixel owned (colour index) by a stable
// Gaussian, and which lies within `max_plane_dist` of that Gaussian's plane, is attached: attach[i] = 1.
//   uv = (K (R p + t))[:2] / z, truncated toward zero like `.long()` (scene/cameras.py:161-168); inside the image;
//   owner = stable colour index at (v, u) >= 0;  |(x_owner - p) . n_owner| < max_plane_dist.
__global__ void __launch_bounds__(256) attach_test_kernel(const float* __restrict__ pts, int n, const float* __restrict__ w2c,
                                                          float fx, float fy, float cx, float cy, int H, int W,
                                                          const int32_t* __restrict__ cidx, const float* __restrict__ sxyz,
                                                          const float* __restrict__ snrm, float max_dist, uint8_t* __restrict__ attach) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
  float c[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) c[r] = ((x * w2c[4 * r] + y * w2c[4 * r + 1]) + z * w2c[4 * r + 2]) + w2c[4 * r + 3];
  // uv = xyz_c @ K.T: u = fx x + 0 y + cx z (torch sums the three products in order), then / z, then truncation
  const float un = (c[0] * fx + c[1] * 0.f) + c[2] * cx, vn = (c[0] * 0.f + c[1] * fy) + c[2] * cy;
  const float uf = un / c[2], vf = vn / c[2];
  uint8_t a = 0;
  if (uf == uf && vf == vf && fabsf(uf) < 1e9f && fabsf(vf) < 1e9f) {
    const long long u = (long long)uf, v = (long long)vf;
    if (u >= 0 && u < W && v >= 0 && v < H) {
      const int o = cidx[(size_t)v * W + u];
      if (o >= 0) {
        const float d = ((sxyz[3 * (size_t)o] - x) * snrm[3 * (size_t)o] + (sxyz[3 * (size_t)o + 1] - y) * snrm[3 * (size_t)o + 1]) +
                        (sxyz[3 * (size_t)o + 2] - z) * snrm[3 * (size_t)o + 2];
        a = fabsf(d) < max_dist ? 1 : 0;
      }
    }
  }
  attach[i] = a;
}
int rtgs_error_counters(int32_t nf, const float* g_color, const float* g_depth, float color_strike_thr, float depth_strike_thr,
                        int32_t* depth_counter, int32_t* color_counter, int32_t limit, uint8_t* delete_mask, uint8_t* release_mask,
                        uint32_t* counts2, void* stream) {
  if (nf < 0 || !counts2) return -1;
  hipStream_t st = (hipStream_t)stream;
  SLAM_TRY(hipMemsetAsync(counts2, 0, 2 * sizeof(uint32_t), st));
  if (nf == 0) return 0;
  if (!g_color || !g_depth || !depth_counter || !color_counter || !delete_mask || !release_mask) return -1;
  hipLaunchKernelGGL(error_counters_kernel, dim3(grid1(nf)), dim3(256), 0, st, (int)nf, g_color, g_depth, color_strike_thr,
                     depth_strike_thr, depth_counter, color_counter, (int)limit, delete_mask, release_mask, counts2);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_delete_mask(int32_t n, const float* scales, const int32_t* add_tick, int32_t time_now, int32_t window, uint8_t* mask,
                     uint32_t* count1, void* stream) {
  if (n <= 0 || !scales || !mask || !count1) return -1;
  hipLaunchKernelGGL(delete_mask_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)n, scales, add_tick, (int)time_now,
                     (int)window, mask, count1);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_gather_new_points(const int64_t* pick, int32_t n, const float* vertex_map, const float* normal_map, const float* color_map,
                           int32_t identity_rot, float* xyz, float* normal, float* color, float* rots, void* stream) {
  if (n < 0) return -1;
  if (n == 0) return 0;
  if (!pick || !vertex_map || !normal_map || !color_map || !xyz || !normal || !color || !rots) return -1;
  hipLaunchKernelGGL(gather_new_points_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, pick, (int)n, vertex_map, normal_map,
                     color_map, (int)identity_rot, xyz, normal, color, rots);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_draw_new_points(const int32_t* cand, int32_t n_cand, int32_t k, uint64_t key, const float* vertex_map, const float* normal_map,
                         const float* color_map, int32_t identity_rot, float* xyz, float* normal, float* color, float* rots,
                         int32_t* pick_out, void* stream) {
  if (n_cand < 0 || k < 0 || k > n_cand) return -1;
  if (k == 0) return 0;
  if (!cand || !vertex_map || !normal_map || !color_map || !xyz || !normal || !color || !rots) return -1;
  uint32_t bits = 2;
  while (bits < 32 && (1ull << bits) < (uint64_t)n_cand) ++bits;
  const uint32_t half_bits = (bits + 1) / 2;
  hipLaunchKernelGGL(draw_new_points_kernel, dim3(grid1(k)), dim3(256), 0, (hipStream_t)stream, cand, (uint32_t)n_cand, (int)k, half_bits,
                     key, vertex_map, normal_map, color_map, (int)identity_rot, xyz, normal, color, rots, pick_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_filter_keep(int32_t n, const float* dist2, const int32_t* idx, const float* scales, float ratio, uint8_t* keep, void* stream) {
  if (n < 0) return -1;
  if (n == 0) return 0;
  if (!dist2 || !idx || !scales || !keep) return -1;
  hipLaunchKernelGGL(filter_keep_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, (int)n, dist2, idx, scales, ratio, keep);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_bbox_pad(int32_t n, const float* xyz, float pad, float* out6, void* stream) {
  if (n <= 0 || !xyz || !out6) return -1;
  hipLaunchKernelGGL(bbox_pad_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)n, xyz, pad, out6);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_compact_points(int32_t n, const uint8_t* keep, const float* xyz, const float* color, const float* opacity_raw,
                        const float* rots, float* out_xyz, float* out_color, float* out_opacity_raw, float* out_rots,
                        int32_t* count_out, void* stream) {
  if (n < 0 || !count_out) return -1;
  if (n > 0 && (!keep || !xyz || !color || !opacity_raw || !rots || !out_xyz || !out_color || !out_opacity_raw || !out_rots)) return -1;
  hipLaunchKernelGGL(compact_points_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)n, keep, xyz, color, opacity_raw, rots,
                     out_xyz, out_color, out_opacity_raw, out_rots, count_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_append_valid_rows(int32_t n, const uint8_t* valid, const float* rows59, float* xyz_dst, float* shs_dst, float* raw8_dst,
                           int32_t n_aux, void* const* aux_dst, const uint32_t* aux_fill_bits, int32_t* count_out, void* stream) {
  if (n < 0 || n_aux < 0 || n_aux > 8 || !count_out) return -1;
  if (n > 0 && (!valid || !rows59 || !xyz_dst || !shs_dst || !raw8_dst)) return -1;
  if (n_aux > 0 && (!aux_dst || !aux_fill_bits)) return -1;
  AuxFill aux;
  aux.n = n_aux;
  for (int k = 0; k < 8; ++k) {
    aux.dst[k] = k < n_aux ? (uint32_t*)aux_dst[k] : nullptr;
    aux.bits[k] = k < n_aux ? aux_fill_bits[k] : 0u;
    if (k < n_aux && !aux.dst[k]) return -1;
  }
  hipLaunchKernelGGL(append_valid_rows_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (int)n, valid, rows59, xyz_dst, shs_dst,
                     raw8_dst, aux, count_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_new_rows(int32_t n, const float* xyz, const float* color, const float* opacity_raw, const float* rots, const float* dist2,
                  const int32_t* idx, const float* exist_scales, float min_radius, float max_radius, float scale_factor,
                  float factor_x, float factor_y, float factor_z, float* packed59, uint8_t* valid, void* stream) {
  if (n < 0) return -1;
  if (n == 0) return 0;
  if (!xyz || !color || !opacity_raw || !rots || !dist2 || !idx || !packed59 || !valid) return -1;
  hipLaunchKernelGGL(new_rows_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, (int)n, xyz, color, opacity_raw, rots, dist2,
                     idx, exist_scales, min_radius, max_radius, scale_factor, factor_x, factor_y, factor_z, packed59, valid);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_attach_test(const float* points, int32_t n, const float* w2c16, float fx, float fy, float cx, float cy, int32_t H,
                     int32_t W, const int32_t* stable_color_index, const float* stable_xyz, const float* stable_normal,
                     float max_plane_dist, uint8_t* attach_out, void* stream) {
  if (n < 0 || H <= 0 || W <= 0) return -1;
  if (n == 0) return 0;
  if (!points || !w2c16 || !stable_color_index || !stable_xyz || !stable_normal || !attach_out) return -1;
  hipLaunchKernelGGL(attach_test_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, points, n, w2c16, fx, fy, cx, cy, H,
                     W, stable_color_index, stable_xyz, stable_normal, max_plane_dist, attach_out);
  SLAM_TRY(hipGetLastError());
  return 0;
}

// transform_map (SLAM/utils.py:56-63): every 3-vector of a map through a 4x4 transform (homogeneous 1 appended; for normals
// the caller passes get_rot(c2w), whose translation column is zero).  The reference does it as a batched 4x4 matmul per
// pixel; `map @ R.T + t` in torch becomes a Tensile GEMM with K = 3 - 106 us per 1200x680 map.  12 B in, 12 B out.
__global__ void __launch_bounds__(256) transform_map_kernel(const float* __restrict__ in, int64_t n, const float* __restrict__ T,
                                                            float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float x = in[3 * i], y = in[3 * i + 1], z = in[3 * i + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) out[3 * i + r] = ((T[4 * r] * x + T[4 * r + 1] * y) + T[4 * r + 2] * z) + T[4 * r + 3];
}
int rtgs_transform_map(const float* map3, int64_t n, const float* transform16, float* out3, void* stream) {
  if (n < 0 || (n > 0 && (!map3 || !transform16 || !out3))) return -1;
  if (n == 0) return 0;
  hipLaunchKernelGGL(transform_map_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, map3, n,
                     transform16, out3);
  SLAM_TRY(hipGetLastError());
  return 0;
}

int rtgs_gather_rows3(const float* rows, const int32_t* index, int32_t n, float* out, void* stream) {
  if (!index || !out || n < 0) return -1;
  if (n == 0) return 0;
  if (!rows) return -1;
  hipLaunchKernelGGL(gather_rows3_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, rows, index, n, out);
  SLAM_TRY(hipGetLastError());
  return 0;
}
int rtgs_scatter_rows3(const float* g, const int32_t* index, int32_t n, float* grad_rows, void* stream) {
  if (!g || !index || n < 0) return -1;
  if (n == 0) return 0;
  if (!grad_rows) return -1;
  hipLaunchKernelGGL(scatter_rows3_kernel, dim3(grid1(n)), dim3(256), 0, (hipStream_t)stream, g, index, n, grad_rows);
  SLAM_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
