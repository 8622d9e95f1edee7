// gfx950 kernels + C ABI of the projective point-to-plane ICP tracker (include/rtgs_icp.h).
// Restates /root/reference/SLAM/icp.py and the helpers it takes from SLAM/utils.py:65-122,
// 511-527, fused: one kernel per pyramid stage, one residual/Jacobian/6x6-reduction kernel and
// one single-workgroup solve+update kernel per Gauss-Newton iteration, pose resident on the
// device.  Compiled with -ffp-contract=off so the float32 op sequence of the gating
// arithmetic (projection, nearest-neighbour association, thresholds) follows the reference's.
#include "../../include/rtgs_icp.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>

namespace rtgs_icp {

// Residual workgroups per launch, at most.  One per CU measured best at 1200x680, alone and beside the mapper (track alone /
// unit, us: 64: 457 / 556, 128: 317 / 445, 192: 281 / 437, 256: 263 / 420, 384: 270 / 427, 512: 283 / 431): fewer tickets and
// partial rows for the last arriver, and - the kernel needs 151 VGPRs - fewer wave slots to find beside the mapper's.
#ifndef RTGS_ICP_MAX_BLOCKS
#define RTGS_ICP_MAX_BLOCKS 256
#endif
constexpr int MAX_BLOCKS = RTGS_ICP_MAX_BLOCKS;
constexpr int NACC = 28;            // 21 upper-triangular JtJ + 6 Jtr + 1 valid count
constexpr int PSTRIDE = 32;         // floats per block partial

struct Scratch {
  uint32_t minmax[2 * RTGS_ICP_MAX_LEVELS];   // per level: enc(min), ~enc(max)
  float partials[MAX_BLOCKS * PSTRIDE];
  uint32_t ticket;                            // arrival counter of the residual kernel's workgroups
  uint32_t pad[15];                           // pad[0]: abort flag of the persistent track kernel
  double sums[24 * 8 * PSTRIDE];              // persistent track: 8 rows of float64 totals per grid barrier
  unsigned long long dbg[96];                 // RTGS_ICP_DEBUG_TIMING=1: wall-clock stamps of workgroup 0 (5 per barrier)
  uint32_t minmax_b[2 * RTGS_ICP_MAX_LEVELS]; // second set: a build re-arms the set the NEXT build uses (no memset launch)
};

// The tracker's kernels are a chain the frame waits for; beside the mapper (VALU-issue-bound tile walks on every SIMD)
// their waves would take turns with five or six others.  Raising the wave's issue priority (s_setprio, an SQ arbitration
// hint) lets the chain through; the mapper has the slack.  RTGS_ICP_WAVE_PRIO=0 at build time turns it off.
#ifndef RTGS_ICP_WAVE_PRIO
#define RTGS_ICP_WAVE_PRIO 3
#endif
__device__ __forceinline__ void wave_priority_high() {
#if RTGS_ICP_WAVE_PRIO > 0
  __builtin_amdgcn_s_setprio(RTGS_ICP_WAVE_PRIO);
#endif
}

__device__ __forceinline__ uint32_t enc_f(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f(uint32_t e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7fffffffu) : ~e);
}

struct PyrDesc {
  int levels;
  int H, W;                                   // full resolution
  int Hl[RTGS_ICP_MAX_LEVELS], Wl[RTGS_ICP_MAX_LEVELS], shift[RTGS_ICP_MAX_LEVELS];
  int block_start[RTGS_ICP_MAX_LEVELS + 1];
  float* vertex[RTGS_ICP_MAX_LEVELS];
  float* normal[RTGS_ICP_MAX_LEVELS];
  uint32_t* mm;                               // this build's min / max words
  uint32_t* mm_rearm;                         // nullable: the other set, re-armed (0xffffffff) for the next build
};

// ---- K10: max-pool depth pyramid + back-projection (SLAM/utils.py:511-521, :65-75) ------------
// Grid-stride over 256-pixel chunks (a chunk never straddles two levels).  The per-level depth
// min / max needed by compute_normal_map's invalid mask are reduced wave -> LDS -> ONE pair of
// global atomics per workgroup and level (same-address global atomics cost ~12 ns each).
__global__ void __launch_bounds__(256) icp_vertex_kernel(PyrDesc d, const float* __restrict__ depth,
                                                         const float* __restrict__ K, Scratch* sc) {
  __shared__ uint32_t s_mm[2 * RTGS_ICP_MAX_LEVELS];
  if (threadIdx.x < 2 * RTGS_ICP_MAX_LEVELS) s_mm[threadIdx.x] = 0xffffffffu;
  __syncthreads();
  const int total = d.block_start[d.levels];
  for (int chunk = blockIdx.x; chunk < total; chunk += gridDim.x) {
    int l = 0;
    while (l + 1 < d.levels && chunk >= d.block_start[l + 1]) ++l;
    const int Hl = d.Hl[l], Wl = d.Wl[l], sh = d.shift[l];
    const int idx = (chunk - d.block_start[l]) * 256 + (int)threadIdx.x;
    float dmax = 0.f;
    const bool live = idx < Hl * Wl;
    if (live) {
      const int y = idx / Wl, x = idx % Wl;
      const int f = 1 << sh;
      dmax = -INFINITY;
      for (int a = 0; a < f; ++a)
        for (int b = 0; b < f; ++b) dmax = fmaxf(dmax, depth[(size_t)(y * f + a) * d.W + (x * f + b)]);
      const float ds = 1.f / (float)f;                       // K * downscale, K[2][2] = 1
      const float fx = K[0] * ds, fy = K[4] * ds, cx = K[2] * ds, cy = K[5] * ds;
      float* v = d.vertex[l] + (size_t)idx * 3;
      v[0] = (((float)x - cx) / fx) * dmax;
      v[1] = (((float)y - cy) / fy) * dmax;
      v[2] = dmax;                                           // 1 * depth
    }
    uint32_t emin = 0xffffffffu, emaxinv = 0xffffffffu;
    if (live) { emin = enc_f(dmax); emaxinv = ~emin; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      emin = min(emin, (uint32_t)__shfl_xor((int)emin, off));
      emaxinv = min(emaxinv, (uint32_t)__shfl_xor((int)emaxinv, off));
    }
    if ((threadIdx.x & 63) == 0 && emin != 0xffffffffu) {
      atomicMin(&s_mm[2 * l], emin);
      atomicMin(&s_mm[2 * l + 1], emaxinv);
    }
  }
  __syncthreads();
  if (threadIdx.x < 2 * d.levels && s_mm[threadIdx.x] != 0xffffffffu) atomicMin(&d.mm[threadIdx.x], s_mm[threadIdx.x]);
  if (d.mm_rearm && blockIdx.x == 0 && threadIdx.x < 2 * d.levels) d.mm_rearm[threadIdx.x] = 0xffffffffu;
}

// ---- K10, three levels in ONE pass: a lane owns a 4x4 block of the full-resolution depth (four 16-B row loads when the
// width allows) and writes its 16 full-resolution vertices, its four half-resolution ones and its quarter-resolution
// one - the depth image is read once instead of three times with 16-/4-element strided loops, and the stores of a lane
// are 48-byte runs.  Same arithmetic as icp_vertex_kernel (max is exact in any order), same min / max reduction.
__device__ __forceinline__ void vertex_of(float* __restrict__ out, int x, int y, float d, float fx, float fy, float cx, float cy) {
  out[0] = (((float)x - cx) / fx) * d;
  out[1] = (((float)y - cy) / fy) * d;
  out[2] = d;
}
__global__ void __launch_bounds__(256) icp_vertex3_kernel(PyrDesc d, const float* __restrict__ depth,
                                                          const float* __restrict__ K, Scratch* sc) {
  wave_priority_high();
  __shared__ uint32_t s_mm[6];
  if (threadIdx.x < 6) s_mm[threadIdx.x] = 0xffffffffu;
  __syncthreads();
  const int H = d.H, W = d.W, bw = (W + 3) >> 2, bh = (H + 3) >> 2;
  const int H1 = d.Hl[1], W1 = d.Wl[1], H0 = d.Hl[0], W0 = d.Wl[0];
  uint32_t mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mxi[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu};
  for (int b = blockIdx.x * 256 + (int)threadIdx.x; b < bw * bh; b += gridDim.x * 256) {
    const int by = b / bw, bx = b % bw;
    const int x0 = bx * 4, y0 = by * 4;
    float t[4][4];
    const bool full = (x0 + 4 <= W) && (y0 + 4 <= H);
    if (full && (W & 3) == 0) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const float4 r = *reinterpret_cast<const float4*>(depth + (size_t)(y0 + a) * W + x0);
        t[a][0] = r.x; t[a][1] = r.y; t[a][2] = r.z; t[a][3] = r.w;
      }
    } else {
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          t[a][c] = (y0 + a < H && x0 + c < W) ? depth[(size_t)(y0 + a) * W + (x0 + c)] : -INFINITY;
    }
    // level 2 (full resolution): K * 1
    {
      const float ds = 1.f / (float)1;
      const float fx = K[0] * ds, fy = K[4] * ds, cx = K[2] * ds, cy = K[5] * ds;
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        if (y0 + a >= H) continue;
        float* row = d.vertex[2] + ((size_t)(y0 + a) * W + x0) * 3;
        float o[12];
#pragma unroll
        for (int c = 0; c < 4; ++c) vertex_of(o + 3 * c, x0 + c, y0 + a, t[a][c], fx, fy, cx, cy);
        if (full && (W & 3) == 0) {
          float4* r4 = reinterpret_cast<float4*>(row);
          r4[0] = make_float4(o[0], o[1], o[2], o[3]); r4[1] = make_float4(o[4], o[5], o[6], o[7]);
          r4[2] = make_float4(o[8], o[9], o[10], o[11]);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (x0 + c < W) { row[3 * c] = o[3 * c]; row[3 * c + 1] = o[3 * c + 1]; row[3 * c + 2] = o[3 * c + 2]; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (x0 + c < W) { const uint32_t e = enc_f(t[a][c]); mn[2] = min(mn[2], e); mxi[2] = min(mxi[2], ~e); }
      }
    }
    // level 1 (MaxPool 2x2, floor mode): K * 0.5
    float h[2][2];
    {
      const float ds = 1.f / (float)2;
      const float fx = K[0] * ds, fy = K[4] * ds, cx = K[2] * ds, cy = K[5] * ds;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float m = -INFINITY;
          m = fmaxf(m, t[2 * j][2 * i]); m = fmaxf(m, t[2 * j][2 * i + 1]);
          m = fmaxf(m, t[2 * j + 1][2 * i]); m = fmaxf(m, t[2 * j + 1][2 * i + 1]);
          h[j][i] = m;
          const int y1 = 2 * by + j, x1 = 2 * bx + i;
          if (y1 < H1 && x1 < W1) {
            vertex_of(d.vertex[1] + ((size_t)y1 * W1 + x1) * 3, x1, y1, m, fx, fy, cx, cy);
            const uint32_t e = enc_f(m); mn[1] = min(mn[1], e); mxi[1] = min(mxi[1], ~e);
          }
        }
    }
    // level 0 (MaxPool 4x4): K * 0.25
    if (by < H0 && bx < W0) {
      const float ds = 1.f / (float)4;
      const float fx = K[0] * ds, fy = K[4] * ds, cx = K[2] * ds, cy = K[5] * ds;
      const float m = fmaxf(fmaxf(h[0][0], h[0][1]), fmaxf(h[1][0], h[1][1]));
      vertex_of(d.vertex[0] + ((size_t)by * W0 + bx) * 3, bx, by, m, fx, fy, cx, cy);
      const uint32_t e = enc_f(m); mn[0] = min(mn[0], e); mxi[0] = min(mxi[0], ~e);
    }
  }
#pragma unroll
  for (int l = 0; l < 3; ++l) {
    uint32_t a = mn[l], bq = mxi[l];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      a = min(a, (uint32_t)__shfl_xor((int)a, off));
      bq = min(bq, (uint32_t)__shfl_xor((int)bq, off));
    }
    if ((threadIdx.x & 63) == 0 && a != 0xffffffffu) { atomicMin(&s_mm[2 * l], a); atomicMin(&s_mm[2 * l + 1], bq); }
  }
  __syncthreads();
  if (threadIdx.x < 6 && s_mm[threadIdx.x] != 0xffffffffu) atomicMin(&d.mm[threadIdx.x], s_mm[threadIdx.x]);
  if (d.mm_rearm && blockIdx.x == 0 && threadIdx.x < 6) d.mm_rearm[threadIdx.x] = 0xffffffffu;
}

// ---- K11: Sobel normals (SLAM/utils.py:77-122): replicate pad, cross(dy, dx), / (|n| + 1e-8),
//      zero where depth <= min or depth >= max --------------------------------------------------
__global__ void __launch_bounds__(256) icp_normal_kernel(PyrDesc d, const Scratch* sc) {
  wave_priority_high();
  int l = 0;
  while (l + 1 < d.levels && (int)blockIdx.x >= d.block_start[l + 1]) ++l;
  const int Hl = d.Hl[l], Wl = d.Wl[l];
  const int idx = ((int)blockIdx.x - d.block_start[l]) * 256 + (int)threadIdx.x;
  if (idx >= Hl * Wl) return;
  const int y = idx / Wl, x = idx % Wl;
  const float* V = d.vertex[l];
  const int ym = max(y - 1, 0), yp = min(y + 1, Hl - 1), xm = max(x - 1, 0), xp = min(x + 1, Wl - 1);
  float gx[3], gy[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float a00 = V[((size_t)ym * Wl + xm) * 3 + c], a01 = V[((size_t)ym * Wl + x) * 3 + c], a02 = V[((size_t)ym * Wl + xp) * 3 + c];
    const float a10 = V[((size_t)y * Wl + xm) * 3 + c], a12 = V[((size_t)y * Wl + xp) * 3 + c];
    const float a20 = V[((size_t)yp * Wl + xm) * 3 + c], a21 = V[((size_t)yp * Wl + x) * 3 + c], a22 = V[((size_t)yp * Wl + xp) * 3 + c];
    gx[c] = -a00 + a02 - 2.f * a10 + 2.f * a12 - a20 + a22;
    gy[c] = -a00 - 2.f * a01 - a02 + a20 + 2.f * a21 + a22;
  }
  // cross(dy, dx), rounded as torch.cross rounds it: a b - c d = fma(a, b, -(c d)); and |n| as torch.norm does:
  // sqrt(fma(z, z, fma(y, y, x x))) - both probed against torch 2.10 CPU, bit for bit (the Sobel sums above already
  // follow conv2d's row-major tap order).  With that the normal pyramid equals the reference's exactly, and with it
  // every gate decision of the tracker that tests a normal.
  float nx = __fmaf_rn(gy[1], gx[2], -(gy[2] * gx[1]));
  float ny = __fmaf_rn(gy[2], gx[0], -(gy[0] * gx[2]));
  float nz = __fmaf_rn(gy[0], gx[1], -(gy[1] * gx[0]));
  const float mag = sqrtf(__fmaf_rn(nz, nz, __fmaf_rn(ny, ny, nx * nx))) + 1e-8f;
  nx /= mag; ny /= mag; nz /= mag;
  const float dep = V[(size_t)idx * 3 + 2];
  const float dmin = dec_f(d.mm[2 * l]), dmax = dec_f(~d.mm[2 * l + 1]);
  if (dep <= dmin || dep >= dmax) { nx = 0.f; ny = 0.f; nz = 0.f; }
  float* N = d.normal[l] + (size_t)idx * 3;
  N[0] = nx; N[1] = ny; N[2] = nz;
}

// ---- wave64 sum via DPP, result in lane 63 ---------------------------------------------------
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp_add(float v) {
  const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false);
  return v + __int_as_float(moved);
}
__device__ __forceinline__ float wave_sum63(float v) {
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  v = dpp_add<0x142, 0xa>(v);
  v = dpp_add<0x143, 0xc>(v);
  return v;
}

__device__ __forceinline__ void block_write_partials(float (&acc)[NACC], float* __restrict__ partials) {
  __shared__ float s_part[4 * PSTRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const float r = wave_sum63(acc[k]);
    if (lane == 63) s_part[wave * PSTRIDE + k] = r;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    const int k = threadIdx.x;
    // write-through (sc1) store: visible to the electing workgroup without an L2 write-back fence
    __hip_atomic_store(&partials[(size_t)blockIdx.x * PSTRIDE + k],
                       (s_part[k] + s_part[PSTRIDE + k]) + (s_part[2 * PSTRIDE + k] + s_part[3 * PSTRIDE + k]),
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Persistent track: the workgroup's 28 sums go straight into a row of float64 totals with native f64 atomics - nobody
// has to read 256 partial rows back.  Same-address atomics serialise at ~17 ns each, so the workgroups spread over 8
// rows (32 adds per address instead of 256); readers add the 8 rows in a fixed order.  The order of the adds inside
// a row varies from run to run at the 1e-16 level, which the float32 pose does not see.
__device__ __forceinline__ void block_add_totals(float (&acc)[NACC], double* __restrict__ totals) {
  __shared__ float s_part2[4 * PSTRIDE];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NACC; ++k) {
    const float r = wave_sum63(acc[k]);
    if (lane == 63) s_part2[wave * PSTRIDE + k] = r;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    const int k = threadIdx.x;
    const float v = (s_part2[k] + s_part2[PSTRIDE + k]) + (s_part2[2 * PSTRIDE + k] + s_part2[3 * PSTRIDE + k]);
    if (v != 0.f) unsafeAtomicAdd(&totals[k], (double)v);
  }
}

// ---- K13: final reduction + damped 6x6 solve + SE(3) exp update, one workgroup ------------------
enum { MODE_SOLVE = 0, MODE_EQUATIONS = 1, MODE_P2P = 2 };

__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FinalArgs {
  int mode;
  float damping, inv_pixels;
  float* pose;
  float* stats;
  float* JtJ_out;
  float* Jtr_out;
  float* nvalid_out;
  int first;       // bit 0: first iteration of a track (the last arriver clears the failure counters - no memset launch);
                   // bit 1: the track starts from the identity (`pose` is not read before the last arriver writes it)
  int f32_solve;   // RTGS_ICP_FLAG_F32_SOLVE: damping, inverse, exp and pose product in float32 in the reference's order
};

// Sum of the per-workgroup partial rows in float64, by ONE whole workgroup (256 threads); the 28 totals land in
// s_tot (shared, PSTRIDE doubles) and are visible to every thread on return.
// 256 threads = 32 row groups x 8 four-float columns: every thread issues all of its (<= 16 x 4) write-through loads
// before the first add, so the whole partial matrix costs ~2 memory round trips.
__device__ __forceinline__ void sum_partials(const float* __restrict__ partials, int nblocks, double* s_sum, double* s_tot) {
  {
    const int q = threadIdx.x & 7, grp = threadIdx.x >> 3;
    double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
    if (q < 7) {
      constexpr int MAXR = (MAX_BLOCKS + 31) / 32;
      float t[MAXR][4];
#pragma unroll
      for (int i = 0; i < MAXR; ++i) {
        const int r = grp + 32 * i;
        if (r < nblocks) {
          const float* row = partials + (size_t)r * PSTRIDE + 4 * q;
          t[i][0] = ld_sc1(row); t[i][1] = ld_sc1(row + 1); t[i][2] = ld_sc1(row + 2); t[i][3] = ld_sc1(row + 3);
        } else {
          t[i][0] = t[i][1] = t[i][2] = t[i][3] = 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < MAXR; ++i) { a0 += (double)t[i][0]; a1 += (double)t[i][1]; a2 += (double)t[i][2]; a3 += (double)t[i][3]; }
    }
    s_sum[grp * PSTRIDE + 4 * q] = a0; s_sum[grp * PSTRIDE + 4 * q + 1] = a1;
    s_sum[grp * PSTRIDE + 4 * q + 2] = a2; s_sum[grp * PSTRIDE + 4 * q + 3] = a3;
  }
  __syncthreads();
  if (threadIdx.x < NACC) {
    double t = 0.0;
#pragma unroll
    for (int g = 0; g < 32; ++g) t += s_sum[g * PSTRIDE + threadIdx.x];
    s_tot[threadIdx.x] = t;
  }
  __syncthreads();
}

// One Gauss-Newton update from the 28 sums, by ONE thread: lev_mar_H damping (icp.py:248-256), xi = -H^-1 J^T r
// through a register-resident Cholesky (icp.py:328-334 inverts H on the CPU), pose <- exp(xi) @ pose (icp.py:271-310,
// :259-268).  Returns false (pose untouched) when the damped H is not positive definite.
__device__ __forceinline__ bool gn_update(const double (&S)[NACC], float damping, float* pose) {
  double Hm[6][6], bvec[6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      const int idx = r * 6 - (r * (r - 1)) / 2 + (c - r);     // position in the packed upper triangle
      Hm[r][c] = S[idx]; Hm[c][r] = S[idx];
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) bvec[r] = S[21 + r];
  double tr = 0.0;
#pragma unroll
  for (int r = 0; r < 6; ++r) tr += Hm[r][r];
#pragma unroll
  for (int r = 0; r < 6; ++r) Hm[r][r] += tr * (double)damping;
  // Cholesky H = L L^T, fully unrolled so every array stays in registers (no scratch); the diagonal is kept as its
  // reciprocal (one division per row instead of one per entry)
  double L[6][6], inv[6];
  bool spd = true;
#pragma unroll
  for (int r = 0; r < 6; ++r) {
#pragma unroll
    for (int c = 0; c <= r; ++c) {
      double s = Hm[r][c];
#pragma unroll
      for (int m = 0; m < c; ++m) s -= L[r][m] * L[c][m];
      if (r == c) {
        spd = spd && (s > 0.0);
        L[r][r] = sqrt(s > 0.0 ? s : 1.0);
        inv[r] = 1.0 / L[r][r];
      } else {
        L[r][c] = s * inv[c];
      }
    }
  }
  if (!spd) return false;
  double yv[6], xi[6];
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    double s = -bvec[r];                                     // xi = -H^-1 Jtr (icp.py:328-334)
#pragma unroll
    for (int m = 0; m < r; ++m) s -= L[r][m] * yv[m];
    yv[r] = s * inv[r];
  }
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    double s = yv[r];
#pragma unroll
    for (int m = r + 1; m < 6; ++m) s -= L[m][r] * xi[m];
    xi[r] = s * inv[r];
  }
  // exp_se3 (icp.py:271-310): Rodrigues + left Jacobian.  The three coefficients sin(t)/t,
  // (1-cos t)/t^2, (t-sin t)/t^3 are evaluated by their Maclaurin series in double (|xi_w| of an
  // ICP step is far below 1; 10 terms are exact to double rounding for t < 1 and the identity
  // limit of the reference's eps test falls out), libm sin/cos only beyond that.
  const double w0 = xi[0], w1 = xi[1], w2 = xi[2];
  const double t2 = w0 * w0 + w1 * w1 + w2 * w2;
  double ka, kb, kc;
  if (t2 < 1.0) {
    ka = 1.0; kb = 0.5; kc = 1.0 / 6.0;
    double ta = 1.0, tb = 0.5, tc = 1.0 / 6.0;
#pragma unroll
    for (int n = 1; n <= 10; ++n) {
      ta *= -t2 * (1.0 / (double)((2 * n) * (2 * n + 1)));        // reciprocals fold at compile time
      tb *= -t2 * (1.0 / (double)((2 * n + 1) * (2 * n + 2)));
      tc *= -t2 * (1.0 / (double)((2 * n + 2) * (2 * n + 3)));
      ka += ta; kb += tb; kc += tc;
    }
  } else {
    const float th = sqrtf((float)t2);
    ka = (double)(sinf(th) / th);
    kb = (double)((1.f - cosf(th)) / (th * th));
    kc = (double)((th - sinf(th)) / (th * th * th));
  }
  // W = [w]x, W2 = W W
  const double W2_00 = -(w1 * w1 + w2 * w2), W2_11 = -(w0 * w0 + w2 * w2), W2_22 = -(w0 * w0 + w1 * w1);
  const double W2_01 = w0 * w1, W2_02 = w0 * w2, W2_12 = w1 * w2;
  double E[3][4];
  double Jl[3][3];
  E[0][0] = 1.0 + kb * W2_00; E[0][1] = -ka * w2 + kb * W2_01; E[0][2] = ka * w1 + kb * W2_02;
  E[1][0] = ka * w2 + kb * W2_01; E[1][1] = 1.0 + kb * W2_11; E[1][2] = -ka * w0 + kb * W2_12;
  E[2][0] = -ka * w1 + kb * W2_02; E[2][1] = ka * w0 + kb * W2_12; E[2][2] = 1.0 + kb * W2_22;
  Jl[0][0] = 1.0 + kc * W2_00; Jl[0][1] = -kb * w2 + kc * W2_01; Jl[0][2] = kb * w1 + kc * W2_02;
  Jl[1][0] = kb * w2 + kc * W2_01; Jl[1][1] = 1.0 + kc * W2_11; Jl[1][2] = -kb * w0 + kc * W2_12;
  Jl[2][0] = -kb * w1 + kc * W2_02; Jl[2][1] = kb * w0 + kc * W2_12; Jl[2][2] = 1.0 + kc * W2_22;
#pragma unroll
  for (int r = 0; r < 3; ++r) E[r][3] = Jl[r][0] * xi[3] + Jl[r][1] * xi[4] + Jl[r][2] * xi[5];
  double Pm[3][4];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) Pm[r][c] = (double)pose[r * 4 + c];
  // pose <- exp(xi) @ pose (bottom row stays 0 0 0 1)
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double o = E[r][0] * Pm[0][c] + E[r][1] * Pm[1][c] + E[r][2] * Pm[2][c];
      if (c == 3) o += E[r][3];
      pose[r * 4 + c] = (float)o;
    }
  return true;
}

// The same update in FLOAT32 and in the reference's own order of operations (RTGS_ICP_FLAG_F32_SOLVE):
// lev_mar_H (icp.py:248-256) -> invH = torch.inverse(H) on the CPU (:313-325: LAPACK sgetrf + sgetri, i.e. an LU with row
// pivoting, U inverted, then inv(A) L = inv(U) solved column by column from the right) -> xi = -invH @ Rhs (:328-334) ->
// exp_se3 with float32 sin / cos and its eps test (:271-310) -> exp(xi) @ pose.  The 27 sums still arrive in float64 (with
// float64 sums and the float32 solve the reference's own code reproduces its float32 answer to 2e-9).  What this
// cannot reproduce is the blocked / vectorised order INSIDE the library calls (sgetri's triangular solves, the 8-lane
// horizontal sum behind torch.sum): differences of one float32 rounding remain.  Measured: tests/test_icp_gpu.py.
__device__ __forceinline__ bool gn_update_f32(const double (&S)[NACC], float damping, float* pose) {
  float A[6][6], b[6];
#pragma unroll
  for (int r = 0; r < 6; ++r)
#pragma unroll
    for (int c = r; c < 6; ++c) {
      const int idx = r * 6 - (r * (r - 1)) / 2 + (c - r);
      A[r][c] = (float)S[idx]; A[c][r] = (float)S[idx];
    }
#pragma unroll
  for (int r = 0; r < 6; ++r) b[r] = (float)S[21 + r];
  float tr = 0.f;
#pragma unroll
  for (int r = 0; r < 6; ++r) tr = __fadd_rn(tr, A[r][r]);
  const float eps = __fmul_rn(tr, damping);
#pragma unroll
  for (int r = 0; r < 6; ++r) A[r][r] = __fadd_rn(A[r][r], eps);
  // sgetf2: right-looking LU with partial pivoting (column scaled by the reciprocal pivot, rank-1 update)
  int piv[6];
  for (int k = 0; k < 6; ++k) {
    int pk = k;
    float mx = fabsf(A[k][k]);
    for (int r = k + 1; r < 6; ++r) if (fabsf(A[r][k]) > mx) { mx = fabsf(A[r][k]); pk = r; }
    piv[k] = pk;
    if (mx == 0.f) return false;
    if (pk != k) for (int c = 0; c < 6; ++c) { const float t = A[k][c]; A[k][c] = A[pk][c]; A[pk][c] = t; }
    const float rp = __fdiv_rn(1.f, A[k][k]);
    for (int r = k + 1; r < 6; ++r) A[r][k] = __fmul_rn(A[r][k], rp);
    for (int r = k + 1; r < 6; ++r)
      for (int c = k + 1; c < 6; ++c) A[r][c] = __fsub_rn(A[r][c], __fmul_rn(A[r][k], A[k][c]));
  }
  // strtri: inv(U) in place, column by column (upper, non-unit)
  for (int j = 0; j < 6; ++j) {
    A[j][j] = __fdiv_rn(1.f, A[j][j]);
    const float ajj = -A[j][j];
    // x = U(0:j, 0:j) * U(0:j, j)  (strmv, upper, no transpose: rows from the top, each using the already inverted block)
    for (int r = 0; r < j; ++r) {
      float t = 0.f;
      bool first = true;
      for (int c = r; c < j; ++c) {
        const float pr = __fmul_rn(A[r][c], A[c][j]);
        t = first ? pr : __fadd_rn(t, pr);
        first = false;
      }
      // (strmv works in place top-down: row r of the product needs only entries r..j-1 of the vector, still untouched)
      A[r][j] = t;
    }
    for (int r = 0; r < j; ++r) A[r][j] = __fmul_rn(A[r][j], ajj);
  }
  // sgetri: for j = n-1 .. 0: save L's column j, zero it, A(:, j) -= A(:, j+1:) * l(j+1:)
  for (int j = 4; j >= 0; --j) {
    float l[6];
    for (int r = j + 1; r < 6; ++r) { l[r] = A[r][j]; A[r][j] = 0.f; }
    for (int r = 0; r < 6; ++r) {
      float t = A[r][j];
      for (int c = j + 1; c < 6; ++c) t = __fsub_rn(t, __fmul_rn(A[r][c], l[c]));
      A[r][j] = t;
    }
  }
  // the column interchanges that undo the row pivoting
  for (int j = 4; j >= 0; --j)
    if (piv[j] != j) for (int r = 0; r < 6; ++r) { const float t = A[r][j]; A[r][j] = A[r][piv[j]]; A[r][piv[j]] = t; }
  float xi[6];
  for (int r = 0; r < 6; ++r) {
    float t = __fmul_rn(-A[r][0], b[0]);
    for (int c = 1; c < 6; ++c) t = __fadd_rn(t, __fmul_rn(-A[r][c], b[c]));
    xi[r] = t;
  }
  // exp_se3, icp.py:271-310, float32
  const float w0 = xi[0], w1 = xi[1], w2 = xi[2];
  const float Wm[3][3] = {{0.f, -w2, w1}, {w2, 0.f, -w0}, {-w1, w0, 0.f}};
  float W2[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) W2[r][c] = __fadd_rn(__fadd_rn(__fmul_rn(Wm[r][0], Wm[0][c]), __fmul_rn(Wm[r][1], Wm[1][c])), __fmul_rn(Wm[r][2], Wm[2][c]));
  const float theta = sqrtf(__fadd_rn(__fadd_rn(__fmul_rn(w0, w0), __fmul_rn(w1, w1)), __fmul_rn(w2, w2)));
  float E[3][3], Jl[3][3];
  if (theta <= 1e-8f) {
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) { E[r][c] = r == c ? 1.f : 0.f; Jl[r][c] = E[r][c]; }
  } else {
    const float t2 = __fmul_rn(theta, theta), t3 = __fmul_rn(t2, theta), st = sinf(theta), ct = cosf(theta);
    const float k1 = __fdiv_rn(__fsub_rn(1.f, ct), t2), k2 = __fdiv_rn(__fsub_rn(theta, st), t3);
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) {
        const float I = r == c ? 1.f : 0.f;
        // eye + w_hat * sin / theta + w_hat_second * (1 - cos) / theta^2   (left to right, as the expression is written)
        E[r][c] = __fadd_rn(__fadd_rn(I, __fdiv_rn(__fmul_rn(Wm[r][c], st), theta)), __fdiv_rn(__fmul_rn(W2[r][c], __fsub_rn(1.f, ct)), t2));
        Jl[r][c] = __fadd_rn(__fadd_rn(I, __fmul_rn(k1, Wm[r][c])), __fmul_rn(k2, W2[r][c]));
      }
  }
  float tv[3];
  for (int r = 0; r < 3; ++r) tv[r] = __fadd_rn(__fadd_rn(__fmul_rn(Jl[r][0], xi[3]), __fmul_rn(Jl[r][1], xi[4])), __fmul_rn(Jl[r][2], xi[5]));
  float Pm[4][4];
  for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) Pm[r][c] = pose[r * 4 + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 4; ++c) {
      float o = __fmul_rn(E[r][0], Pm[0][c]);
      o = __fadd_rn(o, __fmul_rn(E[r][1], Pm[1][c]));
      o = __fadd_rn(o, __fmul_rn(E[r][2], Pm[2][c]));
      o = __fadd_rn(o, __fmul_rn(tv[r], Pm[3][c]));
      pose[r * 4 + c] = o;
    }
  return true;
}

// Executed by ONE whole workgroup (256 threads): the last one to arrive in the residual kernel.
__device__ __forceinline__ void final_stage(const float* __restrict__ partials, int nblocks, const FinalArgs& fa) {
  __shared__ double s_sum[32 * PSTRIDE];
  __shared__ double s_tot[PSTRIDE];
  sum_partials(partials, nblocks, s_sum, s_tot);
  if (threadIdx.x != 0) return;
  double S[NACC];
#pragma unroll
  for (int c = 0; c < NACC; ++c) S[c] = s_tot[c];
  if (fa.mode == MODE_P2P) { fa.stats[1] = (float)(S[0] * (double)fa.inv_pixels); return; }
  if (fa.mode == MODE_EQUATIONS) {
#pragma unroll
    for (int r = 0; r < 6; ++r) {
#pragma unroll
      for (int c = 0; c < 6; ++c) {
        const int lo = r < c ? r : c, hi = r < c ? c : r;
        fa.JtJ_out[r * 6 + c] = (float)S[lo * 6 - (lo * (lo - 1)) / 2 + (hi - lo)];
      }
      fa.Jtr_out[r] = (float)S[21 + r];
    }
    fa.nvalid_out[0] = (float)S[27];
    return;
  }
  if (fa.first & 1) { fa.stats[1] = 0.f; fa.stats[2] = 0.f; fa.stats[3] = 0.f; }
  if (fa.first & 2) {
#pragma unroll
    for (int k = 0; k < 16; ++k) fa.pose[k] = (k % 5 == 0) ? 1.f : 0.f;
  }
  fa.stats[0] = (float)(S[27] * (double)fa.inv_pixels);            // valid_ratio (icp.py:46-47)
  if (!(fa.f32_solve ? gn_update_f32(S, fa.damping, fa.pose) : gn_update(S, fa.damping, fa.pose))) fa.stats[2] += 1.f;
}

// Publish this workgroup's partial row and elect the last arriver.  Hand-off form (cdna_hip_programming.md
// Guideline 16, write-through variant): the producer stores its row with sc1 (agent-scope relaxed atomic
// stores), drains them (vmcnt(0)), then takes a ticket; the last arriver reads every row with sc1 loads -
// no L2 write-back / invalidate fence on either side.  The ticket is re-armed by the last workgroup
// (and zeroed once per call by a memset node on the stream).
__device__ __forceinline__ bool arrive_and_elect_last(uint32_t* ticket) {
  __shared__ uint32_t s_is_last;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const uint32_t last = (t == gridDim.x - 1) ? 1u : 0u;
    if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_is_last = last;
  }
  __syncthreads();
  return s_is_last != 0;
}

// ---- K12: residual + Jacobian + 6x6 reduction (icp.py:52-119) ---------------------------------
struct LevelGeom {
  float R00, R01, R02, t0, R10, R11, R12, t1, R20, R21, R22, t2;
  float fx, fy, cx, cy, Wm1, Hm1, hw, hh, dist_thr, cos_thr;
  int W;
};

// One source pixel: transform, project, nearest-neighbour association, the three gates, and the rank-1 update of the
// 27 sums + valid count.  The float32 op sequence of the gating arithmetic follows the reference's (icp.py:52-104,
// warp_features :132-148) so that associations and gate decisions match it bit for bit.
__device__ __forceinline__ void accumulate_pixel(const LevelGeom& g, float v0, float v1, float v2, float n0, float n1,
                                                 float n2, const float* __restrict__ vt, const float* __restrict__ nt,
                                                 float (&acc)[NACC]) {
  const float px = (g.R00 * v0 + g.R01 * v1 + g.R02 * v2) + g.t0;
  const float py = (g.R10 * v0 + g.R11 * v1 + g.R12 * v2) + g.t1;
  const float pz = (g.R20 * v0 + g.R21 * v1 + g.R22 * v2) + g.t2;
  const float u = (px / pz) * g.fx + g.cx;
  const float v = (py / pz) * g.fy + g.cy;
  const bool inview = (u > 0.f) && (u < g.Wm1) && (v > 0.f) && (v < g.Hm1);
  if (!inview || !(v2 > 0.f)) return;
  // grid_sample(nearest, border, align_corners=True) of warp_features (icp.py:132-148)
  const float un = u / g.hw - 1.f, vn = v / g.hh - 1.f;
  float ix = ((un + 1.f) / 2.f) * g.Wm1, iy = ((vn + 1.f) / 2.f) * g.Hm1;
  ix = fminf(g.Wm1, fmaxf(ix, 0.f));
  iy = fminf(g.Hm1, fmaxf(iy, 0.f));
  const int xi = (int)nearbyintf(ix), yi = (int)nearbyintf(iy);
  const size_t j = ((size_t)yi * g.W + xi) * 3;
  const float q0 = vt[j], q1 = vt[j + 1], q2 = vt[j + 2];
  if (!(q2 > 0.f)) return;
  const float m0 = nt[j], m1 = nt[j + 1], m2 = nt[j + 2];
  const float rn0 = g.R00 * n0 + g.R01 * n1 + g.R02 * n2;
  const float rn1 = g.R10 * n0 + g.R11 * n1 + g.R12 * n2;
  const float rn2 = g.R20 * n0 + g.R21 * n1 + g.R22 * n2;
  if (!(rn0 * m0 + rn1 * m1 + rn2 * m2 > g.cos_thr)) return;
  const float d0 = px - q0, d1 = py - q1, d2 = pz - q2;
  if (sqrtf(d0 * d0 + d1 * d1 + d2 * d2) > g.dist_thr) return;
  const float r = m0 * d0 + m1 * d1 + m2 * d2;
  float J[6];
  J[0] = py * m2 - pz * m1;        // -(m^T [p]x) = p x m
  J[1] = pz * m0 - px * m2;
  J[2] = px * m1 - py * m0;
  J[3] = m0; J[4] = m1; J[5] = m2;
  int k = 0;
#pragma unroll
  for (int a = 0; a < 6; ++a)
#pragma unroll
    for (int b = a; b < 6; ++b) acc[k++] += J[a] * J[b];
#pragma unroll
  for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r;
  acc[27] += 1.f;
}

__device__ const float c_identity[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};

__device__ __forceinline__ LevelGeom make_geom(const float* pose, const float* __restrict__ K, float ds, int H, int W,
                                               float dist_thr, float cos_thr) {
  LevelGeom g;
  g.R00 = pose[0]; g.R01 = pose[1]; g.R02 = pose[2]; g.t0 = pose[3];
  g.R10 = pose[4]; g.R11 = pose[5]; g.R12 = pose[6]; g.t1 = pose[7];
  g.R20 = pose[8]; g.R21 = pose[9]; g.R22 = pose[10]; g.t2 = pose[11];
  g.fx = K[0] * ds; g.fy = K[4] * ds; g.cx = K[2] * ds; g.cy = K[5] * ds;
  g.Wm1 = (float)(W - 1); g.Hm1 = (float)(H - 1);
  g.hw = g.Wm1 / 2.f; g.hh = g.Hm1 / 2.f;
  g.dist_thr = dist_thr; g.cos_thr = cos_thr; g.W = W;
  return g;
}

// The source maps are [n,3] float32: a lane takes FOUR consecutive pixels = 48 contiguous bytes per map = three 16-B
// loads (the 12-B pixel stride rules out one vector load per pixel); the remaining n % 4 pixels go to one lane each.
__device__ __forceinline__ void accumulate_range(const LevelGeom& g, const float* __restrict__ vs,
                                                 const float* __restrict__ ns, const float* __restrict__ vt,
                                                 const float* __restrict__ nt, int n, int first, int stride,
                                                 float (&acc)[NACC]) {
  const int n4 = n >> 2;
  const float4* vs4 = reinterpret_cast<const float4*>(vs);
  const float4* ns4 = reinterpret_cast<const float4*>(ns);
  for (int q = first; q < n4; q += stride) {
    const float4 a0 = vs4[3 * q], a1 = vs4[3 * q + 1], a2 = vs4[3 * q + 2];
    const float4 b0 = ns4[3 * q], b1 = ns4[3 * q + 1], b2 = ns4[3 * q + 2];
    accumulate_pixel(g, a0.x, a0.y, a0.z, b0.x, b0.y, b0.z, vt, nt, acc);
    accumulate_pixel(g, a0.w, a1.x, a1.y, b0.w, b1.x, b1.y, vt, nt, acc);
    accumulate_pixel(g, a1.z, a1.w, a2.x, b1.z, b1.w, b2.x, vt, nt, acc);
    accumulate_pixel(g, a2.y, a2.z, a2.w, b2.y, b2.z, b2.w, vt, nt, acc);
  }
  const int idx = 4 * n4 + first;
  if (idx < n)
    accumulate_pixel(g, vs[(size_t)idx * 3], vs[(size_t)idx * 3 + 1], vs[(size_t)idx * 3 + 2], ns[(size_t)idx * 3],
                     ns[(size_t)idx * 3 + 1], ns[(size_t)idx * 3 + 2], vt, nt, acc);
}

// K points at the FULL-resolution intrinsics; `ds` is the level's downscale (icp.py:431-433).
__global__ void __launch_bounds__(256) icp_reduce_kernel(
    const float* __restrict__ vs, const float* __restrict__ ns, const float* __restrict__ vt,
    const float* __restrict__ nt, int H, int W, const float* __restrict__ K, float ds,
    const float* __restrict__ pose, float dist_thr, float cos_thr, float* __restrict__ partials,
    uint32_t* __restrict__ ticket, FinalArgs fa) {
  wave_priority_high();
  const LevelGeom g = make_geom((fa.first & 2) ? c_identity : pose, K, ds, H, W, dist_thr, cos_thr);
  float acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
  accumulate_range(g, vs, ns, vt, nt, H * W, blockIdx.x * 256 + threadIdx.x, gridDim.x * 256, acc);
  block_write_partials(acc, partials);
  if (arrive_and_elect_last(ticket)) final_stage(partials, (int)gridDim.x, fa);
}

// ---- point2plane_loss (icp.py:7-13) at one level: sum(((R v1 + t - v0) . n0)^2) ---------------
__global__ void __launch_bounds__(256) icp_p2p_kernel(const float* __restrict__ vs, const float* __restrict__ vt,
                                                      const float* __restrict__ nt, int n,
                                                      const float* __restrict__ pose, float* __restrict__ partials,
                                                      uint32_t* __restrict__ ticket, FinalArgs fa) {
  wave_priority_high();
  const float R00 = pose[0], R01 = pose[1], R02 = pose[2], t0 = pose[3];
  const float R10 = pose[4], R11 = pose[5], R12 = pose[6], t1 = pose[7];
  const float R20 = pose[8], R21 = pose[9], R22 = pose[10], t2 = pose[11];
  float acc[NACC];
#pragma unroll
  for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < n; idx += gridDim.x * 256) {
    const size_t j = (size_t)idx * 3;
    const float v0 = vs[j], v1 = vs[j + 1], v2 = vs[j + 2];
    const float px = (v0 * R00 + v1 * R01 + v2 * R02) + t0;
    const float py = (v0 * R10 + v1 * R11 + v2 * R12) + t1;
    const float pz = (v0 * R20 + v1 * R21 + v2 * R22) + t2;
    const float l = (px - vt[j]) * nt[j] + (py - vt[j + 1]) * nt[j + 1] + (pz - vt[j + 2]) * nt[j + 2];
    acc[0] += l * l;
  }
  block_write_partials(acc, partials);
  if (arrive_and_elect_last(ticket)) final_stage(partials, (int)gridDim.x, fa);
}

// ---- the whole track as ONE persistent kernel ---------------------------------------------------
// IcpTracker.predict_pose's level loop (icp.py:428-447: 3 levels x 5 Gauss-Newton iterations, then the p2p loss) used
// to be 16 dependent launches, each with a whole-grid fan-in to a last-arriver workgroup and a fan-out through the next
// launch.  Here the workgroups stay resident (one per CU) and meet at a grid barrier per iteration:
//   every workgroup: partial sums of its pixels -> write-through row -> ticket -> spin until all rows are in ->
//   EVERY workgroup sums all rows (float64, same order everywhere) and takes the same Gauss-Newton step on its own
//   LDS copy of the pose.
// One global round trip per iteration instead of three (fan-in, solve, fan-out), no launch gaps, and nothing for the
// other stream's kernels to squeeze between.  The sums of an iteration meet in a row of float64 totals (native f64
// atomics; a fresh row per barrier, zeroed by the host).  The spin is bounded: a workgroup that waits too long raises an abort
// flag and everybody leaves (stats[3] = 1; the host reports it) - a scheduling pathology must not hang the device.
struct TrackLevel {
  const float *vs, *ns, *vt, *nt;
  int H, W, iters;
  float ds;
};
struct TrackArgs {
  TrackLevel lv[RTGS_ICP_MAX_LEVELS];
  int n_levels;
  const float* K;
  float dist_thr, cos_thr, damping;
  float* pose;
  float* stats;
  double* sums;                // [barriers][8][PSTRIDE], zeroed by the host
  uint32_t* ticket;
  uint32_t* abort_flag;
  unsigned long long* dbg;     // nullptr = no timing stamps
  // Cluster form: only the first `levels_here` levels run in this kernel, on `cluster` workgroups that all sit on ONE XCD
  // (workgroups are dealt round-robin over the 8 XCDs: of a launch of 8 x cluster workgroups those with
  // blockIdx % 8 == 0 stay, the rest leave at once) - the barrier's atomics and the coarse maps stay in that XCD's L2
  // and the other 224 CUs are free for the mapper's stream.  0 = every workgroup of the launch, every level + the p2p loss.
  int cluster, levels_here;
};

__device__ __forceinline__ bool grid_arrive_wait(uint32_t* ticket, uint32_t target, uint32_t* abort_flag, float* stats) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this workgroup's write-through row is out
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int ok = 0;
    for (int spin = 0; spin < (1 << 22); ++spin) {
      if (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
      if ((spin & 63) == 63 && __hip_atomic_load(abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
      __builtin_amdgcn_s_sleep(1);
    }
    if (!ok) {
      __hip_atomic_store(abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      stats[3] = 1.f;
    }
    s_ok = ok;
  }
  __syncthreads();
  return s_ok != 0;
}

__device__ __forceinline__ double load_totals(const double* totals, int k) {
  double t[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) t[r] = __hip_atomic_load(&totals[r * PSTRIDE + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
}

__global__ void __launch_bounds__(256) icp_track_kernel(TrackArgs a) {
  __shared__ float s_pose[16];
  __shared__ float s_stat[4];
  __shared__ double s_tot[PSTRIDE];
  int G = (int)gridDim.x, bid = (int)blockIdx.x;
  if (a.cluster > 0) {
    if ((blockIdx.x & 7u) != 0u) return;       // the other XCDs' workgroups: not part of the cluster
    G = a.cluster; bid = (int)(blockIdx.x >> 3);
  }
  const int n_levels = a.cluster > 0 ? a.levels_here : a.n_levels;
  if (threadIdx.x < 16) s_pose[threadIdx.x] = a.pose[threadIdx.x];
  if (threadIdx.x < 4) s_stat[threadIdx.x] = 0.f;
  __syncthreads();
  uint32_t epoch = 0;
  float acc[NACC];
  for (int l = 0; l < n_levels; ++l) {
    const TrackLevel L = a.lv[l];
    const int n = L.H * L.W;
    const float inv = 1.f / ((float)L.H * (float)L.W);
    for (int it = 0; it < L.iters; ++it) {
      const bool stamp = a.dbg && bid == 0 && threadIdx.x == 0 && epoch < 19;
      if (stamp) a.dbg[5 * epoch] = wall_clock64();
      const LevelGeom g = make_geom(s_pose, a.K, L.ds, L.H, L.W, a.dist_thr, a.cos_thr);
#pragma unroll
      for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
      accumulate_range(g, L.vs, L.ns, L.vt, L.nt, n, bid * 256 + (int)threadIdx.x, G * 256, acc);
      double* totals = a.sums + (size_t)epoch * 8 * PSTRIDE;
      if (stamp) a.dbg[5 * epoch + 1] = wall_clock64();
      block_add_totals(acc, totals + ((uint32_t)bid & 7u) * PSTRIDE);
      ++epoch;
      if (!grid_arrive_wait(a.ticket, epoch * (uint32_t)G, a.abort_flag, a.stats)) return;
      if (stamp) a.dbg[5 * epoch - 3] = wall_clock64();
      if (threadIdx.x < NACC) s_tot[threadIdx.x] = load_totals(totals, threadIdx.x);
      __syncthreads();
      if (stamp) a.dbg[5 * epoch - 2] = wall_clock64();
      if (threadIdx.x == 0) {
        double S[NACC];
#pragma unroll
        for (int c = 0; c < NACC; ++c) S[c] = s_tot[c];
        s_stat[0] = (float)(S[27] * (double)inv);                // valid_ratio of this iteration (icp.py:46-47)
        if (!gn_update(S, a.damping, s_pose)) s_stat[2] += 1.f;
        if (stamp) a.dbg[5 * epoch - 1] = wall_clock64();
      }
      __syncthreads();
    }
  }
  if (a.cluster > 0) {          // the finest level and the p2p loss follow as launches of their own: hand the pose over
    if (bid == 0) {
      if (threadIdx.x < 12) a.pose[threadIdx.x] = s_pose[threadIdx.x];
      if (threadIdx.x == 0) { a.stats[0] = s_stat[0]; a.stats[2] = s_stat[2]; }
    }
    return;
  }
  // point2plane_loss (icp.py:7-13, :443-447) of the final pose at the finest level, no association
  {
    const TrackLevel F = a.lv[a.n_levels - 1];
    const int n = F.H * F.W;
    const float R00 = s_pose[0], R01 = s_pose[1], R02 = s_pose[2], t0 = s_pose[3];
    const float R10 = s_pose[4], R11 = s_pose[5], R12 = s_pose[6], t1 = s_pose[7];
    const float R20 = s_pose[8], R21 = s_pose[9], R22 = s_pose[10], t2 = s_pose[11];
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = 0.f;
    for (int idx = bid * 256 + (int)threadIdx.x; idx < n; idx += G * 256) {
      const size_t j = (size_t)idx * 3;
      const float v0 = F.vs[j], v1 = F.vs[j + 1], v2 = F.vs[j + 2];
      const float px = (v0 * R00 + v1 * R01 + v2 * R02) + t0;
      const float py = (v0 * R10 + v1 * R11 + v2 * R12) + t1;
      const float pz = (v0 * R20 + v1 * R21 + v2 * R22) + t2;
      const float lp = (px - F.vt[j]) * F.nt[j] + (py - F.vt[j + 1]) * F.nt[j + 1] + (pz - F.vt[j + 2]) * F.nt[j + 2];
      acc[0] += lp * lp;
    }
    double* totals = a.sums + (size_t)epoch * 8 * PSTRIDE;
    block_add_totals(acc, totals + ((uint32_t)bid & 7u) * PSTRIDE);
    ++epoch;
    if (!grid_arrive_wait(a.ticket, epoch * (uint32_t)G, a.abort_flag, a.stats)) return;
    if (bid != 0) return;
    if (threadIdx.x < 12) a.pose[threadIdx.x] = s_pose[threadIdx.x];
    if (threadIdx.x == 0) {
      a.stats[0] = s_stat[0];
      a.stats[1] = (float)(load_totals(totals, 0) * (1.0 / (double)n));
      a.stats[2] = s_stat[2];
    }
  }
}

// ---- model-depth hole filling (icp.py:397-415) ------------------------------------------------
__global__ void __launch_bounds__(256) icp_fill_kernel(float* __restrict__ rd, const float* __restrict__ fd,
                                                       const float* __restrict__ rn, const float* __restrict__ fn,
                                                       int n, float dist_thr, float normal_thr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float r = rd[i], f = fd[i];
  const float a0 = rn[3 * i], a1 = rn[3 * i + 1], a2 = rn[3 * i + 2];
  const float b0 = fn[3 * i], b1 = fn[3 * i + 1], b2 = fn[3 * i + 2];
  // F.cosine_similarity(dim=-1, eps=1e-8): (a / max(|a|, eps)) . (b / max(|b|, eps))
  const float na = fmaxf(sqrtf(a0 * a0 + a1 * a1 + a2 * a2), 1e-8f);
  const float nb = fmaxf(sqrtf(b0 * b0 + b1 * b1 + b2 * b2), 1e-8f);
  const float cs = (a0 / na) * (b0 / nb) + (a1 / na) * (b1 / nb) + (a2 / na) * (b2 / nb);
  const bool fill = ((fabsf(r - f) > dist_thr) || (r == 0.f) || ((1.f - cs) > normal_thr)) && (f > 0.f);
  if (fill) rd[i] = f;
}

static int grid_for(int n) {
  int g = (n + 255) / 256;
  return g > MAX_BLOCKS ? MAX_BLOCKS : (g < 1 ? 1 : g);
}

}  // namespace rtgs_icp

using namespace rtgs_icp;

#define ICP_TRY(expr)                \
  do {                               \
    if ((expr) != hipSuccess) return -2; \
  } while (0)

extern "C" {

size_t rtgs_icp_scratch_bytes(void) { return sizeof(Scratch); }

int rtgs_icp_build_pyramids(const float* depth, int32_t H, int32_t W, const float* K, int32_t levels,
                            float* const* vertex_out, float* const* normal_out, void* scratch, void* stream) {
  return rtgs_icp_build_pyramids_ex(depth, H, W, K, levels, vertex_out, normal_out, scratch, 0, stream);
}

int rtgs_icp_scratch_init(void* scratch, void* stream) {
  if (!scratch) return -1;
  Scratch* sc = (Scratch*)scratch;
  hipStream_t st = (hipStream_t)stream;
  ICP_TRY(hipMemsetAsync(sc, 0, sizeof(Scratch), st));
  ICP_TRY(hipMemsetAsync(sc->minmax, 0xff, sizeof(sc->minmax), st));
  ICP_TRY(hipMemsetAsync(sc->minmax_b, 0xff, sizeof(sc->minmax_b), st));
  return 0;
}

int rtgs_icp_build_pyramids_ex(const float* depth, int32_t H, int32_t W, const float* K, int32_t levels,
                               float* const* vertex_out, float* const* normal_out, void* scratch, int32_t flags,
                               void* stream) {
  if (!depth || !K || !vertex_out || !normal_out || !scratch || H <= 0 || W <= 0 || levels < 1 ||
      levels > RTGS_ICP_MAX_LEVELS)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  Scratch* sc = (Scratch*)scratch;
  PyrDesc d{};
  d.levels = levels; d.H = H; d.W = W;
  int blocks = 0;
  for (int l = 0; l < levels; ++l) {
    const int sh = levels - 1 - l;
    d.shift[l] = sh; d.Hl[l] = H >> sh; d.Wl[l] = W >> sh;     // MaxPool2d(2^sh, 2^sh), floor mode
    if (d.Hl[l] < 1 || d.Wl[l] < 1 || !vertex_out[l] || !normal_out[l]) return -1;
    d.vertex[l] = vertex_out[l]; d.normal[l] = normal_out[l];
    d.block_start[l] = blocks;
    blocks += (d.Hl[l] * d.Wl[l] + 255) / 256;
  }
  d.block_start[levels] = blocks;
  if (flags & RTGS_ICP_PYR_SCRATCH_READY) {       // both sets were armed once (rtgs_icp_scratch_init): use one, re-arm the other
    const bool second = (flags & RTGS_ICP_PYR_SECOND_SET) != 0;
    d.mm = second ? sc->minmax_b : sc->minmax;
    d.mm_rearm = second ? sc->minmax : sc->minmax_b;
  } else {
    d.mm = sc->minmax; d.mm_rearm = nullptr;
    ICP_TRY(hipMemsetAsync(sc->minmax, 0xff, sizeof(sc->minmax), st));
  }
  // the one-pass kernel moves 16-byte rows: it needs the depth image and the vertex maps 16-byte aligned (torch
  // allocations are; a view with a storage offset may not be) - anything else takes the scalar kernel
  uintptr_t align_bits = (uintptr_t)depth;
  for (int l = 0; l < levels; ++l) align_bits |= (uintptr_t)vertex_out[l];
  if (levels == 3 && (align_bits & 15u) == 0) {   // the shipped configuration (icp_downscales [0.25, 0.5, 1.0]): all three levels in one pass
    const int nb = ((H + 3) / 4) * ((W + 3) / 4);
    hipLaunchKernelGGL(icp_vertex3_kernel, dim3(grid_for(nb)), dim3(256), 0, st, d, depth, K, sc);
  } else {
    hipLaunchKernelGGL(icp_vertex_kernel, dim3(blocks < 512 ? blocks : 512), dim3(256), 0, st, d, depth, K, sc);
  }
  hipLaunchKernelGGL(icp_normal_kernel, dim3(blocks), dim3(256), 0, st, d, (const Scratch*)sc);
  ICP_TRY(hipGetLastError());
  return 0;
}

int rtgs_icp_step(const float* vs, const float* ns, const float* vt, const float* nt, int32_t H, int32_t W,
                  const float* K, const float* pose, float dist_thr, float cos_thr, float* JtJ_out, float* Jtr_out,
                  float* nvalid_out, void* scratch, void* stream) {
  if (!vs || !ns || !vt || !nt || !K || !pose || !JtJ_out || !Jtr_out || !nvalid_out || !scratch || H <= 0 || W <= 0)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  Scratch* sc = (Scratch*)scratch;
  const int g = grid_for(H * W);
  ICP_TRY(hipMemsetAsync(&sc->ticket, 0, sizeof(uint32_t), st));
  FinalArgs fa{(int)MODE_EQUATIONS, 0.f, 0.f, nullptr, nullptr, JtJ_out, Jtr_out, nvalid_out};
  hipLaunchKernelGGL(icp_reduce_kernel, dim3(g), dim3(256), 0, st, vs, ns, vt, nt, H, W, K, 1.0f, pose, dist_thr,
                     cos_thr, sc->partials, &sc->ticket, fa);
  ICP_TRY(hipGetLastError());
  return 0;
}

int rtgs_icp_track(const rtgs_icp_level* lv, int32_t n_levels, const float* K, float dist_thr, float cos_thr,
                   float damping, float* pose, float* stats, void* scratch, int32_t flags, void* stream) {
  if (!lv || n_levels < 1 || n_levels > RTGS_ICP_MAX_LEVELS || !K || !pose || !stats || !scratch) return -1;
  hipStream_t st = (hipStream_t)stream;
  Scratch* sc = (Scratch*)scratch;
  for (int l = 0; l < n_levels; ++l) {
    const rtgs_icp_level& L = lv[l];
    if (!L.vertex_src || !L.normal_src || !L.vertex_tgt || !L.normal_tgt || L.H <= 0 || L.W <= 0 || L.iters < 0)
      return -1;
  }
  // launch-per-iteration form on a scratch that was initialised once (RTGS_ICP_FLAG_SCRATCH_READY): no memset launches -
  // the ticket is re-armed by every last arriver, the failure counters are cleared by the first iteration's
  const bool ready = (flags & RTGS_ICP_FLAG_SCRATCH_READY) != 0;
  const bool from_identity = (flags & RTGS_ICP_FLAG_FROM_IDENTITY) != 0;
  // Two forms, same arithmetic.  One launch per Gauss-Newton iteration (default): 16 short kernels that leave the GPU
  // to whatever else is queued between them - measured best when the tracker overlaps the map optimisation on another
  // stream (1 690 vs 1 511 frames/s in bench.py).  One persistent kernel (RTGS_ICP_FLAG_PERSISTENT, or the environment
  // override RTGS_ICP_PERSISTENT=1/0): ~8 % faster when the tracker has the device to itself (287 vs 312 us per
  // 1200x680 track), but its resident, mostly waiting workgroups slow co-running kernels down.
  static const int env_persistent = [] { const char* e = getenv("RTGS_ICP_PERSISTENT"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  const bool persistent = env_persistent >= 0 ? env_persistent == 1 : (flags & RTGS_ICP_FLAG_PERSISTENT) != 0;
  static const int env_cluster = [] { const char* e = getenv("RTGS_ICP_CLUSTER"); return e ? atoi(e) : -1; }();
  const int cluster = env_cluster >= 0 ? env_cluster : ((flags & RTGS_ICP_FLAG_CLUSTER) ? 64 : 0);
  if ((flags & RTGS_ICP_FLAG_F32_SOLVE) && (persistent || (cluster > 0 && n_levels >= 2))) return -1;   // the launch-per-iteration chain only
  const bool plain = !persistent && !(cluster > 0 && n_levels >= 2);
  if (!(plain && ready)) {
    ICP_TRY(hipMemsetAsync(stats, 0, 4 * sizeof(float), st));
    ICP_TRY(hipMemsetAsync(&sc->ticket, 0, 2 * sizeof(uint32_t), st));       // ticket + abort flag
  }
  if (from_identity && !plain) {                   // the one-kernel forms read the pose: give them the identity
    static const float eye[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};
    ICP_TRY(hipMemcpyAsync(pose, eye, sizeof(eye), hipMemcpyHostToDevice, st));
  }
  if (persistent) {
    int dev = 0, cus = 0;
    ICP_TRY(hipGetDevice(&dev));
    ICP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int G = cus > 0 ? cus : 64;
    if (G > MAX_BLOCKS) G = MAX_BLOCKS;
    TrackArgs a{};
    for (int l = 0; l < n_levels; ++l)
      a.lv[l] = TrackLevel{lv[l].vertex_src, lv[l].normal_src, lv[l].vertex_tgt, lv[l].normal_tgt, lv[l].H, lv[l].W,
                           lv[l].iters, lv[l].downscale};
    a.n_levels = n_levels; a.K = K; a.dist_thr = dist_thr; a.cos_thr = cos_thr; a.damping = damping;
    a.pose = pose; a.stats = stats; a.sums = sc->sums; a.ticket = &sc->ticket; a.abort_flag = &sc->pad[0];
    int n_barriers = 1;
    for (int l = 0; l < n_levels; ++l) n_barriers += lv[l].iters;
    if (n_barriers > 24) return -1;                                        // rows of Scratch::sums
    ICP_TRY(hipMemsetAsync(sc->sums, 0, (size_t)n_barriers * 8 * PSTRIDE * sizeof(double), st));
    static const bool dbg_timing = [] { const char* e = getenv("RTGS_ICP_DEBUG_TIMING"); return e && atoi(e) != 0; }();
    a.dbg = dbg_timing ? sc->dbg : nullptr;
    hipLaunchKernelGGL(icp_track_kernel, dim3(G), dim3(256), 0, st, a);
    // leave the scratch ARMED (ticket = 0, no abort) for a later launch-per-iteration track that skips its memsets
    ICP_TRY(hipMemsetAsync(&sc->ticket, 0, 2 * sizeof(uint32_t), st));
    ICP_TRY(hipGetLastError());
    return 0;
  }
  // Cluster form (RTGS_ICP_FLAG_CLUSTER / RTGS_ICP_CLUSTER=1): every level but the finest - 10 of the 15 iterations, 51 k
  // and 204 k pixels at 1200x680 - in ONE launch on a cluster of workgroups of one XCD, with a barrier inside the XCD per
  // iteration instead of a kernel boundary, a 50-200-way ticket fan-in and a relaunch; the finest level keeps one launch
  // per iteration (816 k pixels want the whole chip).
  int first_launched = 0;
  if (cluster > 0 && n_levels >= 2) {
    TrackArgs a{};
    int n_barriers = 0;
    for (int l = 0; l < n_levels - 1; ++l) {
      a.lv[l] = TrackLevel{lv[l].vertex_src, lv[l].normal_src, lv[l].vertex_tgt, lv[l].normal_tgt, lv[l].H, lv[l].W,
                           lv[l].iters, lv[l].downscale};
      n_barriers += lv[l].iters;
    }
    if (n_barriers > 24) return -1;
    a.n_levels = n_levels; a.K = K; a.dist_thr = dist_thr; a.cos_thr = cos_thr; a.damping = damping;
    a.pose = pose; a.stats = stats; a.sums = sc->sums; a.ticket = &sc->ticket; a.abort_flag = &sc->pad[0];
    a.dbg = nullptr;
    a.cluster = cluster > 64 ? 64 : cluster; a.levels_here = n_levels - 1;
    ICP_TRY(hipMemsetAsync(sc->sums, 0, (size_t)(n_barriers + 1) * 8 * PSTRIDE * sizeof(double), st));
    hipLaunchKernelGGL(icp_track_kernel, dim3(8 * a.cluster), dim3(256), 0, st, a);
    ICP_TRY(hipMemsetAsync(&sc->ticket, 0, sizeof(uint32_t), st));       // the launches below elect their last arriver from 0
    first_launched = n_levels - 1;
  }
  bool launched_any = false;
  for (int l = first_launched; l < n_levels; ++l) {
    const rtgs_icp_level& L = lv[l];
    // a lane takes FOUR pixels (accumulate_range): one workgroup per 1024 pixels - at the coarse levels that is 50 / 200
    // workgroups instead of 200 / 512 with three lanes in four idle, and as many fewer tickets and partial rows for the
    // last arriver to collect
    const int g = grid_for((L.H * L.W + 3) / 4);
    const float inv = 1.f / ((float)L.H * (float)L.W);
    FinalArgs fa{(int)MODE_SOLVE, damping, inv, pose, stats, nullptr, nullptr, nullptr, 0, (flags & RTGS_ICP_FLAG_F32_SOLVE) ? 1 : 0};
    for (int it = 0; it < L.iters; ++it) {   // ONE launch per Gauss-Newton iteration: residuals + solve + pose update
      fa.first = (plain && !launched_any) ? (1 | (from_identity ? 2 : 0)) : 0;
      launched_any = true;
      hipLaunchKernelGGL(icp_reduce_kernel, dim3(g), dim3(256), 0, st, L.vertex_src, L.normal_src, L.vertex_tgt,
                         L.normal_tgt, L.H, L.W, K, L.downscale, (const float*)pose, dist_thr, cos_thr, sc->partials,
                         &sc->ticket, fa);
    }
  }
  if (plain && !launched_any) {                    // no iteration anywhere: nobody cleared / initialised anything
    ICP_TRY(hipMemsetAsync(stats, 0, 4 * sizeof(float), st));
    if (from_identity) {
      static const float eye0[16] = {1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f, 0.f, 0.f, 0.f, 0.f, 1.f};
      ICP_TRY(hipMemcpyAsync(pose, eye0, sizeof(eye0), hipMemcpyHostToDevice, st));
    }
  }
  const rtgs_icp_level& F = lv[n_levels - 1];
  const int n = F.H * F.W;
  const int g = grid_for(n);
  FinalArgs fp{(int)MODE_P2P, 0.f, 1.f / (float)n, pose, stats, nullptr, nullptr, nullptr, 0, 0};
  hipLaunchKernelGGL(icp_p2p_kernel, dim3(g), dim3(256), 0, st, F.vertex_src, F.vertex_tgt, F.normal_tgt, n,
                     (const float*)pose, sc->partials, &sc->ticket, fp);
  ICP_TRY(hipGetLastError());
  return 0;
}

int rtgs_icp_fill_model_depth(float* rd, const float* fd, const float* rn, const float* fn, int32_t H, int32_t W,
                              float dist_thr, float normal_thr, void* stream) {
  if (!rd || !fd || !rn || !fn || H <= 0 || W <= 0) return -1;
  const int n = H * W;
  hipLaunchKernelGGL(icp_fill_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, rd, fd, rn, fn, n,
                     dist_thr, normal_thr);
  ICP_TRY(hipGetLastError());
  return 0;
}

}  // extern "C"
