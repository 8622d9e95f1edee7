// Map-state kernels around the rasterizer (SURVEY.md §8f "next-2"): the activations that turn the
// raw map parameters into the rasterizer's inputs, and their backward, each as ONE streaming
// kernel instead of ~40 elementwise / gather launches (SLAM/gaussian_pointcloud.py:16-25 exp /
// sigmoid / normalize, :538-550 get_normal).
#include "../../include/rtgs_raster.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "adam_common.h"
#include "activate_common.h"


// ---------------------------------------------------------------------------------------------
// Block-SoA variant used by the optimisation step: the map keeps xyz [N,3] and SH [N,48] as the
// rasterizer reads them (no activation, no copy) and only the 8 raw columns
// raw8 = (opacity | scaling xyz | rotation wxyz) go through an activation kernel:
//   forward : raw8 -> opacity[N,1], scales[N,3], rotations[N,4], normal[N,3]   (32 B in, 44 B out)
//   backward: gradients of those four -> g_raw8[N,8]
// ---------------------------------------------------------------------------------------------
namespace rtgs {

__global__ void __launch_bounds__(256) activate8_fwd_kernel(const float4* __restrict__ raw8, int64_t n,
                                                            float* __restrict__ opacity, float* __restrict__ scales,
                                                            float4* __restrict__ rots, float* __restrict__ normal) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    activate8_row_store(raw8[2 * i], raw8[2 * i + 1], i, opacity, scales, rots, normal);   // (o, s0, s1, s2), (qw, qx, qy, qz)
}

__global__ void __launch_bounds__(256) activate8_bwd_kernel(const float4* __restrict__ raw8, int64_t n,
                                                            const float* __restrict__ g_op, const float* __restrict__ g_sc,
                                                            const float4* __restrict__ g_rot, const float* __restrict__ g_nrm,
                                                            const uint8_t* __restrict__ row_state,
                                                            float4* __restrict__ g_raw8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    if (row_state) {     // rtgs_raster_backward_rows protocol: 0 = row already zero, 2 = zeroed upstream by this pass
      const uint8_t s = row_state[i];
      if (s == 0) continue;
      if (s == 2) { g_raw8[2 * i] = g_raw8[2 * i + 1] = make_float4(0.f, 0.f, 0.f, 0.f); continue; }
    }
    float4 lo, hi;
    activate8_bwd_row(raw8[2 * i], raw8[2 * i + 1], g_op[i], g_sc[i * 3], g_sc[i * 3 + 1], g_sc[i * 3 + 2], g_rot[i],
                      g_nrm[i * 3], g_nrm[i * 3 + 1], g_nrm[i * 3 + 2], lo, hi);
    g_raw8[2 * i] = lo;
    g_raw8[2 * i + 1] = hi;
  }
}

}  // namespace rtgs

extern "C" int rtgs_map_activate8_forward(const float* raw8, int64_t n, float* opacity, float* scales, float* rotations,
                                          float* normal, void* stream) {
  if (n < 0 || (n > 0 && (!raw8 || !opacity || !scales || !rotations || !normal))) return -1;
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::activate8_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)raw8, n, opacity, scales, (float4*)rotations, normal);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_map_activate8_backward(const float* raw8, int64_t n, const float* g_opacity, const float* g_scales,
                                           const float* g_rotations, const float* g_normal, float* g_raw8, void* stream) {
  if (n < 0 || (n > 0 && (!raw8 || !g_opacity || !g_scales || !g_rotations || !g_normal || !g_raw8))) return -1;
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::activate8_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)raw8, n, g_opacity, g_scales, (const float4*)g_rotations, g_normal,
                     (const uint8_t*)nullptr, (float4*)g_raw8);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_map_activate8_backward_rows(const float* raw8, int64_t n, const float* g_opacity, const float* g_scales,
                                                const float* g_rotations, const float* g_normal, const uint8_t* row_state,
                                                float* g_raw8, void* stream) {
  if (n < 0 || (n > 0 && (!raw8 || !g_opacity || !g_scales || !g_rotations || !g_normal || !row_state || !g_raw8))) return -1;
  if (n == 0) return 0;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::activate8_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)raw8, n, g_opacity, g_scales, (const float4*)g_rotations, g_normal, row_state,
                     (float4*)g_raw8);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Tail of the one-call map step (rtgs_slam_map_step): activation backward + Adam on xyz, SH and raw8 in ONE launch
// instead of four (each of which is latency-bound when only a few thousand rows carry gradient).  Same arithmetic
// as activate8_bwd_kernel and fused_adam_rows_kernel, same row-state protocol: state 1 rows are computed, state 2
// rows of g_raw8 are zeroed, a row is stepped iff it carries gradient now or its moments have ever left zero.
// ---------------------------------------------------------------------------------------------
namespace rtgs {

struct TailArgs {
  const float4* raw8_in;          // = raw8 (read before it is stepped)
  float *xyz, *shs, *raw8;
  const float *g_op, *g_sc, *g_nrm, *g_xyz, *g_shs;
  const float4* g_rot;
  float4* g_raw8;
  const uint8_t* row_state;
  float *m_xyz, *v_xyz, *m_shs, *v_shs, *m_raw8, *v_raw8;
  const float *lr_xyz, *lr_shs, *lr_raw8;
  uint8_t *ever_xyz, *ever_shs, *ever_raw8;
  long long rows;
  float beta1, beta2, eps, bc1, bc2_sqrt;
  // attach regulariser (mapper.py:384-401) and confidence increment (:454-456), both optional
  const float* init_xyz;
  const float4* init_raw8;
  const float* attach_info;       // [0] = number of selected rows
  float* confidence;
  const uint32_t* skip_flag;      // multi-GPU: non-zero = the gradient exchange of this step overflowed, do nothing
  // optional: the caller's activated copies of raw8, re-activated for every row whose raw8 is stepped
  float *act_opacity, *act_scales, *act_normal;
  float4* act_rots;
};

__device__ __forceinline__ bool attach_selected(const float4 init_lo) {
  return 1.f / (1.f + __expf(-init_lo.x)) < 0.9f;                // opacity_activation(init_stat["opacity"]) < 0.9
}

__global__ void __launch_bounds__(256) map_tail_rows_kernel(TailArgs a) {
  __shared__ int s_rows[4][64];
  if (a.skip_flag && a.skip_flag[0] != 0u) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wave0 = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
  for (long long base = wave0 * 64; base < a.rows; base += nwaves * 64) {
    const long long r = base + lane;
    bool need_sh = false, has_grad = false;
    if (r < a.rows) {
      // the four flag bytes of the row up front: independent loads, one memory latency instead of four dependent ones
      // (on a depth-complex map 99.6 % of the rows end here)
      const uint8_t st = a.row_state[r];
      const uint8_t e_raw8 = a.ever_raw8[r], e_xyz = a.ever_xyz[r], e_shs = a.ever_shs[r];
      const bool rgrad = st == 1;                                // gradient from the rasterizer
      has_grad = rgrad;
      float4 at_lo = make_float4(0.f, 0.f, 0.f, 0.f), at_hi = at_lo;
      float at_x = 0.f, at_y = 0.f, at_z = 0.f;
      bool sel = false;
      // a row that was never stepped still equals its snapshot: its attach gradient is exactly 0 and it is skipped
      // without reading the snapshot (the common case: most low-opacity Gaussians never see a render gradient)
      if (a.init_raw8 && (rgrad || e_raw8 != 0 || e_xyz != 0)) {
        const float4 i_lo = a.init_raw8[2 * r];
        sel = attach_selected(i_lo);
        if (sel) {
          // d/dp of 1000 * mean_{selected rows x cols} (p - p0)^2 = 2000 (p - p0) / (n_sel * cols)
          const float4 i_hi = a.init_raw8[2 * r + 1], c_lo = a.raw8_in[2 * r], c_hi = a.raw8_in[2 * r + 1];
          const float n = fmaxf(a.attach_info[0], 1.f);
          const float k3 = 2000.f / (n * 3.f), k4 = 2000.f / (n * 4.f);
          at_lo = make_float4(0.f, k3 * (c_lo.y - i_lo.y), k3 * (c_lo.z - i_lo.z), k3 * (c_lo.w - i_lo.w));
          at_hi = make_float4(k4 * (c_hi.x - i_hi.x), k4 * (c_hi.y - i_hi.y), k4 * (c_hi.z - i_hi.z), k4 * (c_hi.w - i_hi.w));
          at_x = k3 * (a.xyz[(size_t)r * 3] - a.init_xyz[(size_t)r * 3]);
          at_y = k3 * (a.xyz[(size_t)r * 3 + 1] - a.init_xyz[(size_t)r * 3 + 1]);
          at_z = k3 * (a.xyz[(size_t)r * 3 + 2] - a.init_xyz[(size_t)r * 3 + 2]);
        }
      }
      const bool grad = rgrad || sel;
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;      // (a row in state 0 without attach term has a zero gradient row)
      if (st != 0 || sel) {                                      // raw8 gradient row: value (1 / attached) or zero (2)
        if (rgrad)
          activate8_bwd_row(a.raw8_in[2 * r], a.raw8_in[2 * r + 1], a.g_op[r], a.g_sc[r * 3], a.g_sc[r * 3 + 1],
                            a.g_sc[r * 3 + 2], a.g_rot[r], a.g_nrm[r * 3], a.g_nrm[r * 3 + 1], a.g_nrm[r * 3 + 2], lo, hi);
        lo.y += at_lo.y; lo.z += at_lo.z; lo.w += at_lo.w;
        hi.x += at_hi.x; hi.y += at_hi.y; hi.z += at_hi.z; hi.w += at_hi.w;
        a.g_raw8[2 * r] = lo; a.g_raw8[2 * r + 1] = hi;
      }
      if (grad || e_raw8 != 0) {                                 // raw8: 8 columns, this lane
        if (e_raw8 == 0) a.ever_raw8[r] = 1;
        const float4 g0 = lo, g1 = hi;                           // what was just stored - or the all-zero row of state 0
        float4* p4 = reinterpret_cast<float4*>(a.raw8) + 2 * r;
        float4* m4 = reinterpret_cast<float4*>(a.m_raw8) + 2 * r;
        float4* v4 = reinterpret_cast<float4*>(a.v_raw8) + 2 * r;
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        float4 pp[2] = {p4[0], p4[1]}, mm[2] = {m4[0], m4[1]}, vv[2] = {v4[0], v4[1]};
        float* pf = reinterpret_cast<float*>(pp); float* mf = reinterpret_cast<float*>(mm); float* vf = reinterpret_cast<float*>(vv);
#pragma unroll
        for (int c = 0; c < 8; ++c) pf[c] = adam1(pf[c], gg[c], mf[c], vf[c], a.lr_raw8[c], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
        p4[0] = pp[0]; p4[1] = pp[1]; m4[0] = mm[0]; m4[1] = mm[1]; v4[0] = vv[0]; v4[1] = vv[1];
        // keep the caller's activated arrays in step with the rows that moved: no full activation pass next iteration
        if (a.act_opacity) activate8_row_store(pp[0], pp[1], r, a.act_opacity, a.act_scales, a.act_rots, a.act_normal);
      }
      if (grad || e_xyz != 0) {                                  // xyz: 3 columns, this lane
        if (e_xyz == 0) a.ever_xyz[r] = 1;
        const float at[3] = {at_x, at_y, at_z};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const size_t o = (size_t)r * 3 + c;
          float mi = a.m_xyz[o], vi = a.v_xyz[o];
          a.xyz[o] = adam1(a.xyz[o], (rgrad ? a.g_xyz[o] : 0.f) + at[c], mi, vi, a.lr_xyz[c], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
          a.m_xyz[o] = mi; a.v_xyz[o] = vi;
        }
      }
      if (a.confidence && rgrad) {                               // confidence += 1 where the f_dc gradient is non-zero
        const size_t o = (size_t)r * 48;
        if (a.g_shs[o] != 0.f || a.g_shs[o + 1] != 0.f || a.g_shs[o + 2] != 0.f) a.confidence[r] += 1.f;
      }
      need_sh = rgrad || e_shs != 0;
      if (need_sh && e_shs == 0) a.ever_shs[r] = 1;
    }
    // SH: 48 columns = 12 lanes x float4 per live row, five rows per sweep (as fused_adam_rows_kernel<48>)
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(need_sh);
    if (mask == 0ull) continue;
    const int n = __popcll(mask);
    if (need_sh) s_rows[wv][__popcll(mask & ((1ull << lane) - 1ull))] = lane | (has_grad ? 64 : 0);
    __builtin_amdgcn_wave_barrier();
    const int slot = lane / 12, sub = lane - slot * 12;
    for (int k0 = 0; k0 < n; k0 += 5) {
      const int k = k0 + slot;
      if (slot < 5 && k < n) {
        const int rk = s_rows[wv][k];                            // lane of the row | 64 if it carries a gradient
        const size_t o = (size_t)(base + (rk & 63)) * 12 + sub;
        // a row without gradient (state 0 / 2) is all-zero by the arena's invariant: not read
        const float4 gi = (rk & 64) ? reinterpret_cast<const float4*>(a.g_shs)[o] : make_float4(0.f, 0.f, 0.f, 0.f);
        const float4 pi = reinterpret_cast<const float4*>(a.shs)[o];
        float4 mi = reinterpret_cast<float4*>(a.m_shs)[o], vi = reinterpret_cast<float4*>(a.v_shs)[o], po;
        po.x = adam1(pi.x, gi.x, mi.x, vi.x, a.lr_shs[4 * sub], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
        po.y = adam1(pi.y, gi.y, mi.y, vi.y, a.lr_shs[4 * sub + 1], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
        po.z = adam1(pi.z, gi.z, mi.z, vi.z, a.lr_shs[4 * sub + 2], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
        po.w = adam1(pi.w, gi.w, mi.w, vi.w, a.lr_shs[4 * sub + 3], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
        reinterpret_cast<float4*>(a.shs)[o] = po;
        reinterpret_cast<float4*>(a.m_shs)[o] = mi;
        reinterpret_cast<float4*>(a.v_shs)[o] = vi;
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}

}  // namespace rtgs

namespace rtgs {
// number of attach-selected rows and the value of the regulariser: info[0] += n, info[1] = loss (second launch)
__global__ void __launch_bounds__(256) attach_sums_kernel(const float* __restrict__ xyz, const float4* __restrict__ raw8,
                                                          const float* __restrict__ init_xyz, const float4* __restrict__ init_raw8,
                                                          long long rows, float* __restrict__ sums4) {
  float n = 0.f, ss = 0.f, sx = 0.f, sr = 0.f;
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (long long)gridDim.x * blockDim.x) {
    const float4 i_lo = init_raw8[2 * r];
    if (!attach_selected(i_lo)) continue;
    const float4 i_hi = init_raw8[2 * r + 1], c_lo = raw8[2 * r], c_hi = raw8[2 * r + 1];
    n += 1.f;
    ss += (c_lo.y - i_lo.y) * (c_lo.y - i_lo.y) + (c_lo.z - i_lo.z) * (c_lo.z - i_lo.z) + (c_lo.w - i_lo.w) * (c_lo.w - i_lo.w);
    sr += (c_hi.x - i_hi.x) * (c_hi.x - i_hi.x) + (c_hi.y - i_hi.y) * (c_hi.y - i_hi.y) + (c_hi.z - i_hi.z) * (c_hi.z - i_hi.z) +
          (c_hi.w - i_hi.w) * (c_hi.w - i_hi.w);
#pragma unroll
    for (int c = 0; c < 3; ++c) { const float d = xyz[(size_t)r * 3 + c] - init_xyz[(size_t)r * 3 + c]; sx += d * d; }
  }
  float v[4] = {n, ss, sx, sr};
  __shared__ float sh[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_xor(v[k], off);
    if ((threadIdx.x & 63) == 0) sh[k][threadIdx.x >> 6] = v[k];
  }
  __syncthreads();
  if (threadIdx.x < 4) unsafeAtomicAdd(&sums4[threadIdx.x], (sh[threadIdx.x][0] + sh[threadIdx.x][1]) + (sh[threadIdx.x][2] + sh[threadIdx.x][3]));
}
__global__ void attach_finish_kernel(const float* __restrict__ sums4, float* __restrict__ info) {
  const float n = sums4[0];
  info[0] = n;
  info[1] = n > 0.f ? 1000.f * (sums4[1] / (n * 3.f) + sums4[2] / (n * 3.f) + sums4[3] / (n * 4.f)) : 0.f;
}
}  // namespace rtgs

// ---------------------------------------------------------------------------------------------
// Mapping.history_merge (mapper.py:212-251).  One lane per row for xyz / scaling / rotation, then the wave's rows 12 lanes
// x float4 each for the 48 SH columns (as the Adam tail moves them).
// ---------------------------------------------------------------------------------------------
namespace rtgs {
__global__ void __launch_bounds__(256) history_merge_kernel(float* __restrict__ xyz, float4* __restrict__ shs, float4* __restrict__ raw8,
                                                            const float* __restrict__ t_xyz, const float4* __restrict__ t_shs,
                                                            const float4* __restrict__ t_raw8, const float* __restrict__ c_then,
                                                            const float* __restrict__ c_now, long long rows, float max_weight) {
  const float w0 = max_weight * c_then[0] / (c_now[0] + 1e-6f);          // history_weight[0]: the reference's indexing
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long base = wave0 * 64; base < rows; base += nwaves * 64) {
    const long long r = base + lane;
    if (r < rows) {
      const float w = max_weight * c_then[r] / (c_now[r] + 1e-6f);
#pragma unroll
      for (int c = 0; c < 3; ++c) xyz[r * 3 + c] = t_xyz[r * 3 + c] * w + (1.f - w) * xyz[r * 3 + c];
      float4 lo = raw8[2 * r], hi = raw8[2 * r + 1];
      const float4 tlo = t_raw8[2 * r], thi = t_raw8[2 * r + 1];
      lo.y = tlo.y * w0 + (1.f - w0) * lo.y; lo.z = tlo.z * w0 + (1.f - w0) * lo.z; lo.w = tlo.w * w0 + (1.f - w0) * lo.w;
      // slerp(v0 = get_rotation then, v1 = get_rotation now, t = 1 - w), both normalised (F.normalize: / max(|q|, 1e-12))
      const float n0 = fmaxf(sqrtf(thi.x * thi.x + thi.y * thi.y + thi.z * thi.z + thi.w * thi.w), 1e-12f);
      const float n1 = fmaxf(sqrtf(hi.x * hi.x + hi.y * hi.y + hi.z * hi.z + hi.w * hi.w), 1e-12f);
      const float v0[4] = {thi.x / n0, thi.y / n0, thi.z / n0, thi.w / n0}, v1[4] = {hi.x / n1, hi.y / n1, hi.z / n1, hi.w / n1};
      // slerp normalises its inputs again (utils.py:608-612) before the dot product
      const float m0 = sqrtf(v0[0] * v0[0] + v0[1] * v0[1] + v0[2] * v0[2] + v0[3] * v0[3]);
      const float m1 = sqrtf(v1[0] * v1[0] + v1[1] * v1[1] + v1[2] * v1[2] + v1[3] * v1[3]);
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < 4; ++c) dot += (v0[c] / m0) * (v1[c] / m1);
      const float t = 1.f - w;
      float o[4];
      if (!(fabsf(dot) <= 0.9995f)) {                                    // colinear (or NaN): torch.lerp(v0, v1, t)
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = v0[c] + t * (v1[c] - v0[c]);
      } else {
        const float th0 = acosf(dot), s0n = sinf(th0), tht = th0 * t;
        const float s0 = sinf(th0 - tht) / s0n, s1 = sinf(tht) / s0n;
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = s0 * v0[c] + s1 * v1[c];
      }
      raw8[2 * r] = lo;
      raw8[2 * r + 1] = make_float4(o[0], o[1], o[2], o[3]);
    }
    // SH: 64 rows x 12 float4, five rows per sweep
    const long long nrow = rows - base < 64 ? rows - base : 64;
    const int slot = lane / 12, sub = lane - slot * 12;
    for (long long k0 = 0; k0 < nrow; k0 += 5) {
      const long long k = k0 + slot;
      if (slot < 5 && k < nrow) {
        const size_t o = (size_t)(base + k) * 12 + sub;
        const float4 a = t_shs[o], b = shs[o];
        shs[o] = make_float4(a.x * w0 + (1.f - w0) * b.x, a.y * w0 + (1.f - w0) * b.y, a.z * w0 + (1.f - w0) * b.z,
                             a.w * w0 + (1.f - w0) * b.w);
      }
    }
  }
}
}  // namespace rtgs

extern "C" int rtgs_history_merge(float* xyz, float* shs, float* raw8, const float* then_xyz, const float* then_shs,
                                  const float* then_raw8, const float* conf_then, const float* conf_now, int64_t rows,
                                  float max_weight, void* stream) {
  if (rows < 0) return -1;
  if (rows == 0 || !(max_weight > 0.f)) return 0;
  if (!xyz || !shs || !raw8 || !then_xyz || !then_shs || !then_raw8 || !conf_then || !conf_now) return -1;
  long long blocks = (rows + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::history_merge_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, xyz, (float4*)shs,
                     (float4*)raw8, then_xyz, (const float4*)then_shs, (const float4*)then_raw8, conf_then, conf_now,
                     (long long)rows, max_weight);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// attach->attach_info must point at 6 floats: [0] n_selected, [1] loss, [2..5] scratch sums
extern "C" int rtgs_attach_prepare(const float* xyz, const float* raw8, const rtgs_attach* attach, int64_t rows, void* stream) {
  if (!attach || !attach->init_xyz || !attach->init_raw8 || !attach->attach_info || rows < 0) return -1;
  if (rows > 0 && (!xyz || !raw8)) return -1;
  hipStream_t st = (hipStream_t)stream;
  float* info = attach->attach_info;
  if (hipMemsetAsync(info, 0, 6 * sizeof(float), st) != hipSuccess) return -2;
  if (rows > 0) {
    long long blocks = (rows + 255) / 256;
    if (blocks > 192) blocks = 192;
    hipLaunchKernelGGL(rtgs::attach_sums_kernel, dim3((unsigned)blocks), dim3(256), 0, st, xyz, (const float4*)raw8,
                       attach->init_xyz, (const float4*)attach->init_raw8, (long long)rows, info + 2);
  }
  hipLaunchKernelGGL(rtgs::attach_finish_kernel, dim3(1), dim3(1), 0, st, (const float*)(info + 2), info);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_map_tail_rows(float* xyz, float* shs, float* raw8, const float* g_opacity, const float* g_scales,
                                  const float* g_rotations, const float* g_normal, const float* g_xyz, const float* g_shs,
                                  float* g_raw8, const uint8_t* row_state, float* m_xyz, float* v_xyz, float* m_shs,
                                  float* v_shs, float* m_raw8, float* v_raw8, const float* lr_xyz, const float* lr_shs,
                                  const float* lr_raw8, uint8_t* ever_xyz, uint8_t* ever_shs, uint8_t* ever_raw8,
                                  int64_t rows, int32_t step, float beta1, float beta2, float eps,
                                  const rtgs_attach* attach, float* confidence, const uint32_t* skip_flag,
                                  const rtgs_activated* refresh, void* stream) {
  if (rows < 0 || step < 1) return -1;
  if (rows == 0) return 0;
  if (!xyz || !shs || !raw8 || !g_opacity || !g_scales || !g_rotations || !g_normal || !g_xyz || !g_shs || !g_raw8 ||
      !row_state || !m_xyz || !v_xyz || !m_shs || !v_shs || !m_raw8 || !v_raw8 || !lr_xyz || !lr_shs || !lr_raw8 ||
      !ever_xyz || !ever_shs || !ever_raw8)
    return -1;
  rtgs::TailArgs a;
  a.raw8_in = (const float4*)raw8; a.xyz = xyz; a.shs = shs; a.raw8 = raw8;
  a.g_op = g_opacity; a.g_sc = g_scales; a.g_nrm = g_normal; a.g_xyz = g_xyz; a.g_shs = g_shs;
  a.g_rot = (const float4*)g_rotations; a.g_raw8 = (float4*)g_raw8; a.row_state = row_state;
  a.m_xyz = m_xyz; a.v_xyz = v_xyz; a.m_shs = m_shs; a.v_shs = v_shs; a.m_raw8 = m_raw8; a.v_raw8 = v_raw8;
  a.lr_xyz = lr_xyz; a.lr_shs = lr_shs; a.lr_raw8 = lr_raw8;
  a.ever_xyz = ever_xyz; a.ever_shs = ever_shs; a.ever_raw8 = ever_raw8;
  a.rows = rows; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
  a.init_xyz = nullptr; a.init_raw8 = nullptr; a.attach_info = nullptr; a.confidence = confidence;
  a.skip_flag = skip_flag;
  a.act_opacity = a.act_scales = a.act_normal = nullptr; a.act_rots = nullptr;
  if (refresh) {
    if (!refresh->opacity || !refresh->scales || !refresh->rotations || !refresh->normal) return -1;
    a.act_opacity = refresh->opacity; a.act_scales = refresh->scales; a.act_normal = refresh->normal;
    a.act_rots = (float4*)refresh->rotations;
  }
  if (attach) {
    if (!attach->init_xyz || !attach->init_raw8 || !attach->attach_info) return -1;
    a.init_xyz = attach->init_xyz; a.init_raw8 = (const float4*)attach->init_raw8; a.attach_info = attach->attach_info;
  }
  a.bc1 = 1.f - powf(beta1, (float)step);
  a.bc2_sqrt = sqrtf(1.f - powf(beta2, (float)step));
  long long blocks = (rows + 255) / 256;
  if (blocks > 16384) blocks = 16384;
  hipLaunchKernelGGL(rtgs::map_tail_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Sparse gradient exchange of the multi-GPU map step (map_optim.ShardedMapOptimizer.step_slam with more than one
// rank): only the rows that received gradient travel, in fixed-capacity lists whose length rides IN BAND - the host
// never reads a count before it launches the collective.  A list is [1 + capacity] rows of 64 words:
//   row 0 (header)  [0] number of rows the sender HAD (may exceed the capacity: the list then holds the first
//                       `capacity` of them and is useless - see overflow below)   [1] capacity
//   data row        [0] Gaussian id  [1..3] d_xyz  [4..51] d_shs  [52] d_opacity  [53..55] d_scales  [56..59] d_rotations
//                   [60..62] d_normal  [63] unused
// rows_pack compacts the state-1 rows of the row-state arena (ids staged in LDS, one global atomic per workgroup);
// rows_overflow looks at the W gathered headers and raises a device flag if any sender overflowed; rows_apply zeroes
// (mode 0) or adds (mode 1, also sets the row's state to 1) a list into an arena - and does nothing when the flag is
// up, as does the Adam tail: an overflowing step leaves parameters, arena and states untouched on every rank (all see
// the same headers), and the host - which learns of it one step later, from an asynchronous copy - repeats the
// exchange with a larger capacity.  Every rank zeroes its own rows and then adds the lists of ranks 0..W-1 in that
// order, so all replicas sum in the same order and stay bit-identical.
// ---------------------------------------------------------------------------------------------
namespace rtgs {

constexpr int PACK_CHUNK = 8192;
constexpr int ROW_WORDS = 64;

struct ArenaPtrs {
  float *d_xyz, *d_shs, *d_opac, *d_scales, *d_rots, *d_normal;
};
__device__ __forceinline__ float* arena_word(const ArenaPtrs& a, uint32_t id, int w) {
  // w in [1, 62]
  if (w < 4) return a.d_xyz + (size_t)id * 3 + (w - 1);
  if (w < 52) return a.d_shs + (size_t)id * 48 + (w - 4);
  if (w < 53) return a.d_opac + id;
  if (w < 56) return a.d_scales + (size_t)id * 3 + (w - 53);
  if (w < 60) return a.d_rots + (size_t)id * 4 + (w - 56);
  return a.d_normal + (size_t)id * 3 + (w - 60);
}

__global__ void __launch_bounds__(256) rows_pack_kernel(const uint8_t* __restrict__ row_state, int P, ArenaPtrs a,
                                                        float* __restrict__ list, uint32_t capacity,
                                                        uint32_t* __restrict__ out_count) {
  __shared__ uint32_t s_ids[PACK_CHUNK];
  __shared__ uint32_t s_n, s_base;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int begin = blockIdx.x * PACK_CHUNK;
  for (int i = begin + (int)threadIdx.x; i < begin + PACK_CHUNK; i += 256) {     // uniform trip count per wave
    const bool in = i < P && row_state[i] == 1;
    const unsigned long long mk = __builtin_amdgcn_ballot_w64(in);
    if (mk == 0ull) continue;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(mk));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (in) s_ids[base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = (uint32_t)i;
  }
  __syncthreads();
  const uint32_t n = s_n;
  if (n == 0u) return;
  if (threadIdx.x == 0) s_base = atomicAdd(out_count, n);
  __syncthreads();
  const uint32_t base = s_base;
  const int w = threadIdx.x & 63;
  for (uint32_t k = threadIdx.x >> 6; k < n; k += 4) {                             // one wave per row, one word per lane
    if (base + k >= capacity) break;                                               // overflow: the header will tell
    const uint32_t id = s_ids[k];
    float v = 0.f;
    if (w == 0) v = __uint_as_float(id);
    else if (w < 63) v = *arena_word(a, id, w);
    list[(size_t)(1 + base + k) * ROW_WORDS + w] = v;
  }
}
__global__ void rows_header_kernel(float* __restrict__ list, const uint32_t* __restrict__ count, uint32_t capacity) {
  list[0] = __uint_as_float(count[0]);
  list[1] = __uint_as_float(capacity);
}

// flag_out[0] = 1 if any of the W gathered lists overflowed its capacity, else 0; flag_out[1 + r] = count of rank r
__global__ void rows_overflow_kernel(const float* __restrict__ gathered, int W, uint32_t capacity, uint32_t* __restrict__ flag_out) {
  uint32_t over = 0;
  for (int r = 0; r < W; ++r) {
    const uint32_t c = __float_as_uint(gathered[(size_t)r * (1 + capacity) * ROW_WORDS]);
    flag_out[1 + r] = c;
    over |= c > capacity ? 1u : 0u;
  }
  flag_out[0] = over;
}

__global__ void __launch_bounds__(256) rows_apply_kernel(const float* __restrict__ list, uint32_t capacity, int mode, ArenaPtrs a,
                                                         uint8_t* __restrict__ row_state, const uint32_t* __restrict__ skip_flag) {
  if (skip_flag && skip_flag[0] != 0u) return;
  const uint32_t n = min(__float_as_uint(list[0]), capacity);
  const float* rows = list + ROW_WORDS;
  const int w = threadIdx.x & 63;
  for (uint32_t k = blockIdx.x * 4 + (threadIdx.x >> 6); k < n; k += gridDim.x * 4) {
    const uint32_t id = __float_as_uint(rows[(size_t)k * ROW_WORDS]);
    if (w == 0) { if (mode == 1) row_state[id] = 1; }
    else if (w < 63) {
      float* dst = arena_word(a, id, w);
      if (mode == 0) *dst = 0.f;
      else *dst += rows[(size_t)k * ROW_WORDS + w];          // ids are unique within one list: no atomics
    }
  }
}

}  // namespace rtgs

extern "C" int rtgs_rows_pack(const uint8_t* row_state, int32_t P, float* d_xyz, float* d_shs, float* d_opacity,
                              float* d_scales, float* d_rotations, float* d_normal, float* out_list, int32_t capacity,
                              uint32_t* count_scratch, void* stream) {
  if (P < 0 || capacity < 1 || !out_list || !count_scratch) return -1;
  if (P > 0 && (!row_state || !d_xyz || !d_shs || !d_opacity || !d_scales || !d_rotations || !d_normal)) return -1;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(count_scratch, 0, sizeof(uint32_t), st) != hipSuccess) return -2;
  if (P > 0) {
    const rtgs::ArenaPtrs a{d_xyz, d_shs, d_opacity, d_scales, d_rotations, d_normal};
    hipLaunchKernelGGL(rtgs::rows_pack_kernel, dim3((P + rtgs::PACK_CHUNK - 1) / rtgs::PACK_CHUNK), dim3(256), 0, st,
                       row_state, P, a, out_list, (uint32_t)capacity, count_scratch);
  }
  hipLaunchKernelGGL(rtgs::rows_header_kernel, dim3(1), dim3(1), 0, st, out_list, (const uint32_t*)count_scratch,
                     (uint32_t)capacity);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_rows_overflow(const float* gathered, int32_t world, int32_t capacity, uint32_t* flag_and_counts,
                                  void* stream) {
  if (!gathered || world < 1 || world > 64 || capacity < 1 || !flag_and_counts) return -1;
  hipLaunchKernelGGL(rtgs::rows_overflow_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, gathered, world, (uint32_t)capacity,
                     flag_and_counts);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_rows_apply(const float* list, int32_t capacity, int32_t mode, float* d_xyz, float* d_shs,
                               float* d_opacity, float* d_scales, float* d_rotations, float* d_normal, uint8_t* row_state,
                               const uint32_t* skip_flag, void* stream) {
  if (capacity < 1 || (mode != 0 && mode != 1)) return -1;
  if (!list || !d_xyz || !d_shs || !d_opacity || !d_scales || !d_rotations || !d_normal || !row_state) return -1;
  const rtgs::ArenaPtrs a{d_xyz, d_shs, d_opacity, d_scales, d_rotations, d_normal};
  int blocks = (capacity + 3) / 4;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rtgs::rows_apply_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, list, (uint32_t)capacity, mode,
                     a, row_state, skip_flag);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Fused SLAM loss: the image terms of Mapping.loss_update (mapper.py:402-448), value and both image gradients.
//   mask       = render_mask, or every pixel when it is NULL - and only then the SSIM term is live (:411-417)
//   colour     = mean over mask (x 3 channels) of |C - C_gt|                                          (:421)
//   depth      = mean over {depth_index != -1, D_gt > 0, (D - D_gt) < add_depth_thres, mask} of |D - D_gt|  (:423-431)
//                (the threshold is on the SIGNED error, as written there; an empty set contributes 0 - torch's mean
//                 of an empty selection is nan)
//   ssim       = 1 - mean(ssim_map), 11x11 Gaussian window sigma 1.5, zero padding (utils/loss_utils.py:58-100)
//   total      = depth_weight depth + color_weight colour + ssim_weight ssim
// (normal_weight is 0 in every shipped config, configs/base.yaml:81: the normal term is its own pair of kernels below.)
// Launches: block partial sums -> a few device atomics; [SSIM forward: per-pixel statistics and the three derivative
// maps]; gradients (+ SSIM backward: the derivative maps convolved with the same window).
// ---------------------------------------------------------------------------------------------
namespace rtgs {

__device__ __forceinline__ float wave_sum_shfl(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// sums: [0] sum |dC| over mask  [1] sum |dD| over valid  [2] #valid  [3] #mask pixels  [4] sum of the SSIM map
// V pixels per lane and step: V = 4 reads every plane with 16-B loads (the image must then hold a multiple of four
// pixels); V = 1 is the general form.  The pixel order of the per-lane sums differs between the two, as it does between
// grid sizes: the loss value is a float sum either way.
template <int V> struct PixVec;
template <> struct PixVec<1> {
  float v[1];
  __device__ __forceinline__ void load(const float* p, int64_t i) { v[0] = p[i]; }
  __device__ __forceinline__ void store(float* p, int64_t i) const { p[i] = v[0]; }
};
template <> struct PixVec<4> {
  float v[4];
  __device__ __forceinline__ void load(const float* p, int64_t i) {
    const float4 t = *reinterpret_cast<const float4*>(p + i); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  }
  __device__ __forceinline__ void store(float* p, int64_t i) const { *reinterpret_cast<float4*>(p + i) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <int V>
__device__ __forceinline__ void load_flags(const uint8_t* __restrict__ rmask, const int32_t* __restrict__ didx, int64_t i, bool (&m)[V],
                                           bool (&hit)[V]) {
  if constexpr (V == 4) {
    const uint32_t mm = rmask ? *reinterpret_cast<const uint32_t*>(rmask + i) : 0x01010101u;
    const int4 d = *reinterpret_cast<const int4*>(didx + i);
    m[0] = (mm & 0xffu) != 0; m[1] = (mm & 0xff00u) != 0; m[2] = (mm & 0xff0000u) != 0; m[3] = (mm & 0xff000000u) != 0;
    hit[0] = d.x != -1; hit[1] = d.y != -1; hit[2] = d.z != -1; hit[3] = d.w != -1;
  } else {
    m[0] = !rmask || rmask[i] != 0;
    hit[0] = didx[i] != -1;
  }
}

template <int V>
__global__ void __launch_bounds__(256) slam_loss_sums_kernel(const float* __restrict__ color, const float* __restrict__ depth,
                                                             const int32_t* __restrict__ didx, const float* __restrict__ gt_c,
                                                             const float* __restrict__ gt_d, const uint8_t* __restrict__ rmask,
                                                             int64_t hw, float depth_thr, float* __restrict__ sums) {
  float s_c = 0.f, s_d = 0.f, s_m = 0.f, s_n = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < hw; i += (int64_t)gridDim.x * blockDim.x * V) {
    bool m[V], hit[V];
    load_flags<V>(rmask, didx, i, m, hit);
    PixVec<V> c0, c1, c2, t0, t1, t2, dp, gd;
    c0.load(color, i); c1.load(color + hw, i); c2.load(color + 2 * hw, i);
    t0.load(gt_c, i); t1.load(gt_c + hw, i); t2.load(gt_c + 2 * hw, i);
    dp.load(depth, i); gd.load(gt_d, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      if (!m[k]) continue;
      s_n += 1.f;
      s_c += fabsf(c0.v[k] - t0.v[k]) + fabsf(c1.v[k] - t1.v[k]) + fabsf(c2.v[k] - t2.v[k]);
      const float g = gd.v[k], e = dp.v[k] - g;
      if (hit[k] && g > 0.f && e < depth_thr) { s_d += fabsf(e); s_m += 1.f; }
    }
  }
  s_c = wave_sum_shfl(s_c); s_d = wave_sum_shfl(s_d); s_m = wave_sum_shfl(s_m); s_n = wave_sum_shfl(s_n);
  __shared__ float sh[4][4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (lane == 0) { sh[0][w] = s_c; sh[1][w] = s_d; sh[2][w] = s_m; sh[3][w] = s_n; }
  __syncthreads();
  if (threadIdx.x < 4) {
    const float t = sh[threadIdx.x][0] + sh[threadIdx.x][1] + sh[threadIdx.x][2] + sh[threadIdx.x][3];
    unsafeAtomicAdd(&sums[threadIdx.x], t);
  }
}

struct SsimWin { float g[11]; };
constexpr int SS_R = 5, SS_T = 16 + 2 * SS_R;     // window radius, tile + halo edge

// SSIM forward of one 16x16 tile: windowed means / second moments by a separable pass through LDS, the map value, and
// its derivatives w.r.t. the three windowed quantities that depend on the rendered image:
//   dA = dS/d mu1 (total, through sigma1^2 and sigma12 too), dB = dS/d conv(x1^2), dC = dS/d conv(x1 x2).
__global__ void __launch_bounds__(256) ssim_fwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int H, int W,
                                                       SsimWin win, float* __restrict__ dmaps, float* __restrict__ sums) {
  __shared__ float s_a[SS_T * SS_T], s_b[SS_T * SS_T];
  __shared__ float s_h[5][SS_T * 16];
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int px = blockIdx.x * 16 + lx, py = blockIdx.y * 16 + ly;
  const size_t HW = (size_t)H * W;
  float ssum = 0.f;
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
    for (int q = threadIdx.x; q < SS_T * SS_T; q += 256) {
      const int x = (int)blockIdx.x * 16 - SS_R + q % SS_T, y = (int)blockIdx.y * 16 - SS_R + q / SS_T;
      const bool in = x >= 0 && x < W && y >= 0 && y < H;
      s_a[q] = in ? x1[c * HW + (size_t)y * W + x] : 0.f;
      s_b[q] = in ? x2[c * HW + (size_t)y * W + x] : 0.f;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < SS_T * 16; q += 256) {         // horizontal pass: SS_T rows x 16 columns
      const int r = q / 16, col = q % 16;
      float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
      for (int k = 0; k < 11; ++k) {
        const float a = s_a[r * SS_T + col + k], b = s_b[r * SS_T + col + k], w = win.g[k];
        m1 += w * a; m2 += w * b; e11 += w * (a * a); e22 += w * (b * b); e12 += w * (a * b);
      }
      s_h[0][q] = m1; s_h[1][q] = m2; s_h[2][q] = e11; s_h[3][q] = e22; s_h[4][q] = e12;
    }
    __syncthreads();
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; ++k) {
      const float w = win.g[k];
      const int q = (ly + k) * 16 + lx;
      m1 += w * s_h[0][q]; m2 += w * s_h[1][q]; e11 += w * s_h[2][q]; e22 += w * s_h[3][q]; e12 += w * s_h[4][q];
    }
    if (px < W && py < H) {
      const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
      const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
      const float n1 = 2.f * m1 * m2 + C1, n2 = 2.f * s12 + C2, d1 = m1 * m1 + m2 * m2 + C1, d2 = s1 + s2 + C2;
      const float S = (n1 * n2) / (d1 * d2);
      ssum += S;
      const float dB = -S / d2;                                  // via sigma1^2 in d2
      const float dC = 2.f * n1 / (d1 * d2);                     // via sigma12 in n2
      const float dA = (2.f * m2 * n2 - 2.f * m2 * n1) / (d1 * d2) - S * (2.f * m1 / d1 - 2.f * m1 / d2);
      const size_t o = c * HW + (size_t)py * W + px;
      dmaps[o] = dA; dmaps[3 * HW + o] = dB; dmaps[6 * HW + o] = dC;
    }
  }
  ssum = wave_sum_shfl(ssum);
  __shared__ float s_w[4];
  if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = ssum;
  __syncthreads();
  if (threadIdx.x == 0) unsafeAtomicAdd(&sums[4], (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]));
}

// g_color += k * (conv(dA) + 2 x1 conv(dB) + x2 conv(dC)),  k = ssim_weight * d(1 - mean S)/dS = -ssim_weight / (3 H W)
__global__ void __launch_bounds__(256) ssim_bwd_kernel(const float* __restrict__ x1, const float* __restrict__ x2, int H, int W,
                                                       SsimWin win, const float* __restrict__ dmaps, float k,
                                                       float* __restrict__ g_color) {
  __shared__ float s_t[3][SS_T * SS_T];
  __shared__ float s_h[3][SS_T * 16];
  const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
  const int px = blockIdx.x * 16 + lx, py = blockIdx.y * 16 + ly;
  const size_t HW = (size_t)H * W;
  for (int c = 0; c < 3; ++c) {
    __syncthreads();
    for (int q = threadIdx.x; q < SS_T * SS_T; q += 256) {
      const int x = (int)blockIdx.x * 16 - SS_R + q % SS_T, y = (int)blockIdx.y * 16 - SS_R + q / SS_T;
      const bool in = x >= 0 && x < W && y >= 0 && y < H;
      const size_t o = c * HW + (size_t)y * W + x;
      s_t[0][q] = in ? dmaps[o] : 0.f; s_t[1][q] = in ? dmaps[3 * HW + o] : 0.f; s_t[2][q] = in ? dmaps[6 * HW + o] : 0.f;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < SS_T * 16; q += 256) {
      const int r = q / 16, col = q % 16;
      float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
      for (int t = 0; t < 11; ++t) {
        const float w = win.g[t];
        a += w * s_t[0][r * SS_T + col + t]; b += w * s_t[1][r * SS_T + col + t]; cc += w * s_t[2][r * SS_T + col + t];
      }
      s_h[0][q] = a; s_h[1][q] = b; s_h[2][q] = cc;
    }
    __syncthreads();
    float a = 0.f, b = 0.f, cc = 0.f;
#pragma unroll
    for (int t = 0; t < 11; ++t) {
      const float w = win.g[t];
      const int q = (ly + t) * 16 + lx;
      a += w * s_h[0][q]; b += w * s_h[1][q]; cc += w * s_h[2][q];
    }
    if (px < W && py < H) {
      const size_t o = c * HW + (size_t)py * W + px;
      g_color[o] += k * (a + 2.f * x1[o] * b + x2[o] * cc);
    }
  }
}

// loss4: [0] total [1] colour [2] depth [3] ssim term (1 - mean S, or 0)
template <int V>
__global__ void __launch_bounds__(256) slam_loss_grads_kernel(const float* __restrict__ color, const float* __restrict__ depth,
                                                              const int32_t* __restrict__ didx, const float* __restrict__ gt_c,
                                                              const float* __restrict__ gt_d, const uint8_t* __restrict__ rmask,
                                                              int64_t hw, float cw, float dw, float sw, float depth_thr,
                                                              const float* __restrict__ sums, float* __restrict__ loss4,
                                                              float* __restrict__ g_color, float* __restrict__ g_depth) {
  const float inv_c = 1.f / (3.f * fmaxf(sums[3], 1.f));
  const float inv_m = 1.f / fmaxf(sums[2], 1.f);
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const float lc = sums[0] * inv_c, ld = sums[1] * inv_m, ls = sw != 0.f ? 1.f - sums[4] / (3.f * (float)hw) : 0.f;
    loss4[1] = lc; loss4[2] = ld; loss4[3] = ls;
    loss4[0] = dw * ld + cw * lc + sw * ls;
  }
  const float kc = cw * inv_c, kd = dw * inv_m;
  for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * V; i < hw; i += (int64_t)gridDim.x * blockDim.x * V) {
    bool m[V], hit[V];
    load_flags<V>(rmask, didx, i, m, hit);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      PixVec<V> a, b, o;
      a.load(color + c * hw, i); b.load(gt_c + c * hw, i);
#pragma unroll
      for (int k = 0; k < V; ++k) {
        const float d = a.v[k] - b.v[k];
        o.v[k] = !m[k] ? 0.f : (d > 0.f ? kc : (d < 0.f ? -kc : 0.f));     // sign(0) = 0 as torch.abs' backward
      }
      o.store(g_color + c * hw, i);
    }
    PixVec<V> dp, gt, o;
    dp.load(depth, i); gt.load(gt_d, i);
#pragma unroll
    for (int k = 0; k < V; ++k) {
      const float g = gt.v[k], e = dp.v[k] - g;
      o.v[k] = (m[k] && hit[k] && g > 0.f && e < depth_thr) ? (e > 0.f ? kd : (e < 0.f ? -kd : 0.f)) : 0.f;
    }
    o.store(g_depth, i);
  }
}

// 16-B path: a multiple of four pixels (so every plane starts 16-B aligned relative to its tensor) and aligned bases
static inline bool loss_vec4_ok(int64_t hw, const void* a, const void* b, const void* c, const void* d, const void* e,
                                const void* mask, const void* f, const void* g) {
  if (hw % 4 != 0) return false;
  const void* ptrs[7] = {a, b, c, d, e, f, g};
  for (const void* p : ptrs) if (p && ((uintptr_t)p & 15u) != 0) return false;
  return !mask || ((uintptr_t)mask & 3u) == 0;
}

}  // namespace rtgs

extern "C" size_t rtgs_slam_loss_scratch_bytes(int32_t H, int32_t W, int32_t with_ssim) {
  return 8 * sizeof(float) + (with_ssim ? (size_t)9 * (size_t)H * (size_t)W * sizeof(float) : 0);
}

static void ssim_window(rtgs::SsimWin& win) {
  float gf[11], totf = 0.f;
  for (int i = 0; i < 11; ++i) { gf[i] = (float)exp(-(double)((i - 5) * (i - 5)) / (2.0 * 1.5 * 1.5)); totf += gf[i]; }
  for (int i = 0; i < 11; ++i) win.g[i] = gf[i] / totf;          // torch.Tensor([...]) / sum, in float32 (loss_utils.py:39-46)
}

// First half: the five sums (scratch[0..4]); with the SSIM term live also the per-pixel derivative maps.
extern "C" int rtgs_slam_loss_sums(const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                                   const float* gt_depth, int32_t H, int32_t W, const rtgs_loss_cfg* cfg, void* scratch,
                                   void* stream) {
  if (!color || !depth || !depth_index || !gt_color || !gt_depth || !cfg || !scratch || H <= 0 || W <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int64_t hw = (int64_t)H * W;
  float* sums = (float*)scratch;
  const bool ssim = cfg->render_mask == nullptr && cfg->ssim_weight != 0.f;
  if (!cfg->sums_zeroed && hipMemsetAsync(sums, 0, 8 * sizeof(float), st) != hipSuccess) return -2;
  int64_t blocks = (hw + 255) / 256;
  // the sums kernel ends with four same-address global atomics per workgroup (~20 ns each, serialised): keep it to
  // 192 workgroups (1 024 of them cost 20 us for a 23 MB read)
  if (blocks > 192) blocks = 192;
  if (rtgs::loss_vec4_ok(hw, color, depth, depth_index, gt_color, gt_depth, cfg->render_mask, nullptr, nullptr))
    hipLaunchKernelGGL(rtgs::slam_loss_sums_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, color, depth, depth_index,
                       gt_color, gt_depth, cfg->render_mask, hw, cfg->add_depth_thres, sums);
  else
    hipLaunchKernelGGL(rtgs::slam_loss_sums_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, color, depth, depth_index,
                       gt_color, gt_depth, cfg->render_mask, hw, cfg->add_depth_thres, sums);
  if (ssim) {
    rtgs::SsimWin win;
    ssim_window(win);
    hipLaunchKernelGGL(rtgs::ssim_fwd_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, st, color, gt_color, H, W, win,
                       sums + 8, sums);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// Second half: loss values and both image gradients from the sums in scratch[0..4] - which a multi-GPU caller that
// splits ONE view across ranks all-reduces in between (the normalisers are counts over the whole image).
extern "C" int rtgs_slam_loss_grads(const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                                    const float* gt_depth, int32_t H, int32_t W, const rtgs_loss_cfg* cfg, void* scratch,
                                    float* loss_out4, float* g_color, float* g_depth, void* stream) {
  if (!color || !depth || !depth_index || !gt_color || !gt_depth || !cfg || !scratch || !loss_out4 || !g_color || !g_depth ||
      H <= 0 || W <= 0)
    return -1;
  hipStream_t st = (hipStream_t)stream;
  const int64_t hw = (int64_t)H * W;
  float* sums = (float*)scratch;
  const bool ssim = cfg->render_mask == nullptr && cfg->ssim_weight != 0.f;
  int64_t blocks = (hw + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  if (rtgs::loss_vec4_ok(hw, color, depth, depth_index, gt_color, gt_depth, cfg->render_mask, g_color, g_depth)) {
    blocks = (hw / 4 + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(rtgs::slam_loss_grads_kernel<4>, dim3((unsigned)blocks), dim3(256), 0, st, color, depth, depth_index,
                       gt_color, gt_depth, cfg->render_mask, hw, cfg->color_weight, cfg->depth_weight,
                       ssim ? cfg->ssim_weight : 0.f, cfg->add_depth_thres, (const float*)sums, loss_out4, g_color, g_depth);
  } else {
    hipLaunchKernelGGL(rtgs::slam_loss_grads_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, st, color, depth, depth_index,
                       gt_color, gt_depth, cfg->render_mask, hw, cfg->color_weight, cfg->depth_weight,
                       ssim ? cfg->ssim_weight : 0.f, cfg->add_depth_thres, (const float*)sums, loss_out4, g_color, g_depth);
  }
  if (ssim) {
    rtgs::SsimWin win;
    ssim_window(win);
    hipLaunchKernelGGL(rtgs::ssim_bwd_kernel, dim3((W + 15) / 16, (H + 15) / 16), dim3(256), 0, st, color, gt_color, H, W, win,
                       (const float*)(sums + 8), -cfg->ssim_weight / (3.f * (float)hw), g_color);
  }
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_slam_loss(const float* color, const float* depth, const int32_t* depth_index, const float* gt_color,
                              const float* gt_depth, int32_t H, int32_t W, const rtgs_loss_cfg* cfg, void* scratch,
                              float* loss_out4, float* g_color, float* g_depth, void* stream) {
  int rc = rtgs_slam_loss_sums(color, depth, depth_index, gt_color, gt_depth, H, W, cfg, scratch, stream);
  if (rc != 0) return rc;
  return rtgs_slam_loss_grads(color, depth, depth_index, gt_color, gt_depth, H, W, cfg, scratch, loss_out4, g_color, g_depth,
                              stream);
}


// ---------------------------------------------------------------------------------------------
// The normal term of Mapping.loss_update (mapper.py:433-442): 1 - F.cosine_similarity(render normal, gt normal) averaged
// over {render mask & depth_index != -1 & gt normal not all-zero}; the render normal of a pixel is the world normal of
// the Gaussian that owns its depth (render.py:130-133), so the gradient goes to that Gaussian's d_normal row.
// ---------------------------------------------------------------------------------------------
namespace rtgs {

__device__ __forceinline__ bool normal_pixel(const float* __restrict__ normal_w, const int32_t* __restrict__ didx,
                                             const float* __restrict__ gtn, const uint8_t* __restrict__ mask, int64_t i,
                                             int& owner, float (&n)[3], float (&g)[3]) {
  owner = didx[i];
  if (owner < 0 || (mask && mask[i] == 0)) return false;
  g[0] = gtn[3 * i]; g[1] = gtn[3 * i + 1]; g[2] = gtn[3 * i + 2];
  if (g[0] == 0.f && g[1] == 0.f && g[2] == 0.f) return false;
  n[0] = normal_w[3 * (int64_t)owner]; n[1] = normal_w[3 * (int64_t)owner + 1]; n[2] = normal_w[3 * (int64_t)owner + 2];
  return true;
}

__global__ void __launch_bounds__(256) normal_loss_sums_kernel(const float* __restrict__ normal_w, const int32_t* __restrict__ didx,
                                                               const float* __restrict__ gtn, const uint8_t* __restrict__ mask,
                                                               int64_t hw, float* __restrict__ sums2,
                                                               const uint32_t* __restrict__ skip) {
  if (skip && skip[0] != 0u) return;
  float s = 0.f, c = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    int owner; float n[3], g[3];
    if (!normal_pixel(normal_w, didx, gtn, mask, i, owner, n, g)) continue;
    // F.cosine_similarity(dim=-1, eps=1e-8): x1 . x2 / (max(|x1|, eps) * max(|x2|, eps))
    const float nn = fmaxf(sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), 1e-8f);
    const float gg = fmaxf(sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]), 1e-8f);
    s += 1.f - (n[0] * g[0] + n[1] * g[1] + n[2] * g[2]) / (nn * gg);
    c += 1.f;
  }
  s = wave_sum_shfl(s); c = wave_sum_shfl(c);
  __shared__ float sh[2][4];
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsafeAtomicAdd(&sums2[0], (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]));
    unsafeAtomicAdd(&sums2[1], (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]));
  }
}

__global__ void __launch_bounds__(256) normal_loss_grads_kernel(const float* __restrict__ normal_w, const int32_t* __restrict__ didx,
                                                                const float* __restrict__ gtn, const uint8_t* __restrict__ mask,
                                                                int64_t hw, float weight, const float* __restrict__ sums2,
                                                                float* __restrict__ loss4, float* __restrict__ d_normal,
                                                                uint8_t* __restrict__ row_state,
                                                                const uint32_t* __restrict__ skip, uint32_t t0, uint32_t tn) {
  if (skip && skip[0] != 0u) return;
  const float cnt = fmaxf(sums2[1], 1.f);                         // an empty set gives 0 (nan in the reference)
  if (blockIdx.x == 0 && threadIdx.x == 0) loss4[0] += weight * (sums2[0] / cnt);
  const float k = weight / cnt;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    int owner; float n[3], g[3];
    if (!normal_pixel(normal_w, didx, gtn, mask, i, owner, n, g)) continue;
    if ((uint32_t)owner - t0 >= tn) continue;                     // frozen owner: in the value, no gradient
    const float nr = sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]), gr = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
    const float nn = fmaxf(nr, 1e-8f), gg = fmaxf(gr, 1e-8f);
    const float dot = n[0] * g[0] + n[1] * g[1] + n[2] * g[2];
    // d/dn of -(n . g) / (nn gg), nn = max(|n|, eps): the norm term only where it is not clamped
    const float a = 1.f / (nn * gg), b = nr > 1e-8f ? dot / (nn * nn * nn * gg) : 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) unsafeAtomicAdd(&d_normal[3 * (int64_t)owner + c], -k * (g[c] * a - n[c] * b));
    if (row_state && row_state[owner] != 1) row_state[owner] = 1;   // idempotent; the row was all-zero (arena invariant)
  }
}

}  // namespace rtgs

extern "C" int rtgs_slam_normal_loss_sums(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                                          const uint8_t* render_mask, int32_t H, int32_t W, float* scratch2,
                                          const uint32_t* skip_flag, void* stream) {
  if (!normal_w || !depth_index || !gt_normal || !scratch2 || H <= 0 || W <= 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  const int64_t hw = (int64_t)H * W;
  int blocks = (int)((hw + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  if (hipMemsetAsync(scratch2, 0, 2 * sizeof(float), st) != hipSuccess) return -2;
  hipLaunchKernelGGL(rtgs::normal_loss_sums_kernel, dim3(blocks), dim3(256), 0, st, normal_w, depth_index, gt_normal, render_mask,
                     hw, scratch2, skip_flag);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_slam_normal_loss_grads(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                                           const uint8_t* render_mask, int32_t H, int32_t W, float normal_weight,
                                           const float* sums2, float* loss_out4, float* d_normal, uint8_t* row_state,
                                           const uint32_t* skip_flag, int32_t train_begin, int32_t train_end, void* stream) {
  if (!normal_w || !depth_index || !gt_normal || !sums2 || !loss_out4 || !d_normal || H <= 0 || W <= 0) return -1;
  if (train_begin < 0 || train_end < train_begin) return -1;
  const int64_t hw = (int64_t)H * W;
  int blocks = (int)((hw + 255) / 256);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(rtgs::normal_loss_grads_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, normal_w, depth_index,
                     gt_normal, render_mask, hw, normal_weight, sums2, loss_out4, d_normal, row_state, skip_flag,
                     (uint32_t)train_begin, (uint32_t)(train_end - train_begin));
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_slam_normal_loss_range(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                                           const uint8_t* render_mask, int32_t H, int32_t W, float normal_weight,
                                           float* scratch2, float* loss_out4, float* d_normal, uint8_t* row_state,
                                           const uint32_t* skip_flag, int32_t train_begin, int32_t train_end, void* stream) {
  if (!loss_out4 || !d_normal) return -1;
  const int rc = rtgs_slam_normal_loss_sums(normal_w, depth_index, gt_normal, render_mask, H, W, scratch2, skip_flag, stream);
  if (rc != 0) return rc;
  return rtgs_slam_normal_loss_grads(normal_w, depth_index, gt_normal, render_mask, H, W, normal_weight, scratch2, loss_out4,
                                     d_normal, row_state, skip_flag, train_begin, train_end, stream);
}

extern "C" int rtgs_slam_normal_loss(const float* normal_w, const int32_t* depth_index, const float* gt_normal,
                                     const uint8_t* render_mask, int32_t H, int32_t W, float normal_weight, float* scratch2,
                                     float* loss_out4, float* d_normal, uint8_t* row_state, const uint32_t* skip_flag,
                                     void* stream) {
  return rtgs_slam_normal_loss_range(normal_w, depth_index, gt_normal, render_mask, H, W, normal_weight, scratch2, loss_out4,
                                     d_normal, row_state, skip_flag, 0, 0x7fffffff, stream);
}
