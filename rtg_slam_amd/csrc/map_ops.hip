// Map-state kernels around the rasterizer (SURVEY.md §8f "next-2"): the activations that turn the
// packed raw parameter buffer into the rasterizer's inputs, and their backward, each as ONE
// streaming kernel instead of ~40 elementwise / gather launches.
//   forward : SLAM/gaussian_pointcloud.py:16-25 (exp / sigmoid / normalize), :538-550 (get_normal),
//             :573-577 (get_features)                    packed [N,59] -> six contiguous tensors
//   backward: the chain rule of the above                six gradients -> packed gradient [N,59]
// Packed columns: xyz 0:3 | f_dc 3:6 | f_rest 6:51 | opacity 51 | scaling 52:55 | rotation 55:59.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtgs {

constexpr int COLS = 59;

__device__ __forceinline__ void rot_col(int k, float r, float x, float y, float z, float (&c)[3]) {
  if (k == 0) { c[0] = 1.f - 2.f * (y * y + z * z); c[1] = 2.f * (x * y + r * z); c[2] = 2.f * (x * z - r * y); }
  else if (k == 1) { c[0] = 2.f * (x * y - r * z); c[1] = 1.f - 2.f * (x * x + z * z); c[2] = 2.f * (y * z + r * x); }
  else { c[0] = 2.f * (x * z + r * y); c[1] = 2.f * (y * z - r * x); c[2] = 1.f - 2.f * (x * x + y * y); }
}

__global__ void __launch_bounds__(256) activate_fwd_kernel(const float* __restrict__ packed, int64_t n,
                                                           float* __restrict__ xyz, float* __restrict__ opacity,
                                                           float* __restrict__ shs, float* __restrict__ scales,
                                                           float* __restrict__ rots, float* __restrict__ normal) {
  // phase 1: the 51 pass-through columns (xyz + SH), coalesced over the flattened buffer
  const uint32_t total = (uint32_t)n * COLS;          // launcher guarantees n * 59 < 2^32
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t row = e / COLS;                    // constant divisor: mul-hi, no 64-bit division
    const uint32_t c = e - row * COLS;
    const float v = packed[e];
    if (c < 3) xyz[(size_t)row * 3 + c] = v;
    else if (c < 51) shs[(size_t)row * 48 + (c - 3)] = v;
  }
  // phase 2: one lane per Gaussian for the activated columns
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = packed + i * COLS;
    opacity[i] = 1.f / (1.f + __expf(-p[51]));
    const float s0 = __expf(p[52]), s1 = __expf(p[53]), s2 = __expf(p[54]);
    scales[i * 3] = s0; scales[i * 3 + 1] = s1; scales[i * 3 + 2] = s2;
    const float q0 = p[55], q1 = p[56], q2 = p[57], q3 = p[58];
    const float inv = 1.f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);   // F.normalize eps
    const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
    rots[i * 4] = r; rots[i * 4 + 1] = x; rots[i * 4 + 2] = y; rots[i * 4 + 3] = z;
    int k = 0;                                   // torch.argmin: first minimum
    float sm = s0;
    if (s1 < sm) { sm = s1; k = 1; }
    if (s2 < sm) { k = 2; }
    float c[3];
    rot_col(k, r, x, y, z, c);
    const float m = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + 1e-8f;
    normal[i * 3] = c[0] / m; normal[i * 3 + 1] = c[1] / m; normal[i * 3 + 2] = c[2] / m;
  }
}

__global__ void __launch_bounds__(256) activate_bwd_kernel(const float* __restrict__ packed, int64_t n,
                                                           const float* __restrict__ g_xyz, const float* __restrict__ g_op,
                                                           const float* __restrict__ g_shs, const float* __restrict__ g_sc,
                                                           const float* __restrict__ g_rot, const float* __restrict__ g_nrm,
                                                           float* __restrict__ g_packed) {
  const uint32_t total = (uint32_t)n * COLS;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const uint32_t row = e / COLS;
    const uint32_t c = e - row * COLS;
    if (c < 3) g_packed[e] = g_xyz[(size_t)row * 3 + c];
    else if (c < 51) g_packed[e] = g_shs[(size_t)row * 48 + (c - 3)];
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float* p = packed + i * COLS;
    float* o = g_packed + i * COLS;
    const float sg = 1.f / (1.f + __expf(-p[51]));
    o[51] = g_op[i] * sg * (1.f - sg);
    const float s0 = __expf(p[52]), s1 = __expf(p[53]), s2 = __expf(p[54]);
    o[52] = g_sc[i * 3] * s0; o[53] = g_sc[i * 3 + 1] * s1; o[54] = g_sc[i * 3 + 2] * s2;
    const float q0 = p[55], q1 = p[56], q2 = p[57], q3 = p[58];
    const float nq = fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
    const float inv = 1.f / nq;
    const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
    int k = 0;
    float sm = s0;
    if (s1 < sm) { sm = s1; k = 1; }
    if (s2 < sm) { k = 2; }
    // normal = c / (|c| + eps), c = column k of R(q^)
    float c[3];
    rot_col(k, r, x, y, z, c);
    const float m = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    const float me = m + 1e-8f;
    const float gn0 = g_nrm[i * 3], gn1 = g_nrm[i * 3 + 1], gn2 = g_nrm[i * 3 + 2];
    const float cg = c[0] * gn0 + c[1] * gn1 + c[2] * gn2;
    const float kk = (m > 0.f) ? cg / (m * me * me) : 0.f;
    const float dc0 = gn0 / me - c[0] * kk, dc1 = gn1 / me - c[1] * kk, dc2 = gn2 / me - c[2] * kk;
    // d column_k / d (r,x,y,z)
    float dr, dx, dy, dz;
    if (k == 0) {
      dr = 2.f * (z * dc1 - y * dc2);
      dx = 2.f * (y * dc1 + z * dc2);
      dy = 2.f * (-2.f * y * dc0 + x * dc1 - r * dc2);
      dz = 2.f * (-2.f * z * dc0 + r * dc1 + x * dc2);
    } else if (k == 1) {
      dr = 2.f * (-z * dc0 + x * dc2);
      dx = 2.f * (y * dc0 - 2.f * x * dc1 + r * dc2);
      dy = 2.f * (x * dc0 + z * dc2);
      dz = 2.f * (-r * dc0 - 2.f * z * dc1 + y * dc2);
    } else {
      dr = 2.f * (y * dc0 - x * dc1);
      dx = 2.f * (z * dc0 - r * dc1 - 2.f * x * dc2);
      dy = 2.f * (r * dc0 + z * dc1 - 2.f * y * dc2);
      dz = 2.f * (x * dc0 + y * dc1);
    }
    const float t0 = g_rot[i * 4] + dr, t1 = g_rot[i * 4 + 1] + dx, t2 = g_rot[i * 4 + 2] + dy, t3 = g_rot[i * 4 + 3] + dz;
    // q^ = q / |q|
    const float dot = r * t0 + x * t1 + y * t2 + z * t3;
    o[55] = (t0 - r * dot) * inv; o[56] = (t1 - x * dot) * inv; o[57] = (t2 - y * dot) * inv; o[58] = (t3 - z * dot) * inv;
  }
}

}  // namespace rtgs

extern "C" int rtgs_map_activate_forward(const float* packed, int64_t n, float* xyz, float* opacity, float* shs,
                                         float* scales, float* rotations, float* normal, void* stream) {
  if (n < 0 || n > 72000000 || (n > 0 && (!packed || !xyz || !opacity || !shs || !scales || !rotations || !normal)))
    return -1;                                       // 72 M x 59 < 2^32 (32-bit flattened index)
  if (n == 0) return 0;
  int64_t blocks = (n * rtgs::COLS + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::activate_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed, n, xyz,
                     opacity, shs, scales, rotations, normal);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_map_activate_backward(const float* packed, int64_t n, const float* g_xyz, const float* g_opacity,
                                          const float* g_shs, const float* g_scales, const float* g_rotations,
                                          const float* g_normal, float* g_packed, void* stream) {
  if (n < 0 || n > 72000000 || (n > 0 && (!packed || !g_xyz || !g_opacity || !g_shs || !g_scales || !g_rotations || !g_normal || !g_packed)))
    return -1;
  if (n == 0) return 0;
  int64_t blocks = (n * rtgs::COLS + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::activate_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed, n,
                     g_xyz, g_opacity, g_shs, g_scales, g_rotations, g_normal, g_packed);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

// ---------------------------------------------------------------------------------------------
// Single-GPU map step: activation backward + Adam + activation forward of the UPDATED parameters
// in one streaming kernel (the rasterizer inputs of the next iteration are produced here).
// Reads 6 gradient tensors + packed + m + v, writes packed + m + v + the 6 rasterizer inputs:
// 1.9 kB per Gaussian instead of 2.9 kB for the three separate kernels.  With more than one rank
// the reduce-scatter sits between the backward and Adam, so the unfused kernels are used there.
// ---------------------------------------------------------------------------------------------
namespace rtgs {

struct AdamC { float beta1, beta2, eps, bc1, bc2_sqrt; };

__device__ __forceinline__ float adam_update(float p, float g, float& m, float& v, float lr, const AdamC& a) {
  m = a.beta1 * m + (1.f - a.beta1) * g;
  v = a.beta2 * v + (1.f - a.beta2) * g * g;
  const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
  return p - (lr / a.bc1) * (m / denom);
}

__global__ void __launch_bounds__(256) map_fused_step_kernel(
    float* __restrict__ packed, float* __restrict__ em, float* __restrict__ ev, const float* __restrict__ lr_col,
    uint32_t n, AdamC a, const float* __restrict__ g_xyz, const float* __restrict__ g_op,
    const float* __restrict__ g_shs, const float* __restrict__ g_sc, const float* __restrict__ g_rot,
    const float* __restrict__ g_nrm, float* __restrict__ xyz, float* __restrict__ opacity, float* __restrict__ shs,
    float* __restrict__ scales, float* __restrict__ rots, float* __restrict__ normal) {
  // phase A: the 51 pass-through columns (xyz + SH), flattened
  const uint32_t totalA = n * 51u;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < totalA; e += gridDim.x * blockDim.x) {
    const uint32_t row = e / 51u, c = e - row * 51u;
    const size_t pi = (size_t)row * COLS + c;
    const float g = (c < 3) ? g_xyz[(size_t)row * 3 + c] : g_shs[(size_t)row * 48 + (c - 3)];
    float m = em[pi], v = ev[pi];
    const float pn = adam_update(packed[pi], g, m, v, lr_col[c], a);
    packed[pi] = pn; em[pi] = m; ev[pi] = v;
    if (c < 3) xyz[(size_t)row * 3 + c] = pn; else shs[(size_t)row * 48 + (c - 3)] = pn;
  }
  // phase B: one lane per Gaussian for opacity / scaling / rotation (chain rule, Adam, re-activation)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float* p = packed + (size_t)i * COLS;
    float* pm = em + (size_t)i * COLS;
    float* pv = ev + (size_t)i * COLS;
    float gp[8];
    {
      const float sg = 1.f / (1.f + __expf(-p[51]));
      gp[0] = g_op[i] * sg * (1.f - sg);
      const float s0 = __expf(p[52]), s1 = __expf(p[53]), s2 = __expf(p[54]);
      gp[1] = g_sc[(size_t)i * 3] * s0; gp[2] = g_sc[(size_t)i * 3 + 1] * s1; gp[3] = g_sc[(size_t)i * 3 + 2] * s2;
      const float q0 = p[55], q1 = p[56], q2 = p[57], q3 = p[58];
      const float inv = 1.f / fmaxf(sqrtf(q0 * q0 + q1 * q1 + q2 * q2 + q3 * q3), 1e-12f);
      const float r = q0 * inv, x = q1 * inv, y = q2 * inv, z = q3 * inv;
      int k = 0;
      float sm = s0;
      if (s1 < sm) { sm = s1; k = 1; }
      if (s2 < sm) { k = 2; }
      float c[3];
      rot_col(k, r, x, y, z, c);
      const float mm = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
      const float me = mm + 1e-8f;
      const float gn0 = g_nrm[(size_t)i * 3], gn1 = g_nrm[(size_t)i * 3 + 1], gn2 = g_nrm[(size_t)i * 3 + 2];
      const float cg = c[0] * gn0 + c[1] * gn1 + c[2] * gn2;
      const float kk = (mm > 0.f) ? cg / (mm * me * me) : 0.f;
      const float dc0 = gn0 / me - c[0] * kk, dc1 = gn1 / me - c[1] * kk, dc2 = gn2 / me - c[2] * kk;
      float dr, dx, dy, dz;
      if (k == 0) {
        dr = 2.f * (z * dc1 - y * dc2); dx = 2.f * (y * dc1 + z * dc2);
        dy = 2.f * (-2.f * y * dc0 + x * dc1 - r * dc2); dz = 2.f * (-2.f * z * dc0 + r * dc1 + x * dc2);
      } else if (k == 1) {
        dr = 2.f * (-z * dc0 + x * dc2); dx = 2.f * (y * dc0 - 2.f * x * dc1 + r * dc2);
        dy = 2.f * (x * dc0 + z * dc2); dz = 2.f * (-r * dc0 - 2.f * z * dc1 + y * dc2);
      } else {
        dr = 2.f * (y * dc0 - x * dc1); dx = 2.f * (z * dc0 - r * dc1 - 2.f * x * dc2);
        dy = 2.f * (r * dc0 + z * dc1 - 2.f * y * dc2); dz = 2.f * (x * dc0 + y * dc1);
      }
      const float t0 = g_rot[(size_t)i * 4] + dr, t1 = g_rot[(size_t)i * 4 + 1] + dx, t2 = g_rot[(size_t)i * 4 + 2] + dy,
                  t3 = g_rot[(size_t)i * 4 + 3] + dz;
      const float dot = r * t0 + x * t1 + y * t2 + z * t3;
      gp[4] = (t0 - r * dot) * inv; gp[5] = (t1 - x * dot) * inv; gp[6] = (t2 - y * dot) * inv; gp[7] = (t3 - z * dot) * inv;
    }
    float np_[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      float m = pm[51 + c], v = pv[51 + c];
      np_[c] = adam_update(p[51 + c], gp[c], m, v, lr_col[51 + c], a);
      p[51 + c] = np_[c]; pm[51 + c] = m; pv[51 + c] = v;
    }
    // re-activate the updated row
    opacity[i] = 1.f / (1.f + __expf(-np_[0]));
    const float s0 = __expf(np_[1]), s1 = __expf(np_[2]), s2 = __expf(np_[3]);
    scales[(size_t)i * 3] = s0; scales[(size_t)i * 3 + 1] = s1; scales[(size_t)i * 3 + 2] = s2;
    const float inv = 1.f / fmaxf(sqrtf(np_[4] * np_[4] + np_[5] * np_[5] + np_[6] * np_[6] + np_[7] * np_[7]), 1e-12f);
    const float r = np_[4] * inv, x = np_[5] * inv, y = np_[6] * inv, z = np_[7] * inv;
    rots[(size_t)i * 4] = r; rots[(size_t)i * 4 + 1] = x; rots[(size_t)i * 4 + 2] = y; rots[(size_t)i * 4 + 3] = z;
    int k = 0;
    float sm = s0;
    if (s1 < sm) { sm = s1; k = 1; }
    if (s2 < sm) { k = 2; }
    float c[3];
    rot_col(k, r, x, y, z, c);
    const float mq = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + 1e-8f;
    normal[(size_t)i * 3] = c[0] / mq; normal[(size_t)i * 3 + 1] = c[1] / mq; normal[(size_t)i * 3 + 2] = c[2] / mq;
  }
}

}  // namespace rtgs

extern "C" int rtgs_map_fused_step(float* packed, float* exp_avg, float* exp_avg_sq, const float* lr_per_column,
                                   int64_t n, int32_t step, float beta1, float beta2, float eps, const float* g_xyz,
                                   const float* g_opacity, const float* g_shs, const float* g_scales,
                                   const float* g_rotations, const float* g_normal, float* xyz, float* opacity,
                                   float* shs, float* scales, float* rotations, float* normal, void* stream) {
  if (n < 0 || n > 72000000 || step < 1) return -1;
  if (n == 0) return 0;
  if (!packed || !exp_avg || !exp_avg_sq || !lr_per_column || !g_xyz || !g_opacity || !g_shs || !g_scales ||
      !g_rotations || !g_normal || !xyz || !opacity || !shs || !scales || !rotations || !normal)
    return -1;
  rtgs::AdamC a{beta1, beta2, eps, 1.f - powf(beta1, (float)step), sqrtf(1.f - powf(beta2, (float)step))};
  int64_t blocks = (n * 51 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(rtgs::map_fused_step_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, packed,
                     exp_avg, exp_avg_sq, lr_per_column, (uint32_t)n, a, g_xyz, g_opacity, g_shs, g_scales, g_rotations,
                     g_normal, xyz, opacity, shs, scales, rotations, normal);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
