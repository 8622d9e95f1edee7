// raw8 = (opacity | scaling xyz | rotation wxyz) activations of one row and their backward
// (SLAM/gaussian_pointcloud.py:16-25 exp / sigmoid / normalize, :538-550 get_normal): ONE definition for the
// activation kernels, the Adam tail (map_ops.hip) and the fused tail of the one-call step (map_fused.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtgs {


__device__ __forceinline__ void rot_col(int k, float r, float x, float y, float z, float (&c)[3]) {
  if (k == 0) { c[0] = 1.f - 2.f * (y * y + z * z); c[1] = 2.f * (x * y + r * z); c[2] = 2.f * (x * z - r * y); }
  else if (k == 1) { c[0] = 2.f * (x * y - r * z); c[1] = 1.f - 2.f * (x * x + z * z); c[2] = 2.f * (y * z + r * x); }
  else { c[0] = 2.f * (x * z + r * y); c[1] = 2.f * (y * z - r * x); c[2] = 1.f - 2.f * (x * x + y * y); }
}

}  // namespace rtgs

namespace rtgs {

// one row: a = (o, s0, s1, s2) raw, q = raw quaternion (w, x, y, z).  ONE definition for the full activation pass and
// for the rows the tail kernel re-activates after stepping them (bit-identical by construction).
__device__ __forceinline__ void activate8_row_store(const float4 a, const float4 q, int64_t i, float* __restrict__ opacity,
                                                    float* __restrict__ scales, float4* __restrict__ rots,
                                                    float* __restrict__ normal) {
  opacity[i] = 1.f / (1.f + __expf(-a.x));
  const float s0 = __expf(a.y), s1 = __expf(a.z), s2 = __expf(a.w);
  scales[i * 3] = s0; scales[i * 3 + 1] = s1; scales[i * 3 + 2] = s2;
  const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  const float r = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
  rots[i] = make_float4(r, x, y, z);
  int k = 0;
  float sm = s0;
  if (s1 < sm) { sm = s1; k = 1; }
  if (s2 < sm) { k = 2; }
  float c[3];
  rot_col(k, r, x, y, z, c);
  const float m = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]) + 1e-8f;
  normal[i * 3] = c[0] / m; normal[i * 3 + 1] = c[1] / m; normal[i * 3 + 2] = c[2] / m;
}

// gradient of (opacity, scales, rotations, normal) w.r.t. the raw8 row (a = o s0 s1 s2, q = quaternion wxyz)
__device__ __forceinline__ void activate8_bwd_row(const float4 a, const float4 q, float g_op, float gs0, float gs1, float gs2,
                                                  const float4 gr, float gn0, float gn1, float gn2, float4& lo, float4& hi) {
  const float sg = 1.f / (1.f + __expf(-a.x));
  const float s0 = __expf(a.y), s1 = __expf(a.z), s2 = __expf(a.w);
  const float nq = fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
  const float inv = 1.f / nq;
  const float r = q.x * inv, x = q.y * inv, y = q.z * inv, z = q.w * inv;
  int k = 0;
  float sm = s0;
  if (s1 < sm) { sm = s1; k = 1; }
  if (s2 < sm) { k = 2; }
  float c[3];
  rot_col(k, r, x, y, z, c);
  const float m = sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
  const float me = m + 1e-8f;
  const float cg = c[0] * gn0 + c[1] * gn1 + c[2] * gn2;
  const float kk = (m > 0.f) ? cg / (m * me * me) : 0.f;
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
  const float t0 = gr.x + dr, t1 = gr.y + dx, t2 = gr.z + dy, t3 = gr.w + dz;
  const float dot = r * t0 + x * t1 + y * t2 + z * t3;
  lo = make_float4(g_op * sg * (1.f - sg), gs0 * s0, gs1 * s1, gs2 * s2);
  hi = make_float4((t0 - r * dot) * inv, (t1 - x * dot) * inv, (t2 - y * dot) * inv, (t3 - z * dot) * inv);
}

}  // namespace rtgs
