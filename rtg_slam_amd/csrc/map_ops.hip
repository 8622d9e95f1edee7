// Map-state kernels around the rasterizer (SURVEY.md §8f "next-2"): the activations that turn the
// raw map parameters into the rasterizer's inputs, and their backward, each as ONE streaming
// kernel instead of ~40 elementwise / gather launches (SLAM/gaussian_pointcloud.py:16-25 exp /
// sigmoid / normalize, :538-550 get_normal).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rtgs {


__device__ __forceinline__ void rot_col(int k, float r, float x, float y, float z, float (&c)[3]) {
  if (k == 0) { c[0] = 1.f - 2.f * (y * y + z * z); c[1] = 2.f * (x * y + r * z); c[2] = 2.f * (x * z - r * y); }
  else if (k == 1) { c[0] = 2.f * (x * y - r * z); c[1] = 1.f - 2.f * (x * x + z * z); c[2] = 2.f * (y * z + r * x); }
  else { c[0] = 2.f * (x * z + r * y); c[1] = 2.f * (y * z - r * x); c[2] = 1.f - 2.f * (x * x + y * y); }
}

}  // namespace rtgs

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
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = raw8[2 * i], q = raw8[2 * i + 1];          // (o, s0, s1, s2), (qw, qx, qy, qz)
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
}

__global__ void __launch_bounds__(256) activate8_bwd_kernel(const float4* __restrict__ raw8, int64_t n,
                                                            const float* __restrict__ g_op, const float* __restrict__ g_sc,
                                                            const float4* __restrict__ g_rot, const float* __restrict__ g_nrm,
                                                            float4* __restrict__ g_raw8) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 a = raw8[2 * i], q = raw8[2 * i + 1];
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
    const float gn0 = g_nrm[i * 3], gn1 = g_nrm[i * 3 + 1], gn2 = g_nrm[i * 3 + 2];
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
    const float4 gr = g_rot[i];
    const float t0 = gr.x + dr, t1 = gr.y + dx, t2 = gr.z + dy, t3 = gr.w + dz;
    const float dot = r * t0 + x * t1 + y * t2 + z * t3;
    g_raw8[2 * i] = make_float4(g_op[i] * sg * (1.f - sg), g_sc[i * 3] * s0, g_sc[i * 3 + 1] * s1, g_sc[i * 3 + 2] * s2);
    g_raw8[2 * i + 1] = make_float4((t0 - r * dot) * inv, (t1 - x * dot) * inv, (t2 - y * dot) * inv, (t3 - z * dot) * inv);
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
                     (const float4*)raw8, n, g_opacity, g_scales, (const float4*)g_rotations, g_normal, (float4*)g_raw8);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
