// Adam update of one scalar, shared by every kernel that applies it (raster_bwd.hip, map_ops.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace rtgs {
// Contractions are spelled out (and automatic ones disabled) so that every kernel that inlines this helper
// rounds identically - the row-skipping kernel must match the dense ones bit for bit.
__device__ __forceinline__ float adam1(float p, float g, float& m, float& v, float lr, float beta1, float beta2,
                                       float eps, float bc1, float bc2_sqrt) {
#pragma clang fp contract(off)
  const float t1 = (1.f - beta1) * g;
  const float t2 = ((1.f - beta2) * g) * g;
  m = __builtin_fmaf(beta1, m, t1);
  v = __builtin_fmaf(beta2, v, t2);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  const float step = lr / bc1;
  return __builtin_fmaf(-step, m / denom, p);
}

}  // namespace rtgs
