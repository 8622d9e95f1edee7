// Probe of the gfx950 wave-level primitives the MFMA backward walk (raster_bwd_mfma.hip) is built from - run once on
// the GPU box to pin their semantics before trusting a kernel that composes them:
//   1. DPP row_newbcast:n  - every lane of a row of 16 reads lane n of its own row;
//   2. in-row inclusive scans with row_shr:1/2/4/8 where lanes without a source keep `old`;
//   3. the operand / result lane maps of v_mfma_f32_16x16x4_f32 (A[m = l & 15][k = l >> 4], B[k = l >> 4][n = l & 15],
//      C[4 (l >> 4) + r][l & 15] in accumulator register r).
// Build: hipcc --offload-arch=gfx950 -O2 tools/probe/wave_ops.hip -o tools/probe/wave_ops ; prints PASS / FAIL lines.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CTRL>
__device__ __forceinline__ float dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, 0xf, 0xf, false));
}

__global__ void probe(float* out) {
  const int l = threadIdx.x;
  const float v = (float)l;
  out[0 * 64 + l] = dpp<0x150 + 3>(0.f, v);          // row_newbcast:3
  out[1 * 64 + l] = dpp<0x150 + 15>(0.f, v);         // row_newbcast:15
  float s = (float)(1 + (l & 15));
  s += dpp<0x111>(0.f, s);                           // row_shr:1
  s += dpp<0x112>(0.f, s);
  s += dpp<0x114>(0.f, s);
  s += dpp<0x118>(0.f, s);
  out[2 * 64 + l] = s;                               // inclusive prefix sum of 1..16 within the row
  float p = 1.f + 0.125f * (float)(l & 15);
  p *= dpp<0x111>(1.f, p);
  p *= dpp<0x112>(1.f, p);
  p *= dpp<0x114>(1.f, p);
  p *= dpp<0x118>(1.f, p);
  out[3 * 64 + l] = p;                               // inclusive prefix product
  out[4 * 64 + l] = dpp<0x111>(-7.f, v);             // exclusive shift: lane 0 of a row keeps old = -7
  const int m = l & 15, k = l >> 4;
  const float a = (float)(m + 1) + 100.f * (float)(k + 1);          // A[m][k]
  const float b = 0.5f * (float)(k + 1) + 0.01f * (float)m;         // B[k][n = m]
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[(5 + r) * 64 + l] = c[r];
  out[9 * 64 + l] = dpp<0x138>(-1.f, v);             // wave_shr:1
  out[10 * 64 + l] = dpp<0x123>(0.f, v);             // row_ror:3
}

int main() {
  float* d;
  float h[11 * 64];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL hip\n"); return 1; }
  int bad[8] = {0};
  for (int l = 0; l < 64; ++l) {
    const int row = l & ~15, n = l & 15;
    if (h[l] != (float)(row + 3)) bad[0]++;
    if (h[64 + l] != (float)(row + 15)) bad[1]++;
    if (h[128 + l] != (float)((n + 1) * (n + 2) / 2)) bad[2]++;
    double pp = 1.0; for (int j = 0; j <= n; ++j) pp *= 1.0 + 0.125 * j;
    if (fabs(h[192 + l] - pp) > 1e-4 * pp) bad[3]++;
    if (h[256 + l] != (n == 0 ? -7.f : (float)(l - 1))) bad[4]++;
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * (l >> 4) + r, j = n;
      double e = 0; for (int k = 0; k < 4; ++k) e += ((i + 1) + 100.0 * (k + 1)) * (0.5 * (k + 1) + 0.01 * j);
      if (fabs(h[(5 + r) * 64 + l] - e) > 1e-3) bad[5]++;
    }
    if (h[9 * 64 + l] != (l == 0 ? -1.f : (float)(l - 1))) bad[6]++;
    if (h[10 * 64 + l] != (float)(row + ((n + 16 - 3) & 15))) bad[7]++;
  }
  const char* names[8] = {"row_newbcast:3", "row_newbcast:15", "row_shr prefix sum", "row_shr prefix product", "row_shr:1 keeps old",
                          "mfma_f32_16x16x4_f32 lane maps", "wave_shr:1", "row_ror:3 reads lane n-3"};
  int fails = 0;
  for (int t = 0; t < 8; ++t) { printf("%s %s (%d mismatches)\n", bad[t] ? "FAIL" : "PASS", names[t], bad[t]); fails += bad[t] != 0; }
  if (fails) { for (int r = 0; r < 11; ++r) { printf("row %d:", r); for (int l = 0; l < 64; ++l) printf(" %g", h[r * 64 + l]); printf("\n"); } }
  return fails;
}
