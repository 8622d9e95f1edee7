// Issue-rate probe for gfx950: how many wave64 VALU instructions per second does ONE SIMD issue, for
// the instruction kinds the blend walks are made of, with 1..8 waves resident per SIMD?
//   indep   : 8 independent accumulator chains per wave (v_fma_f32)             -> the issue ceiling
//   dep     : ONE dependent chain per wave (v_fma_f32 on the same register)     -> latency per instruction
//   dpp     : dependent v_mul_f32 with a row_shr:1 DPP source (the scans of blend_bwd_mfma)
//   exp     : independent v_exp_f32 (transcendental pipe, quarter rate on earlier CDNA)
//   mfma    : v_mfma_f32_16x16x4_f32 back to back on two accumulators, two VALU between (the walk's step shape)
// Every block is 256 threads (one wave per SIMD of its CU); a grid of 256 x k blocks with k = 1..8 puts k waves on each
// SIMD (the kernel uses < 64 VGPRs and no LDS).  Reported per (kind, k): wave-instructions / s / SIMD from the wall time
// of the launch (HIP events), the shader-cycle count of a wave from s_memtime, and cycles per instruction seen by ONE wave
// and by the SIMD (= wave cycles / (instructions x waves on the SIMD)).
// Build + run:  hipcc --offload-arch=gfx950 -O2 tools/probe/valu_rate.hip -o tools/probe/valu_rate && tools/probe/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int ITERS = 2048;      // loop trips
constexpr int PER = 64;          // probed instructions per trip

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ void __launch_bounds__(256) probe(float* out, unsigned long long* cyc, float seed) {
  float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f,
        a7 = a0 + 7.f;
  const float m = 0.999f, c = 1e-3f;
  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
  unsigned sc0 = 1u, sc1 = 2u;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < ITERS; ++it) {
    if (KIND == 0) {                    // 8 chains x 8 = 64 independent-enough FMAs
      REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                        "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
    } else if (KIND == 1) {             // 64 dependent FMAs
      REP8(asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                        : "+v"(a0) : "v"(m), "v"(c));)
    } else if (KIND == 2) {             // 64 dependent DPP multiplies (the assembler inserts nothing: hazards are ours)
      REP8(asm volatile("s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "s_nop 1\n v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(a0));)
    } else if (KIND == 3) {             // 64 independent-enough v_exp_f32
      REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
                        "v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
    } else if (KIND == 4) {             // 16 x (2 MFMA on alternating accumulators + 2 VALU) = 64 instructions
      REP8(asm volatile("v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_fma_f32 %4, %4, %6, %7\n"
                        "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_fma_f32 %5, %5, %6, %7\n"
                        "v_mfma_f32_16x16x4_f32 %0, %2, %3, %0\n v_fma_f32 %4, %4, %6, %7\n"
                        "v_mfma_f32_16x16x4_f32 %1, %2, %3, %1\n v_fma_f32 %5, %5, %6, %7\n"
                        : "+v"(acc0), "+v"(acc1) : "v"(a2), "v"(a3), "v"(a0), "v"(a1), "v"(m), "v"(c));)
    } else if (KIND == 5) {             // THREE interleaved dependent DPP chains (two instructions between a write and its
                                        // DPP read = the two wait states the hazard needs): do the others fill the wait?
      REP8(asm volatile("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2));)
    } else if (KIND == 6) {             // scalar + vector mix as in blend_fwd's mask walk: 1 SALU per 2 VALU
      REP8(asm volatile("v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n s_add_u32 %4, %4, 1\n"
                        "v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n s_add_u32 %5, %5, 1\n"
                        "v_fma_f32 %0, %0, %6, %7\n v_fma_f32 %1, %1, %6, %7\n s_add_u32 %4, %4, 1\n"
                        "v_fma_f32 %2, %2, %6, %7\n v_fma_f32 %3, %3, %6, %7\n s_add_u32 %5, %5, 1\n"
                        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+s"(sc0), "+s"(sc1) : "v"(m), "v"(c) : "scc");)
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc0[0] + acc1[0] + acc0[1] + acc1[3];
  if (r == 12345.678f || sc0 + sc1 == 7u) out[0] = r;                       // keeps the chains alive
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int per_trip_counted, float* out, unsigned long long* cyc, unsigned long long* hcyc, int cus) {
  for (int k = 1; k <= 8; ++k) {
    const int blocks = cus * k;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);      // warm
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<KIND>, dim3(blocks), dim3(256), 0, 0, out, cyc, 1.0f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(hcyc, cyc, sizeof(unsigned long long) * blocks * 4, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks * 4; ++i) mean += (double)hcyc[i];
    mean /= blocks * 4;
    const double insts = (double)ITERS * per_trip_counted;             // per wave
    const double simds = cus * 4.0;
    const double rate = insts * k * simds / (ms * 1e-3) / simds;       // wave-instructions / s / SIMD
    printf("%-8s waves/SIMD %d  launch %.3f ms  %.3f G wave-inst/s/SIMD  (%.1f G/s chip)  s_memtime ticks/inst: wave %.2f  SIMD %.2f\n",
           name, k, ms, rate / 1e9, rate * simds / 1e9, mean / insts, mean / insts / k);
    hipEventDestroy(e0); hipEventDestroy(e1);
  }
}

int main() {
  setvbuf(stdout, nullptr, _IOLBF, 0);
  int dev = 0, cus = 0, clk = 0;
  hipGetDevice(&dev);
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, dev);
  printf("device: %d CUs, peak clock %.0f MHz; s_memtime ticks = shader cycles (ticks / wall time gives the clock the chip "
         "actually ran at under that load)\n", cus, clk / 1e3);
  float* out; unsigned long long *cyc, *hcyc;
  hipMalloc(&out, 64);
  hipMalloc(&cyc, sizeof(unsigned long long) * cus * 8 * 4);
  hcyc = (unsigned long long*)malloc(sizeof(unsigned long long) * cus * 8 * 4);
  run<0>("indep", PER, out, cyc, hcyc, cus);
  run<1>("dep", PER, out, cyc, hcyc, cus);
  run<2>("dpp_dep", PER, out, cyc, hcyc, cus);        // counts the 64 DPP multiplies (the s_nop states ride along)
  run<5>("dpp_3ch", 72, out, cyc, hcyc, cus);
  run<3>("exp", PER, out, cyc, hcyc, cus);
  run<4>("mfma+2v", PER, out, cyc, hcyc, cus);        // 32 MFMA + 32 VALU per trip
  run<6>("v2+s1", 64, out, cyc, hcyc, cus);           // 64 VALU + 32 SALU per trip: VALU counted
  return 0;
}
