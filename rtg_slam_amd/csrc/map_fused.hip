// The per-Gaussian tail of the one-call map step (rtgs_slam_map_step on one GPU) as ONE kernel over ONE compact list of
// live rows:   grad_reduce  +  preprocess_bwd  +  activation backward / attach gradient / Adam / re-activation
// (raster_bwd.hip, map_ops.hip) - the three kernels that were 28 % of the surface-map iteration and moved 1.7x their
// algorithmic bytes (profiles/r03_traffic_surface.json: 887 MB for ~530 MB).  What no longer touches HBM:
//   * the SplatGrad record        (64 B written by grad_reduce, read and re-zeroed by preprocess_bwd)
//   * the 236-B gradient row      (written by preprocess_bwd, read by the tail; + the 32-B raw8 gradient row)
//   * the row-state / touched scans of two of the three kernels.
// A row is live if a tile left it a gradient slot this step (touched) or its Adam moments have ever left zero (the
// row-skipping rule of fused_adam_rows: such a row is stepped with a zero gradient).  Per live row, one lane: slots ->
// SplatGrad (long runs: the whole wave sums them first) -> chain rule in registers (raster_chain.h) -> activation
// backward + attach gradient -> Adam on raw8 and xyz -> re-activation; then the wave sweeps the 48 SH columns of its rows
// with 12 lanes x float4 per row, forming the SH gradient basis[k] * gc[c] on the fly.  Same device functions as the
// three kernels it replaces: results are theirs to the rounding of the slot-summation order (tests/test_trainable_gpu.py).
// Measured and NOT kept (profiles/r04_fused_tail_variants.txt): one global live-row list built by a light first pass (every
// wave of the heavy pass full, but the long slot runs of a depth-complex map then queue up inside a few waves: 105 us
// instead of 55 on the headline scene); a third workgroup per CU at 168 registers with spills (+20 %); the SH block reduced
// group by group to save registers (the serialised loads cost more than the registers saved).
// The gradient rows of the arena are NOT produced here (they stay all-zero, row_state 0): multi-GPU steps, which exchange
// them, and the autograd path keep the three-kernel form.
#include "../../include/rtgs_raster.h"
#include "raster_common.h"
#include "raster_chain.h"
#include "adam_common.h"
#include "activate_common.h"

#include <math.h>

namespace rtgs {

struct FusedArgs {
  RasterParams p;
  // rasterizer inputs of the step (activated values; means = xyz and shs are the parameters themselves)
  const float *opac, *scales, *rots, *normal_w;
  const uint8_t* clamped;
  // gradient partials of blend_bwd
  const BwdInfo* info;
  const uint32_t* gbase;
  uint32_t* slot_count;
  uint8_t* touched;
  SplatGrad* records;            // the atomics fallback accumulates here (info->use_slots == 0); zeroed again as consumed
  // parameters and Adam state; m / v / ever / init_* / confidence start at row t0 (rtgs_map_step_args)
  float *xyz, *shs, *raw8;
  float *m_xyz, *v_xyz, *m_shs, *v_shs, *m_raw8, *v_raw8;
  const float *lr_xyz, *lr_shs, *lr_raw8;
  uint8_t *ever_xyz, *ever_shs, *ever_raw8;
  float beta1, beta2, eps, bc1, bc2_sqrt;
  const float* init_xyz;
  const float4* init_raw8;
  const float* attach_info;
  float* confidence;
  float *act_opacity, *act_scales, *act_normal;
  float4* act_rots;
  uint32_t* live_counts;         // nullable: [0] += rows with gradient, [1] += rows stepped
  int t0, rows;                  // the trainable rows [t0, t0 + rows)
};

// Rows a workgroup compacts per pass - chosen per launch (fused_chunk below).  The kernel holds 2 waves per SIMD (512
// workgroups resident), and a pass is a chain of dependent memory round trips whatever it holds: few live rows (the
// headline map: 4.5 k of 1.2 M) want FEW, LARGE chunks - 586 of 2 048 rows are about one resident round, 341 us per
// iteration against 357 at 1 024 and 375 at 512; many live rows (the surface map: 209 k) want MANY SMALL ones for the
// balance of the last round - 0.643 ms at 512 against 0.657 at 1 024 / 2 048 and 0.72 at 4 096 (profiles/r04_fused_tail_variants.txt).
constexpr int BIG_RUN = 8;       // slot runs longer than this are summed by the whole wave

__device__ __forceinline__ bool attach_sel(const float4 init_lo) {
  return 1.f / (1.f + __expf(-init_lo.x)) < 0.9f;                // opacity_activation(init_stat["opacity"]) < 0.9 (map_ops.hip)
}

// RPW: list entries a wave takes per pass.  64 = every lane a row (large maps: the chunk choice above).  16 = a quarter of the
// lanes (FCHUNK 64): for SMALL trainable sets - the unstable rows of a SLAM map, a few thousand - where 1 024-row chunks
// left 5-10 workgroups on a 256-CU device and the kernel took 112 us for 8 000 rows (r05 sequence trace): spread thin,
// the same rows occupy sixteen times the waves and a wave's SH sweep is 4 sub-passes instead of 13.
template <int FCHUNK, int RPW = 64>
__global__ void __launch_bounds__(256) map_fused_tail_kernel(FusedArgs a) {
  __shared__ uint32_t s_list[FCHUNK];
  __shared__ uint32_t s_n;
  __shared__ float s_big[4][64][16];          // [wave][lane]: SplatGrad of a long run, summed by the wave
  __shared__ float s_sh[4][64][20];           // [wave][k]: gc[3] | basis[16] | -  of the wave's rows that step their SH block
  __shared__ int s_rows[4][64];
  if (spec_failed(a.p.spec_fail)) return;     // nothing persistent may change in a failed speculation
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const bool use_slots = a.info->use_slots != 0;
  const float* __restrict__ slots = reinterpret_cast<const float*>(a.info->slot_grads);
  uint32_t n_grad = 0, n_step = 0;
  for (int c0 = blockIdx.x * FCHUNK; c0 < a.rows; c0 += gridDim.x * FCHUNK) {
    __syncthreads();
    if (tid == 0) s_n = 0;
    __syncthreads();
    // ---- the live rows of this chunk, compacted (four byte reads per row)
    // all flag bytes of the chunk in flight at once (inside the loop below each row's four loads waited for the
    // previous row's ballot: a round trip to memory per 256 rows)
    constexpr int PASSES = FCHUNK >= 256 ? FCHUNK / 256 : 1;
    uint32_t flags = 0;
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int rl = min(c0 + k * 256 + tid, a.rows - 1);
      const uint32_t f = (uint32_t)a.touched[a.t0 + rl] | (uint32_t)a.ever_raw8[rl] | (uint32_t)a.ever_xyz[rl] | (uint32_t)a.ever_shs[rl];
      flags |= (f != 0u ? 1u : 0u) << k;
    }
#pragma unroll
    for (int k = 0; k < PASSES; ++k) {
      const int rl = c0 + k * 256 + tid;                       // row relative to t0
      const bool work = rl < a.rows && k * 256 + tid < FCHUNK && ((flags >> k) & 1u) != 0u;
      const unsigned long long m = __builtin_amdgcn_ballot_w64(work);
      if (m == 0ull) continue;
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (work) s_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)rl;
    }
    __syncthreads();
    const int nlive = (int)s_n;
    for (int q0 = wv * RPW; q0 < nlive; q0 += 4 * RPW) {        // a wave takes RPW list entries per pass
      const int q = q0 + lane;
      const bool have = lane < RPW && q < nlive;
      const int rl = have ? (int)s_list[q] : 0;                 // relative to t0 (Adam state, snapshot, confidence)
      const int r = a.t0 + rl;                                  // absolute (parameters, activated arrays, slots)
      bool rgrad = have && a.touched[r] != 0;
      // ---- SplatGrad of the row: its slots summed (long runs by the whole wave), or the record of the atomics fallback
      SplatGrad g{};
      uint32_t cnt = 0, b0 = 0;
      if (rgrad && use_slots) { cnt = a.slot_count[r]; b0 = a.gbase[r]; }
      if (use_slots) {
        unsigned long long bigm = __builtin_amdgcn_ballot_w64(rgrad && cnt > (uint32_t)BIG_RUN);
        while (bigm != 0ull) {
          const int src = __builtin_ctzll(bigm);
          bigm &= bigm - 1ull;
          const uint32_t n = (uint32_t)__builtin_amdgcn_readlane((int)cnt, src);
          const size_t base = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)b0, src);
          if (n <= 48u) {                                       // 16 lanes per slot, four slots per step
            const int grp = lane >> 4, c = lane & 15;
            float m0 = 0.f, m1 = 0.f;
            uint32_t s = grp;
            for (; s + 4 < n; s += 8) { m0 += slots[(base + s) * 16 + c]; m1 += slots[(base + s + 4) * 16 + c]; }
            if (s < n) m0 += slots[(base + s) * 16 + c];
            float acc = m0 + m1;
            acc += __shfl_xor(acc, 16);
            acc += __shfl_xor(acc, 32);
            if (lane < 16) s_big[wv][src][c] = acc;
          } else {
            // a near, screen-filling Gaussian (hundreds of tiles): four lanes per slot (one 16-B load each), 16 slots per
            // load instruction, four instructions in flight - the run is this wave's critical path (as in grad_reduce)
            const float4* __restrict__ s4 = reinterpret_cast<const float4*>(slots);
            const uint32_t sub = (uint32_t)lane & 3u, sg = (uint32_t)lane >> 2;
            float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
            uint32_t s = sg;
            for (; s + 48 < n; s += 64) {
              const float4 t0 = s4[(base + s) * 4 + sub], t1 = s4[(base + s + 16) * 4 + sub];
              const float4 t2 = s4[(base + s + 32) * 4 + sub], t3 = s4[(base + s + 48) * 4 + sub];
              a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w; a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
              a2.x += t2.x; a2.y += t2.y; a2.z += t2.z; a2.w += t2.w; a3.x += t3.x; a3.y += t3.y; a3.z += t3.z; a3.w += t3.w;
            }
            for (; s < n; s += 16) {
              const float4 t0 = s4[(base + s) * 4 + sub];
              a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
            }
            float4 acc = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                                     (a0.w + a1.w) + (a2.w + a3.w));
#pragma unroll
            for (int off = 4; off < 64; off <<= 1) {
              acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
              acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
            }
            if (lane < 4) { s_big[wv][src][4 * sub] = acc.x; s_big[wv][src][4 * sub + 1] = acc.y; s_big[wv][src][4 * sub + 2] = acc.z; s_big[wv][src][4 * sub + 3] = acc.w; }
          }
        }
        __builtin_amdgcn_wave_barrier();
        if (rgrad) {
          float t[16];
          if (cnt > (uint32_t)BIG_RUN) {
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = s_big[wv][lane][c];
          } else {
#pragma unroll
            for (int c = 0; c < 16; ++c) t[c] = 0.f;
            const float4* s4 = reinterpret_cast<const float4*>(slots) + (size_t)b0 * 4;
            for (uint32_t s = 0; s < cnt; ++s) {
              const float4 x0 = s4[s * 4], x1 = s4[s * 4 + 1], x2 = s4[s * 4 + 2], x3 = s4[s * 4 + 3];
              t[0] += x0.x; t[1] += x0.y; t[2] += x0.z; t[3] += x0.w; t[4] += x1.x; t[5] += x1.y; t[6] += x1.z; t[7] += x1.w;
              t[8] += x2.x; t[9] += x2.y; t[10] += x2.z; t[11] += x2.w; t[12] += x3.x;
            }
          }
          g.du = t[0]; g.dv = t[1]; g.dca = t[2]; g.dcb = t[3]; g.dcc = t[4]; g.dop = t[5]; g.dr = t[6]; g.dg = t[7];
          g.db = t[8]; g.dnx = t[9]; g.dny = t[10]; g.dnz = t[11]; g.dpd = t[12];
          a.slot_count[r] = 0u;                                 // the counters are zero between calls
        }
      } else if (rgrad) {
        g = a.records[r];
        float4* z = reinterpret_cast<float4*>(a.records + r);   // leave the scratch zero for the next call
        z[0] = z[1] = z[2] = z[3] = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (rgrad) {
        a.touched[r] = 0;
        rgrad = (g.du != 0.f) | (g.dv != 0.f) | (g.dca != 0.f) | (g.dcb != 0.f) | (g.dcc != 0.f) | (g.dop != 0.f) |
                (g.dr != 0.f) | (g.dg != 0.f) | (g.db != 0.f) | (g.dnx != 0.f) | (g.dny != 0.f) | (g.dnz != 0.f) | (g.dpd != 0.f);
      }
      bool need_sh = false;
      if (have) {
        const uint8_t e_raw8 = a.ever_raw8[rl], e_xyz = a.ever_xyz[rl], e_shs = a.ever_shs[rl];
        ChainOut o{};
        if (rgrad) chain_rule(a.p, r, a.xyz, a.shs, a.scales, a.rots, a.normal_w, a.clamped, g, o, nullptr);
        // ---- attach regulariser (mapper.py:384-401): a row that was never stepped still equals its snapshot
        const float4* raw8_4 = reinterpret_cast<const float4*>(a.raw8);
        float4 at_lo = make_float4(0.f, 0.f, 0.f, 0.f), at_hi = at_lo;
        float at[3] = {0.f, 0.f, 0.f};
        bool sel = false;
        if (a.init_raw8 && (rgrad || e_raw8 != 0 || e_xyz != 0)) {
          const float4 i_lo = a.init_raw8[2 * (size_t)rl];
          sel = attach_sel(i_lo);
          if (sel) {
            const float4 i_hi = a.init_raw8[2 * (size_t)rl + 1], c_lo = raw8_4[2 * (size_t)r], c_hi = raw8_4[2 * (size_t)r + 1];
            const float n = fmaxf(a.attach_info[0], 1.f);
            const float k3 = 2000.f / (n * 3.f), k4 = 2000.f / (n * 4.f);
            at_lo = make_float4(0.f, k3 * (c_lo.y - i_lo.y), k3 * (c_lo.z - i_lo.z), k3 * (c_lo.w - i_lo.w));
            at_hi = make_float4(k4 * (c_hi.x - i_hi.x), k4 * (c_hi.y - i_hi.y), k4 * (c_hi.z - i_hi.z), k4 * (c_hi.w - i_hi.w));
#pragma unroll
            for (int c = 0; c < 3; ++c) at[c] = k3 * (a.xyz[(size_t)r * 3 + c] - a.init_xyz[(size_t)rl * 3 + c]);
          }
        }
        const bool grad = rgrad || sel;
        // ---- raw8: activation backward (+ attach), Adam, re-activation
        if (grad || e_raw8 != 0) {
          float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
          if (rgrad)
            activate8_bwd_row(raw8_4[2 * (size_t)r], raw8_4[2 * (size_t)r + 1], o.dop, o.ds[0], o.ds[1], o.ds[2], o.dq, o.dn[0],
                              o.dn[1], o.dn[2], lo, hi);
          lo.y += at_lo.y; lo.z += at_lo.z; lo.w += at_lo.w;
          hi.x += at_hi.x; hi.y += at_hi.y; hi.z += at_hi.z; hi.w += at_hi.w;
          if (e_raw8 == 0) a.ever_raw8[rl] = 1;
          float4* p4 = reinterpret_cast<float4*>(a.raw8) + 2 * (size_t)r;
          float4* m4 = reinterpret_cast<float4*>(a.m_raw8) + 2 * (size_t)rl;
          float4* v4 = reinterpret_cast<float4*>(a.v_raw8) + 2 * (size_t)rl;
          const float gg[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          float4 pp[2] = {p4[0], p4[1]}, mm[2] = {m4[0], m4[1]}, vv[2] = {v4[0], v4[1]};
          float* pf = reinterpret_cast<float*>(pp); float* mf = reinterpret_cast<float*>(mm); float* vf = reinterpret_cast<float*>(vv);
#pragma unroll
          for (int c = 0; c < 8; ++c) pf[c] = adam1(pf[c], gg[c], mf[c], vf[c], a.lr_raw8[c], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
          p4[0] = pp[0]; p4[1] = pp[1]; m4[0] = mm[0]; m4[1] = mm[1]; v4[0] = vv[0]; v4[1] = vv[1];
          if (a.act_opacity) activate8_row_store(pp[0], pp[1], r, a.act_opacity, a.act_scales, a.act_rots, a.act_normal);
        }
        // ---- xyz
        if (grad || e_xyz != 0) {
          if (e_xyz == 0) a.ever_xyz[rl] = 1;
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const size_t op = (size_t)r * 3 + c, os = (size_t)rl * 3 + c;
            float mi = a.m_xyz[os], vi = a.v_xyz[os];
            a.xyz[op] = adam1(a.xyz[op], (rgrad ? o.dm[c] : 0.f) + at[c], mi, vi, a.lr_xyz[c], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
            a.m_xyz[os] = mi; a.v_xyz[os] = vi;
          }
        }
        // confidence += 1 where the f_dc gradient is non-zero (mapper.py:454-456): d f_dc = basis[0] * gc
        if (a.confidence && rgrad) {
          if (o.basis[0] * o.gc[0] != 0.f || o.basis[0] * o.gc[1] != 0.f || o.basis[0] * o.gc[2] != 0.f) a.confidence[rl] += 1.f;
        }
        need_sh = rgrad || e_shs != 0;
        if (need_sh && e_shs == 0) a.ever_shs[rl] = 1;
        n_grad += rgrad ? 1u : 0u;
        n_step += (grad || e_raw8 != 0 || e_xyz != 0 || need_sh) ? 1u : 0u;
        // ---- hand the SH block to the wave: 48 columns = 12 lanes x float4 per row
        const unsigned long long mask = __builtin_amdgcn_ballot_w64(need_sh);
        if (need_sh) {
          const int k = __popcll(mask & ((1ull << lane) - 1ull));
          s_rows[wv][k] = rl | (rgrad ? (1 << 30) : 0);
#pragma unroll
          for (int c = 0; c < 3; ++c) s_sh[wv][k][c] = o.gc[c];
#pragma unroll
          for (int c = 0; c < 16; ++c) s_sh[wv][k][3 + c] = o.basis[c];
        }
      }
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(need_sh);
      __builtin_amdgcn_wave_barrier();
      const int n = __popcll(mask);
      const int slot = lane / 12, sub = lane - slot * 12;
      for (int k0 = 0; k0 < n; k0 += 5) {
        const int k = k0 + slot;
        if (slot < 5 && k < n) {
          const int rk = s_rows[wv][k];
          const int rlk = rk & ((1 << 30) - 1);
          const size_t op = (size_t)(a.t0 + rlk) * 12 + sub, os = (size_t)rlk * 12 + sub;
          float4 gi = make_float4(0.f, 0.f, 0.f, 0.f);
          if (rk & (1 << 30)) {                                  // column 4 sub + j = coefficient (4 sub + j) / 3, channel (4 sub + j) % 3
            const float* sh = s_sh[wv][k];
            const int c0 = 4 * sub;
            gi.x = sh[3 + c0 / 3] * sh[c0 % 3];
            gi.y = sh[3 + (c0 + 1) / 3] * sh[(c0 + 1) % 3];
            gi.z = sh[3 + (c0 + 2) / 3] * sh[(c0 + 2) % 3];
            gi.w = sh[3 + (c0 + 3) / 3] * sh[(c0 + 3) % 3];
          }
          const float4 pi = reinterpret_cast<const float4*>(a.shs)[op];
          float4 mi = reinterpret_cast<float4*>(a.m_shs)[os], vi = reinterpret_cast<float4*>(a.v_shs)[os], po;
          po.x = adam1(pi.x, gi.x, mi.x, vi.x, a.lr_shs[4 * sub], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
          po.y = adam1(pi.y, gi.y, mi.y, vi.y, a.lr_shs[4 * sub + 1], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
          po.z = adam1(pi.z, gi.z, mi.z, vi.z, a.lr_shs[4 * sub + 2], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
          po.w = adam1(pi.w, gi.w, mi.w, vi.w, a.lr_shs[4 * sub + 3], a.beta1, a.beta2, a.eps, a.bc1, a.bc2_sqrt);
          reinterpret_cast<float4*>(a.shs)[op] = po;
          reinterpret_cast<float4*>(a.m_shs)[os] = mi;
          reinterpret_cast<float4*>(a.v_shs)[os] = vi;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  if (a.live_counts) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { n_grad += (uint32_t)__shfl_xor((int)n_grad, off); n_step += (uint32_t)__shfl_xor((int)n_step, off); }
    if (lane == 0 && (n_grad | n_step)) { atomicAdd(&a.live_counts[0], n_grad); atomicAdd(&a.live_counts[1], n_step); }
  }
}

}  // namespace rtgs

extern "C" int rtgs_map_fused_tail(const rtgs_raster_settings* settings, const rtgs_map_step_args* s, void* geom_buffer,
                                   const void* image_buffer, const uint32_t* spec_fail, uint32_t* live_counts2, void* stream) {
  return rtgs_map_fused_tail_hint(settings, s, geom_buffer, image_buffer, spec_fail, live_counts2, 0, stream);
}

extern "C" int rtgs_map_fused_tail_hint(const rtgs_raster_settings* settings, const rtgs_map_step_args* s, void* geom_buffer,
                                        const void* image_buffer, const uint32_t* spec_fail, uint32_t* live_counts2,
                                        uint32_t listed_hint, void* stream) {
  using namespace rtgs;
  if (!settings || !s || !geom_buffer || !image_buffer) return RTGS_E_INVALID;
  const int32_t P = s->P;
  if (P <= 0 || s->sh_coeffs != 16 || s->step < 1) return RTGS_E_INVALID;
  int32_t t0 = 0, t1 = P;
  if (s->train_begin != 0 || s->train_end != 0) { t0 = s->train_begin; t1 = s->train_end; }   // only (0, 0) means every row
  if (t0 < 0 || t1 > P || t1 < t0) return RTGS_E_INVALID;
  if (t1 == t0) return RTGS_OK;                            // empty range: nothing to step
  FusedArgs a{};
  RasterParams& p = a.p;
  p.H = settings->image_height; p.W = settings->image_width;
  p.gx = (p.W + TILE - 1) / TILE; p.gy = (p.H + TILE - 1) / TILE;
  p.P = P; p.M = 16; p.deg = settings->sh_degree;
  p.tanfovx = settings->tanfovx; p.tanfovy = settings->tanfovy;
  p.fx = (float)(p.W / (2.0 * (double)settings->tanfovx));
  p.fy = (float)(p.H / (2.0 * (double)settings->tanfovy));
  p.cx = settings->cx > 0.f ? settings->cx : 0.5f * (float)(p.W - 1);
  p.cy = settings->cy > 0.f ? settings->cy : 0.5f * (float)(p.H - 1);
  p.scale_modifier = settings->scale_modifier;
  p.opaque_thr = settings->opaque_threshold; p.depth_thr = settings->depth_threshold; p.normal_thr = settings->normal_threshold;
  p.color_sigma = settings->color_sigma; p.T_thr = settings->T_threshold;
  p.view = settings->viewmatrix; p.campos = settings->campos; p.bg = settings->bg;
  p.spec_fail = spec_fail;
  size_t off[8];
  if (rtgs_raster_backward_buffers(P, p.H, p.W, off) != RTGS_OK) return RTGS_E_INVALID;
  const char* geom = (const char*)geom_buffer;
  const char* img = (const char*)image_buffer;
  a.opac = s->opacity; a.scales = s->scales; a.rots = s->rotations; a.normal_w = s->normal;
  a.clamped = (const uint8_t*)(geom + off[0]);
  a.gbase = (const uint32_t*)(geom + off[1]);
  a.slot_count = (uint32_t*)(geom + off[2]);
  a.info = (const BwdInfo*)(img + off[3]);
  a.records = (SplatGrad*)s->grad_scratch;
  a.touched = (uint8_t*)s->grad_scratch + off[4];
  a.xyz = s->xyz; a.shs = s->shs; a.raw8 = s->raw8;
  a.m_xyz = s->m_xyz; a.v_xyz = s->v_xyz; a.m_shs = s->m_shs; a.v_shs = s->v_shs; a.m_raw8 = s->m_raw8; a.v_raw8 = s->v_raw8;
  a.lr_xyz = s->lr_xyz; a.lr_shs = s->lr_shs; a.lr_raw8 = s->lr_raw8;
  a.ever_xyz = s->ever_xyz; a.ever_shs = s->ever_shs; a.ever_raw8 = s->ever_raw8;
  a.beta1 = s->beta1; a.beta2 = s->beta2; a.eps = s->eps;
  a.bc1 = 1.f - powf(s->beta1, (float)s->step);
  a.bc2_sqrt = sqrtf(1.f - powf(s->beta2, (float)s->step));
  if (s->attach) {
    if (!s->attach->init_xyz || !s->attach->init_raw8 || !s->attach->attach_info) return RTGS_E_INVALID;
    a.init_xyz = s->attach->init_xyz; a.init_raw8 = (const float4*)s->attach->init_raw8; a.attach_info = s->attach->attach_info;
  }
  a.confidence = s->confidence;
  a.act_opacity = s->opacity; a.act_scales = s->scales; a.act_normal = s->normal; a.act_rots = (float4*)s->rotations;
  a.live_counts = live_counts2;
  a.t0 = t0; a.rows = t1 - t0;
  // chunk size from the number of Gaussians the last verified forward LISTED for binning (the rows with gradient are among
  // them; 0 = unknown): the largest chunk that still expects at most one wave pass (64 live rows) per workgroup
  static const int forced = [] { const char* e = getenv("RTGS_FUSED_CHUNK"); return e ? atoi(e) : 0; }();
  int chunk = 1024;
  if (forced == 512 || forced == 1024 || forced == 2048) chunk = forced;
  else if (listed_hint > 0u) {
    const double per_row = (double)listed_hint / (double)(P > 0 ? P : 1);
    chunk = per_row * 2048.0 <= 64.0 ? 2048 : (per_row * 1024.0 <= 64.0 ? 1024 : 512);
  }
  if (forced == 64 || (forced == 0 && a.rows <= 65536)) chunk = 64;     // a small trainable set: spread thin (RPW 16)
  int blocks = (a.rows + chunk - 1) / chunk;
  if (blocks > 4096) blocks = 4096;
  if (chunk == 64) hipLaunchKernelGGL((map_fused_tail_kernel<64, 16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (chunk == 2048) hipLaunchKernelGGL(map_fused_tail_kernel<2048>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else if (chunk == 512) hipLaunchKernelGGL(map_fused_tail_kernel<512>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(map_fused_tail_kernel<1024>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? RTGS_OK : RTGS_E_HIP;
}
