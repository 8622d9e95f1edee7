// Backward kernels of the gfx950 rasterizer: per-tile front-to-back replay with wave64 DPP
// butterfly reductions (one LDS add per wave per entry, one global atomic per tile per quantity),
// and the per-Gaussian chain rule back to {means3D, opacities, shs, scales, rotations, normal_w}.
// Math: SURVEY.md Appendix B "Backward"; checked against autograd of oracle/raster_oracle.py.
#include "raster_common.h"
#include "adam_common.h"
#include "raster_chain.h"

namespace rtgs {

// ---------------------------------------------------------------------------------------------
// K7 blend_bwd: FRONT-TO-BACK replay.  Walking the tile list in the forward's own order makes
// T bit-identical to the forward (no T/(1-alpha) reconstruction), lets a wave leave as soon as
// its 64 pixels have passed their last contributor, and needs only the pixel's final colour:
//   S_k      = C_total - sum_{m<=k} c_m a_m T_m                  (colour behind entry k)
//   dL/da_k  = T_k (c_k . g) - (S_k . g) / (1 - a_k) - T_final (bg . g) / (1 - a_k)
// Per entry the 9 per-Gaussian partials are reduced inside the wave with a 3-step multi-value
// DPP butterfly (lanes end up holding one of 8 quantities, summed over their 8-lane group) and
// land in a wave-private LDS accumulator; the tile then stores ONE 64-byte gradient slot per touched
// (Gaussian, tile) pair with plain stores (one integer atomic picks the slot; grad_reduce sums a Gaussian's
// slots) - no global float atomics.  The opaque-depth gradient rides along: a pixel hands its four plane partials over when
// the walk reaches its owner's entry.  (Fallback when the slot space would be too large: one global
// atomic per (tile, Gaussian, quantity).)
// ---------------------------------------------------------------------------------------------

template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ float dpp_mov(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float xor4(float v) {     // value of lane l^4 (within a row of 16)
  float t = dpp_mov<0x104, 0xf, 0x5>(v, v);          // row_shl:4 into banks 0,2  (lane l <- l+4)
  return dpp_mov<0x114, 0xf, 0xa>(t, v);             // row_shr:4 into banks 1,3  (lane l <- l-4)
}

// in: v[0..7] per lane.  out: lane l holds, summed over its 8-lane group {l ^ 8, l ^ 4, l ^ 1 combinations}, quantity
// idx(l) = ((l >> 3) & 1) << 2 | ((l >> 2) & 1) << 1 | (l & 1).
// The two wide steps (4 and 2 results) exchange at lane distance 8 and 4, where DPP BANK masks (groups of four lanes)
// pick which half of a pair keeps which quantity: two v_add_f32_dpp per result and no v_cndmask (the quad_perm form of
// the same step costs three).  Hand-scheduled so that no DPP source is read within two wait states of its write.
__device__ __forceinline__ float butterfly8(const float (&v)[8], int lane) {
  float w0, w1, w2, w3, x0, x1;
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %8, %8 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %9, %9 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %3, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %0, %0 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %4, %2, %2 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      "v_add_f32_dpp %5, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
      "v_add_f32_dpp %5, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
      : "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&v"(x0), "=&v"(x1)
      : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
  const bool b0 = lane & 1;
  const float keep = b0 ? x1 : x0;
  const float send = b0 ? x0 : x1;
  return keep + dpp_mov<0xB1>(0.f, send);              // quad_perm [1,0,3,2]: lane l^1
}
// the 8 lanes of row 0 with (lane & 2) == 0 end up with the wave totals of the 8 butterfly quantities (lane l holds
// quantity idx(l)): fold the two groups of a row (lane ^ 2), the two rows of a half (xor 16) and the two halves
// (xor 32) - so the LDS store that follows has 8 DISTINCT addresses.
__device__ __forceinline__ float xor16_sum(float v) {     // v[l] + v[l ^ 16]: gfx950 v_permlane16_swap, a VALU op (no LDS pipe)
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float xor32_sum(float v) {     // v[l] + v[l ^ 32]: v_permlane32_swap
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_groups(float v) {
  v += dpp_mov<0x4E>(0.f, v);                                                        // quad_perm [2,3,0,1]: lane ^ 2
  return xor32_sum(xor16_sum(v));
}
// Wave totals of TWO values at once (the opacity partials of the two entries of a round): after the first exchange a
// lane holds the pair sum of d[lane & 1]; the remaining five steps only combine lanes of equal parity.  Every lane
// ends up with the wave total of d[lane & 1] - six cross-lane steps for both values instead of twelve.
__device__ __forceinline__ float wave_sum_pair(float d0, float d1, int lane) {
  const bool odd = lane & 1;
  float v = (odd ? d1 : d0) + dpp_mov<0xB1>(0.f, odd ? d0 : d1);                    // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_mov<0x4E>(0.f, v);                                                        // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_mov<0x124>(0.f, v);                                                       // row_ror:4
  v += dpp_mov<0x128>(0.f, v);                                                       // row_ror:8: all 8 same-parity lanes of the row
  return xor32_sum(xor16_sum(v));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  v += dpp_mov<0xB1>(0.f, v);               // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(0.f, v);               // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(0.f, v);              // row_half_mirror
  v += dpp_mov<0x140>(0.f, v);              // row_mirror: every lane of a row holds the row sum
  v += dpp_mov<0x142, 0xa>(0.f, v);         // row_bcast:15 into rows 1,3
  v += dpp_mov<0x143, 0xc>(0.f, v);         // row_bcast:31 into rows 2,3: lane 63 = wave sum
  return v;
}
__device__ __forceinline__ float group8_sum(float v) {   // every lane: sum over its 8-lane group
  v += dpp_mov<0xB1>(0.f, v);
  v += dpp_mov<0x4E>(0.f, v);
  v += xor4(v);
  return v;
}

// ---------------------------------------------------------------------------------------------
// Two walks, chosen PER TILE by blend_fwd (tile_mode, raster_common.h): tiles whose lists are shared by their 4x4
// blocks (large footprints: every block needs nearly every entry) take the TILE-UNIFORM strip walk below - one entry
// for all 64 lanes of a wave, 64-lane reductions, plain LDS stores; tiles whose blocks need a fraction of the list
// (a surface map of small discs) take the ROW-GRANULAR walk further down.
// ---------------------------------------------------------------------------------------------
constexpr int NGS = 9;   // du dv dca dcb dcc dr dg db | dop

__global__ void __launch_bounds__(256, 6) blend_bwd_strip_kernel(
    RasterParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const Splat* __restrict__ splats, const float* __restrict__ out_color, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const int32_t* __restrict__ depth_index,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const uint32_t* __restrict__ gbase, uint32_t* __restrict__ slot_count, const BwdInfo* __restrict__ info,
    SplatGrad* __restrict__ grads, uint8_t* __restrict__ touched, const uint32_t* __restrict__ tile_mode,
    uint32_t t0, uint32_t tn) {
  __shared__ float4 s_rec[BATCH * 3];           // u v ca cb | cc o r g | b hy id slot: three 16-B broadcast reads per entry
  __shared__ float s_grad[4 * BATCH * NGS];      // one private copy per wave: plain stores, no LDS atomics
  __shared__ float s_dep[BATCH * 4];            // depth-plane partials of the batch's entries (rare: LDS float adds)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int tile = blockIdx.y * p.gx + blockIdx.x;
  const int px = blockIdx.x * TILE + (tid & 15);
  const int py = blockIdx.y * TILE + (tid >> 4);
  const bool inside = px < p.W && py < p.H;
  const float pxf = (float)px, pyf = (float)py;
  if (spec_failed(p.spec_fail)) return;     // the forward's speculative sizes did not hold: the host redoes the step
  if ((tile_mode[tile] & 3u) != 0u) return;       // another walk has this tile (rows: block-sparse lists; MFMA)
  const uint2 range = ranges[tile];
  const size_t pix = (size_t)py * p.W + px;
  const size_t HW = (size_t)p.H * p.W;
  const int n = (int)(range.y - range.x);
  if (n == 0) return;      // nothing was blended here (or the tile belongs to the other pass of a two-pass forward)
  const bool use_slots = info->use_slots != 0;
  SplatGrad* const slot_grads = info->slot_grads;

  const uint32_t last = inside ? n_contrib[pix] : 0u;
  const float T_final = inside ? final_T[pix] : 0.f;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, S0 = 0.f, S1 = 0.f, S2 = 0.f;
  // opaque-surface depth: D = pd / (n_c . r); only the pixel's owner Gaussian receives it.  The four partials are
  // computed up front and handed over when the walk reaches the owner's entry (it is one of this pixel's contributors).
  int owner = -1;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (inside) {
    g0 = dL_dcolor[pix]; g1 = dL_dcolor[HW + pix]; g2 = dL_dcolor[2 * HW + pix];
    // colour behind the (not yet started) walk = everything the pixel accumulated, without background
    S0 = out_color[pix] - T_final * p.bg[0];
    S1 = out_color[HW + pix] - T_final * p.bg[1];
    S2 = out_color[2 * HW + pix] - T_final * p.bg[2];
    owner = depth_index[pix];
    const float gD = owner >= 0 ? dL_ddepth[pix] : 0.f;
    if (gD == 0.f) owner = -1;
    if (owner >= 0) {
      const float4 r2 = reinterpret_cast<const float4*>(splats + owner)[2];   // b nx ny nz
      const float pd = reinterpret_cast<const float*>(splats + owner)[12];
      const float rx = (pxf - p.cx) / p.fx, ry = (pyf - p.cy) / p.fy;
      const float den = r2.y * rx + r2.z * ry + r2.w;
      const float iden = 1.f / den;
      const float k = -gD * (pd * iden) * iden;
      a0 = k * rx; a1 = k * ry; a2 = k; a3 = gD * iden;
    }
  }
  const float bgT = T_final * (p.bg[0] * g0 + p.bg[1] * g1 + p.bg[2] * g2);
  float T = 1.f;
  const int gidx = (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | (lane & 1);   // butterfly8's quantity of this lane
  const bool gstore = (lane & 0x32) == 0;            // the 8 lanes that hold the wave totals after fold_groups
  float* const wgrad = s_grad + (tid >> 6) * BATCH * NGS;
  const float strip_y0 = (float)(blockIdx.y * TILE + (tid >> 6) * 4), strip_y1 = strip_y0 + 3.f;

  // only entries below the tile's largest `last` were blended by any pixel: stage and zero no more than that
  __shared__ unsigned int s_nmax;
  if (tid == 0) s_nmax = 0;
  __syncthreads();
  {
    unsigned int wl = last;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) wl = max(wl, (unsigned int)__shfl_xor((int)wl, off));
    if (lane == 0) atomicMax(&s_nmax, wl);
  }
  __syncthreads();
  const int nuse = min(n, (int)s_nmax);

  for (int base = 0; base < nuse; base += BATCH) {
    const int m = min(BATCH, nuse - base);
    __syncthreads();                                   // previous batch fully flushed before its LDS is reused
    if (tid < m) {
      const uint32_t id = point_list[range.x + base + tid];
      const float4* src = reinterpret_cast<const float4*>(splats + id);
      s_rec[tid * 3 + 0] = src[0];
      s_rec[tid * 3 + 1] = src[1];
      const float b = reinterpret_cast<const float*>(splats + id)[8];
      const float hy = reinterpret_cast<const float*>(splats + id)[15];
      const uint32_t slot0 = use_slots ? gbase[id] : 0u;       // first slot of the Gaussian's run
      s_rec[tid * 3 + 2] = make_float4(b, hy, __uint_as_float(id), __uint_as_float(slot0));
    }
    for (int q = tid; q < 4 * BATCH * NGS; q += BLOCK)
      if ((q % (BATCH * NGS)) < m * NGS) s_grad[q] = 0.f;
    for (int q = tid; q < BATCH * 4; q += BLOCK) s_dep[q] = 0.f;
    __syncthreads();

    // Two entries per round: records, alphas and the two 8-value butterflies are independent instruction
    // streams (the single-wave dependent chain, not ALU throughput, bounds this walk); only the
    // T / S recurrences are sequential.
    for (int j = 0; j < m; j += 2) {
      if (__builtin_amdgcn_ballot_w64((uint32_t)(base + j) < last) == 0ull) break;   // wave past its last contributor
      float4 r0[2], r1[2], r2[2];
      float dx[2], dy[2], G[2], alpha[2];
      bool valid[2], own[2];
      int e[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        e[k] = min(j + k, m - 1);                                 // past the end: re-read the last entry, masked out
        r0[k] = s_rec[e[k] * 3 + 0];
        r1[k] = s_rec[e[k] * 3 + 1];
        r2[k] = s_rec[e[k] * 3 + 2];
        const float ehy = r2[k].y;
        const bool in_batch = j + k < m;
        const bool live = in_batch & !((r0[k].y + ehy < strip_y0) | (r0[k].y - ehy > strip_y1));     // wave-uniform, branch-free
        dx[k] = r0[k].x - pxf; dy[k] = r0[k].y - pyf;
        const float power = splat_power(r0[k].z, r0[k].w, r1[k].x, dx[k], dy[k]);
        G[k] = splat_exp(fminf(power, 0.f));
        alpha[k] = fminf(0.99f, r1[k].y * G[k]);
        valid[k] = live & ((uint32_t)(base + j + k) < last) & !(power > 0.f) & !(alpha[k] < 1.f / 255.f);
        own[k] = in_batch & (owner == (int)__float_as_uint(r2[k].z));
      }
      const unsigned long long vm0 = __builtin_amdgcn_ballot_w64(valid[0]);
      const unsigned long long vm1 = __builtin_amdgcn_ballot_w64(valid[1]);
      if ((vm0 | vm1) == 0ull) continue;                          // wave-uniform
      float gda[2], w[2], Tk[2], Sg[2], cg[2], ia[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {                               // sequential part: T and the colour behind
        const float c0 = r1[k].z, c1 = r1[k].w, c2 = r2[k].x;
        w[k] = valid[k] ? alpha[k] * T : 0.f;
        S0 -= c0 * w[k]; S1 -= c1 * w[k]; S2 -= c2 * w[k];       // colour strictly behind this entry
        Tk[k] = T;
        const float oma = 1.f - alpha[k];
        ia[k] = __builtin_amdgcn_rcpf(oma);
        T = valid[k] ? T * oma : T;
        Sg[k] = S0 * g0 + S1 * g1 + S2 * g2 + bgT;
        cg[k] = c0 * g0 + c1 * g1 + c2 * g2;
      }
      float r8[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float dL_dalpha = Tk[k] * cg[k] - Sg[k] * ia[k];
        gda[k] = valid[k] ? G[k] * dL_dalpha : 0.f;   // clamp of alpha is transparent in the backward (upstream 3DGS)
        const float gdl = gda[k] * r1[k].y;
        // moments of gdl over the pixels: the conic enters once per entry (in the flush), not once per pixel
        float v[8];
        v[0] = gdl * dx[k];
        v[1] = gdl * dy[k];
        v[2] = v[0] * dx[k];
        v[3] = v[0] * dy[k];
        v[4] = v[1] * dy[k];
        v[5] = w[k] * g0; v[6] = w[k] * g1; v[7] = w[k] * g2;
        r8[k] = butterfly8(v, lane);
      }
      const float ro = wave_sum_pair(gda[0], gda[1], lane);
      // every (wave, entry, quantity) slot is written at most once per batch -> plain LDS stores
      if (vm0) {
        const float t8 = fold_groups(r8[0]);
        if (gstore) wgrad[e[0] * NGS + gidx] = t8;
        if (lane == 0) wgrad[e[0] * NGS + 8] = ro;
      }
      if (vm1) {
        const float t8 = fold_groups(r8[1]);
        if (gstore) wgrad[e[1] * NGS + gidx] = t8;
        if (lane == 1) wgrad[e[1] * NGS + 8] = ro;
      }
      // depth owners among this wave's pixels (each pixel owns at most one entry of the whole list: rare per round)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        if (__builtin_amdgcn_ballot_w64(own[k]) != 0ull) {
          const float s0 = wave_sum_to_lane63(own[k] ? a0 : 0.f);
          const float s1 = wave_sum_to_lane63(own[k] ? a1 : 0.f);
          const float s2 = wave_sum_to_lane63(own[k] ? a2 : 0.f);
          const float s3 = wave_sum_to_lane63(own[k] ? a3 : 0.f);
          if (lane == 63) {
            atomicAdd(&s_dep[e[k] * 4 + 0], s0); atomicAdd(&s_dep[e[k] * 4 + 1], s1);
            atomicAdd(&s_dep[e[k] * 4 + 2], s2); atomicAdd(&s_dep[e[k] * 4 + 3], s3);
          }
        }
      }
    }
    __syncthreads();
    if (tid < m) {
      float t[NGS + 4];
      bool any = false;
#pragma unroll
      for (int k = 0; k < NGS; ++k) {
        t[k] = (s_grad[tid * NGS + k] + s_grad[(BATCH + tid) * NGS + k]) +
               (s_grad[(2 * BATCH + tid) * NGS + k] + s_grad[(3 * BATCH + tid) * NGS + k]);
        any |= (t[k] != 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { t[NGS + k] = s_dep[tid * 4 + k]; any |= (t[NGS + k] != 0.f); }
      // a frozen row (id outside [t0, t0 + tn): rendered, never differentiated - the stable part of the map) takes no slot
      if (any && __float_as_uint(s_rec[tid * 3 + 2].z) - t0 < tn) {
        // t[0..4] hold the moments sum(gdl dx), sum(gdl dy), sum(gdl dx^2), sum(gdl dx dy), sum(gdl dy^2) with
        // d = centre - pixel; d alpha / d(u, v, conic) of G = exp(-1/2 (ca dx^2 + cc dy^2) - cb dx dy):
        {
          const float4 q0 = s_rec[tid * 3 + 0];            // u v ca cb
          const float ccn = s_rec[tid * 3 + 1].x;          // cc
          const float mx = t[0], my = t[1];
          t[0] = -(q0.z * mx + q0.w * my);                 // du
          t[1] = -(ccn * my + q0.w * mx);                  // dv
          t[2] = -0.5f * t[2];                             // dca
          t[3] = -t[3];                                    // dcb
          t[4] = -0.5f * t[4];                             // dcc
        }
        const uint32_t gid = __float_as_uint(s_rec[tid * 3 + 2].z);
        touched[gid] = 1;     // byte per Gaussian: grad_reduce / the row-state backward skip untouched Gaussians
        if (use_slots) {
          // SplatGrad order: du dv dca dcb | dcc dop dr dg | db dnx dny dnz | dpd - - -
          // next free slot of the run: at most one tile per rect tile asks, so the run (= rect area) cannot overflow
          const uint32_t slot = __float_as_uint(s_rec[tid * 3 + 2].w) + atomicAdd(&slot_count[gid], 1u);
          float4* dst = reinterpret_cast<float4*>(slot_grads + slot);
          dst[0] = make_float4(t[0], t[1], t[2], t[3]);
          dst[1] = make_float4(t[4], t[8], t[5], t[6]);
          dst[2] = make_float4(t[7], t[9], t[10], t[11]);
          dst[3] = make_float4(t[12], 0.f, 0.f, 0.f);
        } else {
          float* dst = reinterpret_cast<float*>(grads + gid);
          if (t[0] != 0.f) unsafeAtomicAdd(dst + 0, t[0]);
          if (t[1] != 0.f) unsafeAtomicAdd(dst + 1, t[1]);
          if (t[2] != 0.f) unsafeAtomicAdd(dst + 2, t[2]);
          if (t[3] != 0.f) unsafeAtomicAdd(dst + 3, t[3]);
          if (t[4] != 0.f) unsafeAtomicAdd(dst + 4, t[4]);
          if (t[8] != 0.f) unsafeAtomicAdd(dst + 5, t[8]);
          if (t[5] != 0.f) unsafeAtomicAdd(dst + 6, t[5]);
          if (t[6] != 0.f) unsafeAtomicAdd(dst + 7, t[6]);
          if (t[7] != 0.f) unsafeAtomicAdd(dst + 8, t[7]);
          if (t[9] != 0.f) unsafeAtomicAdd(dst + 9, t[9]);
          if (t[10] != 0.f) unsafeAtomicAdd(dst + 10, t[10]);
          if (t[11] != 0.f) unsafeAtomicAdd(dst + 11, t[11]);
          if (t[12] != 0.f) unsafeAtomicAdd(dst + 12, t[12]);
        }
      }
    }
  }
}


constexpr int NACC = 13;                 // per-entry accumulator: 5 moments, 3 colour, opacity | 4 depth-plane partials
constexpr int ACC_STRIDE = 16;           // floats per accumulator row (one 64-B line: the address is a shift)
constexpr int BWD_CHUNKS = BATCH / 32;   // 32-entry words of a block's sub-list
#ifndef RTGS_BWD_U
#define RTGS_BWD_U 1
#endif
constexpr int BU = RTGS_BWD_U;           // entries a row takes per pass (independent instruction streams; only T / S are sequential)

// Row totals (16 lanes) of TWO values at once: after the first exchange a lane holds the pair sum of d[lane & 1]; the
// remaining three steps only combine lanes of equal parity.  Every lane ends up with the row total of d[lane & 1].
__device__ __forceinline__ float row_sum_pair(float d0, float d1, int lane) {
  const bool odd = lane & 1;
  float v = (odd ? d1 : d0) + dpp_mov<0xB1>(0.f, odd ? d0 : d1);                    // quad_perm [1,0,3,2]: lane ^ 1
  v += dpp_mov<0x4E>(0.f, v);                                                        // quad_perm [2,3,0,1]: lane ^ 2
  v += dpp_mov<0x124>(0.f, v);                                                       // row_ror:4
  v += dpp_mov<0x128>(0.f, v);                                                       // row_ror:8
  return v;
}

// ROW-GRANULAR walk (see blend_fwd): a wave owns an 8x8 quadrant of the tile, each DPP row of 16 lanes a 4x4 pixel
// block that walks its own sub-list of the staged batch (bit masks built by the staging threads), BU entries per pass.
// The per-entry partials are reduced inside the row with DPP only - the 3-step multi-value butterfly leaves 8
// quantities on 8 lanes, one more quad step completes the 16-lane sum - and reach the entries' LDS accumulators with
// ds_add_f32 (the 8 quantities of the pass's first entry ride on the lanes with (lane & 2) == 0, those of the second
// on the others: one instruction; a second one carries the two opacity sums).  No cross-row step, no per-wave copies.
__global__ void __launch_bounds__(256, 6) blend_bwd_rows_kernel(
    RasterParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const Splat* __restrict__ splats, const float* __restrict__ out_color, const float* __restrict__ final_T,
    const uint32_t* __restrict__ n_contrib, const int32_t* __restrict__ depth_index,
    const float* __restrict__ dL_dcolor, const float* __restrict__ dL_ddepth,
    const uint32_t* __restrict__ gbase, uint32_t* __restrict__ slot_count, const BwdInfo* __restrict__ info,
    SplatGrad* __restrict__ grads, uint8_t* __restrict__ touched, const uint32_t* __restrict__ tile_mode,
    uint32_t t0, uint32_t tn) {
  __shared__ float4 s_rec[BATCH * 4];           // u v ca cb | cc o r g | b id slot - | - : one 64-B line per entry
  __shared__ float s_acc[BATCH * ACC_STRIDE];   // per-entry partial sums of the tile (LDS float adds)
  __shared__ uint32_t s_live[16][BWD_CHUNKS];   // per 4x4 block: the staged entries that reach it

  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int tile = blockIdx.y * p.gx + blockIdx.x;
  if (spec_failed(p.spec_fail)) return;     // the forward's speculative sizes did not hold: the host redoes the step
  if ((tile_mode[tile] & 3u) != 1u) return;       // another walk has this tile (strip: shared lists; MFMA)
  const int bx = ((wv & 1) << 1) | ((lane >> 4) & 1), by = (wv & 2) | (lane >> 5);
  const int blk = by * 4 + bx;
  const int px = blockIdx.x * TILE + bx * 4 + (lane & 3);
  const int py = blockIdx.y * TILE + by * 4 + ((lane >> 2) & 3);
  const int rsh = lane & 48;
  const bool inside = px < p.W && py < p.H;
  const float pxf = (float)px, pyf = (float)py;
  const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
  const uint2 range = ranges[tile];
  const size_t pix = (size_t)py * p.W + px;
  const size_t HW = (size_t)p.H * p.W;
  const int n = (int)(range.y - range.x);
  if (n == 0) return;      // nothing was blended here (or the tile belongs to the other pass of a two-pass forward)
  const bool use_slots = info->use_slots != 0;
  SplatGrad* const slot_grads = info->slot_grads;

  const uint32_t last = inside ? n_contrib[pix] : 0u;
  const float T_final = inside ? final_T[pix] : 0.f;
  float g0 = 0.f, g1 = 0.f, g2 = 0.f, S0 = 0.f, S1 = 0.f, S2 = 0.f;
  // opaque-surface depth: D = pd / (n_c . r); only the pixel's owner Gaussian receives it.  The four partials are
  // computed up front and handed over when the walk reaches the owner's entry (it is one of this pixel's contributors).
  int owner = -1;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
  if (inside) {
    g0 = dL_dcolor[pix]; g1 = dL_dcolor[HW + pix]; g2 = dL_dcolor[2 * HW + pix];
    // colour behind the (not yet started) walk = everything the pixel accumulated, without background
    S0 = out_color[pix] - T_final * p.bg[0];
    S1 = out_color[HW + pix] - T_final * p.bg[1];
    S2 = out_color[2 * HW + pix] - T_final * p.bg[2];
    owner = depth_index[pix];
    const float gD = owner >= 0 ? dL_ddepth[pix] : 0.f;
    if (gD == 0.f) owner = -1;
    if (owner >= 0) {
      const float4 r2 = reinterpret_cast<const float4*>(splats + owner)[2];   // b nx ny nz
      const float pd = reinterpret_cast<const float*>(splats + owner)[12];
      const float rx = (pxf - p.cx) / p.fx, ry = (pyf - p.cy) / p.fy;
      const float den = r2.y * rx + r2.z * ry + r2.w;
      const float iden = 1.f / den;
      const float k = -gD * (pd * iden) * iden;
      a0 = k * rx; a1 = k * ry; a2 = k; a3 = gD * iden;
    }
  }
  const float bgT = T_final * (p.bg[0] * g0 + p.bg[1] * g1 + p.bg[2] * g2);
  float T = 1.f;
  const int gidx = (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 1) | (lane & 1);   // butterfly8's quantity of this lane
  const bool second = (lane & 2) != 0;               // this lane carries the pass's second entry to LDS (BU == 2)

  // a block walks no further than its last contributor; the tile stages no further than its largest
  uint32_t row_last = last;
  row_last = max(row_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)row_last, 0xB1, 0xf, 0xf, false));    // lane ^ 1
  row_last = max(row_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)row_last, 0x4E, 0xf, 0xf, false));    // lane ^ 2
  row_last = max(row_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)row_last, 0x124, 0xf, 0xf, false));   // row_ror:4
  row_last = max(row_last, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)row_last, 0x128, 0xf, 0xf, false));   // row_ror:8
  __shared__ unsigned int s_nmax;
  if (tid == 0) s_nmax = 0;
  __syncthreads();
  {
    unsigned int wl = row_last;
    wl = max(wl, (unsigned int)__shfl_xor((int)wl, 16));
    wl = max(wl, (unsigned int)__shfl_xor((int)wl, 32));
    if (lane == 0) atomicMax(&s_nmax, wl);
  }
  __syncthreads();
  const int nuse = min(n, (int)s_nmax);

  for (int base = 0; base < nuse; base += BATCH) {
    const int m = min(BATCH, nuse - base);
    __syncthreads();                                   // previous batch fully flushed before its LDS is reused
    if (wv < (BATCH + 63) / 64) {                      // the staging waves (wave-uniform: they ballot)
      uint32_t reach = 0;
      if (tid < m) {
        const uint32_t id = point_list[range.x + base + tid];
        const float4* src = reinterpret_cast<const float4*>(splats + id);
        const float4 q0 = src[0];
        s_rec[tid * 4 + 0] = q0;
        const float4 q1 = src[1];
        s_rec[tid * 4 + 1] = q1;
        const float b = reinterpret_cast<const float*>(splats + id)[8];
        const float2 hxy = reinterpret_cast<const float2*>(splats + id)[7];
        const uint32_t slot0 = use_slots ? gbase[id] : 0u;       // first slot of the Gaussian's run
        s_rec[tid * 4 + 2] = make_float4(b, __uint_as_float(id), __uint_as_float(slot0), 0.f);
        reach = blocks_reached(q0.x, q0.y, hxy.x, hxy.y, q0.z, q0.w, q1.x, q1.y, tx0, ty0);
      }
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64((reach >> b) & 1u);
        if (lane == 0) { s_live[b][2 * wv] = (uint32_t)bal; s_live[b][2 * wv + 1] = (uint32_t)(bal >> 32); }
      }
    }
    for (int q = tid; q < m * ACC_STRIDE; q += BLOCK) s_acc[q] = 0.f;
    __syncthreads();

    const int nch = (m + 31) >> 5;
    int c = -1;
    uint32_t cur = 0u;                               // row-uniform: the unread part of the current word of the sub-list
    for (;;) {
      int e[BU];
      bool has[BU];
#pragma unroll
      for (int k = 0; k < BU; ++k) {
        while (cur == 0u && c + 1 < nch) { ++c; cur = s_live[blk][c]; }
        e[k] = (c << 5) + (cur != 0u ? __builtin_ctz(cur) : 0);
        has[k] = cur != 0u && (uint32_t)(base + e[k]) < row_last;      // positions increase: past row_last the block is through
        cur = has[k] ? (cur & (cur - 1u)) : 0u;
        if (!has[k]) { e[k] = 0; c = nch; }
      }
      if (__builtin_amdgcn_ballot_w64(has[0]) == 0ull) break;
      float4 r0[BU], r1[BU];
      float2 r2[BU];
      float dx[BU], dy[BU], G[BU], alpha[BU];
      bool valid[BU];
#pragma unroll
      for (int k = 0; k < BU; ++k) {
        r0[k] = s_rec[e[k] * 4 + 0];
        r1[k] = s_rec[e[k] * 4 + 1];
        r2[k] = *reinterpret_cast<const float2*>(&s_rec[e[k] * 4 + 2]);      // b id
        dx[k] = r0[k].x - pxf; dy[k] = r0[k].y - pyf;
        const float power = splat_power(r0[k].z, r0[k].w, r1[k].x, dx[k], dy[k]);
        G[k] = splat_exp(fminf(power, 0.f));
        alpha[k] = fminf(0.99f, r1[k].y * G[k]);
        valid[k] = has[k] & ((uint32_t)(base + e[k]) < last) & !(power > 0.f) & !(alpha[k] < 1.f / 255.f);
      }
      unsigned long long vm[BU], vany = 0ull;
#pragma unroll
      for (int k = 0; k < BU; ++k) { vm[k] = __builtin_amdgcn_ballot_w64(valid[k]); vany |= vm[k]; }
      if (vany == 0ull) continue;                                   // wave-uniform
      float gda[BU], r8[BU];
#pragma unroll
      for (int k = 0; k < BU; ++k) {                               // sequential part: T and the colour behind
        const float c0 = r1[k].z, c1 = r1[k].w, c2 = r2[k].x;
        const float w = valid[k] ? alpha[k] * T : 0.f;
        S0 -= c0 * w; S1 -= c1 * w; S2 -= c2 * w;                   // colour strictly behind this entry
        const float oma = 1.f - alpha[k];
        const float dL_dalpha = T * (c0 * g0 + c1 * g1 + c2 * g2) - (S0 * g0 + S1 * g1 + S2 * g2 + bgT) * __builtin_amdgcn_rcpf(oma);
        T = valid[k] ? T * oma : T;
        gda[k] = valid[k] ? G[k] * dL_dalpha : 0.f;                 // clamp of alpha is transparent in the backward (upstream 3DGS)
        const float gdl = gda[k] * r1[k].y;
        // moments of gdl over the pixels: the conic enters once per entry (in the flush), not once per pixel
        float v[8];
        v[0] = gdl * dx[k];
        v[1] = gdl * dy[k];
        v[2] = v[0] * dx[k];
        v[3] = v[0] * dy[k];
        v[4] = v[1] * dy[k];
        v[5] = w * g0; v[6] = w * g1; v[7] = w * g2;
        r8[k] = butterfly8(v, lane);                                // sums over the lane's 8-lane group
        r8[k] += dpp_mov<0x4E>(0.f, r8[k]);                         // quad_perm [2,3,0,1]: the other group of the row
      }
      if constexpr (BU == 2) {
        const float ro = row_sum_pair(gda[0], gda[1], lane);        // lane parity = entry
        const bool rv0 = (uint32_t)((vm[0] >> rsh) & 0xffffull) != 0u, rv1 = (uint32_t)((vm[1] >> rsh) & 0xffffull) != 0u;
        if (second ? rv1 : rv0) atomicAdd(&s_acc[(second ? e[1] : e[0]) * ACC_STRIDE + gidx], second ? r8[1] : r8[0]);
        if ((lane & 14) == 0 && ((lane & 1) ? rv1 : rv0)) atomicAdd(&s_acc[((lane & 1) ? e[1] : e[0]) * ACC_STRIDE + 8], ro);
      } else {
        float ro = gda[0] + dpp_mov<0xB1>(0.f, gda[0]);
        ro += dpp_mov<0x4E>(0.f, ro);
        ro += dpp_mov<0x124>(0.f, ro);
        ro += dpp_mov<0x128>(0.f, ro);
        const bool rv0 = (uint32_t)((vm[0] >> rsh) & 0xffffull) != 0u;
        if (rv0 && (!second || (lane & 15) == 2)) atomicAdd(&s_acc[e[0] * ACC_STRIDE + (second ? 8 : gidx)], second ? ro : r8[0]);
      }
      // depth owners (each pixel owns at most one entry of the whole list): straight to the accumulator
#pragma unroll
      for (int k = 0; k < BU; ++k)
        if (has[k] && owner == (int)__float_as_uint(r2[k].y)) {
          float* const acc = &s_acc[e[k] * ACC_STRIDE];
          atomicAdd(acc + 9, a0); atomicAdd(acc + 10, a1); atomicAdd(acc + 11, a2); atomicAdd(acc + 12, a3);
        }
    }
    __syncthreads();
    if (tid < m) {
      float t[NACC];
      bool any = false;
#pragma unroll
      for (int k = 0; k < NACC; ++k) { t[k] = s_acc[tid * ACC_STRIDE + k]; any |= (t[k] != 0.f); }
      if (any && __float_as_uint(s_rec[tid * 4 + 2].y) - t0 < tn) {     // frozen rows take no slot
        // t[0..4] hold the moments sum(gdl dx), sum(gdl dy), sum(gdl dx^2), sum(gdl dx dy), sum(gdl dy^2) with
        // d = centre - pixel; d alpha / d(u, v, conic) of G = exp(-1/2 (ca dx^2 + cc dy^2) - cb dx dy):
        {
          const float4 q0 = s_rec[tid * 4 + 0];            // u v ca cb
          const float ccn = s_rec[tid * 4 + 1].x;          // cc
          const float mx = t[0], my = t[1];
          t[0] = -(q0.z * mx + q0.w * my);                 // du
          t[1] = -(ccn * my + q0.w * mx);                  // dv
          t[2] = -0.5f * t[2];                             // dca
          t[3] = -t[3];                                    // dcb
          t[4] = -0.5f * t[4];                             // dcc
        }
        const uint32_t gid = __float_as_uint(s_rec[tid * 4 + 2].y);
        touched[gid] = 1;     // byte per Gaussian: grad_reduce / the row-state backward skip untouched Gaussians
        if (use_slots) {
          // SplatGrad order: du dv dca dcb | dcc dop dr dg | db dnx dny dnz | dpd - - -
          // next free slot of the run: at most one tile per rect tile asks, so the run (= rect area) cannot overflow
          const uint32_t slot = __float_as_uint(s_rec[tid * 4 + 2].z) + atomicAdd(&slot_count[gid], 1u);
          float4* dst = reinterpret_cast<float4*>(slot_grads + slot);
          dst[0] = make_float4(t[0], t[1], t[2], t[3]);
          dst[1] = make_float4(t[4], t[8], t[5], t[6]);
          dst[2] = make_float4(t[7], t[9], t[10], t[11]);
          dst[3] = make_float4(t[12], 0.f, 0.f, 0.f);
        } else {
          float* dst = reinterpret_cast<float*>(grads + gid);
          if (t[0] != 0.f) unsafeAtomicAdd(dst + 0, t[0]);
          if (t[1] != 0.f) unsafeAtomicAdd(dst + 1, t[1]);
          if (t[2] != 0.f) unsafeAtomicAdd(dst + 2, t[2]);
          if (t[3] != 0.f) unsafeAtomicAdd(dst + 3, t[3]);
          if (t[4] != 0.f) unsafeAtomicAdd(dst + 4, t[4]);
          if (t[8] != 0.f) unsafeAtomicAdd(dst + 5, t[8]);
          if (t[5] != 0.f) unsafeAtomicAdd(dst + 6, t[5]);
          if (t[6] != 0.f) unsafeAtomicAdd(dst + 7, t[6]);
          if (t[7] != 0.f) unsafeAtomicAdd(dst + 8, t[7]);
          if (t[9] != 0.f) unsafeAtomicAdd(dst + 9, t[9]);
          if (t[10] != 0.f) unsafeAtomicAdd(dst + 10, t[10]);
          if (t[11] != 0.f) unsafeAtomicAdd(dst + 11, t[11]);
          if (t[12] != 0.f) unsafeAtomicAdd(dst + 12, t[12]);
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Gradient slots (BwdInfo, raster_common.h): grad_reduce sums, for every Gaussian a tile flagged as touched, the
// slots its run holds into its SplatGrad record.  16 lanes per slot, lane = component of the 64-byte record: a slot is
// one coalesced 64-B read and the sum needs no cross-lane step inside a group.  Runs of up to 8 slots (the common case
// on a surface map) are summed by one group, four Gaussians in flight per wave; longer runs (near, screen-filling
// Gaussians: dozens to hundreds of tiles) get the whole wave, four slots per step.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) grad_reduce_kernel(int P, const uint8_t* __restrict__ touched,
                                                          const uint32_t* __restrict__ gbase,
                                                          uint32_t* __restrict__ count,
                                                          const BwdInfo* __restrict__ info,
                                                          SplatGrad* __restrict__ grads, const uint32_t* __restrict__ spec_fail) {
  if (spec_failed(spec_fail)) return;
  if (info->use_slots == 0) return;
  __shared__ uint8_t s_small[4][64], s_big[4][64];
  const float* __restrict__ slots = reinterpret_cast<const float*>(info->slot_grads);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int grp = lane >> 4, c = lane & 15;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const long long wave0 = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
  for (long long base = wave0 * 64; base < P; base += nwaves * 64) {
    const long long i = base + lane;
    const bool t = i < P && touched[i] != 0;
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(t);
    if (mask == 0ull) continue;
    const bool big = t && count[i] > 8u;
    const unsigned long long mbig = __builtin_amdgcn_ballot_w64(big), msmall = mask & ~mbig;
    if (t) {
      if (big) s_big[wv][__popcll(mbig & lt)] = (uint8_t)lane;
      else s_small[wv][__popcll(msmall & lt)] = (uint8_t)lane;
    }
    __builtin_amdgcn_wave_barrier();
    const int nbig = __popcll(mbig), nsmall = __popcll(msmall);
    for (int k = 0; k < nbig; ++k) {
      // long run (a near, screen-filling Gaussian: hundreds of tiles): four lanes per slot (one 16-B load each), sixteen
      // slots per load instruction, four instructions in flight - the run is the kernel's critical path, so what counts
      // is bytes in flight per wave, not lanes per slot
      const size_t id = (size_t)(base + s_big[wv][k]);
      const size_t b0 = gbase[id];
      const uint32_t n = count[id];
      if (n <= 48u) {        // medium run: sixteen lanes per slot, four slots per step, two shuffles at the end
        float m0 = 0.f, m1 = 0.f;
        uint32_t q = grp;
        for (; q + 4 < n; q += 8) { m0 += slots[(b0 + q) * 16 + c]; m1 += slots[(b0 + q + 4) * 16 + c]; }
        if (q < n) m0 += slots[(b0 + q) * 16 + c];
        float acc = m0 + m1;
        acc += __shfl_xor(acc, 16);
        acc += __shfl_xor(acc, 32);
        if (lane < 16) reinterpret_cast<float*>(grads + id)[c] = acc;
        if (lane == 0) count[id] = 0u;
        continue;
      }
      const float4* __restrict__ s4 = reinterpret_cast<const float4*>(slots);
      const uint32_t sub = (uint32_t)lane & 3u, sg = (uint32_t)lane >> 2;
      float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0, a2 = a0, a3 = a0;
      uint32_t q = sg;
      for (; q + 48 < n; q += 64) {
        const float4 t0 = s4[(b0 + q) * 4 + sub], t1 = s4[(b0 + q + 16) * 4 + sub];
        const float4 t2 = s4[(b0 + q + 32) * 4 + sub], t3 = s4[(b0 + q + 48) * 4 + sub];
        a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
        a1.x += t1.x; a1.y += t1.y; a1.z += t1.z; a1.w += t1.w;
        a2.x += t2.x; a2.y += t2.y; a2.z += t2.z; a2.w += t2.w;
        a3.x += t3.x; a3.y += t3.y; a3.z += t3.z; a3.w += t3.w;
      }
      for (; q < n; q += 16) {
        const float4 t0 = s4[(b0 + q) * 4 + sub];
        a0.x += t0.x; a0.y += t0.y; a0.z += t0.z; a0.w += t0.w;
      }
      float4 acc = make_float4((a0.x + a1.x) + (a2.x + a3.x), (a0.y + a1.y) + (a2.y + a3.y), (a0.z + a1.z) + (a2.z + a3.z),
                               (a0.w + a1.w) + (a2.w + a3.w));
#pragma unroll
      for (int off = 4; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off); acc.y += __shfl_xor(acc.y, off);
        acc.z += __shfl_xor(acc.z, off); acc.w += __shfl_xor(acc.w, off);
      }
      if (lane < 4) reinterpret_cast<float4*>(grads + id)[sub] = acc;
      if (lane == 0) count[id] = 0u;          // the counters are zero between calls (the forward clears them too)
    }
    for (int k0 = 0; k0 < nsmall; k0 += 4) {
      const int k = k0 + grp;
      const bool act = k < nsmall;
      const size_t id = (size_t)(base + (act ? s_small[wv][k] : 0));
      const size_t b0 = act ? gbase[id] : 0u;
      const uint32_t n = act ? count[id] : 0u;
      float acc = 0.f;
      for (uint32_t q = 0; q < n; ++q) acc += slots[(b0 + q) * 16 + c];
      if (act) reinterpret_cast<float*>(grads + id)[c] = acc;
      if (act && c == 0) count[id] = 0u;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// K8 preprocess_bwd: one lane per Gaussian; recomputes the forward intermediates from the
// inputs (cheaper than storing them) and applies the chain rule.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void preprocess_bwd_one(
    const RasterParams& p, const int i, const float* __restrict__ means, const float* __restrict__ opac,
    const float* __restrict__ shs, const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ normal_w, const int32_t* __restrict__ radii,
    const uint8_t* __restrict__ clamped, SplatGrad* __restrict__ grads, uint8_t* __restrict__ touched_b,
    uint8_t* __restrict__ row_state,
    float* __restrict__ d_means, float* __restrict__ d_opac, float* __restrict__ d_shs,
    float* __restrict__ d_scales, float* __restrict__ d_rots, float* __restrict__ d_normal) {
  float* dsh = d_shs + (size_t)i * p.M * 3;
  // Row-state mode (rtgs_raster_backward_rows): the gradient buffers and the SplatGrad scratch persist between
  // calls and rows whose state is not 1 are already zero, so an untouched Gaussian costs two byte reads here.
  // Dense mode (row_state == nullptr): every row is written, as the reference's backward does.
  SplatGrad g;
  bool touched;
  uint8_t old_state = 1;
  if (row_state) {
    old_state = row_state[i];
    touched = touched_b[i] != 0;
    if (touched) {
      touched_b[i] = 0;
      g = grads[i];
      float4* z = reinterpret_cast<float4*>(grads + i);        // leave the scratch zero for the next call
      z[0] = z[1] = z[2] = z[3] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  } else {
    touched = radii[i] > 0;
    if (touched) g = grads[i];
  }
  if (touched) {
    touched = (g.du != 0.f) | (g.dv != 0.f) | (g.dca != 0.f) | (g.dcb != 0.f) | (g.dcc != 0.f) | (g.dop != 0.f) |
              (g.dr != 0.f) | (g.dg != 0.f) | (g.db != 0.f) | (g.dnx != 0.f) | (g.dny != 0.f) | (g.dnz != 0.f) |
              (g.dpd != 0.f);
  }
  if (row_state) {
    if (touched) row_state[i] = 1;
    else if (old_state != 0) row_state[i] = old_state == 1 ? 2 : 0;     // 2: zeroed by THIS call (consumers clean too)
    if (!touched && old_state != 1) return;                              // row is already zero
  }
  if (!touched) {   // exact zeros for Gaussians that reached no pixel (mapper.py:455)
    d_means[3 * i] = d_means[3 * i + 1] = d_means[3 * i + 2] = 0.f;
    d_opac[i] = 0.f;
    if (p.M == 16) {
      float4* d4 = reinterpret_cast<float4*>(dsh);
#pragma unroll
      for (int q = 0; q < 12; ++q) d4[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      for (int k = 0; k < p.M * 3; ++k) dsh[k] = 0.f;
    }
    d_scales[3 * i] = d_scales[3 * i + 1] = d_scales[3 * i + 2] = 0.f;
    reinterpret_cast<float4*>(d_rots)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    d_normal[3 * i] = d_normal[3 * i + 1] = d_normal[3 * i + 2] = 0.f;
    return;
  }

  ChainOut o;
  chain_rule(p, i, means, shs, scales, rots, normal_w, clamped, g, o, dsh);
  d_normal[3 * i] = o.dn[0]; d_normal[3 * i + 1] = o.dn[1]; d_normal[3 * i + 2] = o.dn[2];
  d_scales[3 * i] = o.ds[0]; d_scales[3 * i + 1] = o.ds[1]; d_scales[3 * i + 2] = o.ds[2];
  reinterpret_cast<float4*>(d_rots)[i] = o.dq;
  d_means[3 * i] = o.dm[0]; d_means[3 * i + 1] = o.dm[1]; d_means[3 * i + 2] = o.dm[2];
  d_opac[i] = o.dop;
}

// Dense mode: one lane per Gaussian.  Row-state mode: only Gaussians that received gradient, or whose row still holds
// the previous call's gradient, have anything to do (a few thousand of 1.2 M on a depth-complex map, ~17 % on a
// surface map) - a workgroup first compacts the work of its chunk of ids into LDS (two coalesced byte reads per id),
// then walks that list with DENSE lanes: the chain rule below is ~600 instructions and 250 B of scattered reads per
// Gaussian, far too much to run with one lane in six active.
constexpr int PBWD_CHUNK = 2048;
__global__ void __launch_bounds__(256) preprocess_bwd_kernel(
    RasterParams p, const float* __restrict__ means, const float* __restrict__ opac,
    const float* __restrict__ shs, const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ normal_w, const int32_t* __restrict__ radii,
    const uint8_t* __restrict__ clamped, SplatGrad* __restrict__ grads, uint8_t* __restrict__ touched_b,
    uint8_t* __restrict__ row_state,
    float* __restrict__ d_means, float* __restrict__ d_opac, float* __restrict__ d_shs,
    float* __restrict__ d_scales, float* __restrict__ d_rots, float* __restrict__ d_normal) {
  if (spec_failed(p.spec_fail)) return;     // nothing persistent (gradient rows, row states) may change in a failed speculation
  if (!row_state) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < p.P)
      preprocess_bwd_one(p, i, means, opac, shs, scales, rots, normal_w, radii, clamped, grads, touched_b, row_state, d_means,
                         d_opac, d_shs, d_scales, d_rots, d_normal);
    return;
  }
  __shared__ uint32_t s_list[PBWD_CHUNK];
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int c0 = blockIdx.x * PBWD_CHUNK;
#pragma unroll
  for (int k = 0; k < PBWD_CHUNK / 256; ++k) {
    const int i = c0 + k * 256 + (int)threadIdx.x;
    const bool work = i < p.P && ((touched_b[i] != 0) | (row_state[i] == 1));
    const unsigned long long m = __builtin_amdgcn_ballot_w64(work);
    if (m == 0ull) continue;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&s_n, (uint32_t)__popcll(m));
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    if (work) s_list[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)i;
  }
  __syncthreads();
  const uint32_t n = s_n;
  for (uint32_t q = threadIdx.x; q < n; q += 256)
    preprocess_bwd_one(p, (int)s_list[q], means, opac, shs, scales, rots, normal_w, radii, clamped, grads, touched_b,
                       row_state, d_means, d_opac, d_shs, d_scales, d_rots, d_normal);
}

void launch_blend_bwd(const RasterParams& p, const uint2* ranges, const uint32_t* point_list, const Splat* splats,
                      const float* out_color, const float* final_T, const uint32_t* n_contrib,
                      const int32_t* depth_index, const float* dL_dcolor, const float* dL_ddepth, const uint32_t* gbase,
                      uint32_t* slot_count, const BwdInfo* info, SplatGrad* grads, uint8_t* touched,
                      const uint32_t* tile_mode, int which, uint32_t t0, uint32_t tn, hipStream_t st) {
  // which: bit 0 = tiles on the strip walk exist (or unknown), bit 1 = tiles on the row-granular walk exist (or unknown)
  if (which & 1)
    hipLaunchKernelGGL(blend_bwd_strip_kernel, dim3(p.gx, p.gy), dim3(BLOCK), 0, st, p, ranges, point_list, splats, out_color,
                       final_T, n_contrib, depth_index, dL_dcolor, dL_ddepth, gbase, slot_count, info, grads, touched, tile_mode,
                       t0, tn);
  if (which & 2)
    hipLaunchKernelGGL(blend_bwd_rows_kernel, dim3(p.gx, p.gy), dim3(BLOCK), 0, st, p, ranges, point_list, splats, out_color,
                       final_T, n_contrib, depth_index, dL_dcolor, dL_ddepth, gbase, slot_count, info, grads, touched, tile_mode,
                       t0, tn);
}
void launch_grad_reduce(int P, const uint8_t* touched, const uint32_t* gbase, uint32_t* count, const BwdInfo* info,
                        SplatGrad* grads, const uint32_t* spec_fail, hipStream_t st) {
  if (P == 0) return;
  int blocks = (P + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(grad_reduce_kernel, dim3(blocks), dim3(256), 0, st, P, touched, gbase, count, info, grads, spec_fail);
}
void launch_preprocess_bwd(const RasterParams& p, const float* means, const float* opac, const float* shs,
                           const float* scales, const float* rots, const float* normal_w, const int32_t* radii,
                           const uint8_t* clamped, SplatGrad* grads, uint8_t* touched, uint8_t* row_state,
                           float* d_means, float* d_opac, float* d_shs, float* d_scales, float* d_rots, float* d_normal,
                           hipStream_t st) {
  if (p.P == 0) return;
  const int blocks = row_state ? (p.P + PBWD_CHUNK - 1) / PBWD_CHUNK : (p.P + 255) / 256;
  hipLaunchKernelGGL(preprocess_bwd_kernel, dim3(blocks), dim3(256), 0, st, p, means, opac, shs, scales,
                     rots, normal_w, radii, clamped, grads, touched, row_state, d_means, d_opac, d_shs, d_scales, d_rots,
                     d_normal);
}

}  // namespace rtgs

// ---------------------------------------------------------------------------------------------
// Fused Adam over a packed [n, C] float32 parameter shard with a per-column learning rate
// (xyz / f_dc / f_rest / opacity / scaling / rotation groups of
// SLAM/gaussian_pointcloud.py:245-284; torch.optim.Adam(eps=1e-15) semantics, mapper.py:156).
// ---------------------------------------------------------------------------------------------
namespace rtgs {
// 16 B per lane per stream (7 streams: p g m v in, p m v out); n_elems % 4 == 0 on this path
__global__ void __launch_bounds__(256) fused_adam_vec4_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                              float4* __restrict__ m, float4* __restrict__ v,
                                                              const float* __restrict__ lr_col, long long n_vec, int C,
                                                              float beta1, float beta2, float eps, float bc1,
                                                              float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += (long long)gridDim.x * blockDim.x) {
    const float4 pi = p[i], gi = g[i];
    float4 mi = m[i], vi = v[i];
    const unsigned c0 = (unsigned)((unsigned long long)(4 * i) % (unsigned)C);
    const unsigned c1 = c0 + 1 == (unsigned)C ? 0u : c0 + 1, c2 = c1 + 1 == (unsigned)C ? 0u : c1 + 1,
                   c3 = c2 + 1 == (unsigned)C ? 0u : c2 + 1;
    float4 po;
    po.x = adam1(pi.x, gi.x, mi.x, vi.x, lr_col[c0], beta1, beta2, eps, bc1, bc2_sqrt);
    po.y = adam1(pi.y, gi.y, mi.y, vi.y, lr_col[c1], beta1, beta2, eps, bc1, bc2_sqrt);
    po.z = adam1(pi.z, gi.z, mi.z, vi.z, lr_col[c2], beta1, beta2, eps, bc1, bc2_sqrt);
    po.w = adam1(pi.w, gi.w, mi.w, vi.w, lr_col[c3], beta1, beta2, eps, bc1, bc2_sqrt);
    p[i] = po; m[i] = mi; v[i] = vi;
  }
}

__global__ void __launch_bounds__(256) fused_adam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                         float* __restrict__ m, float* __restrict__ v,
                                                         const float* __restrict__ lr_col, long long n_elems, int C,
                                                         float beta1, float beta2, float eps, float bc1, float bc2_sqrt) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems;
       i += (long long)gridDim.x * blockDim.x) {
    float mi = m[i], vi = v[i];
    p[i] = adam1(p[i], g[i], mi, vi, lr_col[(int)((unsigned long long)i % (unsigned)C)], beta1, beta2, eps, bc1, bc2_sqrt);
    m[i] = mi; v[i] = vi;
  }
}

// Row-skipping Adam: a row whose gradient is entirely zero AND whose moments have never left zero
// (`ever[row] == 0`) is left untouched - for such a row dense Adam computes m = v = 0 and an update of
// exactly 0, so the result is bit-identical to fused_adam_kernel while the untouched rows cost one read of
// their gradient (4 B / parameter) instead of 28 B / parameter.  In RTG-SLAM only Gaussians that reach a
// rendered pixel receive gradient (mapper.py:455 relies on the exact zeros) and the optimiser is re-created
// for every local optimisation (mapper.py:156), so most rows of a large map stay in this state.
template <int C>
__global__ void __launch_bounds__(256) fused_adam_rows_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                              float* __restrict__ m, float* __restrict__ v,
                                                              const float* __restrict__ lr_col, uint8_t* __restrict__ ever,
                                                              const uint8_t* __restrict__ row_state,
                                                              long long rows, float beta1, float beta2, float eps,
                                                              float bc1, float bc2_sqrt) {
  // Phase 1 (one lane per row): does the row need an update?  Phase 2: the wave compacts its live rows and
  // sweeps them with LPR lanes per row, so the p/g/m/v accesses of a live row are contiguous 16-B (or 4-B) lanes.
  constexpr bool VEC = (C % 4 == 0);
  constexpr int LPR = VEC ? C / 4 : C;          // lanes per row
  constexpr int RPP = 64 / LPR;                 // rows per pass
  __shared__ int s_rows[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long long wave0 = (long long)blockIdx.x * 4 + wv, nwaves = (long long)gridDim.x * 4;
  for (long long base = wave0 * 64; base < rows; base += nwaves * 64) {
    const long long r = base + lane;
    bool need = false;
    if (r < rows) {
      bool any = false;
      if (row_state) {             // the producer of g already knows which rows are non-zero: no gradient scan
        any = row_state[r] == 1;
      } else if constexpr (VEC) {
        const float4* g4 = reinterpret_cast<const float4*>(g + (size_t)r * C);
#pragma unroll
        for (int q = 0; q < C / 4; ++q) { const float4 t = g4[q]; any |= (t.x != 0.f) | (t.y != 0.f) | (t.z != 0.f) | (t.w != 0.f); }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) any |= g[(size_t)r * C + c] != 0.f;
      }
      need = any || ever[r] != 0;
      if (need) ever[r] = 1;
    }
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(need);
    if (mask == 0ull) continue;
    const int n = __popcll(mask);
    if (need) s_rows[wv][__popcll(mask & ((1ull << lane) - 1ull))] = lane;
    __builtin_amdgcn_wave_barrier();
    const int slot = lane / LPR, sub = lane - slot * LPR;
    for (int k0 = 0; k0 < n; k0 += RPP) {
      const int k = k0 + slot;
      if (slot < RPP && k < n) {
        const size_t row = (size_t)(base + s_rows[wv][k]);
        if constexpr (VEC) {
          const size_t o = row * (C / 4) + sub;
          const float4 gi = reinterpret_cast<const float4*>(g)[o], pi = reinterpret_cast<const float4*>(p)[o];
          float4 mi = reinterpret_cast<float4*>(m)[o], vi = reinterpret_cast<float4*>(v)[o], po;
          po.x = adam1(pi.x, gi.x, mi.x, vi.x, lr_col[4 * sub], beta1, beta2, eps, bc1, bc2_sqrt);
          po.y = adam1(pi.y, gi.y, mi.y, vi.y, lr_col[4 * sub + 1], beta1, beta2, eps, bc1, bc2_sqrt);
          po.z = adam1(pi.z, gi.z, mi.z, vi.z, lr_col[4 * sub + 2], beta1, beta2, eps, bc1, bc2_sqrt);
          po.w = adam1(pi.w, gi.w, mi.w, vi.w, lr_col[4 * sub + 3], beta1, beta2, eps, bc1, bc2_sqrt);
          reinterpret_cast<float4*>(p)[o] = po;
          reinterpret_cast<float4*>(m)[o] = mi;
          reinterpret_cast<float4*>(v)[o] = vi;
        } else {
          const size_t o = row * C + sub;
          float mi = m[o], vi = v[o];
          p[o] = adam1(p[o], g[o], mi, vi, lr_col[sub], beta1, beta2, eps, bc1, bc2_sqrt);
          m[o] = mi; v[o] = vi;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
  }
}
}  // namespace rtgs

extern "C" int rtgs_fused_adam(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                               const float* lr_per_column, int64_t rows, int32_t cols, int32_t step, float beta1,
                               float beta2, float eps, void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_per_column || rows < 0 || cols < 1 || step < 1) return -1;
  const long long n = (long long)rows * cols;
  if (n == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  const bool vec = (n % 4 == 0) && ((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0);
  long long blocks = ((vec ? n / 4 : n) + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  if (vec)
    hipLaunchKernelGGL(rtgs::fused_adam_vec4_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       (float4*)params, (const float4*)grads, (float4*)exp_avg, (float4*)exp_avg_sq, lr_per_column, n / 4,
                       (int)cols, beta1, beta2, eps, bc1, bc2s);
  else
    hipLaunchKernelGGL(rtgs::fused_adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, params, grads,
                       exp_avg, exp_avg_sq, lr_per_column, n, (int)cols, beta1, beta2, eps, bc1, bc2s);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

extern "C" int rtgs_fused_adam_rows(float* params, const float* grads, float* exp_avg, float* exp_avg_sq,
                                    const float* lr_per_column, uint8_t* ever_touched, const uint8_t* row_state,
                                    int64_t rows, int32_t cols, int32_t step, float beta1, float beta2, float eps,
                                    void* stream) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !lr_per_column || !ever_touched || rows < 0 || step < 1) return -1;
  if (cols != 3 && cols != 8 && cols != 48) return -1;      // the three block tensors of the map (xyz, raw8, SH)
  if (rows == 0) return 0;
  const float bc1 = 1.f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
  long long blocks = (rows + 255) / 256;      // one wave per 64 rows
  if (blocks > 16384) blocks = 16384;
  hipStream_t st = (hipStream_t)stream;
  if (cols == 3)
    hipLaunchKernelGGL(rtgs::fused_adam_rows_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg,
                       exp_avg_sq, lr_per_column, ever_touched, row_state, (long long)rows, beta1, beta2, eps, bc1, bc2s);
  else if (cols == 8)
    hipLaunchKernelGGL(rtgs::fused_adam_rows_kernel<8>, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg,
                       exp_avg_sq, lr_per_column, ever_touched, row_state, (long long)rows, beta1, beta2, eps, bc1, bc2s);
  else
    hipLaunchKernelGGL(rtgs::fused_adam_rows_kernel<48>, dim3((unsigned)blocks), dim3(256), 0, st, params, grads, exp_avg,
                       exp_avg_sq, lr_per_column, ever_touched, row_state, (long long)rows, beta1, beta2, eps, bc1, bc2s);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
