// Forward kernels of the gfx950 rasterizer: per-Gaussian preprocess (one pass, or geometry / lazy shading for the
// two-pass forward), (Gaussian, tile) instance emission and tile ranges of the fallback path, and the per-tile
// front-to-back alpha blend with opaque-surface depth.
// Arithmetic follows SURVEY.md Appendix B (frozen in oracle/raster_oracle.py); the reference's
// own CUDA sources are an un-vendored submodule (/root/reference/.gitmodules:1-4), its call
// contract is /root/reference/SLAM/render.py:68-128.
#include "raster_common.h"
#include <stdlib.h>

namespace rtgs {

// ---------------------------------------------------------------------------------------------
// tile-mask summed-area table: sat[(y+1)*(gx+1)+(x+1)] = #{mask != 0 in [0..y]x[0..x]}
// One workgroup; the grid is at most a few thousand tiles (43x75 for Replica).
// ---------------------------------------------------------------------------------------------
constexpr int SAT_LDS = 12288;     // entries staged in LDS (Replica: 76 x 44 = 3 344); larger grids work in global memory
__global__ void __launch_bounds__(256) mask_sat_kernel(const int32_t* __restrict__ mask, int gx, int gy,
                                                       int32_t* __restrict__ sat) {
  __shared__ int32_t s_sat[SAT_LDS];
  const int sw = gx + 1, n = (gy + 1) * sw;
  // the two prefix passes are chains of dependent read-modify-writes: in LDS a step costs ~100 cycles, in global
  // memory ~2 000 (the global form took 25 us for 43 x 75 tiles)
  int32_t* const w = n <= SAT_LDS ? s_sat : sat;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    int y = i / sw, x = i % sw;
    w[i] = (y == 0 || x == 0) ? 0 : (mask[(y - 1) * gx + (x - 1)] != 0);
  }
  __syncthreads();
  for (int y = 1 + threadIdx.x; y <= gy; y += blockDim.x) {      // row prefix
    int acc = 0;
    for (int x = 1; x <= gx; ++x) { acc += w[y * sw + x]; w[y * sw + x] = acc; }
  }
  __syncthreads();
  for (int x = 1 + threadIdx.x; x <= gx; x += blockDim.x) {      // column prefix
    int acc = 0;
    for (int y = 1; y <= gy; ++y) { acc += w[y * sw + x]; w[y * sw + x] = acc; }
  }
  if (w != sat) {
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += blockDim.x) sat[i] = w[i];
  }
}

// ---------------------------------------------------------------------------------------------
// K1 preprocess_fwd: one lane per Gaussian.  Reads 62 floats (248 B), writes one 64-B Splat +
// radius + tiles_touched + clamp flags.
// ---------------------------------------------------------------------------------------------
// MODE 0: everything for every Gaussian (single-pass forward).
// MODE 1: geometry only - depth cull, radius, tile rect, depth bin, (u, v) - for every Gaussian: 52 B read and
//         21 B written per Gaussian instead of 248 B + 77 B.  The two-pass forward shades lazily:
// MODE 2: the rest (conic, SH colour, plane, the 64-B Splat) for the Gaussians of a work list (the near slice),
//         or - list.ids == nullptr - for every visible Gaussian NOT in the slice, and only if the slice left tiles
//         unfinished.  (u, v) and the radius are taken from MODE 1 so that every kernel derives the same tile rect.
template <int MODE>
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(
    RasterParams p, const float* __restrict__ means, const float* __restrict__ opac,
    const float* __restrict__ shs, const float* __restrict__ scales, const float* __restrict__ rots,
    const float* __restrict__ normal_w, const int32_t* __restrict__ sat,
    Splat* __restrict__ splats, uint32_t* __restrict__ tiles_touched, int32_t* __restrict__ radii,
    uint8_t* __restrict__ clamped, int32_t* __restrict__ out_radii, uint32_t* __restrict__ zero_words, int zero_n,
    uint8_t* __restrict__ zbin, float2* __restrict__ uv, SliceList list, SliceSel sel) {
  // no automatic FMA contraction: the three instantiations must round identically (the two-pass forward is tested
  // bit for bit against the single pass)
#pragma clang fp contract(off)
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if constexpr (MODE != 2) {
    // clears the per-tile counters bin_count accumulates into (saves a memset launch on the critical path)
    for (int t = i; t < zero_n; t += gridDim.x * blockDim.x) zero_words[t] = 0u;
    if (i >= p.P) return;
    tiles_touched[i] = 0;
    radii[i] = 0;
    if (out_radii) out_radii[i] = 0;
    if (zbin) zbin[i] = 255;
  } else {
    if (spec_failed(p.spec_fail)) return;      // speculative forward already known to be wrong: the work list is not valid
    if (list.ids) {
      if (i >= (int)*list.count) return;
      i = (int)list.ids[i];
    } else {
      if (sel.ctr && sel.ctr[0] == 0u) return;   // the slice finished every tile: nobody will read the other Splats
      const int cut = slice_cut(sel);            // (all 256 threads of the workgroup call)
      if (i >= p.P) return;
      const int zb = (int)sel.zbin[i];
      if (zb == 255 || zb <= cut) return;        // invisible, or shaded with the slice already
      if (sel.sat) {                              // no unfinished tile under its rect: bin_count will not read it either
        const float2 c = uv[i];
        int x0, y0, x1, y1;
        tile_rect_of(c.x, c.y, radii[i], p.gx, p.gy, x0, y0, x1, y1);
        if (sat_count(sel.sat, p.gx, x0, y0, x1, y1) == 0) return;
      }
    }
  }

  const float* V = p.view;   // V[j*4+i] = W2C[i][j]
  const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  const float pcx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
  const float pcy = V[1] * mx + V[5] * my + V[9] * mz + V[13];
  const float pcz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
  if (!(pcz > 0.2f)) return;

  // Sigma3D = (R S)(R S)^T
  const float4 q4 = reinterpret_cast<const float4*>(rots)[i];
  const float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
  const float sx = scales[3 * i] * p.scale_modifier, sy = scales[3 * i + 1] * p.scale_modifier,
              sz = scales[3 * i + 2] * p.scale_modifier;
  const float R00 = 1.f - 2.f * (qy * qy + qz * qz), R01 = 2.f * (qx * qy - qr * qz), R02 = 2.f * (qx * qz + qr * qy);
  const float R10 = 2.f * (qx * qy + qr * qz), R11 = 1.f - 2.f * (qx * qx + qz * qz), R12 = 2.f * (qy * qz - qr * qx);
  const float R20 = 2.f * (qx * qz - qr * qy), R21 = 2.f * (qy * qz + qr * qx), R22 = 1.f - 2.f * (qx * qx + qy * qy);
  const float M00 = R00 * sx, M01 = R01 * sy, M02 = R02 * sz;
  const float M10 = R10 * sx, M11 = R11 * sy, M12 = R12 * sz;
  const float M20 = R20 * sx, M21 = R21 * sy, M22 = R22 * sz;
  const float S00 = M00 * M00 + M01 * M01 + M02 * M02;
  const float S01 = M00 * M10 + M01 * M11 + M02 * M12;
  const float S02 = M00 * M20 + M01 * M21 + M02 * M22;
  const float S11 = M10 * M10 + M11 * M11 + M12 * M12;
  const float S12 = M10 * M20 + M11 * M21 + M12 * M22;
  const float S22 = M20 * M20 + M21 * M21 + M22 * M22;

  // EWA: Sigma2D = T Sigma T^T, T = J Wr
  const float limx = 1.3f * p.tanfovx, limy = 1.3f * p.tanfovy;
  const float txtz = pcx / pcz, tytz = pcy / pcz;
  const float tx = fminf(limx, fmaxf(-limx, txtz)) * pcz;
  const float ty = fminf(limy, fmaxf(-limy, tytz)) * pcz;
  const float iz = 1.f / pcz;
  const float J00 = p.fx * iz, J02 = -p.fx * tx * iz * iz;
  const float J11 = p.fy * iz, J12 = -p.fy * ty * iz * iz;
  // Wr[r][c] = V[c*4+r]
  const float T00 = J00 * V[0] + J02 * V[2], T01 = J00 * V[4] + J02 * V[6], T02 = J00 * V[8] + J02 * V[10];
  const float T10 = J11 * V[1] + J12 * V[2], T11 = J11 * V[5] + J12 * V[6], T12 = J11 * V[9] + J12 * V[10];
  const float a0 = S00 * T00 + S01 * T01 + S02 * T02;
  const float a1 = S01 * T00 + S11 * T01 + S12 * T02;
  const float a2 = S02 * T00 + S12 * T01 + S22 * T02;
  const float b0 = S00 * T10 + S01 * T11 + S02 * T12;
  const float b1 = S01 * T10 + S11 * T11 + S12 * T12;
  const float b2 = S02 * T10 + S12 * T11 + S22 * T12;
  const float ca = T00 * a0 + T01 * a1 + T02 * a2 + 0.3f;
  const float cb = T00 * b0 + T01 * b1 + T02 * b2;
  const float cc = T10 * b0 + T11 * b1 + T12 * b2 + 0.3f;
  const float det = ca * cc - cb * cb;
  if (det == 0.f) return;
  const float idet = 1.f / det;
  const float mid = 0.5f * (ca + cc);
  const float lam = mid + sqrtf(fmaxf(0.1f, mid * mid - det));
  int radius = (int)ceilf(p.color_sigma * sqrtf(lam));
  float u = p.fx * pcx / pcz + p.cx;
  float v = p.fy * pcy / pcz + p.cy;
  int touched = 0;
  if constexpr (MODE == 2) {
    const float2 t = uv[i];
    u = t.x; v = t.y; radius = radii[i];
  } else {
    int x0, y0, x1, y1;
    tile_rect_of(u, v, radius, p.gx, p.gy, x0, y0, x1, y1);
    if ((x1 - x0) * (y1 - y0) == 0) return;
    const int sw = p.gx + 1;
    touched = sat ? sat[y1 * sw + x1] - sat[y0 * sw + x1] - sat[y1 * sw + x0] + sat[y0 * sw + x0]
                  : (x1 - x0) * (y1 - y0);   // LDS binning path recounts exactly; this is a bound
  }
  if constexpr (MODE == 1) {
    radii[i] = radius;
    if (out_radii) out_radii[i] = radius;
    tiles_touched[i] = (uint32_t)touched;
    zbin[i] = (uint8_t)slice_bin_of(pcz);
    uv[i] = make_float2(u, v);
    return;
  }

  // view-dependent colour (utils/sh_utils.py:57-120 basis), clamped at 0
  float dxw = mx - p.campos[0], dyw = my - p.campos[1], dzw = mz - p.campos[2];
  const float il = 1.f / sqrtf(dxw * dxw + dyw * dyw + dzw * dzw);
  dxw *= il; dyw *= il; dzw *= il;
  // SH block: 48 floats (192 B, 16-B aligned) per Gaussian when M == 16 -> twelve 16-B loads per lane
  // (every 64-B line is consumed by 4 back-to-back loads) instead of 48 strided dword loads
  float shv[48];
  if (p.M == 16) {
    const float4* sh4 = reinterpret_cast<const float4*>(shs + (size_t)i * 48);
#pragma unroll
    for (int q = 0; q < 12; ++q) {
      const float4 t = sh4[q];
      shv[4 * q] = t.x; shv[4 * q + 1] = t.y; shv[4 * q + 2] = t.z; shv[4 * q + 3] = t.w;
    }
  } else {
    const float* shp = shs + (size_t)i * p.M * 3;
#pragma unroll
    for (int q = 0; q < 48; ++q) shv[q] = (q < p.M * 3) ? shp[q] : 0.f;
  }
  const float* sh = shv;
  float col[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    float r = RTGS_SH_C0 * sh[c];
    if (p.deg > 0) {
      r = r - RTGS_SH_C1 * dyw * sh[3 + c] + RTGS_SH_C1 * dzw * sh[6 + c] - RTGS_SH_C1 * dxw * sh[9 + c];
      if (p.deg > 1) {
        const float xx = dxw * dxw, yy = dyw * dyw, zz = dzw * dzw, xy = dxw * dyw, yz = dyw * dzw, xz = dxw * dzw;
        r = r + RTGS_SH_C2_0 * xy * sh[12 + c] + RTGS_SH_C2_1 * yz * sh[15 + c] +
            RTGS_SH_C2_2 * (2.f * zz - xx - yy) * sh[18 + c] + RTGS_SH_C2_3 * xz * sh[21 + c] +
            RTGS_SH_C2_4 * (xx - yy) * sh[24 + c];
        if (p.deg > 2) {
          r = r + RTGS_SH_C3_0 * dyw * (3.f * xx - yy) * sh[27 + c] + RTGS_SH_C3_1 * xy * dzw * sh[30 + c] +
              RTGS_SH_C3_2 * dyw * (4.f * zz - xx - yy) * sh[33 + c] +
              RTGS_SH_C3_3 * dzw * (2.f * zz - 3.f * xx - 3.f * yy) * sh[36 + c] +
              RTGS_SH_C3_4 * dxw * (4.f * zz - xx - yy) * sh[39 + c] + RTGS_SH_C3_5 * dzw * (xx - yy) * sh[42 + c] +
              RTGS_SH_C3_6 * dxw * (xx - 3.f * yy) * sh[45 + c];
        }
      }
    }
    col[c] = r + 0.5f;
  }
  uint8_t cl = 0;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (col[c] < 0.f) { cl |= (uint8_t)(1u << c); col[c] = 0.f; }
  }

  const float nwx = normal_w[3 * i], nwy = normal_w[3 * i + 1], nwz = normal_w[3 * i + 2];
  const float ncx = V[0] * nwx + V[4] * nwy + V[8] * nwz;
  const float ncy = V[1] * nwx + V[5] * nwy + V[9] * nwz;
  const float ncz = V[2] * nwx + V[6] * nwy + V[10] * nwz;

  Splat s;
  s.u = u; s.v = v;
  s.ca = cc * idet; s.cb = -cb * idet; s.cc = ca * idet;
  s.o = opac[i];
  s.r = col[0]; s.g = col[1]; s.b = col[2];
  s.nx = ncx; s.ny = ncy; s.nz = ncz;
  s.pd = ncx * pcx + ncy * pcy + ncz * pcz;
  s.z = pcz;
  {
    // |dx| <= sqrt(thr * Sigma2D_xx) wherever alpha >= 1/255 (thr = 2 ln(255 o)); Sigma2D = (ca cb; cb cc).
    // 1 % + 0.5 px margin; a non-positive threshold means the Gaussian is invisible everywhere.
    const float thr = 2.f * __logf(255.f * fmaxf(opac[i], 1e-12f));
    const float m = fmaxf(thr, 0.f) * 1.01f;
    s.hx = sqrtf(m * ca) + 0.5f;
    s.hy = sqrtf(m * cc) + 0.5f;
  }
  float4* dst = reinterpret_cast<float4*>(splats + i);
  const float4* src = reinterpret_cast<const float4*>(&s);
  dst[0] = src[0]; dst[1] = src[1]; dst[2] = src[2]; dst[3] = src[3];
  clamped[i] = cl;
  if constexpr (MODE == 0) {
    radii[i] = radius;
    if (out_radii) out_radii[i] = radius;
    tiles_touched[i] = (uint32_t)touched;
    if (zbin) zbin[i] = (uint8_t)slice_bin_of(pcz);
  }
}

// ---------------------------------------------------------------------------------------------
// K3 emit: key = tile << 32 | f32 depth bits, value = Gaussian id, for unmasked tiles of the rect
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) emit_keys_kernel(
    RasterParams p, const Splat* __restrict__ splats, const int32_t* __restrict__ radii,
    const uint32_t* __restrict__ offsets, const int32_t* __restrict__ mask,
    uint64_t* __restrict__ keys, uint32_t* __restrict__ vals) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.P) return;
  const int radius = radii[i];
  if (radius <= 0) return;
  uint32_t off = (i == 0) ? 0u : offsets[i - 1];
  const uint32_t end = offsets[i];
  if (off == end) return;
  const float u = splats[i].u, v = splats[i].v;
  const uint32_t zbits = __float_as_uint(splats[i].z);
  int x0, y0, x1, y1;
  tile_rect_of(u, v, radius, p.gx, p.gy, x0, y0, x1, y1);
  for (int y = y0; y < y1; ++y)
    for (int x = x0; x < x1; ++x) {
      const int t = y * p.gx + x;
      if (mask[t] != 0) {
        keys[off] = ((uint64_t)(uint32_t)t << 32) | zbits;
        vals[off] = (uint32_t)i;
        ++off;
      }
    }
}

// K5 tile ranges over the sorted keys
__global__ void __launch_bounds__(256) tile_ranges_kernel(int64_t R, const uint64_t* __restrict__ keys,
                                                          uint2* __restrict__ ranges) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= R) return;
  const uint32_t t = (uint32_t)(keys[i] >> 32);
  if (i == 0) ranges[t].x = 0;
  else {
    const uint32_t tp = (uint32_t)(keys[i - 1] >> 32);
    if (tp != t) { ranges[tp].y = (uint32_t)i; ranges[t].x = (uint32_t)i; }
  }
  if (i == R - 1) ranges[t].y = (uint32_t)R;
}

// ---------------------------------------------------------------------------------------------
// K6 blend_fwd: one workgroup (4 wave64) per 16x16 tile, list entries staged through LDS in batches of 256 records
// (one 64-B gather per thread).  The walk is ROW-GRANULAR: a wave owns an 8x8 quadrant, each of its four DPP rows
// (16 lanes) a 4x4 pixel block, and every row walks ITS OWN sub-list of the batch - the entries whose alpha >= 1/255
// bounding box reaches its block.  The sub-lists are bit masks built at staging time (the staging thread holds the
// record in registers and tests it against the 16 blocks of the tile; one ballot per block).  On a surface map of
// ~12-pixel discs a block needs about a third of its tile's list, so a wave makes a third of the passes a
// tile-uniform walk makes, and the lanes that evaluate an entry are mostly inside it (SQ counters, profiles/: the
// tile-uniform walk was VALU-issue bound with 23 % of the lanes lit).  Pixels are independent in the forward, so
// nothing crosses lanes; per pixel the entries still arrive in list order, hence bit-identical outputs.
// ---------------------------------------------------------------------------------------------
constexpr int FWD_BATCH = BLOCK;                // list entries staged through LDS per refill (one per thread)
constexpr int FWD_CHUNKS = FWD_BATCH / 32;      // 32-entry words of a block's sub-list

// STAMP (measurement only, tools/fwd_stamps.py): every wave leaves eight 64-bit words - wall clock (100 MHz) at entry and exit,
// shader cycles until the tile's range is there | until the first batch is staged and its barrier passed | inside the walk
// loops | in the whole kernel, walk steps taken, batches staged.
typedef float f4v __attribute__((ext_vector_type(4)));
typedef float f2v __attribute__((ext_vector_type(2)));

template <int OCC, bool STAMP, bool COUNT, bool ASMW>
__global__ void __launch_bounds__(256, OCC) blend_fwd_kernel(
    RasterParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
    const Splat* __restrict__ splats, float* __restrict__ out_color, float* __restrict__ out_depth,
    int32_t* __restrict__ out_cidx, int32_t* __restrict__ out_didx, float* __restrict__ out_cw,
    float* __restrict__ out_dw, float* __restrict__ out_T, uint32_t* __restrict__ n_contrib,
    unsigned long long* __restrict__ counters, SlicePass sp, uint32_t* __restrict__ tile_mode,
    uint32_t* __restrict__ depth_pos, uint32_t* __restrict__ tile_last, int walk, uint32_t* __restrict__ aux_zero,
    TileCache tc, unsigned long long* __restrict__ stamps, uint32_t seg) {
  __shared__ float4 s_rec[FWD_BATCH * 4];     // u v ca cb | cc o r g | b - - id | nx ny nz pd: the walk reads the first three
  // (entry-major, 64 B per entry.  Four planes of FWD_BATCH float4 - the four rows of a wave read four DIFFERENT entries per
  // step, and entry-major two of them collide on their banks whenever their indices agree mod 4 - were measured in round 6:
  // 82.1 / 131.3 us against 80.9 / 131.0.  The walk is not bound by LDS bank conflicts.)
#define RI(e, j) ((e) * 4 + (j))
  __shared__ float s_z[FWD_BATCH];            // centre depth (opaque-surface test only)
  __shared__ uint32_t s_live[16][FWD_CHUNKS]; // per 4x4 block: the staged entries that reach it

  unsigned long long st_wall = 0, st_cyc = 0, st_range = 0, st_first = 0, st_walk = 0, st_steps = 0, st_batches = 0;
  if constexpr (STAMP) { st_wall = wall_clock64(); st_cyc = __builtin_readcyclecounter(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63, wv = tid >> 6;
  const int tile = blockIdx.y * p.gx + blockIdx.x;
  // The words this tile waits for first - its range, the speculation word, its pass-2 mask, and (lists in per-tile segments:
  // `seg` entries per tile, raster_bin.hip) the ids of its first batch, whose addresses need no range - leave as VECTOR loads
  // in one go.  As scalar loads each is a wait of its own in front of the next (round 6's stamps: six serialised scalar waits
  // and the id hop were 5.4 us of a 27-us tile).  `z` is a zero the compiler cannot see.
  int z = 0;
  asm volatile("" : "+v"(z));
  const uint2 range_v = ranges[tile + z];
  const uint32_t fail_v = p.spec_fail ? p.spec_fail[z] : 0u;
  const int32_t m2_v = sp.mode == 2 ? sp.mask2[tile + z] : 1;
  const int32_t on_v = sp.mode == 1 ? sp.user_mask[tile + z] : 1;
  uint32_t id_first = 0u;
  if (seg != 0u) id_first = point_list[(size_t)tile * seg + (size_t)tid];     // (inside the tile's segment whatever its count)
  // eight words a later kernel of the caller's step accumulates into (the loss sums of rtgs_slam_map_step): cleared here,
  // stream-ordered before that kernel, instead of by a memset launch of their own
  if (aux_zero && blockIdx.x == 0 && blockIdx.y == 0 && tid < 8) aux_zero[tid] = 0u;
  // wave = 8x8 quadrant, DPP row = 4x4 block, lane = pixel of the block
  const int bx = ((wv & 1) << 1) | ((lane >> 4) & 1), by = (wv & 2) | (lane >> 5);
  const int blk = by * 4 + bx;
  const int px = blockIdx.x * TILE + bx * 4 + (lane & 3);
  const int py = blockIdx.y * TILE + by * 4 + ((lane >> 2) & 3);
  const bool inside = px < p.W && py < p.H;
  const float pxf = (float)px, pyf = (float)py;
  const float tx0 = (float)(blockIdx.x * TILE), ty0 = (float)(blockIdx.y * TILE);
  if (__builtin_amdgcn_readfirstlane((int)fail_v) != 0) return;   // lists were not built (speculative sizes did not hold): redone by the host
  if (__builtin_amdgcn_readfirstlane(m2_v) == 0) return;          // finished by the near slice (or masked off): outputs stay
  const uint2 range = make_uint2((uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.x), (uint32_t)__builtin_amdgcn_readfirstlane((int)range_v.y));
  int n = (int)(range.y - range.x);
  if (sp.mode == 1 && n > SLICE_MAX_LIST) n = 0;       // near-slice list too long to have been sorted: leave the tile to pass 2
  if constexpr (STAMP) { st_range = __builtin_readcyclecounter() - st_cyc; }

  bool done = !inside;
  unsigned long long Dm = __builtin_amdgcn_ballot_w64(!inside);     // ASMW: the wave's stopped pixels as a mask in SGPRs
  unsigned long long evals_w = 0;                                     // ASMW + COUNT: wave-uniform
  float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
  float best_w = 0.f; int best_id = -1;
  float D = 0.f, d_w = 0.f; int d_id = -1;
  float d_iden = 0.f;                        // 1 / (n_c . r) of the depth owner: with D, all the backward's plane partials need beyond the pixel's ray
  uint32_t d_pos = 0u;                       // list position of the depth owner (the backward's entry walk hands the owner its partials by position)
  uint32_t last_contributor = 0;
  uint32_t evals = 0;
  uint32_t reach_sum = 0, staged = 0;        // wave-uniform: (block, entry) pairs of the sub-lists / entries this wave staged

  // The near-slice pass runs only where the scene has depth complexity: its walks stop after a fraction of the list (52 of
  // ~250 entries on the headline scene), so its FIRST batch is a quarter batch - staging (a 64-B gather + the block test
  // per entry) is not paid for entries no pixel will reach.  Batching does not change a result.
  int bs = sp.mode == 1 ? FWD_BATCH / 4 : FWD_BATCH;
  for (int base = 0; base < n; base += bs, bs = FWD_BATCH) {
    if (base > 0 && __syncthreads_and(ASMW ? (Dm == ~0ull) : done)) break;      // (nothing is done before the first batch; wave-uniform condition)
    const int m = min(bs, n - base);
    {
      uint32_t reach = 0;
      if (tid < m) {
        const uint32_t id = (seg != 0u && base == 0) ? id_first : point_list[range.x + base + tid];
        const float4* src = reinterpret_cast<const float4*>(splats + id);
        const float4 q0 = src[0];
        s_rec[RI(tid, 0)] = q0;
        const float4 q1 = src[1];
        s_rec[RI(tid, 1)] = q1;
        const float4 q2 = src[2];               // b nx ny nz
        const float4 q3 = src[3];               // pd z hx hy
        s_rec[RI(tid, 2)] = make_float4(q2.x, 0.f, 0.f, __uint_as_float(id));
        s_rec[RI(tid, 3)] = make_float4(q2.y, q2.z, q2.w, q3.x);
        s_z[tid] = q3.y;
        // For the backward (TileCache, raster_common.h): the record and the block mask of list position base + tid at a place
        // that depends on the TILE only - its first loads need no tile range, no list id, no gather and no block test.
        // (The record goes out before the block test, the mask after it: nothing but the position stays live across the test.)
        const bool keep = tc.recs != nullptr && base + tid < TILE_RECS;
        if (keep) {
          float4* const dst = tc.recs + ((size_t)tile * TILE_RECS + (size_t)(base + tid)) * 3;
          dst[0] = q0; dst[1] = q1;
          dst[2] = make_float4(q2.x, __uint_as_float(id), 0.f, 0.f);
        }
        reach = blocks_reached(q0.x, q0.y, q3.z, q3.w, q0.z, q0.w, q1.x, q1.y, tx0, ty0);
        if (keep) tc.masks[(size_t)tile * TILE_RECS + (size_t)(base + tid)] = (uint16_t)reach;
      }
      if constexpr (STAMP) st_batches += 1;
#pragma unroll
      for (int b = 0; b < 16; ++b) {
        const unsigned long long bal = __builtin_amdgcn_ballot_w64((reach >> b) & 1u);
        if (lane == 0) { s_live[b][2 * wv] = (uint32_t)bal; s_live[b][2 * wv + 1] = (uint32_t)(bal >> 32); }
        reach_sum += (uint32_t)__popcll(bal);
      }
      staged += (uint32_t)max(0, min(64, m - wv * 64));
    }
    __syncthreads();
    unsigned long long w_in = 0;
    if constexpr (STAMP) { w_in = __builtin_readcyclecounter(); if (base == 0) st_first = w_in - st_cyc; }
    const int nch = (m + 31) >> 5;
    int c = -1;
    uint32_t cur = 0u;                               // row-uniform: the unread part of the current word of the sub-list
    if constexpr (ASMW) {
      // The step, hand-scheduled (round 6).  The compiler's form of the loop below is 86 instructions per step on the common
      // path - every ballot as v_cndmask + v_cmp, every "contrib ? x : y" as a select, masks shuffled between SGPR pairs - and
      // the walk is bound by instruction issue (profiles/r05_valu_rate.txt).  Here the lanes that have an entry ARE the exec
      // mask: the two tests narrow it (v_cmpx), the stop test splits it, and what is left updates T / C / best with plain
      // instructions.  ~45 instructions per step.  Same arithmetic in the same order: outputs are bit-identical.
      uint32_t rec_base = (uint32_t)(uintptr_t)&s_rec[0];
      const uint32_t live_base = (uint32_t)(uintptr_t)&s_live[blk][0];     // per lane: its row's sub-list words
      uint32_t base1 = (uint32_t)base + 1u;
      float thr = p.T_thr, opq = p.opaque_thr;
      uint32_t steps_w = 0u, evals_lo = 0u;                                  // wave-uniform counters (STAMP / COUNT variants)
      // (opaque to the compiler: as kernel arguments it RELOADS them inside the loop - an s_load and a wait for everything in
      // flight, the step's LDS reads included - rather than hold four more SGPRs)
      asm volatile("" : "+s"(rec_base), "+s"(base1), "+s"(thr), "+s"(opq));
      // The WHOLE walk loop is one statement: refill of a row's sub-list word, the loop condition, the step.  It leaves for
      // two reasons: nobody has an entry left (WD == 0), or the step met opaque-surface depth candidates (WD = their lanes; a
      // handful of times per wave: handled below in C++, then the loop is entered again).  The three record loads land in
      // v[70:79] (named: an operand's sub-registers cannot be; they are listed as clobbers, so nothing else lives there -
      // which is why this kernel exists at 5 and 6 waves per SIMD only, 80+ VGPRs).
#define RTGS_FWD_WALK(EXTRA)                                                                                                     \
      asm volatile(                                                                                                              \
          "s_mov_b64 %[WD], 0\n\t"                                                                                               \
          "1:\n\t"                                                                                                               \
          "v_cmp_eq_u32 vcc, 0, %[cur]\n\t"                     /* rows whose word is used up */                                 \
          "s_cbranch_vccz 3f\n\t"                                                                                                \
          "v_add_u32 %[t1], 1, %[c]\n\t"                                                                                         \
          "v_cmp_gt_i32 %[tm], %[nch], %[t1]\n\t"               /* ... and have another word */                                  \
          "s_and_b64 vcc, vcc, %[tm]\n\t"                                                                                        \
          "s_cbranch_vccz 3f\n\t"                                                                                                \
          "s_mov_b64 exec, vcc\n\t"                                                                                              \
          "v_mov_b32 %[c], %[t1]\n\t"                                                                                            \
          "v_lshl_add_u32 %[addr], %[t1], 2, %[lb]\n\t"                                                                          \
          "ds_read_b32 %[cur], %[addr]\n\t"                                                                                      \
          "s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
          "s_mov_b64 exec, -1\n\t"                                                                                               \
          "s_branch 1b\n\t"                                     /* (the new word may be empty too) */                            \
          "3:\n\t"                                                                                                               \
          "v_cmp_ne_u32 vcc, 0, %[cur]\n\t"                                                                                      \
          "s_andn2_b64 %[tm], vcc, %[D]\n\t"                    /* a lane walks on while ITS pixel is open and its row has entries */ \
          "s_cbranch_scc0 9f\n\t"                                                                                                \
          "s_mov_b64 exec, %[tm]\n\t"                                                                                            \
          EXTRA                                                                                                                  \
          "v_ffbl_b32 %[e], %[cur]\n\t"                                                                                          \
          "v_lshl_or_b32 %[e], %[c], 5, %[e]\n\t"                                                                                \
          "v_lshl_add_u32 %[addr], %[e], 6, %[rb]\n\t"                                                                           \
          "ds_read_b128 v[70:73], %[addr]\n\t"                  /* u v ca cb */                                                  \
          "ds_read_b128 v[74:77], %[addr] offset:16\n\t"        /* cc o r g */                                                   \
          "ds_read2_b32 v[78:79], %[addr] offset0:8 offset1:11\n\t"   /* b id */                                                 \
          "v_add_u32 %[t1], -1, %[cur]\n\t"                                                                                      \
          "v_and_b32 %[cur], %[t1], %[cur]\n\t"                 /* the entry is taken (lanes outside: no entry, or stopped for good) */ \
          "s_waitcnt lgkmcnt(2)\n\t"                                                                                             \
          "v_sub_f32 %[dx], v70, %[px]\n\t"                                                                                      \
          "v_sub_f32 %[dy], v71, %[py]\n\t"                                                                                      \
          "v_mul_f32 %[t1], v72, %[dx]\n\t"                                                                                      \
          "v_mul_f32 %[t2], v73, %[dx]\n\t"                                                                                      \
          "s_waitcnt lgkmcnt(1)\n\t"                                                                                             \
          "v_mul_f32 %[pw], v74, %[dy]\n\t"                                                                                      \
          "v_mul_f32 %[pw], %[pw], %[dy]\n\t"                                                                                    \
          "v_fma_f32 %[t1], %[t1], %[dx], %[pw]\n\t"                                                                             \
          "v_mul_f32 %[t2], %[t2], %[dy]\n\t"                                                                                    \
          "v_fma_f32 %[pw], -0.5, %[t1], -%[t2]\n\t"                                                                             \
          "v_min_f32 %[t1], 0, %[pw]\n\t"                                                                                        \
          "v_mul_f32 %[t1], 0x3fb8aa3b, %[t1]\n\t"                                                                               \
          "v_exp_f32 %[t1], %[t1]\n\t"                                                                                           \
          "v_cmpx_nlt_f32 vcc, 0, %[pw]\n\t"                    /* exec: !(power > 0) */                                         \
          "s_nop 0\n\t"                                                                                                          \
          "v_mul_f32 %[al], v75, %[t1]\n\t"                                                                                      \
          "v_min_f32 %[al], 0x3f7d70a4, %[al]\n\t"                                                                               \
          "v_cmpx_ngt_f32 vcc, 0x3b808081, %[al]\n\t"           /* exec: !(alpha < 1/255) */                                     \
          "v_sub_f32 %[t2], 1.0, %[al]\n\t"                                                                                      \
          "v_mul_f32 %[tT], %[T], %[t2]\n\t"                                                                                     \
          "v_cmp_gt_f32 vcc, %[thr], %[tT]\n\t"                 /* the pixel stops in front of this entry */                     \
          "s_or_b64 %[D], %[D], vcc\n\t"                                                                                         \
          "s_andn2_b64 exec, exec, vcc\n\t"                     /* exec: the contributing lanes */                               \
          "s_cbranch_scc0 5f\n\t"                                                                                                \
          "v_mul_f32 %[w], %[al], %[T]\n\t"                                                                                      \
          "v_mov_b32 %[T], %[tT]\n\t"                                                                                            \
          "v_add_u32 %[last], %[base1], %[e]\n\t"                                                                                \
          "s_waitcnt lgkmcnt(0)\n\t"                                                                                             \
          "v_fmac_f32 %[C0], v76, %[w]\n\t"                                                                                      \
          "v_fmac_f32 %[C1], v77, %[w]\n\t"                                                                                      \
          "v_fmac_f32 %[C2], v78, %[w]\n\t"                                                                                      \
          "v_cmp_gt_f32 vcc, %[w], %[bw]\n\t"                                                                                    \
          "v_cndmask_b32 %[bw], %[bw], %[w], vcc\n\t"                                                                            \
          "v_cndmask_b32 %[bid], %[bid], v79, vcc\n\t"                                                                           \
          "v_cmp_gt_i32 vcc, 0, %[did]\n\t"                     /* no depth owner yet ... */                                     \
          "v_cmp_lt_f32 %[WD], %[opq], %[al]\n\t"               /* ... and this entry is opaque enough to be one */              \
          "s_and_b64 %[WD], %[WD], vcc\n\t"                                                                                      \
          "s_mov_b64 exec, -1\n\t"                                                                                               \
          "s_cmp_eq_u64 %[WD], 0\n\t"                                                                                            \
          "s_cbranch_scc1 1b\n\t"                                                                                                \
          "v_mov_b32 %[gid], v79\n\t"                           /* depth candidates: leave with (e, alpha, id) */                \
          "s_branch 9f\n\t"                                                                                                      \
          "5:\n\t"                                                                                                               \
          "s_mov_b64 exec, -1\n\t"                                                                                               \
          "s_branch 1b\n\t"                                                                                                      \
          "9:\n\t"                                                                                                               \
          : [WD] "=&s"(WD), [tm] "=&s"(tm), [D] "+s"(Dm), [e] "=&v"(e), [addr] "=&v"(addr), [al] "=&v"(al), [gid] "=&v"(gid),    \
            [dx] "=&v"(dx), [dy] "=&v"(dy), [t1] "=&v"(t1), [t2] "=&v"(t2), [pw] "=&v"(pw), [tT] "=&v"(tT), [w] "=&v"(w),         \
            [cur] "+v"(cur), [c] "+v"(c), [T] "+v"(T), [C0] "+v"(C0), [C1] "+v"(C1), [C2] "+v"(C2), [bw] "+v"(best_w),            \
            [bid] "+v"(best_id), [last] "+v"(last_contributor), [steps] "+s"(steps_w), [ev] "+s"(evals_lo)                       \
          : [px] "v"(pxf), [py] "v"(pyf), [thr] "s"(thr), [opq] "s"(opq), [base1] "s"(base1), [did] "v"(d_id), [rb] "s"(rec_base), \
            [lb] "v"(live_base), [nch] "s"(nch)                                                                                  \
          : "vcc", "scc", "memory", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79")
      for (;;) {
        unsigned long long WD, tm;
        uint32_t e, addr, gid;
        float al, dx, dy, t1, t2, pw, tT, w;
        if constexpr (STAMP) RTGS_FWD_WALK("s_add_u32 %[steps], %[steps], 1\n\t");
        else if constexpr (COUNT) RTGS_FWD_WALK("s_bcnt1_i32_b64 %[steps], %[tm]\n\ts_add_u32 %[ev], %[ev], %[steps]\n\t");
        else RTGS_FWD_WALK("");
        if (WD == 0ull) break;
        {                                                          // rare: opaque-surface depth candidates
          unsigned long long mine;                                 // (opaque: the test above stays a SCALAR branch)
          asm volatile("v_lshrrev_b64 %[t], %[l], %[m]" : [t] "=v"(mine) : [l] "v"(lane), [m] "s"(WD));
          if ((uint32_t)mine & 1u) {
            const float rx = (pxf - p.cx) / p.fx, ry = (pyf - p.cy) / p.fy;
            const float rnorm = sqrtf(rx * rx + ry * ry + 1.f);
            const float4 r3 = s_rec[RI(e, 3)];    // nx ny nz pd
            const float den = r3.x * rx + r3.y * ry + r3.z;
            if (fabsf(den) / rnorm > p.normal_thr) {
              const float zhit = r3.w / den;
              if (zhit > 0.f && fabsf(zhit - s_z[e]) < p.depth_thr) {
                D = zhit; d_w = al; d_id = (int)gid; d_pos = (uint32_t)base + e; d_iden = 1.f / den;
              }
            }
          }
        }
      }
#undef RTGS_FWD_WALK
      if constexpr (STAMP) st_steps += steps_w;
      if constexpr (COUNT) evals_w += evals_lo;
    } else
    for (;;) {
      while (cur == 0u && c + 1 < nch) { ++c; cur = s_live[blk][c]; }
      // a lane walks on while ITS pixel is open and its row's sub-list has entries (a row whose pixels have all stopped drops
      // out of the loop condition by itself: no per-row "anyone alive" test)
      const bool has = cur != 0u && !done;
      if (__builtin_amdgcn_ballot_w64(has) == 0ull) break;     // every pixel of the wave is through its row's sub-list (or finished)
      const int e = has ? (c << 5) + __builtin_ctz(cur) : 0;   // (a lane without an entry reads slot 0 - a STAGED record: its finite values times w = 0 leave the sums alone)
      cur &= cur - 1u;
      if constexpr (STAMP) st_steps += 1;
      const float4 r0 = s_rec[RI(e, 0)];        // u v ca cb
      const float4 r1 = s_rec[RI(e, 1)];        // cc o r g
      const float4 r2 = s_rec[RI(e, 2)];        // b - - id
      const float dx = r0.x - pxf, dy = r0.y - pyf;
      const float power = splat_power(r0.z, r0.w, r1.x, dx, dy);
      const float al = fminf(0.99f, r1.y * splat_exp(fminf(power, 0.f)));
      const bool ok = has && !(power > 0.f) && !(al < 1.f / 255.f);
      const float test_T = T * (1.f - al);
      const bool stop = ok && (test_T < p.T_thr);
      const bool contrib = ok && !stop;
      if constexpr (COUNT) evals += has ? 1u : 0u;
      done = done || stop;
      const float w = contrib ? al * T : 0.f;
      T = contrib ? test_T : T;
      last_contributor = contrib ? (uint32_t)(base + e + 1) : last_contributor;
      if (__builtin_amdgcn_ballot_w64(contrib) == 0ull) continue;
      C0 += r1.z * w; C1 += r1.w * w; C2 += r2.x * w;
      const bool better = w > best_w;            // w == 0 for non-contributing lanes, best_w >= 0
      const int gid = (int)__float_as_uint(r2.w);
      best_w = better ? w : best_w;
      best_id = better ? gid : best_id;
      const bool want_depth = contrib && d_id < 0 && al > p.opaque_thr;
      if (__builtin_amdgcn_ballot_w64(want_depth) != 0ull) {   // rare: opaque-surface depth candidates
        if (want_depth) {
          // the pixel's ray, recomputed here (a handful of times per pixel) rather than held in registers across the walk
          const float rx = (pxf - p.cx) / p.fx, ry = (pyf - p.cy) / p.fy;
          const float rnorm = sqrtf(rx * rx + ry * ry + 1.f);
          const float4 r3 = s_rec[RI(e, 3)];    // nx ny nz pd
          const float den = r3.x * rx + r3.y * ry + r3.z;
          if (fabsf(den) / rnorm > p.normal_thr) {
            const float zhit = r3.w / den;
            if (zhit > 0.f && fabsf(zhit - s_z[e]) < p.depth_thr) {
              D = zhit; d_w = al; d_id = gid; d_pos = (uint32_t)(base + e); d_iden = 1.f / den;
            }
          }
        }
      }
    }
    if constexpr (STAMP) st_walk += __builtin_readcyclecounter() - w_in;
  }

  // ---- epilogue: ONE barrier.  Every wave leaves four words - (block, entry) pairs of its sub-lists, entries it staged, its last
  // contributor, whether all its pixels have stopped - and every thread reads the sixteen.
  __shared__ uint32_t s_ep[4][4];
  {
    const uint32_t wl = wave_max_u32(last_contributor);
    const bool wave_done = ASMW ? (Dm == ~0ull) : (__builtin_amdgcn_ballot_w64(done) == ~0ull);
    if (lane == 0) { s_ep[wv][0] = reach_sum; s_ep[wv][1] = staged; s_ep[wv][2] = wl; s_ep[wv][3] = wave_done ? 1u : 0u; }
  }
  __syncthreads();
  const uint32_t reach_all = s_ep[0][0] + s_ep[1][0] + s_ep[2][0] + s_ep[3][0], staged_all = s_ep[0][1] + s_ep[1][1] + s_ep[2][1] + s_ep[3][1];
  const uint32_t tl_all = max(max(s_ep[0][2], s_ep[1][2]), max(s_ep[2][2], s_ep[3][2]));
  bool write_out = true;
  if (sp.mode == 1) {
    // A tile whose every pixel stopped inside the near slice is final: the slice's list is a prefix of the tile's full
    // list (depth bins are monotone in depth), so nothing behind it would have been read.  Anything else is redone
    // from scratch by pass 2 (which overwrites every output of the tile: nothing is written for it here).
    const bool finished = (s_ep[0][3] & s_ep[1][3] & s_ep[2][3] & s_ep[3][3]) != 0u;
    const bool on = __builtin_amdgcn_readfirstlane(on_v) != 0;
    if (tid == 0) {
      sp.mask2[tile] = (on && !finished) ? 1 : 0;
      sp.ranges_bwd[tile] = finished ? range : make_uint2(0u, 0u);
      sp.ranges_main[tile] = make_uint2(0u, 0u);
    }
    write_out = finished || !on;
  }

  if (inside && write_out) {
    const size_t pix = (size_t)py * p.W + px;
    const size_t HW = (size_t)p.H * p.W;
    out_color[pix] = C0 + T * p.bg[0];
    out_color[HW + pix] = C1 + T * p.bg[1];
    out_color[2 * HW + pix] = C2 + T * p.bg[2];
    out_depth[pix] = D;
    out_cidx[pix] = best_id;
    out_didx[pix] = d_id;
    out_cw[pix] = best_w;
    out_dw[pix] = d_w;
    out_T[pix] = T;
    n_contrib[pix] = last_contributor;
    depth_pos[pix] = d_pos;
    if (tc.depth_aux != nullptr) tc.depth_aux[pix] = make_float2(d_iden, D);
  }
  if (tid == 0 && write_out) {
    // For the backward: which walk the tile takes (raster_bwd.hip) - row-granular when its 4x4 blocks need, on average,
    // less than ROWS_MAX_SHARE of the entries staged, measured here on the sub-lists themselves, no heuristic about the
    // scene - and the tile's last contributor (the backward stages no further; it would otherwise have to reduce
    // n_contrib over the tile before it can issue its first load).
    // bits 0..1 = the walk; bit 2: this forward left the tile's records / masks / plane words in the TileCache; bits 8.. =
    // the measured share in 1/1000 (diagnostics).  walk < 0: the choice between the two pixel-per-lane walks from the
    // share; else the walk the context asks for
    const float share = staged_all ? (float)reach_all / (16.f * (float)staged_all) : 1.f;
    tile_mode[tile] = (walk >= 0 ? (uint32_t)walk : (share < ROWS_MAX_SHARE ? 1u : 0u)) | (tc.recs != nullptr ? 4u : 0u) |
                      ((uint32_t)(share * 1000.f) << 8);
    tile_last[tile] = tl_all;
  }
  if (COUNT && counters) {
    // work accounting for the roofline: entries any pixel of this tile consumed, and
    // (entry, pixel) pairs evaluated
    __shared__ unsigned int s_max;
    __shared__ unsigned long long s_ev;
    if (tid == 0) { s_max = 0; s_ev = 0; }
    __syncthreads();
    atomicMax(&s_max, last_contributor);
    if constexpr (ASMW) { if (lane == 0) atomicAdd(&s_ev, evals_w); }
    else atomicAdd(&s_ev, (unsigned long long)evals);
    __syncthreads();
    // one slot pair per tile, plain stores (3 225 same-address atomics cost ~80 us - more than the kernel itself);
    // the second pass of a two-pass forward adds to what the first wrote for the tile
    if (tid == 0) {
      const bool add = sp.mode == 2;
      counters[2 * tile] = (add ? counters[2 * tile] : 0ull) + (unsigned long long)s_max;
      counters[2 * tile + 1] = (add ? counters[2 * tile + 1] : 0ull) + s_ev;
    }
  }
  if constexpr (STAMP) {
    if (lane == 0 && stamps != nullptr) {
      unsigned long long* o = stamps + (size_t)(tile * 4 + wv) * 8;
      o[0] = st_wall; o[1] = wall_clock64(); o[2] = st_range; o[3] = st_first; o[4] = st_walk;
      o[5] = __builtin_readcyclecounter() - st_cyc; o[6] = st_steps; o[7] = st_batches;
    }
  }
}
#undef RI


static int g_f1_occ = [] { const char* e = getenv("RTGS_F1_OCC"); const int v = e ? atoi(e) : 6; return v == 5 ? 5 : 6; }();
static int g_f1_asm = [] { const char* e = getenv("RTGS_F1_ASM"); return e ? atoi(e) : 1; }();
static unsigned long long* g_fwd_stamps = nullptr;   // measurement only (rtgs_raster_set_fwd_stamps)
void set_fwd_stamps(void* dev) { g_fwd_stamps = (unsigned long long*)dev; }

// ------------------------------------------------------------------ host-side launch helpers
void launch_mask_sat(const int32_t* mask, int gx, int gy, int32_t* sat, hipStream_t st) {
  hipLaunchKernelGGL(mask_sat_kernel, dim3(1), dim3(256), 0, st, mask, gx, gy, sat);
}
void launch_preprocess_fwd(const RasterParams& p, const float* means, const float* opac, const float* shs,
                           const float* scales, const float* rots, const float* normal_w, const int32_t* sat,
                           Splat* splats, uint32_t* tiles_touched, int32_t* radii, uint8_t* clamped,
                           int32_t* out_radii, uint32_t* zero_words, int zero_n, uint8_t* zbin, hipStream_t st) {
  if (p.P == 0) return;
  hipLaunchKernelGGL(preprocess_fwd_kernel<0>, dim3((p.P + 255) / 256), dim3(256), 0, st, p, means, opac, shs, scales,
                     rots, normal_w, sat, splats, tiles_touched, radii, clamped, out_radii, zero_words, zero_n, zbin,
                     (float2*)nullptr, SliceList{nullptr, nullptr}, SliceSel{0, nullptr, nullptr, 0u, 0u, nullptr, nullptr, nullptr});
}
// two-pass forward, stage 1: geometry of every Gaussian
void launch_preprocess_cull(const RasterParams& p, const float* means, const float* scales, const float* rots,
                            uint32_t* tiles_touched, int32_t* radii, int32_t* out_radii, uint32_t* zero_words, int zero_n,
                            uint8_t* zbin, float2* uv, hipStream_t st) {
  if (p.P == 0) return;
  hipLaunchKernelGGL(preprocess_fwd_kernel<1>, dim3((p.P + 255) / 256), dim3(256), 0, st, p, means, (const float*)nullptr,
                     (const float*)nullptr, scales, rots, (const float*)nullptr, (const int32_t*)nullptr, (Splat*)nullptr,
                     tiles_touched, radii, (uint8_t*)nullptr, out_radii, zero_words, zero_n, zbin, uv,
                     SliceList{nullptr, nullptr}, SliceSel{0, nullptr, nullptr, 0u, 0u, nullptr, nullptr, nullptr});
}
// two-pass forward, stage 2: Splat records of the work list (max_items bounds its length), or of everything else
void launch_preprocess_shade(const RasterParams& p, const float* means, const float* opac, const float* shs,
                             const float* scales, const float* rots, const float* normal_w, Splat* splats,
                             int32_t* radii, uint8_t* clamped, float2* uv, SliceList list, SliceSel sel, size_t max_items,
                             hipStream_t st) {
  if (p.P == 0) return;
  const size_t n = list.ids ? (max_items < (size_t)p.P ? max_items : (size_t)p.P) : (size_t)p.P;
  hipLaunchKernelGGL(preprocess_fwd_kernel<2>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p, means, opac, shs,
                     scales, rots, normal_w, (const int32_t*)nullptr, splats, (uint32_t*)nullptr, radii, clamped,
                     (int32_t*)nullptr, (uint32_t*)nullptr, 0, (uint8_t*)nullptr, uv, list, sel);
}
void launch_emit_keys(const RasterParams& p, const Splat* splats, const int32_t* radii, const uint32_t* offsets,
                      const int32_t* mask, uint64_t* keys, uint32_t* vals, hipStream_t st) {
  if (p.P == 0) return;
  hipLaunchKernelGGL(emit_keys_kernel, dim3((p.P + 255) / 256), dim3(256), 0, st, p, splats, radii, offsets, mask,
                     keys, vals);
}
void launch_tile_ranges(int64_t R, const uint64_t* keys, uint2* ranges, hipStream_t st) {
  if (R == 0) return;
  hipLaunchKernelGGL(tile_ranges_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, st, R, keys, ranges);
}
// After the near slice's blend: count the tiles it finished / left (from the tile mask it wrote - per-tile atomics on
// one counter would serialise: 3 225 of them cost 55 us) and publish the totals to the host, which spins on host[7].
__global__ void __launch_bounds__(256) slice_publish_kernel(int ntiles, const int32_t* __restrict__ user_mask,
                                                            const int32_t* __restrict__ mask2,
                                                            const uint32_t* __restrict__ r1, uint32_t* __restrict__ ctr,
                                                            uint32_t* __restrict__ host, uint32_t seq,
                                                            uint32_t* __restrict__ spec_fail,
                                                            const uint32_t* __restrict__ seg_count) {
  int left = 0, fin = 0, inst = 0;
  for (int t = threadIdx.x; t < ntiles; t += 256) {
    const bool on = user_mask[t] != 0, l = mask2[t] != 0;
    left += l ? 1 : 0;
    fin += (on && !l) ? 1 : 0;
    if (seg_count) inst += (int)seg_count[t];      // one-pass placement: nobody scanned the counts, the total is summed here
  }
  __shared__ int s_l[4], s_f[4], s_i[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { left += __shfl_xor(left, off); fin += __shfl_xor(fin, off); inst += __shfl_xor(inst, off); }
  if ((threadIdx.x & 63) == 0) { s_l[threadIdx.x >> 6] = left; s_f[threadIdx.x >> 6] = fin; s_i[threadIdx.x >> 6] = inst; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t L = (uint32_t)(s_l[0] + s_l[1] + s_l[2] + s_l[3]), F = (uint32_t)(s_f[0] + s_f[1] + s_f[2] + s_f[3]);
    ctr[0] = L; ctr[1] = F;                      // device copy: pass 2's kernels exit at once when nothing is left
    if (spec_fail) *spec_fail = L != 0u ? 1u : 0u;   // speculative forward: the host assumed the slice finishes every tile
    const uint32_t R1 = seg_count ? (uint32_t)(s_i[0] + s_i[1] + s_i[2] + s_i[3]) : r1[0];
    const uint32_t w[7] = {ctr[2], 0u, L, F, R1, 0u, 0u};     // [0]: length of the slice's work list
    publish_to_host(host, w, seq);
  }
}
void launch_slice_publish(int ntiles, const int32_t* user_mask, const int32_t* mask2, const uint32_t* r1, uint32_t* ctr,
                          uint32_t* host, uint32_t seq, uint32_t* spec_fail, hipStream_t st, const uint32_t* seg_count) {
  hipLaunchKernelGGL(slice_publish_kernel, dim3(1), dim3(256), 0, st, ntiles, user_mask, mask2, r1, ctr, host, seq, spec_fail,
                     seg_count);
}

void launch_blend_fwd(const RasterParams& p, const uint2* ranges, const uint32_t* point_list, const Splat* splats,
                      float* out_color, float* out_depth, int32_t* out_cidx, int32_t* out_didx, float* out_cw,
                      float* out_dw, float* out_T, uint32_t* n_contrib, unsigned long long* counters,
                      SlicePass sp, uint32_t* tile_mode, uint32_t* depth_pos, uint32_t* tile_last, int walk, uint32_t* aux_zero,
                      TileCache tc, uint32_t seg, hipStream_t st) {
#define RTGS_FWD1(OCC, STAMP, COUNT, ASMW) hipLaunchKernelGGL((blend_fwd_kernel<OCC, STAMP, COUNT, ASMW>), dim3(p.gx, p.gy), dim3(BLOCK), 0, st, p, ranges, point_list, splats, \
                       out_color, out_depth, out_cidx, out_didx, out_cw, out_dw, out_T, n_contrib, counters, sp, tile_mode, \
                       depth_pos, tile_last, walk, aux_zero, tc, g_fwd_stamps, seg)
  if (g_f1_asm == 0) RTGS_FWD1(6, false, true, false);      // the compiler's walk loop (A-B, tests): RTGS_F1_ASM=0
  else if (counters) RTGS_FWD1(6, false, true, true);        // work counters asked for (bench.py's roofline, tests): the accounting variant
  else if (g_fwd_stamps) RTGS_FWD1(6, true, false, true);
  else if (g_f1_occ == 5) RTGS_FWD1(5, false, false, true); else RTGS_FWD1(6, false, false, true);   // (7 / 8 waves: 79.6 / 79.0 us against 80.3 in round 5; the walk's named registers need 80 VGPRs)
#undef RTGS_FWD1
}

}  // namespace rtgs
