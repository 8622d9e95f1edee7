// Chain rule of one Gaussian from its SplatGrad record to the gradients of the rasterizer's inputs (K8, SURVEY.md Appendix B
// "Backward"): recomputes the forward intermediates from the inputs (cheaper than storing them).  ONE definition, shared
// by preprocess_bwd (raster_bwd.hip: stores the rows) and the fused tail of the one-call map step (map_fused.hip: keeps
// them in registers).
#pragma once
#include "raster_common.h"

namespace rtgs {

struct ChainOut {
  float dm[3];       // dL/d means3D
  float dop;         // dL/d opacity
  float ds[3];       // dL/d scales
  float4 dq;         // dL/d rotations (w x y z)
  float dn[3];       // dL/d normal_w
  float gc[3];       // dL/d rgb with the clamp applied: the SH gradient row is basis[k] * gc[c]
  float basis[16];   // SH basis at the view direction (0 beyond the active degree)
};

// dsh: the Gaussian's row of dL/d shs to store (nullptr: not stored).
__device__ __forceinline__ void chain_rule(const RasterParams& p, const int i, const float* __restrict__ means,
                                           const float* __restrict__ shs, const float* __restrict__ scales,
                                           const float* __restrict__ rots, const float* __restrict__ normal_w,
                                           const uint8_t* __restrict__ clamped, const SplatGrad& g, ChainOut& o,
                                           float* __restrict__ dsh) {
  const float* V = p.view;
  const float mx = means[3 * i], my = means[3 * i + 1], mz = means[3 * i + 2];
  const float pcx = V[0] * mx + V[4] * my + V[8] * mz + V[12];
  const float pcy = V[1] * mx + V[5] * my + V[9] * mz + V[13];
  const float pcz = V[2] * mx + V[6] * my + V[10] * mz + V[14];
  const float iz = 1.f / pcz, iz2 = iz * iz, iz3 = iz2 * iz;

  // ---- forward recompute: Sigma3D, T = J Wr, Sigma2D
  const float4 q4 = reinterpret_cast<const float4*>(rots)[i];
  const float qr = q4.x, qx = q4.y, qy = q4.z, qz = q4.w;
  const float s0 = scales[3 * i] * p.scale_modifier, s1 = scales[3 * i + 1] * p.scale_modifier,
              s2 = scales[3 * i + 2] * p.scale_modifier;
  const float R[9] = {1.f - 2.f * (qy * qy + qz * qz), 2.f * (qx * qy - qr * qz), 2.f * (qx * qz + qr * qy),
                      2.f * (qx * qy + qr * qz), 1.f - 2.f * (qx * qx + qz * qz), 2.f * (qy * qz - qr * qx),
                      2.f * (qx * qz - qr * qy), 2.f * (qy * qz + qr * qx), 1.f - 2.f * (qx * qx + qy * qy)};
  const float Mm[9] = {R[0] * s0, R[1] * s1, R[2] * s2, R[3] * s0, R[4] * s1, R[5] * s2, R[6] * s0, R[7] * s1, R[8] * s2};
  const float S00 = Mm[0] * Mm[0] + Mm[1] * Mm[1] + Mm[2] * Mm[2];
  const float S01 = Mm[0] * Mm[3] + Mm[1] * Mm[4] + Mm[2] * Mm[5];
  const float S02 = Mm[0] * Mm[6] + Mm[1] * Mm[7] + Mm[2] * Mm[8];
  const float S11 = Mm[3] * Mm[3] + Mm[4] * Mm[4] + Mm[5] * Mm[5];
  const float S12 = Mm[3] * Mm[6] + Mm[4] * Mm[7] + Mm[5] * Mm[8];
  const float S22 = Mm[6] * Mm[6] + Mm[7] * Mm[7] + Mm[8] * Mm[8];
  const float limx = 1.3f * p.tanfovx, limy = 1.3f * p.tanfovy;
  const float txtz = pcx * iz, tytz = pcy * iz;
  const bool in_x = (txtz >= -limx) && (txtz <= limx);
  const bool in_y = (tytz >= -limy) && (tytz <= limy);
  const float tx = fminf(limx, fmaxf(-limx, txtz)) * pcz;
  const float ty = fminf(limy, fmaxf(-limy, tytz)) * pcz;
  const float J00 = p.fx * iz, J02 = -p.fx * tx * iz2;
  const float J11 = p.fy * iz, J12 = -p.fy * ty * iz2;
  const float T0[3] = {J00 * V[0] + J02 * V[2], J00 * V[4] + J02 * V[6], J00 * V[8] + J02 * V[10]};
  const float T1[3] = {J11 * V[1] + J12 * V[2], J11 * V[5] + J12 * V[6], J11 * V[9] + J12 * V[10]};
  const float ST0[3] = {S00 * T0[0] + S01 * T0[1] + S02 * T0[2], S01 * T0[0] + S11 * T0[1] + S12 * T0[2],
                        S02 * T0[0] + S12 * T0[1] + S22 * T0[2]};
  const float ST1[3] = {S00 * T1[0] + S01 * T1[1] + S02 * T1[2], S01 * T1[0] + S11 * T1[1] + S12 * T1[2],
                        S02 * T1[0] + S12 * T1[1] + S22 * T1[2]};
  const float a = T0[0] * ST0[0] + T0[1] * ST0[1] + T0[2] * ST0[2] + 0.3f;
  const float b = T0[0] * ST1[0] + T0[1] * ST1[1] + T0[2] * ST1[2];
  const float c = T1[0] * ST1[0] + T1[1] * ST1[1] + T1[2] * ST1[2] + 0.3f;
  const float det = a * c - b * b;
  const float id2 = 1.f / (det * det);

  // ---- conic (A,B,C) = (c,-b,a)/det  ->  (a,b,c)
  const float gA = g.dca, gB = g.dcb, gC = g.dcc;
  const float da = id2 * (-c * c * gA + b * c * gB - b * b * gC);
  const float db = id2 * (2.f * b * c * gA - (a * c + b * b) * gB + 2.f * a * b * gC);
  const float dc = id2 * (-b * b * gA + a * b * gB - a * a * gC);

  // ---- Sigma2D -> Sigma3D (symmetric full-matrix gradient) and -> T
  float dS[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int cidx = 0; cidx < 3; ++cidx)
      dS[r * 3 + cidx] = da * T0[r] * T0[cidx] + 0.5f * db * (T0[r] * T1[cidx] + T1[r] * T0[cidx]) + dc * T1[r] * T1[cidx];
  float dT0[3], dT1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    dT0[k] = 2.f * da * ST0[k] + db * ST1[k];
    dT1[k] = 2.f * dc * ST1[k] + db * ST0[k];
  }
  // T = J Wr: dJ[r][k] = sum_j dT[r][j] Wr[k][j],  Wr[k][j] = V[j*4+k]
  const float dJ00 = dT0[0] * V[0] + dT0[1] * V[4] + dT0[2] * V[8];
  const float dJ02 = dT0[0] * V[2] + dT0[1] * V[6] + dT0[2] * V[10];
  const float dJ11 = dT1[0] * V[1] + dT1[1] * V[5] + dT1[2] * V[9];
  const float dJ12 = dT1[0] * V[2] + dT1[1] * V[6] + dT1[2] * V[10];
  float dpx = in_x ? dJ02 * (-p.fx * iz2) : 0.f;
  float dpy = in_y ? dJ12 * (-p.fy * iz2) : 0.f;
  float dpz = -p.fx * iz2 * dJ00 - p.fy * iz2 * dJ11 + 2.f * p.fx * tx * iz3 * dJ02 + 2.f * p.fy * ty * iz3 * dJ12;

  // ---- pixel centre
  dpx += g.du * p.fx * iz;
  dpy += g.dv * p.fy * iz;
  dpz += -g.du * p.fx * pcx * iz2 - g.dv * p.fy * pcy * iz2;

  // ---- depth plane: pd = n_c . p_c, n_c = Wr n_w
  const float nwx = normal_w[3 * i], nwy = normal_w[3 * i + 1], nwz = normal_w[3 * i + 2];
  const float ncx = V[0] * nwx + V[4] * nwy + V[8] * nwz;
  const float ncy = V[1] * nwx + V[5] * nwy + V[9] * nwz;
  const float ncz = V[2] * nwx + V[6] * nwy + V[10] * nwz;
  dpx += g.dpd * ncx; dpy += g.dpd * ncy; dpz += g.dpd * ncz;
  const float dncx = g.dnx + g.dpd * pcx, dncy = g.dny + g.dpd * pcy, dncz = g.dnz + g.dpd * pcz;
  o.dn[0] = V[0] * dncx + V[1] * dncy + V[2] * dncz;        // Wr^T
  o.dn[1] = V[4] * dncx + V[5] * dncy + V[6] * dncz;
  o.dn[2] = V[8] * dncx + V[9] * dncy + V[10] * dncz;

  // p_c = Wr mu + t
  float dmx = V[0] * dpx + V[1] * dpy + V[2] * dpz;
  float dmy = V[4] * dpx + V[5] * dpy + V[6] * dpz;
  float dmz = V[8] * dpx + V[9] * dpy + V[10] * dpz;

  // ---- Sigma3D = M M^T, M = R diag(s)
  float dM[9];
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int k = 0; k < 3; ++k)
      dM[r * 3 + k] = 2.f * (dS[r * 3 + 0] * Mm[0 * 3 + k] + dS[r * 3 + 1] * Mm[1 * 3 + k] + dS[r * 3 + 2] * Mm[2 * 3 + k]);
  o.ds[0] = p.scale_modifier * (dM[0] * R[0] + dM[3] * R[3] + dM[6] * R[6]);
  o.ds[1] = p.scale_modifier * (dM[1] * R[1] + dM[4] * R[4] + dM[7] * R[7]);
  o.ds[2] = p.scale_modifier * (dM[2] * R[2] + dM[5] * R[5] + dM[8] * R[8]);
  const float G00 = dM[0] * s0, G01 = dM[1] * s1, G02 = dM[2] * s2;
  const float G10 = dM[3] * s0, G11 = dM[4] * s1, G12 = dM[5] * s2;
  const float G20 = dM[6] * s0, G21 = dM[7] * s1, G22 = dM[8] * s2;
  o.dq = make_float4(
      2.f * (-qz * G01 + qy * G02 + qz * G10 - qx * G12 - qy * G20 + qx * G21),
      2.f * (qy * G01 + qz * G02 + qy * G10 - 2.f * qx * G11 - qr * G12 + qz * G20 + qr * G21 - 2.f * qx * G22),
      2.f * (-2.f * qy * G00 + qx * G01 + qr * G02 + qx * G10 + qz * G12 - qr * G20 + qz * G21 - 2.f * qy * G22),
      2.f * (-2.f * qz * G00 - qr * G01 + qx * G02 + qr * G10 - 2.f * qz * G11 + qy * G12 + qx * G20 + qy * G21));

  // ---- colour: rgb = max(SH(dir) + 0.5, 0)
  float ddx = mx - p.campos[0], ddy = my - p.campos[1], ddz = mz - p.campos[2];
  const float len = sqrtf(ddx * ddx + ddy * ddy + ddz * ddz);
  const float il = 1.f / len;
  const float x = ddx * il, y = ddy * il, z = ddz * il;
  const uint8_t cl = clamped[i];
  const float gc[3] = {(cl & 1) ? 0.f : g.dr, (cl & 2) ? 0.f : g.dg, (cl & 4) ? 0.f : g.db};
  float dRdx = 0.f, dRdy = 0.f, dRdz = 0.f;   // dL/d(dir)
  const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
  float basis[16];
  basis[0] = RTGS_SH_C0;
  basis[1] = -RTGS_SH_C1 * y; basis[2] = RTGS_SH_C1 * z; basis[3] = -RTGS_SH_C1 * x;
  basis[4] = RTGS_SH_C2_0 * xy; basis[5] = RTGS_SH_C2_1 * yz; basis[6] = RTGS_SH_C2_2 * (2.f * zz - xx - yy);
  basis[7] = RTGS_SH_C2_3 * xz; basis[8] = RTGS_SH_C2_4 * (xx - yy);
  basis[9] = RTGS_SH_C3_0 * y * (3.f * xx - yy); basis[10] = RTGS_SH_C3_1 * xy * z;
  basis[11] = RTGS_SH_C3_2 * y * (4.f * zz - xx - yy); basis[12] = RTGS_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
  basis[13] = RTGS_SH_C3_4 * x * (4.f * zz - xx - yy); basis[14] = RTGS_SH_C3_5 * z * (xx - yy);
  basis[15] = RTGS_SH_C3_6 * x * (xx - 3.f * yy);
  const int ncoef = (p.deg + 1) * (p.deg + 1);
#pragma unroll
  for (int k = 0; k < 16; ++k) o.basis[k] = (k < ncoef) ? basis[k] : 0.f;
  o.gc[0] = gc[0]; o.gc[1] = gc[1]; o.gc[2] = gc[2];
  if (dsh) {                                  // the SH gradient row basis[k] * gc[c]; the fused tail forms it on the fly
    if (p.M == 16) {
      float ov[48];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float bk = o.basis[k];
        ov[3 * k] = bk * gc[0]; ov[3 * k + 1] = bk * gc[1]; ov[3 * k + 2] = bk * gc[2];
      }
      float4* d4 = reinterpret_cast<float4*>(dsh);
#pragma unroll
      for (int q = 0; q < 12; ++q) d4[q] = make_float4(ov[4 * q], ov[4 * q + 1], ov[4 * q + 2], ov[4 * q + 3]);
    } else {
      for (int k = 0; k < p.M; ++k) {
        const float bk = (k < ncoef) ? basis[k] : 0.f;
        dsh[3 * k] = bk * gc[0]; dsh[3 * k + 1] = bk * gc[1]; dsh[3 * k + 2] = bk * gc[2];
      }
    }
  }
  // (the SH block is read only now, after the 48 gradient values above are stored and dead)
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
  // d(basis_k)/d(x,y,z) contracted with sum_c gc[c] * sh[k][c]
  auto shg = [&](int k) { return gc[0] * sh[3 * k] + gc[1] * sh[3 * k + 1] + gc[2] * sh[3 * k + 2]; };
  if (p.deg > 0) {
    dRdx += -RTGS_SH_C1 * shg(3); dRdy += -RTGS_SH_C1 * shg(1); dRdz += RTGS_SH_C1 * shg(2);
    if (p.deg > 1) {
      const float h4 = shg(4), h5 = shg(5), h6 = shg(6), h7 = shg(7), h8 = shg(8);
      dRdx += RTGS_SH_C2_0 * y * h4 + RTGS_SH_C2_2 * (-2.f * x) * h6 + RTGS_SH_C2_3 * z * h7 + RTGS_SH_C2_4 * 2.f * x * h8;
      dRdy += RTGS_SH_C2_0 * x * h4 + RTGS_SH_C2_1 * z * h5 + RTGS_SH_C2_2 * (-2.f * y) * h6 + RTGS_SH_C2_4 * (-2.f * y) * h8;
      dRdz += RTGS_SH_C2_1 * y * h5 + RTGS_SH_C2_2 * 4.f * z * h6 + RTGS_SH_C2_3 * x * h7;
      if (p.deg > 2) {
        const float h9 = shg(9), h10 = shg(10), h11 = shg(11), h12 = shg(12), h13 = shg(13), h14 = shg(14), h15 = shg(15);
        dRdx += RTGS_SH_C3_0 * h9 * 6.f * xy + RTGS_SH_C3_1 * h10 * yz + RTGS_SH_C3_2 * h11 * (-2.f * xy) +
                RTGS_SH_C3_3 * h12 * (-6.f * xz) + RTGS_SH_C3_4 * h13 * (4.f * zz - 3.f * xx - yy) +
                RTGS_SH_C3_5 * h14 * 2.f * xz + RTGS_SH_C3_6 * h15 * 3.f * (xx - yy);
        dRdy += RTGS_SH_C3_0 * h9 * 3.f * (xx - yy) + RTGS_SH_C3_1 * h10 * xz + RTGS_SH_C3_2 * h11 * (4.f * zz - xx - 3.f * yy) +
                RTGS_SH_C3_3 * h12 * (-6.f * yz) + RTGS_SH_C3_4 * h13 * (-2.f * xy) + RTGS_SH_C3_5 * h14 * (-2.f * yz) +
                RTGS_SH_C3_6 * h15 * (-6.f * xy);
        dRdz += RTGS_SH_C3_1 * h10 * xy + RTGS_SH_C3_2 * h11 * 8.f * yz + RTGS_SH_C3_3 * h12 * (6.f * zz - 3.f * xx - 3.f * yy) +
                RTGS_SH_C3_4 * h13 * 8.f * xz + RTGS_SH_C3_5 * h14 * (xx - yy);
      }
    }
  }
  // dir = d / |d|: dL/dd = (g - dir (dir.g)) / |d|
  const float dot = x * dRdx + y * dRdy + z * dRdz;
  dmx += (dRdx - x * dot) * il;
  dmy += (dRdy - y * dot) * il;
  dmz += (dRdz - z * dot) * il;

  o.dm[0] = dmx; o.dm[1] = dmy; o.dm[2] = dmz;
  o.dop = g.dop;
}

}  // namespace rtgs
