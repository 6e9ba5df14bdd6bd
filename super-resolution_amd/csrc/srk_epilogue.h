// srk_epilogue.h -- fused conv epilogue shared by the fp32 and bf16x3 convolution kernels (device code).
#pragma once
#include "srk_internal.h"

// Fused epilogue shared by the conv kernels.  acc[m][t][reg]: pixel i = (reg&3) + 8*(reg>>2) + 4*hl of M tile m
// (rows 2wv, 2wv+1 of the m-th 8-row group), channel = n0 + 32t + l32.
template <int BN, int MT, bool ROWTILE = false>
__device__ __forceinline__ void conv_epilogue(const srk_conv_args& a, f32x16 (&acc)[MT][BN / 32], float4* smem, int n, int oh0,
                                              int ow0, int n0, int wv, int lane, int lds_slot = -1) {
  constexpr int NTN = BN / 32;
  constexpr int TILE_H = ROWTILE ? 4 * MT : SRK_TH * MT;   // ROWTILE: a 32-pixel M tile is ONE image row (tile 4MT x 32)
  constexpr int TILE_W = ROWTILE ? 32 : SRK_TW;
  const int hl = lane >> 5, l32 = lane & 31;
  // ---- epilogue.  acc[m][t][reg]: pixel i = (reg&3) + 8*(reg>>2) + 4*hl of the M tile, channel = n0 + 32t + l32.
  // Inside one M tile the lane's 16 registers are 2 rows x 8 columns {c, c+1, c+2, c+3, c+8, .., c+11} with
  // c = 4*hl: every address is  base + (reg>>3)*row_stride + col(reg)*col_stride  (32-bit offsets from one
  // 64-bit base per tensor), and interior tiles skip all bounds checks.
  const int Cps_out = a.Cout >> 2;
  const bool interior = (oh0 + TILE_H <= a.OH) && (ow0 + TILE_W <= a.OW);
  const int rowmul = a.ps_out ? 4 * a.OW : a.OW, colmul = a.ps_out ? 2 : 1;   // physical pixel steps per logical row/col
  constexpr int NG = NTN * MT;
  const bool has_r1 = a.r1 != nullptr, has_r2 = a.r2 != nullptr, has_m = a.mask != nullptr;
  // 16-byte path: every tensor the epilogue touches is float4-addressable per pixel
  const bool vec_out = ((a.Cout & 3) == 0) && (!a.ps_out || (Cps_out & 3) == 0) &&
                       ((a.y_ldc | a.y_coff) & 3) == 0 && (((uintptr_t)a.y) & 15) == 0 &&
                       (!a.bias || (((uintptr_t)a.bias) & 15) == 0) &&
                       (!has_r1 || ((((a.r1_ldc | a.r1_coff) & 3) == 0) && (((uintptr_t)a.r1) & 15) == 0)) &&
                       (!has_r2 || ((((a.r2_ldc | a.r2_coff) & 3) == 0) && (((uintptr_t)a.r2) & 15) == 0)) &&
                       (!has_m || ((((a.m_ldc | a.m_coff) & 3) == 0) && (((uintptr_t)a.mask) & 15) == 0));
  if (vec_out) {
    // Transpose each 32 px x 32 ch accumulator tile through this wave's private 4 KB of LDS so that a lane
    // owns 4 consecutive channels of one pixel: 16-byte loads/stores, 4x fewer store instructions (the
    // store tail is issue-bound).  The main loop's last barrier has retired every other use of the LDS.
    float* ls = reinterpret_cast<float*>(smem) + (lds_slot < 0 ? wv : lds_slot) * 1024;   // private 4 KB per wave
    float4* ls4 = reinterpret_cast<float4*>(ls);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int t = g / MT, m = g % MT;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
        ls[i * 32 + l32] = acc[m][t][reg];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int idx = j * 64 + lane;
        const int pl = idx >> 3, c4 = idx & 7;
        const int oh = ROWTILE ? oh0 + wv + 4 * m : oh0 + 8 * m + 2 * wv + (pl >> 4);
        const int ow = ROWTILE ? ow0 + pl : ow0 + (pl & 15);
        const int co = n0 + t * 32 + 4 * c4;
        float4 v = ls4[idx];
        if (co < a.Cout && (interior || (oh < a.OH && ow < a.OW))) {
          int ch = co, pi = 0, pj = 0;
          long pix;
          if (a.ps_out) {
            const int ij = co / Cps_out; ch = co - ij * Cps_out; pi = ij >> 1; pj = ij & 1;
            pix = ((long)(n * 2 * a.OH) + 2 * oh + pi) * (2 * a.OW) + 2 * ow + pj;
          } else {
            pix = ((long)n * a.OH + oh) * a.OW + ow;
          }
          if (a.bias) { const float4 bq = *reinterpret_cast<const float4*>(a.bias + co); v.x += bq.x; v.y += bq.y; v.z += bq.z; v.w += bq.w; }
          v.x *= a.alpha; v.y *= a.alpha; v.z *= a.alpha; v.w *= a.alpha;
          if (has_r1) { const float4 r = *reinterpret_cast<const float4*>(a.r1 + pix * a.r1_ldc + a.r1_coff + ch);
                        v.x += a.beta1 * r.x; v.y += a.beta1 * r.y; v.z += a.beta1 * r.z; v.w += a.beta1 * r.w; }
          if (has_r2) { const float4 r = *reinterpret_cast<const float4*>(a.r2 + pix * a.r2_ldc + a.r2_coff + ch);
                        v.x += a.beta2 * r.x; v.y += a.beta2 * r.y; v.z += a.beta2 * r.z; v.w += a.beta2 * r.w; }
          v.x = v.x > 0.f ? v.x : v.x * a.slope; v.y = v.y > 0.f ? v.y : v.y * a.slope;
          v.z = v.z > 0.f ? v.z : v.z * a.slope; v.w = v.w > 0.f ? v.w : v.w * a.slope;
          if (has_m) { const float4 q = *reinterpret_cast<const float4*>(a.mask + pix * a.m_ldc + a.m_coff + ch);
                       v.x *= (q.x > 0.f ? 1.f : a.mask_slope); v.y *= (q.y > 0.f ? 1.f : a.mask_slope);
                       v.z *= (q.z > 0.f ? 1.f : a.mask_slope); v.w *= (q.w > 0.f ? 1.f : a.mask_slope); }
#ifdef SRK_NO_STORE
          asm volatile("" :: "v"(v.x), "v"(v.y), "v"(v.z), "v"(v.w));
#else
          *reinterpret_cast<float4*>(a.y + pix * a.y_ldc + a.y_coff + ch) = v;
#endif
        }
      }
    }
  } else {
    // scalar path (Cout not a multiple of 4, e.g. the F->1 tail conv, or unaligned views): one dword per lane
    const int y_rs = rowmul * a.y_ldc, y_cs = colmul * a.y_ldc;
    const int r1_rs = rowmul * a.r1_ldc, r1_cs = colmul * a.r1_ldc;
    const int r2_rs = rowmul * a.r2_ldc, r2_cs = colmul * a.r2_ldc;
    const int m_rs = rowmul * a.m_ldc, m_cs = colmul * a.m_ldc;
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      const int co = n0 + t * 32 + l32;
      if (co >= a.Cout) continue;
      const float bz = a.bias ? a.bias[co] : 0.f;
      int ch = co, pi = 0, pj = 0;
      if (a.ps_out) { const int ij = co / Cps_out; ch = co - ij * Cps_out; pi = ij >> 1; pj = ij & 1; }
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const int ohb = ROWTILE ? oh0 + wv + 4 * m : oh0 + 8 * m + 2 * wv, owb = ow0 + 4 * hl;
        long pix0;
        if (a.ps_out) pix0 = ((long)(n * 2 * a.OH) + 2 * ohb + pi) * (2 * a.OW) + 2 * owb + pj;
        else pix0 = ((long)n * a.OH + ohb) * a.OW + owb;
        float* yb = a.y + pix0 * a.y_ldc + a.y_coff + ch;
        const float* r1b = has_r1 ? a.r1 + pix0 * a.r1_ldc + a.r1_coff + ch : nullptr;
        const float* r2b = has_r2 ? a.r2 + pix0 * a.r2_ldc + a.r2_coff + ch : nullptr;
        const float* mb = has_m ? a.mask + pix0 * a.m_ldc + a.m_coff + ch : nullptr;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int rr = ROWTILE ? 0 : reg >> 3, cc = ROWTILE ? (reg & 3) + 8 * (reg >> 2) : (reg & 3) + 8 * ((reg >> 2) & 1);
          if (!interior && (ohb + rr >= a.OH || owb + cc >= a.OW)) continue;
          float v = a.alpha * (acc[m][t][reg] + bz);
          if (has_r1) v += a.beta1 * r1b[rr * r1_rs + cc * r1_cs];
          if (has_r2) v += a.beta2 * r2b[rr * r2_rs + cc * r2_cs];
          v = v > 0.f ? v : v * a.slope;
          if (has_m) v *= (mb[rr * m_rs + cc * m_cs] > 0.f ? 1.f : a.mask_slope);
          yb[rr * y_rs + cc * y_cs] = v;
        }
      }
    }
  }
}

