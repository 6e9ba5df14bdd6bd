// srk_epilogue.h -- fused conv epilogue shared by the fp32 and bf16x3 convolution kernels (device code).
#pragma once
#include "srk_internal.h"
#include <type_traits>

// wave-uniform value through v_readfirstlane (folds away when it already sits in an SGPR): an opaque copy of a kernel
// argument field, so that later selects operate on VALUES
__device__ __forceinline__ int srk_sgpr_opaque(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float srk_sgpr_opaque(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
__device__ __forceinline__ const float* srk_sgpr_opaque(const float* p) {
  const unsigned long long u = (unsigned long long)p;
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)u), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(u >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}

// The 16-byte epilogue proper (see conv_epilogue).  NS = number of per-pixel tensors besides the output (r1, r2, mask: the
// "slots"), NBT = items per batch.  Straight-line code: every access is a buffer load / store through a per-image resource
// whose out-of-range offsets (W_OOB: tile pixels outside the image, channels >= Cout) load 0 / store nothing, and the
// slot roles are uniform coefficients instead of branches -- one basic block, so the compiler counts vmcnt exactly.
// SIGNS (kernels that support srk_conv_args.signs): tile = the workgroup's tile index; SRK_CONV_WRITE_SIGNS collects (o > 0) of the lane's
// NI items x 4 channels (NI <= 32: 128 bits) and stores them with one 16-byte store; SRK_CONV_MASK_SIGNS takes the LeakyReLU' mask from
// such bits (one 16-byte load) -- the mask is then NOT one of the NS tensors.
template <int BN, int MT, bool ROWTILE, int NBT, int NS, int SAUX = 0, bool SIGNS = false>
__device__ __forceinline__ void conv_epilogue_vec(const srk_conv_args& a, f32x16 (&acc)[MT][BN / 32], float* ls, int n, int oh0, int ow0,
                                                  int n0, int wv, int lane, bool interior, int tile = 0) {
  constexpr int NTN = BN / 32, NG = NTN * MT, NI = NG * 4;
  constexpr int NB = NBT < NI ? NBT : NI;           // items per batch: whole accumulator tiles
  static_assert(NB % 4 == 0 && NI % NB == 0, "prefetch batch = whole accumulator tiles");
  constexpr unsigned E_OOB = 0x80000000u;
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  const int hl = lane >> 5, l32 = lane & 31;
  const int Cps_out = a.Cout >> 2;
  const f32x4* ls4 = reinterpret_cast<const f32x4*>(ls);
  const int c4 = lane & 7, plb = lane >> 3;        // item (g, j): pixel pl = 8 j + plb of tile g, channels 4 c4 .. 4 c4 + 3
  int ch[NTN], pij[NTN];
  bool cok[NTN];
  f32x4 bq[NTN];
#pragma unroll
  for (int t = 0; t < NTN; ++t) {
    const int co = n0 + t * 32 + 4 * c4;
    cok[t] = co < a.Cout;
    ch[t] = co; pij[t] = 0;
    if (a.ps_out) { const int ij = co / Cps_out; ch[t] = co - ij * Cps_out; pij[t] = ij; }
    bq[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias && cok[t]) bq[t] = *reinterpret_cast<const f32x4*>(a.bias + co);
  }
  // (the fields go through an opaque SGPR copy first: selects between fields of the by-value argument struct otherwise
  // make SROA keep 16-byte slices of it in scratch memory)
  const float* r1p = srk_sgpr_opaque(a.r1); const float* r2p = srk_sgpr_opaque(a.r2); const float* mkp = srk_sgpr_opaque(a.mask);
  const bool has_r1 = r1p != nullptr, has_r2 = r2p != nullptr;
  const int r1l = srk_sgpr_opaque(a.r1_ldc), r1c = srk_sgpr_opaque(a.r1_coff), r2l = srk_sgpr_opaque(a.r2_ldc), r2c = srk_sgpr_opaque(a.r2_coff);
  const int mkl = srk_sgpr_opaque(a.m_ldc), mkc = srk_sgpr_opaque(a.m_coff);
  const float alpha = srk_sgpr_opaque(a.alpha), beta1 = srk_sgpr_opaque(a.beta1), beta2 = srk_sgpr_opaque(a.beta2);
  const float slope = srk_sgpr_opaque(a.slope), mask_slope = srk_sgpr_opaque(a.mask_slope);
  // slots in the order r1, r2, mask, absent ones skipped (NS = how many are present)
  const int psr = a.ps_out ? 2 : 1;
  const long img_px = (long)a.OH * a.OW * (a.ps_out ? 4 : 1);       // pixels of one output image (host-checked: bytes < 2^31)
  auto rsrc_of = [&](const float* p, int ldc, int coff) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p + (long)n * img_px * ldc + coff), 0, (unsigned)((img_px * ldc - coff) * 4), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t yrs = rsrc_of(a.y, a.y_ldc, a.y_coff);
  __amdgpu_buffer_rsrc_t srs[NS > 0 ? NS : 1];
  int sld[NS > 0 ? NS : 1];
  float scoef[NS > 0 ? NS : 1], sms[NS > 0 ? NS : 1];        // residual weight (0 for the mask); LeakyReLU' slope (1 for a residual)
  bool sres[NS > 0 ? NS : 1];                                 // slot is a residual (else: the mask)
  if constexpr (NS >= 1) {
    const bool m0 = !has_r1 && !has_r2;
    srs[0] = rsrc_of(has_r1 ? r1p : (has_r2 ? r2p : mkp), has_r1 ? r1l : (has_r2 ? r2l : mkl), has_r1 ? r1c : (has_r2 ? r2c : mkc));
    sld[0] = has_r1 ? r1l : (has_r2 ? r2l : mkl);
    scoef[0] = has_r1 ? beta1 : (has_r2 ? beta2 : 0.f);
    sms[0] = m0 ? mask_slope : 1.f; sres[0] = !m0;
  }
  if constexpr (NS >= 2) {
    const bool is2 = has_r1 && has_r2;                 // second slot: r2 when both residuals are there, else the mask
    srs[1] = rsrc_of(is2 ? r2p : mkp, is2 ? r2l : mkl, is2 ? r2c : mkc);
    sld[1] = is2 ? r2l : mkl;
    scoef[1] = is2 ? beta2 : 0.f;
    sms[1] = is2 ? 1.f : mask_slope; sres[1] = is2;
  }
  if constexpr (NS >= 3) {
    srs[2] = rsrc_of(mkp, mkl, mkc);
    sld[2] = mkl; scoef[2] = 0.f; sms[2] = mask_slope; sres[2] = false;
  }
  typedef unsigned sg_u32x4 __attribute__((ext_vector_type(4)));
  sg_u32x4 sbits = {0u, 0u, 0u, 0u};
  bool wsigns = false, msigns = false;
  float sg_slope = 1.f;
  if constexpr (SIGNS) {
    static_assert(NI <= 32, "128 sign bits per lane");
    wsigns = (a.flags & SRK_CONV_WRITE_SIGNS) != 0;
    msigns = (a.flags & SRK_CONV_MASK_SIGNS) != 0;
    sg_slope = mask_slope;
    if (msigns) sbits = *(reinterpret_cast<const sg_u32x4*>(a.signs) + ((long)tile * 4 + wv) * 64 + lane);
  }
#pragma unroll
  for (int b0 = 0; b0 < NI; b0 += NB) {
    int pix[NB];
    unsigned valid = 0;
    f32x4 sv[NS > 0 ? NS : 1][NB], v[NB];
    // ---- A: addresses + loads
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int i = b0 + k, g = i >> 2, j = i & 3, t = g / MT, m = g % MT;
      const int pl = 8 * j + plb;
      const int oh = ROWTILE ? oh0 + wv + 4 * m : oh0 + 8 * m + 2 * wv + (pl >> 4);
      const int ow = ROWTILE ? ow0 + pl : ow0 + (pl & 15);
      const bool ok = cok[t] && (interior || (oh < a.OH && ow < a.OW));
      pix[k] = (psr * oh + (pij[t] >> 1)) * (psr * a.OW) + psr * ow + (pij[t] & 1);     // pij = 0 without PixelShuffle
      valid |= (ok ? 1u : 0u) << k;
    }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
      for (int k = 0; k < NB; ++k) {
        const unsigned off = ((valid >> k) & 1) ? (unsigned)(pix[k] * sld[sidx] + ch[((b0 + k) >> 2) / MT]) * 4u : E_OOB;
        sv[sidx][k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs[sidx], off, 0, 0));
      }
    // ---- B: accumulator tiles -> LDS -> one pixel's 4 channels per lane
#pragma unroll
    for (int k0 = 0; k0 < NB; k0 += 4) {
      const int g = (b0 + k0) >> 2, t = g / MT, m = g % MT;
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
        ls[i * 32 + l32] = acc[m][t][reg];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) v[k0 + j] = ls4[j * 64 + lane];
    }
    // ---- C: arithmetic + stores
#pragma unroll
    for (int k = 0; k < NB; ++k) {
      const int t = ((b0 + k) >> 2) / MT;
      f32x4 o = (v[k] + bq[t]) * alpha;
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx) o += scoef[sidx] * (sres[sidx] ? sv[sidx][k] : f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = o[e] > 0.f ? o[e] : o[e] * slope;
#pragma unroll
      for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] *= (sv[sidx][k][e] > 0.f ? 1.f : sms[sidx]);
      if constexpr (SIGNS) {
        const int i = b0 + k;                       // item i: bits 4 i .. 4 i + 3 of the lane's 128
        if (msigns) {
          const unsigned nib = sbits[i >> 3] >> (4 * (i & 7));
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] *= ((nib >> e) & 1u) ? 1.f : sg_slope;
        }
        if (wsigns) {
          unsigned nib = 0;
#pragma unroll
          for (int e = 0; e < 4; ++e) nib |= (o[e] > 0.f ? 1u : 0u) << e;
          sbits[i >> 3] |= nib << (4 * (i & 7));
        }
      }
      const unsigned off = ((valid >> k) & 1) ? (unsigned)(pix[k] * a.y_ldc + ch[t]) * 4u : E_OOB;
#ifdef SRK_NO_STORE
      asm volatile("" :: "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]), "v"(off));
#else
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), yrs, off, 0, SAUX);
#endif
    }
  }
  if constexpr (SIGNS) {
    if (wsigns) *(reinterpret_cast<sg_u32x4*>(a.signs) + ((long)tile * 4 + wv) * 64 + lane) = sbits;
  }
}

// Fused epilogue shared by the conv kernels.  acc[m][t][reg]: pixel i = (reg&3) + 8*(reg>>2) + 4*hl of M tile m
// (rows 2wv, 2wv+1 of the m-th 8-row group), channel = n0 + 32t + l32.
// SAUX = cache-policy bits of the 16-byte stores (0; the chain kernels write through to device scope, srk_chain.h: their host side
// admits only calls that take the 16-byte path).
template <int BN, int MT, bool ROWTILE = false, int PF = 8, int PF2 = (PF > 4 ? PF / 2 : PF), int SAUX = 0, bool SIGNS = false>
__device__ __forceinline__ void conv_epilogue(const srk_conv_args& a, f32x16 (&acc)[MT][BN / 32], float4* smem, int n, int oh0,
                                              int ow0, int n0, int wv, int lane, int lds_slot = -1, int tile = 0) {
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
    //
    // A lane handles NI = 4 * NG (pixel, 4-channel) items.  They are processed in batches of PF items in three phases:
    //   A  compute the item addresses and ISSUE every residual / mask load of the batch (no waits in between),
    //   B  transpose the batch's accumulator tiles through LDS into registers,
    //   C  finish the arithmetic and store.
    // vmcnt retires in issue order, so a load issued behind a store cannot be waited for without waiting for that store's
    // round trip to HBM as well: with loads and stores interleaved item by item (one dependent round trip per tensor and
    // item, 16 items) the epilogue of a 256-workgroup launch took longer than five K chunks of its main loop.
    float* ls = reinterpret_cast<float*>(smem) + (lds_slot < 0 ? wv : lds_slot) * 1024;   // private 4 KB per wave
    // Batch depth PF when at most one of r1 / r2 / mask is present (the usual case: dense-block convs carry a bias only,
    // data-gradient convs a mask only), PF2 with two of them, 4 with all three: the prefetched values stay within ~4 PF floats.
    const int n_aux = (has_r1 ? 1 : 0) + (has_r2 ? 1 : 0) + (has_m ? 1 : 0);
    if (n_aux == 0) conv_epilogue_vec<BN, MT, ROWTILE, PF, 0, SAUX, SIGNS>(a, acc, ls, n, oh0, ow0, n0, wv, lane, interior, tile);
    else if (n_aux == 1) conv_epilogue_vec<BN, MT, ROWTILE, PF, 1, SAUX, SIGNS>(a, acc, ls, n, oh0, ow0, n0, wv, lane, interior, tile);
    else if (n_aux == 2) conv_epilogue_vec<BN, MT, ROWTILE, PF2, 2, SAUX, SIGNS>(a, acc, ls, n, oh0, ow0, n0, wv, lane, interior, tile);
    else conv_epilogue_vec<BN, MT, ROWTILE, 4, 3, SAUX, SIGNS>(a, acc, ls, n, oh0, ow0, n0, wv, lane, interior, tile);
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

