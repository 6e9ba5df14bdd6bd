// srk_conv_h16.hip -- the fused 3x3 convolution with 16-BIT ACTIVATION STORAGE (fp16: wp_format 7, bf16: wp_format 8) on
// v_mfma_f32_32x32x16_{f16,bf16} for gfx950: BASELINE configs[4]'s reduced-precision path.
//
// x, y, r1, r2 and the mask are 16-bit NHWC views (ldc / coff in ELEMENTS), the bias stays fp32, accumulation is fp32, the
// master weights stay fp32 (srk_pack_weights packs them into 16-bit fragments, fmt 7 / 8, every step).  Halving the bytes per
// activation is what this mode is for: a dense-block conv at 8 x 128 x 128 x 192 moves 50 + 17 MB instead of 101 + 34, which
// puts it (barely) on the compute side of the bf16/fp16 matrix roof (430 FLOP/B against a ridge of ~315).
//
// Shape.  Workgroup = 4 waves = TH x 32 output pixels x 64 output channels, TH = 4 MT rows (MT = 4: 16 x 32, 76 KB of LDS;
// MT = 2: 8 x 32, 58 KB), two workgroups per CU (<= 256 registers).  Wave w owns the MT ADJACENT rows MT w .. MT w + MT - 1 for
// both 32-channel halves: 2 MT accumulator tiles.  A 32-pixel M tile is one image row, so an A fragment (32 pixels x 16
// channels) is 32 consecutive 16-byte slots of the halo image in LDS (conflict-free ds_read_b128) and a kernel tap is an
// address shift.  Per 16-channel chunk the (TH + 2) x 34 halo ([k-half][pixel][8 ch]: two planes) and the nine tap slices of the
// weights ([tap][k-half][64][8]) are DMA'd global -> LDS (buffer_load ... lds, 16 B per lane) by the MFMA waves themselves, one
// chunk ahead, two buffers, one barrier per chunk.  Within a chunk the loop runs over (column shift s, input row): an A fragment is
// read once and feeds up to three kernel rows x two channel halves = six MFMAs; 9 MT x 2 MFMAs per 3 (MT + 2) + 18 fragment reads.
// Epilogue: each accumulator row (32 px x 64 ch fp32) is transposed through the wave's private 8 KB of LDS so that a lane owns 8
// consecutive channels of one pixel: bias, alpha, two residuals, LeakyReLU, LeakyReLU' mask, then ONE 16-byte store (PixelShuffle
// folded in); the residual / mask tensors are read the same way, 16 bytes per lane, all loads of a batch ahead of its stores.
#include "srk_internal.h"
#include "srk_epilogue.h"
#include "srk_chain.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <utility>

typedef _Float16 h16_f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 h16_bf16x8 __attribute__((ext_vector_type(8)));
typedef float h16_f32x8 __attribute__((ext_vector_type(8)));
typedef unsigned h16_u32x4 __attribute__((ext_vector_type(4)));

// Cache-policy bits of the one-conv kernels' 16-byte output stores: 16 = sc1, write-through.  Outputs kept dirty in the L2 displace the
// weights and halo rows the other workgroups of the XCD are about to read: a dense block as five launches takes 160 us with plain
// stores, 153 with sc1 or nt (data gradients: 168 either way); the chain form with plain stores for the tile interior 164 instead of 135.
#ifndef H16_STORE_AUX
#define H16_STORE_AUX 16
#endif
// 16x16x32 chain epilogue: 1 = the two 64-byte halves of a pixel's 128-byte line meet in ONE store instruction (two DPP row shifts per
// dword), 0 (default) = two half-line store instructions.  Interleaved same-box A/B (tools/debug/build_h16_var.sh,
// profiles/r04_ab_h16_epilogue_full_vs_half_lines.txt): dominant kernel of the c4 step 127.5-128.7 us (1) vs 126.1-128.7 (0), HBM traffic equal
#ifndef H16_EPI_FULL_LINES
#define H16_EPI_FULL_LINES 0
#endif
// H16_EPI_SIGNS_PK: 1 (default) = the epilogue's sign bits from packed 16-bit integer min / max on the halves that are stored, 0 = round 3's
// convert back + compare per element.
#ifndef H16_EPI_SIGNS_PK
#define H16_EPI_SIGNS_PK 1
#endif
// H16_EPI_EXP (diagnostic variant builds only, results WRONG): 1 = the 16x16x32 epilogue's stores dropped, 2 = no sign-bit / mask / LeakyReLU work
// H16_EPI_MASK_ARITH: 1 = the LeakyReLU' factors of the sign-bit mask by bit arithmetic, 0 = compare + select per element
#ifndef H16_EPI_MASK_ARITH
#define H16_EPI_MASK_ARITH 0
#endif
#ifndef H16_EPI_EXP
#define H16_EPI_EXP 0
#endif
// H16_ML_EXP (diagnostic variant builds only, results WRONG): 1 = the chain kernel's loader waves fetch nothing behind the launch's first stage,
// 2 = the 16x16x32 main loop reads no fragments behind a conv's head (MFMAs on stale registers), 4 = no flag waits
#ifndef H16_ML_EXP
#define H16_ML_EXP 0
#endif

namespace {

#ifdef SRK_STAMP       // diagnostic build only (make stamp; tools/stamp_h16.py): phase stamps of wave 0 of every workgroup
__device__ unsigned long long* g_h16_stamps = nullptr;
#define H16_STAMP(k) do { if (threadIdx.x == 0 && g_h16_stamps) { g_h16_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); g_h16_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define H16_STAMP_L(k) do { if (threadIdx.x == 256 && g_h16_stamps) { g_h16_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); g_h16_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
// chain form: 8 stamps per conv (0: conv start, 1: main loop done, 2: epilogue issued; loader wave 4: 4 / 5 around the flag wait), 64 per workgroup
#define H16C_STAMP(T0, c, k) do { if (threadIdx.x == (T0) && g_h16_stamps) g_h16_stamps[(long)blockIdx.x * 64 + (c) * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define H16_STAMP(k) do { } while (0)
#define H16_STAMP_L(k) do { } while (0)
#define H16C_STAMP(T0, c, k) do { } while (0)
#endif

template <typename T> struct H16;
template <> struct H16<_Float16> {
  typedef h16_f16x8 v8;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};
template <> struct H16<__bf16> {
  typedef h16_bf16x8 v8;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ f32x4 mfma16(v8 a, v8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};

// one 1-KB LDS-DMA piece: 16 bytes per lane, LDS destination = dst + 16 * lane, global source = resource base + vo (per lane) + so.
// (A plain function on purpose: with the builtin called from inside the kernel TEMPLATE the host pass of hipcc 7.2 silently dropped
// the kernels' host stubs -- undefined __device_stub__ symbols at load time, no diagnostic.)
__device__ __forceinline__ void h16_dma(__amdgpu_buffer_rsrc_t rs, float4* dst, unsigned vo, unsigned so) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, 0);
}
// the same with device-scope coherence (sc1: the load is served from behind the XCD's own L2, so it sees what a workgroup on ANOTHER
// XCD has written through during this kernel -- the chain form's newest slice)
constexpr int H16_AUX_SC1 = 16;
__device__ __forceinline__ void h16_dma_dev(__amdgpu_buffer_rsrc_t rs, float4* dst, unsigned vo, unsigned so) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, vo, so, 0, H16_AUX_SC1);
}

constexpr int HW_TW = 32, HW_IW = HW_TW + 2;       // tile width / halo width
// order of the output channels inside a 32-group of the packed weights (see the weight packing below): position 16 j + 4 a + b holds
// channel 8 a + 4 j + b
__host__ __device__ constexpr int h16_chan_of_pos(int pos) { return 8 * ((pos >> 2) & 3) + 4 * (pos >> 4) + (pos & 3); }
constexpr unsigned H_OOB = 0x80000000u;

// A stage = 32 input channels.  LDS image of its halo: PIXEL-major, 64 bytes per halo pixel = four 16-byte slots, slot j of pixel p
// holds the 8-channel group g = j ^ ((p >> 2) & 3).  Pixel-major so that four consecutive DMA lanes fetch 64 contiguous bytes of one
// pixel (a k-group-major image makes every lane of a piece touch its own cache line: 64 lines per instruction instead of 16); the
// XOR so that the A-fragment read (32 consecutive pixels, one group per lane half) is conflict-free: over any 16 consecutive
// pixels (4 (p & 3) + slot) mod 16 takes every value once.  A fragment's byte address: 64 p + 16 ((2 kk + hl) ^ ((p >> 2) & 3)), i.e.
// the second k-step of the stage is the first one's address XOR 32.
template <int MT> struct HGeo {
  static constexpr int TH = 4 * MT, IH = TH + 2, NHP = IH * HW_IW;        // halo pixels (612 / 340)
  static constexpr int HPIECES = (4 * NHP + 63) / 64;                    // 1-KB DMA pieces (16 pixels each) of the halo (39 / 22)
  static constexpr int WPIECES = 36;                                      // [k-step][tap][k-half] x 64 couts x 16 B
  static constexpr int STAGE4 = (HPIECES + WPIECES) * 64;                 // 16-byte slots per stage
  static constexpr int WBASE = HPIECES * 64;
};

// ------------------------------------------------------------------------------------------------------------------ epilogue
// acc[m][t][reg]: output row MT wv + m of the tile, pixel i = (reg & 3) + 8 (reg >> 2) + 4 hl of that row, channel n0 + 32 t + l32.
// Item (m, j), j = 0..3: pixel pl = 8 j + (lane >> 3), channels n0 + 8 (lane & 7) .. + 7.  NS = how many of r1 / r2 / mask exist.
// SAUX = cache-policy bits of the stores (16 = sc1, write-through: see H16_STORE_AUX above and the chain form below).
template <typename T, int MT, int NS, bool OUTF32, int SAUX = 0>
__device__ __forceinline__ void h16_epilogue(const srk_conv_args& a, f32x16 (&acc)[MT][2], float* ls, int n, int oh0, int ow0, int n0, int wv, int lane, int tile) {
  typedef typename H16<T>::v8 v8;
  // sign bits (srk_conv_args.signs): item (m, j) of the lane = bits 8 (4 m + j) .. + 7 of its 128; tile = the workgroup's index in the launch
  h16_u32x4 sbits = {0u, 0u, 0u, 0u};
  const bool wsigns = !OUTF32 && (a.flags & SRK_CONV_WRITE_SIGNS) != 0, msigns = !OUTF32 && (a.flags & SRK_CONV_MASK_SIGNS) != 0;
  h16_u32x4* const sgp = reinterpret_cast<h16_u32x4*>(a.signs) + ((long)tile * 4 + wv) * 64 + lane;
  if (msigns) sbits = *sgp;
  constexpr int TB = OUTF32 ? 1 : (NS <= 1 ? MT : (NS == 2 ? (MT >= 2 ? MT / 2 : 1) : 1));    // M tiles per batch (loads ahead of stores)
  const int hl = lane >> 5, l32 = lane & 31;
  const int lch = h16_chan_of_pos(l32);
  const int c8 = lane & 7, plb = lane >> 3;
  const int co = n0 + 8 * c8;
  const int Cps_out = a.Cout >> 2;
  int ch = co, pij = 0;
  if (a.ps_out) { pij = co / Cps_out; ch = co - pij * Cps_out; }
  const bool cok = OUTF32 ? (co < a.Cout) : (co + 7 < a.Cout);
  float bq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) bq[e] = 0.f;
  if (a.bias) {
    if (OUTF32) {
#pragma unroll
      for (int e = 0; e < 8; ++e) if (co + e < a.Cout) bq[e] = a.bias[co + e];
    } else if (cok) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co), b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { bq[e] = b0[e]; bq[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) bq[e] *= a.alpha;          // (acc + bias) * alpha = fma(acc, alpha, alpha * bias)
  }
  const float* r1p = srk_sgpr_opaque(a.r1); const float* r2p = srk_sgpr_opaque(a.r2); const float* mkp = srk_sgpr_opaque(a.mask);
  const bool has_r1 = r1p != nullptr, has_r2 = r2p != nullptr;
  const int r1l = srk_sgpr_opaque(a.r1_ldc), r1c = srk_sgpr_opaque(a.r1_coff), r2l = srk_sgpr_opaque(a.r2_ldc), r2c = srk_sgpr_opaque(a.r2_coff);
  const int mkl = srk_sgpr_opaque(a.m_ldc), mkc = srk_sgpr_opaque(a.m_coff);
  const float alpha = srk_sgpr_opaque(a.alpha), beta1 = srk_sgpr_opaque(a.beta1), beta2 = srk_sgpr_opaque(a.beta2);
  const float slope = srk_sgpr_opaque(a.slope), mask_slope = srk_sgpr_opaque(a.mask_slope);
  const int psr = a.ps_out ? 2 : 1;
  const long img_px = (long)a.OH * a.OW * (a.ps_out ? 4 : 1);
  constexpr int YB = OUTF32 ? 4 : 2;                     // bytes per output element
  auto rsrc16 = [&](const float* p, int ldc, int coff) {    // a 16-bit tensor behind the ABI's float* fields
    const T* q = reinterpret_cast<const T*>(p) + (long)n * img_px * ldc + coff;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(q), 0, (unsigned)((img_px * ldc - coff) * 2), 0x00020000);
  };
  __amdgpu_buffer_rsrc_t yrs;
  if (OUTF32) yrs = __builtin_amdgcn_make_buffer_rsrc(a.y + (long)n * img_px * a.y_ldc + a.y_coff, 0, (unsigned)((img_px * a.y_ldc - a.y_coff) * 4), 0x00020000);
  else yrs = rsrc16(a.y, a.y_ldc, a.y_coff);
  __amdgpu_buffer_rsrc_t srs[NS > 0 ? NS : 1];
  int sld[NS > 0 ? NS : 1];
  float scoef[NS > 0 ? NS : 1], sms[NS > 0 ? NS : 1];
  bool sres[NS > 0 ? NS : 1];
  if constexpr (NS >= 1) {
    const bool m0 = !has_r1 && !has_r2;
    srs[0] = rsrc16(has_r1 ? r1p : (has_r2 ? r2p : mkp), has_r1 ? r1l : (has_r2 ? r2l : mkl), has_r1 ? r1c : (has_r2 ? r2c : mkc));
    sld[0] = has_r1 ? r1l : (has_r2 ? r2l : mkl);
    scoef[0] = has_r1 ? beta1 : (has_r2 ? beta2 : 0.f);
    sms[0] = m0 ? mask_slope : 1.f; sres[0] = !m0;
  }
  if constexpr (NS >= 2) {
    const bool is2 = has_r1 && has_r2;
    srs[1] = rsrc16(is2 ? r2p : mkp, is2 ? r2l : mkl, is2 ? r2c : mkc);
    sld[1] = is2 ? r2l : mkl;
    scoef[1] = is2 ? beta2 : 0.f;
    sms[1] = is2 ? 1.f : mask_slope; sres[1] = is2;
  }
  if constexpr (NS >= 3) {
    srs[2] = rsrc16(mkp, mkl, mkc);
    sld[2] = mkl; scoef[2] = 0.f; sms[2] = mask_slope; sres[2] = false;
  }
  const f32x4* ls4 = reinterpret_cast<const f32x4*>(ls);
#pragma unroll
  for (int m0 = 0; m0 < MT; m0 += TB) {
    int pix[TB][4];
    unsigned valid = 0;
    v8 sv[NS > 0 ? NS : 1][TB][4];
    // ---- A: addresses + every residual / mask load of the batch
#pragma unroll
    for (int mm = 0; mm < TB; ++mm)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int oh = oh0 + MT * wv + m0 + mm, ow = ow0 + 8 * j + plb;
        const bool ok = cok && oh < a.OH && ow < a.OW;
        pix[mm][j] = (psr * oh + (pij >> 1)) * (psr * a.OW) + psr * ow + (pij & 1);
        valid |= (ok ? 1u : 0u) << (4 * mm + j);
      }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
      for (int mm = 0; mm < TB; ++mm)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned off = ((valid >> (4 * mm + j)) & 1) ? (unsigned)(pix[mm][j] * sld[sidx] + ch) * 2u : H_OOB;
          sv[sidx][mm][j] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(srs[sidx], off, 0, 0));
        }
    // ---- B / C per tile: transpose through LDS, arithmetic, store
#pragma unroll
    for (int mm = 0; mm < TB; ++mm) {
      const int m = m0 + mm;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
          ls[i * 64 + 32 * t + lch] = acc[m][t][reg];           // (column l32 of half t is channel 32 t + lch: packed order)
        }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int pl = 8 * j + plb;
        const f32x4 v0 = ls4[pl * 16 + 2 * c8], v1 = ls4[pl * 16 + 2 * c8 + 1];
        // One wave per SIMD: every VALU instruction of the epilogue costs its full 4 cycles and nothing hides it (stamps: 3.2 us of
        // a 20-50 us launch).  Hence: bias folded into one fused multiply-add (bq holds alpha * bias), wave-uniform branches around
        // what a launch does not use (no LeakyReLU on the data-gradient convs, no residual arithmetic for a mask slot and vice
        // versa), LeakyReLU as max(o, slope * o) (0 <= slope <= 1, host-checked) instead of compare + select, whose SGPR mask
        // costs wait states.
        float o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = __builtin_fmaf(v0[e], alpha, bq[e]); o[4 + e] = __builtin_fmaf(v1[e], alpha, bq[4 + e]); }
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
          if (sres[sidx]) {
            const h16_f32x8 rv = __builtin_convertvector(sv[sidx][mm][j], h16_f32x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(scoef[sidx], rv[e], o[e]);
          }
        }
        if (slope != 1.f) {
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaxf(o[e], o[e] * slope);
        }
#pragma unroll
        for (int sidx = 0; sidx < NS; ++sidx) {
          if (!sres[sidx]) {
            const h16_f32x8 rv = __builtin_convertvector(sv[sidx][mm][j], h16_f32x8);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rv[e] > 0.f ? o[e] : o[e] * sms[sidx];
          }
        }
        if (msigns) {
          const unsigned byte = sbits[(4 * m + j) >> 2] >> (8 * ((4 * m + j) & 3));
          const float ms = srk_sgpr_opaque(a.mask_slope);
#pragma unroll
          for (int e = 0; e < 8; ++e) o[e] = ((byte >> e) & 1u) ? o[e] : o[e] * ms;
        }
        const bool ok = (valid >> (4 * mm + j)) & 1;
        if constexpr (OUTF32) {
          // the fp32-output form (final conv of the generator, Cout = image channels): one dword per channel that exists
          const unsigned base = (unsigned)(pix[mm][j] * a.y_ldc + ch) * 4u;
#pragma unroll
          for (int e = 0; e < 8; ++e)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o[e]), yrs, (ok && co + e < a.Cout) ? base + 4u * e : H_OOB, 0, 0);
        } else {
          h16_f32x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = o[e];
          const v8 hv = __builtin_convertvector(ov, v8);
          if (wsigns) {                                    // the sign of what is STORED (a value that rounds to zero counts as not positive)
            const h16_f32x8 back = __builtin_convertvector(hv, h16_f32x8);
            unsigned byte = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) byte |= (back[e] > 0.f ? 1u : 0u) << e;
            sbits[(4 * m + j) >> 2] |= (ok ? byte : 0u) << (8 * ((4 * m + j) & 3));         // (no bits for pixels outside the image)
          }
          const unsigned off = ok ? (unsigned)(pix[mm][j] * a.y_ldc + ch) * (unsigned)YB : H_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(h16_u32x4, hv), yrs, off, 0, SAUX);
        }
      }
    }
  }
  if (wsigns) *sgp = sbits;
}

// ---- epilogue of the 16x16x32 form of the chain kernel (conv3x3_h16_chain_kernel<T, true>): NO transposition.
// acc[m][ph][2 p + j][reg] = D[row 4 G + reg][column n16] of MFMA (p, j) on output row MT wv + m, pixels 16 ph .. 16 ph + 15, with the
// weights as the row operand in packed order: rows 4 G + reg of MFMAs j = 0, 1 are channels 32 p + 8 G + 4 j + reg, i.e. the lane
// (n16 = lane & 15, G = lane >> 4) holds the 8 CONSECUTIVE channels 32 p + 8 G .. + 7 of pixel 16 ph + n16: item (m, ph, p) = one
// 16-byte store, residual / mask loads likewise.  Sign bits: item (m, ph, p) = bits 8 (4 m + 2 ph + p) .. + 7 of the lane's 128
// (a layout of this form only: srk_conv3x3_seq_signs_tag tells the forms apart).
// The accumulators START at the bias (conv3x3_h16_chain_kernel stages it through the LDS), so the output is acc * alpha; and every scalar the
// epilogue needs was fetched when the conv began (h16_epi_pre): stamps showed 1.9-2.2 us of a 4.8 us epilogue going to the argument loads
// and the bias round trip, with the matrix pipes idle.
struct h16_epi_pre {            // (13 scalar registers across the conv.  All 27 the epilogue can need -- the residual / mask views too -- did not fit:
  const float* y;               //  the allocator spilled them, through vector registers, to SCRATCH: 43 MB of stores per dense block, PMC)
  void* signs;
  int y_ldc, y_coff, OH, OW, Cout, flags;
  float alpha, slope, mask_slope;
};
__device__ __forceinline__ h16_epi_pre h16_epi_fetch(const srk_conv_args& a) {
  h16_epi_pre e;
  e.y = a.y; e.signs = a.signs;
  e.y_ldc = a.y_ldc; e.y_coff = a.y_coff; e.OH = a.OH; e.OW = a.OW; e.Cout = a.Cout; e.flags = a.flags;
  e.alpha = a.alpha; e.slope = a.slope; e.mask_slope = a.mask_slope;
  // (pinned in scalar registers HERE: left alone, the compiler loads each field where it is first used -- in the epilogue)
  asm volatile("" : "+s"(e.y), "+s"(e.signs), "+s"(e.y_ldc), "+s"(e.y_coff), "+s"(e.OH), "+s"(e.OW));
  asm volatile("" : "+s"(e.Cout), "+s"(e.flags), "+s"(e.alpha), "+s"(e.slope), "+s"(e.mask_slope));
  return e;
}
template <typename T, int MT, int NS, int SAUX>
__device__ __forceinline__ void h16_epilogue16(const h16_epi_pre& a, const srk_conv_args& ax, f32x4 (&acc)[MT][2][4], int n, int oh0, int ow0, int wv, int lane,
                                               int tile, const h16_u32x4* lsig, int cstamp = 0) {
  typedef typename H16<T>::v8 v8;
  // (everything derived from the lane id is formed HERE, from a copy the compiler cannot see through: hoisted out of the conv loop these
  // values live across the stage loops, where every register is taken, and come back from scratch in the middle of the epilogue)
  asm volatile("" : "+v"(lane));
  h16_u32x4 sbits = {0u, 0u, 0u, 0u};
  const bool wsigns = !(H16_EPI_EXP & 2) && (a.flags & SRK_CONV_WRITE_SIGNS) != 0, msigns = !(H16_EPI_EXP & 2) && (a.flags & SRK_CONV_MASK_SIGNS) != 0;
  h16_u32x4* const sgp = reinterpret_cast<h16_u32x4*>(a.signs) + ((long)tile * 4 + wv) * 64 + lane;
  if (msigns) sbits = lsig[wv * 64 + lane];          // (a loader wave fetched the workgroup's 4 KB of bits during the conv: no global round trip here)
  constexpr int TB = NS <= 1 ? MT : (NS == 2 ? (MT >= 2 ? MT / 2 : 1) : 1);      // rows per batch (loads ahead of stores)
  const int n16 = lane & 15, G = lane >> 4;
  bool cok[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) cok[p] = 32 * p + 8 * G + 7 < a.Cout;
  // (the residual / mask views of the few convs that have any -- a block's last conv; data gradients without sign bits -- are read here)
  const bool has_r1 = NS > 0 && ax.r1 != nullptr, has_r2 = NS > 0 && ax.r2 != nullptr;
  const float alpha = a.alpha, beta1 = ax.beta1, beta2 = ax.beta2, slope = (H16_EPI_EXP & 2) ? 1.f : a.slope, mask_slope = a.mask_slope;
  const long img_px = (long)a.OH * a.OW;
  auto rsrc16 = [&](const float* p, int ldc, int coff) {
    const T* q = reinterpret_cast<const T*>(p) + (long)n * img_px * ldc + coff;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(q), 0, (unsigned)((img_px * ldc - coff) * 2), 0x00020000);
  };
  const __amdgpu_buffer_rsrc_t yrs = rsrc16(a.y, a.y_ldc, a.y_coff);
  __amdgpu_buffer_rsrc_t srs[NS > 0 ? NS : 1];
  int sld[NS > 0 ? NS : 1];
  float scoef[NS > 0 ? NS : 1], sms[NS > 0 ? NS : 1];
  bool sres[NS > 0 ? NS : 1];
  if constexpr (NS >= 1) {
    const bool m0 = !has_r1 && !has_r2;
    srs[0] = rsrc16(has_r1 ? ax.r1 : (has_r2 ? ax.r2 : ax.mask), has_r1 ? ax.r1_ldc : (has_r2 ? ax.r2_ldc : ax.m_ldc), has_r1 ? ax.r1_coff : (has_r2 ? ax.r2_coff : ax.m_coff));
    sld[0] = has_r1 ? ax.r1_ldc : (has_r2 ? ax.r2_ldc : ax.m_ldc);
    scoef[0] = has_r1 ? beta1 : (has_r2 ? beta2 : 0.f);
    sms[0] = m0 ? mask_slope : 1.f; sres[0] = !m0;
  }
  if constexpr (NS >= 2) {
    const bool is2 = has_r1 && has_r2;
    srs[1] = rsrc16(is2 ? ax.r2 : ax.mask, is2 ? ax.r2_ldc : ax.m_ldc, is2 ? ax.r2_coff : ax.m_coff);
    sld[1] = is2 ? ax.r2_ldc : ax.m_ldc;
    scoef[1] = is2 ? beta2 : 0.f;
    sms[1] = is2 ? 1.f : mask_slope; sres[1] = is2;
  }
  if constexpr (NS >= 3) {
    srs[2] = rsrc16(ax.mask, ax.m_ldc, ax.m_coff);
    sld[2] = ax.m_ldc; scoef[2] = 0.f; sms[2] = mask_slope; sres[2] = false;
  }
  H16C_STAMP(0, cstamp, 3);            // (stamped build: set-up of the epilogue done)
#pragma unroll
  for (int m0 = 0; m0 < MT; m0 += TB) {
    int pix[TB][2];
    unsigned valid = 0;
    v8 sv[NS > 0 ? NS : 1][TB][2][2];
#pragma unroll
    for (int mm = 0; mm < TB; ++mm)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        const int oh = oh0 + MT * wv + m0 + mm, ow = ow0 + 16 * ph + n16;
        pix[mm][ph] = oh * a.OW + ow;
        valid |= ((oh < a.OH && ow < a.OW) ? 1u : 0u) << (2 * mm + ph);
      }
#pragma unroll
    for (int sidx = 0; sidx < NS; ++sidx)
#pragma unroll
      for (int mm = 0; mm < TB; ++mm)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            const bool ok = ((valid >> (2 * mm + ph)) & 1) && cok[p];
            const unsigned off = ok ? (unsigned)(pix[mm][ph] * sld[sidx] + 32 * p + 8 * G) * 2u : H_OOB;
            sv[sidx][mm][ph][p] = __builtin_bit_cast(v8, __builtin_amdgcn_raw_buffer_load_b128(srs[sidx], off, 0, 0));
          }
#pragma unroll
    for (int mm = 0; mm < TB; ++mm) {
      const int m = m0 + mm;
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) {
        h16_u32x4 hw[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const int it = 4 * m + 2 * ph + p;                 // item of the lane: sign-bit byte
          float o[8];
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = acc[m][ph][2 * p][e]; o[4 + e] = acc[m][ph][2 * p + 1][e]; }
          if (alpha != 1.f) {          // (wave-uniform: only a block's last conv scales)
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] *= alpha;
          }
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            if (sres[sidx]) {
              const h16_f32x8 rv = __builtin_convertvector(sv[sidx][mm][ph][p], h16_f32x8);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaf(scoef[sidx], rv[e], o[e]);
            }
          }
          if (slope != 1.f) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = __builtin_fmaxf(o[e], o[e] * slope);
          }
#pragma unroll
          for (int sidx = 0; sidx < NS; ++sidx) {
            if (!sres[sidx]) {
              const h16_f32x8 rv = __builtin_convertvector(sv[sidx][mm][ph][p], h16_f32x8);
#pragma unroll
              for (int e = 0; e < 8; ++e) o[e] = rv[e] > 0.f ? o[e] : o[e] * sms[sidx];
            }
          }
          if (msigns) {
#if H16_EPI_MASK_ARITH
            // factor = bit ? 1 : mask_slope as a bit select under the sign-extended bit (v_bfe_i32 + v_bitop3 per element, the products in
            // packed pairs): no compare, no condition register, none of the wait states behind them
            const unsigned one = 0x3f800000u, sl = __builtin_bit_cast(unsigned, mask_slope);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
              const unsigned mk = (unsigned)((int)(sbits[it >> 2] << (31 - 8 * (it & 3) - e)) >> 31);
              o[e] *= __builtin_bit_cast(float, (mk & one) | (~mk & sl));
            }
#else
            const unsigned byte = sbits[it >> 2] >> (8 * (it & 3));
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ((byte >> e) & 1u) ? o[e] : o[e] * mask_slope;
#endif
          }
          const bool ok = ((valid >> (2 * mm + ph)) & 1) && cok[p];
          h16_f32x8 ov;
#pragma unroll
          for (int e = 0; e < 8; ++e) ov[e] = o[e];
          const v8 hv = __builtin_convertvector(ov, v8);
          hw[p] = __builtin_bit_cast(h16_u32x4, hv);
          if (wsigns) {
            // The sign of what is STORED (a value that rounds to zero counts as not positive), on the packed halves themselves: a
            // 16-bit float is > 0 exactly when its bits, read as a signed 16-bit integer, are (fp16 and bf16 alike; -0 = 0x8000 is
            // negative, NaNs do not occur), so max(min(h, 1), 0) leaves 1 / 0 in bit 0 of each half -- two packed integer instructions
            // per dword instead of convert back + compare + select per ELEMENT (the round-3 form: ~40 vector instructions per item,
            // with a wait state behind every compare; one wave per SIMD pays each of them in full).
            // (inline assembly: the compiler lowers the packed min / max of <2 x i16> to a compare + select per half.)
#if H16_EPI_SIGNS_PK
            unsigned g = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              unsigned b2;
              asm("v_pk_min_i16 %0, %1, %2" : "=v"(b2) : "v"(hw[p][k]), "v"(0x00010001u));
              asm("v_pk_max_i16 %0, %1, 0" : "=v"(b2) : "v"(b2));
              g |= b2 << (2 * k);                            // element 2 k at bit 2 k, element 2 k + 1 at bit 16 + 2 k
            }
            const unsigned byte = (g & 0x55u) | ((g >> 15) & 0xaau);
#else
            const h16_f32x8 back = __builtin_convertvector(hv, h16_f32x8);
            unsigned byte = 0;
#pragma unroll
            for (int e = 0; e < 8; ++e) byte |= (back[e] > 0.f ? 1u : 0u) << e;
#endif
            sbits[it >> 2] |= (ok ? byte : 0u) << (8 * (it & 3));
          }
        }
        // FULL-LINE STORES.  The lane (n16, G) holds channels 8 G .. of half p = 0 and 32 + 8 G .. of half p = 1 of pixel n16: stored as they
        // are, an instruction writes 64 of a pixel's 128 bytes.  Here the two halves of a pixel first meet in ONE instruction: lanes
        // n16 >= 8 hand their p = 0 half to lane n16 - 8's partner slot and take over the p = 1 half of pixel n16 - 8 (two DPP row shifts by 8
        // within the 16-lane row that shares G): store 1 then writes the whole 128-byte line of pixels 0 .. 7, store 2 of pixels 8 .. 15.
        // (Built on the suspicion that half-line write-through stores were behind 127.8 MB of writes per block for 88 MB of outputs; they
        // were not -- that was a 27-dword argument struct spilled to scratch by every lane, see h16_epi_pre -- the traffic is the same
        // either way.  Kept as a switch; LOG.md has the A/B.)
#if H16_EPI_FULL_LINES
        h16_u32x4 d1, d2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          // row_shr:8 (0x118): lane i reads lane i - 8 of its row; bank_mask 0xc: only lanes 8 .. 15 are written, lanes 0 .. 7 keep `old`
          d1[k] = (unsigned)__builtin_amdgcn_update_dpp((int)hw[0][k], (int)hw[1][k], 0x118, 0xf, 0xc, false);
          // row_shl:8 (0x108): lane i reads lane i + 8; bank_mask 0x3: only lanes 0 .. 7 are written
          d2[k] = (unsigned)__builtin_amdgcn_update_dpp((int)hw[1][k], (int)hw[0][k], 0x108, 0xf, 0x3, false);
        }
        {
          const int oh = oh0 + MT * wv + m, owa = ow0 + 16 * ph + (n16 & 7);
          const int cha = (n16 < 8 ? 0 : 32) + 8 * G;
          const bool cka = cha + 7 < a.Cout && oh < a.OH;
          const unsigned o1 = (cka && owa < a.OW) ? (unsigned)((oh * a.OW + owa) * a.y_ldc + cha) * 2u : H_OOB;
          const unsigned o2 = (cka && owa + 8 < a.OW) ? (unsigned)((oh * a.OW + owa + 8) * a.y_ldc + cha) * 2u : H_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(d1, yrs, o1, 0, SAUX);
          __builtin_amdgcn_raw_buffer_store_b128(d2, yrs, o2, 0, SAUX);
        }
#else
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const bool ok = ((valid >> (2 * mm + ph)) & 1) && cok[p];
          const unsigned off = (ok && !(H16_EPI_EXP & 1)) ? (unsigned)(pix[mm][ph] * a.y_ldc + 32 * p + 8 * G) * 2u : H_OOB;
          __builtin_amdgcn_raw_buffer_store_b128(hw[p], yrs, off, 0, SAUX);
        }
#endif
      }
    }
  }
  if (wsigns) *sgp = sbits;
}

// ------------------------------------------------------------------------------------------------------------------ kernel
constexpr int H16_NLOAD = 2;                            // loader waves
constexpr int H16_THREADS = 64 * (4 + H16_NLOAD);

template <typename T, int MODE, int MT, bool OUTF32>
__global__ __launch_bounds__(H16_THREADS, 2) void conv3x3_h16_kernel(const srk_conv_args a) {
  typedef typename H16<T>::v8 v8;
  typedef HGeo<MT> G;
  constexpr int SMEM4 = 2 * G::STAGE4 > 2048 ? 2 * G::STAGE4 : 2048;       // >= 4 x 8 KB of epilogue scratch
  __shared__ float4 smem[SMEM4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + HW_TW - 1) / HW_TW, tilesH = (a.OH + G::TH - 1) / G::TH;
  int bid = blockIdx.x;
  {
    const int Tn = gridDim.x;                        // XCD-contiguous tile ranges: neighbouring tiles (shared halo rows) meet in one L2
    if ((Tn & 7) == 0) bid = (bid & 7) * (Tn >> 3) + (bid >> 3);
  }
  const int tile = bid + (int)(gridDim.x * blockIdx.y);          // (index of the workgroup's sign bits)
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * G::TH, ow0 = tx * HW_TW, n0 = blockIdx.y * 64;
  const int CoutP = (a.Cout + 63) & ~63;
  const int nq = a.Cin >> 5;                          // stages of 32 input channels
  H16_STAMP(0);

  // ---- DMA addressing shared by the prologue (all six waves) and the loader waves
  const T* xbase = reinterpret_cast<const T*>(a.x);
  const int Cps_in = a.Cin >> 2;
  long img_elems = (long)a.H * a.W * a.x_ldc;
  if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
  const T* ximg = xbase + (long)n * img_elems;
  const unsigned xbytes = (unsigned)(img_elems * 2);
  const unsigned wbytes = (unsigned)((long)nq * 36 * CoutP * 16);
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ximg), 0, xbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);
  const unsigned wvo = (unsigned)((n0 + lane) * 16);
  auto halo_vo = [&](int i) -> unsigned {               // per-lane byte offset of halo piece i (16 pixels x four 8-channel groups)
    const int hp = i * 16 + (lane >> 2);
    const int g = (lane & 3) ^ ((hp >> 2) & 3);           // the 8-channel group this lane's slot holds
    const int hy = hp / HW_IW, hx = hp - hy * HW_IW;
    const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
    // (32-bit arithmetic: the host checks that one image of x stays below 2^31 bytes)
    unsigned off;
    if (MODE == SRK_IN_UNSHUFFLE) off = (unsigned)((2 * ih) * (2 * a.W) + 2 * iw) * (unsigned)a.x_ldc + (unsigned)(a.x_coff + 8 * g);
    else off = (unsigned)(ih * a.W + iw) * (unsigned)a.x_ldc + (unsigned)(a.x_coff + 8 * g);
    const bool ok = hp < G::NHP && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    return ok ? off * 2u : H_OOB;
  };
  // Stages run over the input channels LAST BLOCK FIRST: in a dense block (forward and data gradient alike) the highest channels
  // of the prefix are the slice the previous launch has just written -- by the same tiles on the same XCDs (the tile -> XCD map is
  // the same for every launch of a grid) --, so the first stage, which nothing can hide, comes out of L2 instead of HBM, and the
  // older slices stream in behind the main loop.  (The K order only changes the order of the fp32 sums.)
  auto halo_so = [&](int qs) -> unsigned {              // scalar byte offset of stage qs's 32 input channels
    const int q = nq - 1 - qs;
    if (MODE == SRK_IN_UNSHUFFLE) {
      const int c32 = 32 * q;
      const int ij = c32 / Cps_in, c = c32 - ij * Cps_in;
      return (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 2);
    }
    return (unsigned)(32 * q * 2);
  };
  // ---- prologue: stage 0 is issued by ALL six waves, piece wv + 6 j each (one wave streams ~25 GB/s of DMA at best, and until the
  // first stage has landed the MFMA waves have nothing else to do)
  {
    // (the weight pieces first: their addresses need no arithmetic, so the memory system is at work while the halo offsets are formed)
    constexpr int NW0 = G::WPIECES / 6, NH0 = (G::HPIECES + 5) / 6;
#pragma unroll
    for (int j = 0; j < NW0; ++j) {
      const int w = wv + 6 * j;
      h16_dma(wrs, smem + G::WBASE + w * 64, wvo, (unsigned)(((nq - 1) * 36 + w) * CoutP * 16));
    }
    const unsigned xso0 = halo_so(0);
#pragma unroll
    for (int j = 0; j < NH0; ++j) {
      const int i = wv + 6 * j;
      if (i < G::HPIECES) h16_dma(xrs, smem + i * 64, halo_vo(i), xso0);
    }
    H16_STAMP(6);
  }

  if (wv >= 4) {
    // ------------------------------------------------------------------------------------------------------ loader waves
    // The MFMA waves issue no vector-memory instruction in the main loop: a buffer_load ... lds costs the issuing wave 100-185
    // cycles beside MFMAs and LDS reads (stamps: 19 pieces per wave and stage inflated a 4608-cycle stage to 6200-7300), and with
    // one wave per SIMD nothing hides that.  Two waves of their own (sharing SIMDs 0 and 2 with MFMA waves, <= 256 registers each)
    // issue the 75 pieces of a stage -- piece lw + 2 j: halo pieces first (HBM), then the weight pieces (L2) -- right behind the
    // barrier that frees its buffer, wait for them and meet the MFMA waves at the next stage barrier.
    const int lw = wv - 4;
    constexpr int NXJ = (G::HPIECES + H16_NLOAD - 1) / H16_NLOAD, NWJ = G::WPIECES / H16_NLOAD;
    unsigned xvo[NXJ];
#pragma unroll
    for (int j = 0; j < NXJ; ++j) xvo[j] = halo_vo(lw + H16_NLOAD * j);
    auto stage = [&](int qs, int b) {
      const unsigned xso = halo_so(qs);
      float4* dst = smem + b * G::STAGE4;
#pragma unroll
      for (int j = 0; j < NXJ; ++j) {
        const int i = lw + H16_NLOAD * j;
        if (i < G::HPIECES) h16_dma(xrs, dst + i * 64, xvo[j], xso);
      }
#pragma unroll
      for (int j = 0; j < NWJ; ++j) {
        const int w = lw + H16_NLOAD * j;                                  // (k-step, tap, k-half): contiguous in the packed weights
        h16_dma(wrs, dst + G::WBASE + w * 64, wvo, (unsigned)(((nq - 1 - qs) * 36 + w) * CoutP * 16));
      }
    };
    H16_STAMP_L(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int q = 0; q < nq; ++q) {
      if (q + 1 < nq) stage(q + 1, (q + 1) & 1);        // its buffer held stage q - 1: every read of it returned before the barrier
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------------------- MFMA waves
  f32x16 acc[MT][2];
  // ---- fragment addresses (bytes from smem).  A: halo pixel p = (MT wv + ri) * 34 + l32 + s of the (ri, s) this step reads, k-step 0;
  // k-step 1 = the same XOR 32; the other buffer = + STAGE bytes.  B: [k-step][tap][k-half][64 couts] 16-byte slots behind the halo.
  constexpr int SPS = MT + 2, STEPS1 = 3 * SPS, STEPS = 2 * STEPS1;
  int aaddr[STEPS1];
#pragma unroll
  for (int i = 0; i < STEPS1; ++i) {
    const int s = i / SPS, ri = i % SPS;
    const int hp = (MT * wv + ri) * HW_IW + l32 + s;
    aaddr[i] = hp * 64 + ((hl ^ ((hp >> 2) & 1)) + 2 * ((hp >> 3) & 1)) * 16;
  }
  const int baddr = (G::WBASE + hl * 64 + l32) * 16;            // + ((kk * 18 + tap * 2) * 64 + 32 t) * 16

  // ---- main loop.  A stage is STEPS = 2 x 3 (MT + 2) steps L: k-step kk, column shift s, halo row ri; one A fragment feeds the kernel
  // rows r with output row m = ri - r in range, both channel halves: 2 / 4 / 6 MFMAs.  Fragments are read AHEAD, by hand:
  //   A(L + 4) at step L into a ring of eight register sets (four steps hold at least 12 MFMAs = 384 cycles);  the six B fragments of
  //   the next (k-step, shift) group during steps ri = 1..3 of the current one into the other of two sets.
  // The last FOUR steps of a stage (12 - 18 MFMAs) are issued BEHIND the stage barrier and the first ten reads of the next stage, so
  // that the matrix pipe has work while those reads are in flight (one MFMA wave per SIMD: nothing else would cover them).
  // Straight-line code (sched_barrier between the slots); two stages per loop iteration make the buffer and the ring phase
  // compile-time constants.
  constexpr int DEFER = 4, AHEAD = 4;
  constexpr int RINGP = STEPS & 7;                     // ring phase of the odd stage of a pair (36 steps: 4; 24 steps: 0)
  v8 Af[8], Bf[2][3][2];
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  const char* sm = reinterpret_cast<const char*>(smem);
  auto rdA = [&](int P, int L) {         // (P, L are constants after unrolling)
    const int i = L % STEPS1, kk = L / STEPS1;
    return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + P * (G::STAGE4 * 16) + (kk ? (aaddr[i] ^ 32) : aaddr[i])));
  };
  auto rdB = [&](int P, int S, int r, int t) {      // S = 3 kk + s: the (k-step, shift) group
    const int kk = S / 3, s = S % 3;
    return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + P * (G::STAGE4 * 16) + baddr + ((kk * 18 + (3 * r + s) * 2) * 64 + 32 * t) * 16));
  };
  auto mfma_step = [&](auto pc, auto lc) {
    constexpr int P = decltype(pc)::value, L = decltype(lc)::value;
    constexpr int S = L / SPS, ri = L % SPS;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int m = ri - r;
      if (m >= 0 && m < MT) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = H16<T>::mfma(Af[(RINGP * P + L) & 7], Bf[S & 1][r][t], acc[m][t]);
      }
    }
  };
  auto head = [&](auto pc) {          // first fragments of the stage in buffer P: the six B fragments of group 0, A(0) .. A(AHEAD - 1)
    constexpr int P = decltype(pc)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) Bf[0][r][t] = rdB(P, 0, r, t);
#pragma unroll
    for (int L = 0; L < AHEAD; ++L) Af[(RINGP * P + L) & 7] = rdA(P, L);
  };
  auto stage_fn = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    using Pc = std::integral_constant<int, P>; using Pn = std::integral_constant<int, P ^ 1>;
    auto step = [&](auto lc) {
      constexpr int L = decltype(lc)::value;
      constexpr int S = L / SPS, ri = L % SPS;
      if constexpr (L + AHEAD < STEPS) Af[(RINGP * P + L + AHEAD) & 7] = rdA(P, L + AHEAD);
      if constexpr (ri >= 1 && ri <= 3 && S < 5) {
#pragma unroll
        for (int t = 0; t < 2; ++t) Bf[(S + 1) & 1][ri - 1][t] = rdB(P, S + 1, ri - 1, t);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(Pc{}, lc);
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step(std::integral_constant<int, Ls>{}), ...); }(std::make_integer_sequence<int, STEPS - DEFER>{});
    // every read of this stage's buffer has returned (behind the barrier the loaders overwrite it with stage q + 2); behind the
    // barrier the loaders' pieces of stage q + 1 have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    head(Pn{});          // (behind the LAST stage: reads of a buffer nobody fills any more, into registers nobody uses -- no branch)
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (mfma_step(Pc{}, std::integral_constant<int, STEPS - DEFER + Ls>{}), ...); }(std::make_integer_sequence<int, DEFER>{});
    __builtin_amdgcn_sched_barrier(0);
  };
  static_assert((STEPS & 3) == 0 && ((2 * STEPS) & 7) == 0, "the A ring must close over a pair of stages");

#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // my share of stage 0
  __builtin_amdgcn_s_barrier();          // stage 0 is in LDS
  H16_STAMP(2);
  head(I0{});
  {
    int q = 0;
    for (; q + 1 < nq; q += 2) {
      stage_fn(I0{});
      stage_fn(I1{});
    }
    if (q < nq) stage_fn(I0{});
  }
  // (no barrier in front of the epilogue: behind the last stage barrier no wave reads staged data any more, the loaders have
  // nothing in flight, and every wave's transposition scratch is its own)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  H16_STAMP(3);
  float* ls = reinterpret_cast<float*>(smem) + wv * 2048;
  const int n_aux = (a.r1 ? 1 : 0) + (a.r2 ? 1 : 0) + (a.mask ? 1 : 0);
  if constexpr (OUTF32) {
    h16_epilogue<T, MT, 0, true>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
  } else {
    if (n_aux == 0) h16_epilogue<T, MT, 0, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
    else if (n_aux == 1) h16_epilogue<T, MT, 1, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
    else if (n_aux == 2) h16_epilogue<T, MT, 2, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
    else h16_epilogue<T, MT, 3, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
  }
  H16_STAMP(4);
#ifdef SRK_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  H16_STAMP(5);
#endif
}

// ------------------------------------------------------------------------------------------ the shared-CU form ("h16s")
// The kernel above owns its CU (152 KB of LDS): the ~12 us of every launch that are memory bursts and the kernel boundary -- first
// stage arriving on all CUs at once, 16.8 MB of stores at the end -- overlap with nothing.  This form is built so that TWO
// workgroups share a CU (58 KB of LDS, <= 256 registers, 4 waves): 8 x 32-pixel tiles, 16-channel stages (pixel-major halo image
// of 32 bytes per pixel, slot j of pixel p holds the 8-channel group j ^ ((p >> 3) & 1): conflict-free ds_read_b128 over any 16
// consecutive pixels), the MFMA waves issue their own DMA pieces (what that costs them the co-resident workgroup fills with its
// MFMAs).  It pays when the two workgroups of a CU are at DIFFERENT points of their launches: the engine's two-chain mode runs
// the batch halves as two dependency chains on two streams, so that one chain's bursts sit beside the other chain's main loop.
template <typename T, int MODE, bool OUTF32>
__global__ __launch_bounds__(256, 2) void conv3x3_h16s_kernel(const srk_conv_args a) {
  typedef typename H16<T>::v8 v8;
  constexpr int MT = 2, TH = 4 * MT, IH = TH + 2, NHP = IH * HW_IW;          // 340 halo pixels
  constexpr int HPIECES = (2 * NHP + 63) / 64, WPIECES = 18;                  // 11 + 18 one-KB pieces per 16-channel stage
  constexpr int STAGE4 = (HPIECES + WPIECES + 1) * 64, WBASE = HPIECES * 64, DUMMY = (HPIECES + WPIECES) * 64;
  constexpr int NJH = (HPIECES + 3) / 4, NJW = (WPIECES + 3) / 4, NPIECE = NJH + NJW;
  constexpr int SMEM4 = 2 * STAGE4 > 2048 ? 2 * STAGE4 : 2048;
  __shared__ float4 smem[SMEM4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + HW_TW - 1) / HW_TW, tilesH = (a.OH + TH - 1) / TH;
  int bid = blockIdx.x;
  {
    const int Tn = gridDim.x;
    if ((Tn & 7) == 0) bid = (bid & 7) * (Tn >> 3) + (bid >> 3);
  }
  const int tile = bid + (int)(gridDim.x * blockIdx.y);          // (index of the workgroup's sign bits)
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * TH, ow0 = tx * HW_TW, n0 = blockIdx.y * 64;
  const int CoutP = (a.Cout + 63) & ~63;
  const int nq = a.Cin >> 4;                          // stages of 16 input channels

  const T* xbase = reinterpret_cast<const T*>(a.x);
  const int Cps_in = a.Cin >> 2;
  long img_elems = (long)a.H * a.W * a.x_ldc;
  if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
  const T* ximg = xbase + (long)n * img_elems;
  const unsigned xbytes = (unsigned)(img_elems * 2);
  const unsigned wbytes = (unsigned)((long)nq * 18 * CoutP * 16);
  unsigned xvo[NJH];
#pragma unroll
  for (int j = 0; j < NJH; ++j) {
    const int hp = (wv + 4 * j) * 32 + (lane >> 1);                       // halo pixel of this lane in piece wv + 4 j
    const int g = (lane & 1) ^ ((hp >> 3) & 1);                           // the 8-channel group its slot holds
    const int hy = hp / HW_IW, hx = hp - hy * HW_IW;
    const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
    unsigned off;
    if (MODE == SRK_IN_UNSHUFFLE) off = (unsigned)((2 * ih) * (2 * a.W) + 2 * iw) * (unsigned)a.x_ldc + (unsigned)(a.x_coff + 8 * g);
    else off = (unsigned)(ih * a.W + iw) * (unsigned)a.x_ldc + (unsigned)(a.x_coff + 8 * g);
    const bool ok = hp < NHP && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
    xvo[j] = ok ? off * 2u : H_OOB;
  }
  const unsigned wvo = (unsigned)((n0 + lane) * 16);
  // piece j of stage qs (newest channels first, as above) into buffer b; a piece that does not exist goes through a zero-record
  // descriptor into the DUMMY slot (no branch in the main loop)
  auto piece = [&](int qs, auto bc, auto jc) {
    constexpr int b = decltype(bc)::value, j = decltype(jc)::value;
    float4* dst = smem + b * STAGE4;
    const int q = nq - 1 - qs;
    if constexpr (j < NJH) {
      unsigned xso = (unsigned)(16 * q * 2);
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c16 = 16 * q;
        const int ij = c16 / Cps_in, c = c16 - ij * Cps_in;
        xso = (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 2);
      }
      const int i = wv + 4 * j;
      const bool live = qs < nq && i < HPIECES;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ximg), 0, live ? xbytes : 0u, 0x00020000);
      h16_dma(rs, dst + (i < HPIECES ? i * 64 : DUMMY), xvo[j], live ? xso : 0u);
    } else {
      const int w = wv + 4 * (j - NJH);
      const bool live = qs < nq && w < WPIECES;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, live ? wbytes : 0u, 0x00020000);
      h16_dma(rs, dst + (w < WPIECES ? WBASE + w * 64 : DUMMY), wvo, live ? (unsigned)((q * 18 + w) * CoutP * 16) : 0u);
    }
  };

  f32x16 acc[MT][2];
  constexpr int SPS = MT + 2, STEPS = 3 * SPS;                    // 12 steps per stage: column shift s, halo row ri
  int aaddr[STEPS];
#pragma unroll
  for (int i = 0; i < STEPS; ++i) {
    const int s = i / SPS, ri = i % SPS;
    const int hp = (MT * wv + ri) * HW_IW + l32 + s;
    aaddr[i] = hp * 32 + (hl ^ ((hp >> 3) & 1)) * 16;
  }
  const int baddr = (WBASE + hl * 64 + l32) * 16;                 // + ((tap * 2) * 64 + 32 t) * 16
  constexpr int DEFER = 4, AHEAD = 4;
  constexpr int RINGP = STEPS & 7;                                // 4: the ring closes over a pair of stages
  static_assert(NPIECE <= STEPS - DEFER && ((2 * STEPS) & 7) == 0, "one DMA piece per step; the A ring must close over a pair of stages");
  v8 Af[8], Bf[2][3][2];
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  const char* sm = reinterpret_cast<const char*>(smem);
  auto rdA = [&](int P, int L) { return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + P * (STAGE4 * 16) + aaddr[L])); };
  auto rdB = [&](int P, int s, int r, int t) {
    return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + P * (STAGE4 * 16) + baddr + (((3 * r + s) * 2) * 64 + 32 * t) * 16));
  };
  auto mfma_step = [&](auto pc, auto lc) {
    constexpr int P = decltype(pc)::value, L = decltype(lc)::value;
    constexpr int s = L / SPS, ri = L % SPS;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int m = ri - r;
      if (m >= 0 && m < MT) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = H16<T>::mfma(Af[(RINGP * P + L) & 7], Bf[(P + s) & 1][r][t], acc[m][t]);
      }
    }
  };
  auto head = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) Bf[P & 1][r][t] = rdB(P, 0, r, t);
#pragma unroll
    for (int L = 0; L < AHEAD; ++L) Af[(RINGP * P + L) & 7] = rdA(P, L);
  };
  auto stage_fn = [&](int qs, auto pc) {
    constexpr int P = decltype(pc)::value;
    using Pc = std::integral_constant<int, P>; using Pn = std::integral_constant<int, P ^ 1>;
    auto step = [&](auto lc) {
      constexpr int L = decltype(lc)::value;
      constexpr int s = L / SPS, ri = L % SPS;
      if constexpr (L + AHEAD < STEPS) Af[(RINGP * P + L + AHEAD) & 7] = rdA(P, L + AHEAD);
      if constexpr (ri >= 1 && ri <= 3 && s < 2) {
#pragma unroll
        for (int t = 0; t < 2; ++t) Bf[(P + s + 1) & 1][ri - 1][t] = rdB(P, s + 1, ri - 1, t);
      }
      if constexpr (L + 2 < NPIECE) piece(qs + 1, Pn{}, std::integral_constant<int, (L + 2 < NPIECE ? L + 2 : 0)>{});
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(Pc{}, lc);
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step(std::integral_constant<int, Ls>{}), ...); }(std::make_integer_sequence<int, STEPS - DEFER>{});
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    head(Pn{});
    piece(qs + 2, Pc{}, I0{});
    piece(qs + 2, Pc{}, I1{});
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (mfma_step(Pc{}, std::integral_constant<int, STEPS - DEFER + Ls>{}), ...); }(std::make_integer_sequence<int, DEFER>{});
    __builtin_amdgcn_sched_barrier(0);
  };

  // (weights first: their addresses need no arithmetic)
  [&]<int... Js>(std::integer_sequence<int, Js...>) { (piece(0, I0{}, std::integral_constant<int, NJH + Js>{}), ...); }(std::make_integer_sequence<int, NJW>{});
  [&]<int... Js>(std::integer_sequence<int, Js...>) { (piece(0, I0{}, std::integral_constant<int, Js>{}), ...); }(std::make_integer_sequence<int, NJH>{});
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  head(I0{});
  piece(1, I1{}, I0{});
  piece(1, I1{}, I1{});
  {
    int q = 0;
    for (; q + 1 < nq; q += 2) {
      stage_fn(q, I0{});
      stage_fn(q + 1, I1{});
    }
    if (q < nq) stage_fn(q, I0{});
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // trailing (empty) pieces: their zeros must have landed
  __builtin_amdgcn_s_barrier();                                      // before any wave reuses the staging buffers
  float* ls = reinterpret_cast<float*>(smem) + wv * 2048;
  const int n_aux = (a.r1 ? 1 : 0) + (a.r2 ? 1 : 0) + (a.mask ? 1 : 0);
  if constexpr (OUTF32) {
    h16_epilogue<T, MT, 0, true>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
  } else {
    if (n_aux == 0) h16_epilogue<T, MT, 0, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
    else if (n_aux == 1) h16_epilogue<T, MT, 1, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
    else if (n_aux == 2) h16_epilogue<T, MT, 2, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
    else h16_epilogue<T, MT, 3, false, H16_STORE_AUX>(a, acc, ls, n, oh0, ow0, n0, wv, lane, tile);
  }
}

// ------------------------------------------------------------------------------------------ the chain form
// A DenseResidualBlock is five convolutions over ONE dense buffer: conv k reads channel slices 0..k-1 and writes slice k (forward
// and data gradient alike).  As five launches, 13 us of each (first stage arriving on every CU at once, 17 MB of stores at the end,
// the kernel boundary) overlap with nothing: 39 % of the block's time.  This kernel runs the whole sequence in one launch, one
// workgroup per tile for all of it, as ONE continuous stream of stages (ascending channels: old slices first):
//   * conv k+1's first stage is loaded while conv k's last stage is computed and its epilogue runs -- its channels are OLD slices,
//     complete (halo included) since the tile's neighbours finished conv k-1;
//   * the only data a conv needs from its predecessor is its LAST 64 channels (two stages).  Before the loader waves fetch them they
//     wait until the (up to) eight neighbouring tiles have published that conv: flags[tile] = epoch + k + 1.  The four MFMA waves
//     count themselves in (LDS) once their stores are acknowledged -- a third of the way into the NEXT conv's first stage, where the
//     wait costs them nothing --, loader wave 0 then stores the flag (device scope);
//   * the outputs are stored write-through (sc1) and those two stages are fetched with device-scope loads (sc1): tiles on different
//     XCDs do not share an L2.  s_waitcnt vmcnt(0) behind write-through stores = they have reached memory (what the compiler's own
//     release sequence relies on); no buffer_wbl2 (measured: every workgroup writing the XCD's L2 back costs 30 us per block).
//     Every other load touches data that predates the launch, or that this XCD cannot hold a stale copy of (nobody reads a slice
//     before it is written, and its first reads are the sc1 ones); H16_CHAIN_ALL_DEV makes every halo load of convs >= 1 sc1.
// No workgroup ever waits for a tile that is not resident or on its way: the grid is at most one workgroup per CU (host-checked)
// and the library keeps at most one chain kernel in flight per device.  A wait that still runs into the time limit (30 s) sets
// *err and goes on -- the kernel always drains; the host turns that into an error on the next call.
// Tried and dropped (tools/debug/chain_check.py, 8 x 128 x 128, forward / data-gradient block; five launches: 158 / 168 us; this
// form: 138 / 143): plain stores + buffer_wbl2 before the flag (171 / 164); plain stores and loads where whole images fall to one
// XCD (147 / 148, and correctness would hang on the workgroup -> XCD map); the MFMA waves only DUMPING the tile (16-bit, into the
// free stage buffer) and four loader waves doing residuals / mask / stores beside the next conv's first stage (133 / 148: the
// next stage's DMA then has to wait for the dump to be read out, which costs what the shorter epilogue saves).
typedef srk_chain_args h16_chain_args;
#ifndef H16_CHAIN_ALL_DEV
#define H16_CHAIN_ALL_DEV 0
#endif
#ifndef H16_CHAIN_FRESH_DEV
#define H16_CHAIN_FRESH_DEV 1
#endif
#ifndef H16_CHAIN_SIG_STEP
#define H16_CHAIN_SIG_STEP 12
#endif

#ifndef CH_NLOAD_N
#define CH_NLOAD_N 2
#endif

constexpr int CH_NLOAD = CH_NLOAD_N, CH_THREADS = 64 * (4 + CH_NLOAD);      // loader waves of the chain kernel
// M16 (round 4, the default: H16_CHAIN_M16): the MFMA waves run v_mfma_f32_16x16x32 instead of 32x32x16 -- the same staging, the same LDS
// image, the same number of fragment reads (72 per stage and wave) and the same MFMA cycles (288 x 16), but
//   * under the power limit this chip holds a higher clock on the 16x16x32 shape (MI355X_MICROARCH.md, DVFS give-back item 7: 1.12-1.15 x
//     the FLOP/s at equal cycles, also with LDS-fed operands) -- and this loop runs at 1.4-1.5 GHz, far below 2.4;
//   * with the WEIGHTS as the row operand and the packed channel order of pack_h16_kernel a lane ends up with 8 consecutive output
//     channels of one pixel: the epilogue stores straight from the accumulators (h16_epilogue16), no transposition through the LDS
//     (128 ds_write_b32 + 32 ds_read_b128 per wave and conv, and the waits between them, were most of the 5.5 us epilogue).
template <typename T, bool M16>
__global__ __launch_bounds__(CH_THREADS, 2) void conv3x3_h16_chain_kernel(const h16_chain_args A) {
  constexpr int STORE_AUX = H16_AUX_SC1;
  typedef typename H16<T>::v8 v8;
  constexpr int MT = 4;
  typedef HGeo<MT> G;
  __shared__ float4 smem[2 * G::STAGE4 + 1 + 16 + 256];
  unsigned* const wg_cnt = reinterpret_cast<unsigned*>(smem + 2 * G::STAGE4);
  float* const lds_bias = reinterpret_cast<float*>(smem + 2 * G::STAGE4 + 1);      // (M16) the 64 biases of the conv about to start
  h16_u32x4* const lds_signs = reinterpret_cast<h16_u32x4*>(smem + 2 * G::STAGE4 + 1 + 16);     // (M16) sign bits of the conv in progress: [4 waves][64 lanes]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l32 = lane & 31;
  const int gH = A.c[0].H, gW = A.c[0].W;                 // one geometry for the whole chain (host-checked)
  const int tilesW = (gW + HW_TW - 1) / HW_TW, tilesH = (gH + G::TH - 1) / G::TH;
  int bid = blockIdx.x;
  {
    const int Tn = gridDim.x;
    if ((Tn & 7) == 0) bid = (bid & 7) * (Tn >> 3) + (bid >> 3);
  }
  const int tile = bid;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * G::TH, ow0 = tx * HW_TW;
  const int nconv = A.n;
  if (tid == 0) *wg_cnt = 0;
  srk_chain_skew(A);

  if (wv >= 4) {
    // ------------------------------------------------------------------------------------------------------ loader waves
    const int lw = wv - 4;
    constexpr int NXJ = (G::HPIECES + CH_NLOAD - 1) / CH_NLOAD, NWJ = G::WPIECES / CH_NLOAD;
    const unsigned wvo = (unsigned)(lane * 16);
    const srk_chain_watch watch = srk_chain_watch_of(A.flags, lane, n, ty, tx, tilesH, tilesW);     // lanes 0..8: the eight neighbouring tiles
    // drain: a wait of this launch has run into its bound (here or in another tile) -- no more waiting (srk_chain.h)
    bool drain = false;
    auto wait_flags = [&](unsigned target) { if (!(H16_ML_EXP & 4) && !drain) drain = !srk_chain_wait(watch, target, A, lane); };
    unsigned xvo[NXJ];
    __amdgpu_buffer_rsrc_t xrs, wrs;
    int CoutP = 64;
    auto setup = [&](const srk_conv_args& a) {
      const long img_elems = (long)a.H * a.W * a.x_ldc;
      const T* ximg = reinterpret_cast<const T*>(a.x) + (long)n * img_elems;
      CoutP = (a.Cout + 63) & ~63;
      xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(ximg), 0, (unsigned)(img_elems * 2), 0x00020000);
      wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, (unsigned)((long)(a.Cin >> 5) * 36 * CoutP * 16), 0x00020000);
#pragma unroll
      for (int j = 0; j < NXJ; ++j) {
        const int hp = (lw + CH_NLOAD * j) * 16 + (lane >> 2);
        // slot (lane & 3) of halo pixel hp holds the 8-channel group g.  32x32 form: the one-conv kernels' swizzle.  M16: the swizzle that is
        // conflict-free for ITS reads under the hardware's ds_read_b128 lane groups (see the MFMA waves' frag_addresses)
        const int g = (lane & 3) ^ (M16 ? 2 * ((hp >> 2) & 1) : ((hp >> 2) & 3));
        const int hy = hp / HW_IW, hx = hp - hy * HW_IW;
        const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
        const unsigned off = (unsigned)(ih * a.W + iw) * (unsigned)a.x_ldc + (unsigned)(a.x_coff + 8 * g);
        const bool ok = hp < G::NHP && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        xvo[j] = ok ? off * 2u : H_OOB;
      }
    };
    bool first_stage = true;
    auto stage = [&](int q, int b, bool dev) {           // stage q = input channels 32 q .. 32 q + 31 (ascending here)
      if ((H16_ML_EXP & 1) && !first_stage) return;
      first_stage = false;
      const unsigned xso = (unsigned)(64 * q);
      float4* dst = smem + b * G::STAGE4;
      if (dev) {
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
          const int i = lw + CH_NLOAD * j;
          if (i < G::HPIECES) h16_dma_dev(xrs, dst + i * 64, xvo[j], xso);
        }
      } else {
#pragma unroll
        for (int j = 0; j < NXJ; ++j) {
          const int i = lw + CH_NLOAD * j;
          if (i < G::HPIECES) h16_dma(xrs, dst + i * 64, xvo[j], xso);
        }
      }
#pragma unroll
      for (int j = 0; j < NWJ; ++j) {
        const int w = lw + CH_NLOAD * j;
        h16_dma(wrs, dst + G::WBASE + w * 64, wvo, (unsigned)((q * 36 + w) * CoutP * 16));
      }
    };
    // (M16) the bias of the conv about to start goes through the LDS into the accumulators' initial values: loader wave 0 fetches it in
    // front of the stage that is issued beside it and writes it behind that stage's own vmcnt(0) -- visible behind the barrier that follows
    float bias_v = 0.f;
    bool bias_pending = false, sig_pending_l = false;
    h16_u32x4 sig_v[4];
    auto bias_fetch = [&](const srk_conv_args& a) {
      if (M16 && lw == 0) {
        bias_v = (a.bias && lane < a.Cout) ? a.bias[lane] : 0.f;
        bias_pending = true;
      }
    };
    auto bias_put = [&]() {
      if (M16 && bias_pending) {
        lds_bias[lane] = bias_v;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bias_pending = false;
      }
    };
    setup(A.c[0]);
    bias_fetch(A.c[0]);
    stage(0, 0, false);
    for (int c = 0; c < nconv; ++c) {
      const int nq = A.c[c].Cin >> 5;                     // even, >= 4 behind the first conv (host-checked)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      bias_put();
      __builtin_amdgcn_s_barrier();                       // this conv's stage 0 is in LDS; the epilogue scratch (buffer 1) is free
      for (int q = 0; q < nq; ++q) {
        if (q + 1 < nq) {
          const bool fresh = H16_CHAIN_FRESH_DEV && c > 0 && q + 1 >= nq - 2;    // the slice the previous conv has written
          if (c > 0 && q + 1 == nq - 2) { H16C_STAMP(256, c, 4); wait_flags(A.epoch + (unsigned)c); H16C_STAMP(256, c, 5); }
          stage(q + 1, (q + 1) & 1, fresh || (H16_CHAIN_ALL_DEV && c > 0));
          if (M16 && q == 0 && lw == CH_NLOAD - 1 && (A.c[c].flags & SRK_CONV_MASK_SIGNS)) {
            // the LeakyReLU' sign bits this conv's epilogue masks with: 16 bytes per lane of the four MFMA waves
            const h16_u32x4* sg = reinterpret_cast<const h16_u32x4*>(A.c[c].signs) + (long)tile * 256 + lane;
#pragma unroll
            for (int w = 0; w < 4; ++w) sig_v[w] = sg[w * 64];
            sig_pending_l = true;
          }
          if (c > 0 && q == 0 && lw == 0) {
            // publish conv c - 1 of this tile: once the four MFMA waves have seen their stores acknowledged (they count themselves in
            // a third of the way into this stage), write the XCD's dirty lines back and release the flag -- from here, so that the
            // write-back stalls no MFMA wave
            while (__hip_atomic_load(wg_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4u * (unsigned)c) __builtin_amdgcn_s_sleep(2);
            if (lane == 0) __hip_atomic_store(A.flags + tile, A.epoch + (unsigned)c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            H16C_STAMP(256, c, 6);
          }
        } else if (c + 1 < nconv) {
          setup(A.c[c + 1]);                              // the next conv's first stage (old slices) beside this conv's last one
          bias_fetch(A.c[c + 1]);
          stage(0, 0, H16_CHAIN_ALL_DEV != 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bias_put();           // (the next conv's bias: the MFMA waves read it behind THIS barrier, in front of their accumulators' reset)
        if (M16 && sig_pending_l) {
#pragma unroll
          for (int w = 0; w < 4; ++w) lds_signs[w * 64 + lane] = sig_v[w];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          sig_pending_l = false;
        }
        __builtin_amdgcn_s_barrier();
      }
    }
    return;
  }

  // ---------------------------------------------------------------------------------------------------------- MFMA waves
  if constexpr (M16) {
    // ---- the 16x16x32 form.  Row operand W (weights): lane (i = lane & 15, kg = lane >> 4) reads position 16 j + i of 32-group p, k-group kg
    // (k-step kg >> 1, k-half kg & 1 of the staged image): slot = i mod 16.  Column operand X (activations): lane (pixel n16 = lane & 15, kg)
    // reads the 8-channel group kg of halo pixel hp = (row, 16 ph + n16 + s): slot = 4 (hp & 3) + position of the group within the pixel, mod 16.
    // BANK CONFLICTS.  A ds_read_b128 is served in four NON-contiguous groups of 16 lanes -- {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
    // the same + 32 (MI355X_MICROARCH.md, LDS) --, one cycle per group if its 16 slots differ mod 16.  The W read has that by itself (every
    // group holds each i once).  In the X read a group holds pixels {0-3, 12-15} of one k-group and {4-11} of the next, so the four lanes
    // on one value of hp & 3 carry (kg, quad (hp >> 2) & 3) = (k, q), (k + 1, q + 1), (k + 1, q + 2), (k, q + 3): with the one-conv kernels'
    // swizzle, position = kg ^ quad, two of them meet -- every X read took 8 cycles instead of 4 (PMC, first build of this form:
    // SQ_LDS_BANK_CONFLICT = 29 % of SQ_LDS_IDX_ACTIVE; now 0.  The kernel's time did not move: the LDS array is busy ~15 % of it).  position = kg ^ 2 (quad & 1) is conflict-free for
    // every alignment of hp (exhaustive check of all XOR swizzles by quad: tools/debug/h16_m16_banks.py); the loader waves stage the image
    // that way for this form.  ph = 1 is ph = 0 plus 1024 bytes (the swizzle repeats every 8 pixels).
    // A stage = 3 shifts s x 6 halo rows ri x 2 pixel halves ph = 36 steps; step (s, ri, ph) feeds the kernel rows r with output row
    // m = ri - r in range x the four (p, j) weight fragments: 4 / 8 / 12 MFMAs of 16 cycles.  X fragments: a ring of eight, read four
    // steps ahead.  W fragments: ONE set of 3 taps x 4 (48 registers, as many as the 32x32 form's two sets of six), reloaded as its
    // rows fall dead: tap row 0 is last used at ri = 3, row 1 at ri = 4, row 2 at ri = 5 -- the next shift's W[0] is read at step 8 of
    // the group, W[1] at step 10, W[2] at step 0 of the next group (first use at its step 4): 256-380 cycles ahead.  The last group's
    // steps 8-11 (which read the NEXT stage's buffer) sit behind the stage barrier, as the 32x32 form's last four steps do.
    constexpr int SPS = MT + 2, STEPS1 = 3 * SPS, STEPS = 2 * STEPS1, GRP = 2 * SPS;      // 6, 18, 36, 12
    f32x4 acc[MT][2][4];
    // X addresses.  Halo pixel of (s, ri): hp = hp0 + 34 ri + s, hp0 = 34 MT wv + n16; byte address 64 hp + 16 (kg ^ 2 ((hp >> 2) & 1)).  34 ri =
    // 32 ri + 2 ri, so the swizzle phase only depends on t = 2 ri + s (0 .. 12): ONE register per t holds 64 hp0 + 16 (kg ^ 2 (((hp0 + t) >> 2) & 1)),
    // the rest -- 64 (34 ri + s) and the pixel half -- is the instruction's offset field: 13 address registers instead of 18 (with 18 the
    // allocator spilled two of them INTO the stage loop).
    constexpr int NXA = 2 * (SPS - 1) + 2 + 1;                   // 13
    int xaddr[NXA];
    int waddr = 0;
    auto frag_addresses = [&]() {
      int lo = lane;
      asm volatile("" : "+v"(lo));
      const int n16 = lo & 15, kg = lo >> 4;
      const int hp0 = MT * wv * HW_IW + n16;
#pragma unroll
      for (int t = 0; t < NXA; ++t) xaddr[t] = hp0 * 64 + (kg ^ (2 * (((hp0 + t) >> 2) & 1))) * 16;
      waddr = (G::WBASE + ((kg >> 1) * 18 + (kg & 1)) * 64 + n16) * 16;        // + ((tap * 2) * 64 + 32 p + 16 j) * 16
    };
    constexpr int DEFER = 4, AHEAD = 4;
    constexpr int RING = 6;                                       // X fragments in flight: four ahead + the one in use (36 steps = 6 x 6: no phase)
    static_assert(GRP == 12 && STEPS % RING == 0 && AHEAD < RING - 1, "the schedule below is written for 16-row tiles");
    static_assert(HW_IW == 34, "the address split above uses 34 = 32 + 2");
    v8 Xf[RING], Wf[3][4];
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    const char* sm = reinterpret_cast<const char*>(smem);
    bool sig_pending = false;
    int pofs = 0;
    auto rdX = [&](int P, int L) {       // step L = (s * SPS + ri) * 2 + ph
      const int i = L >> 1, ph = L & 1;
      const int s = i / SPS, ri = i % SPS;
      return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + (P ? pofs : 0) + xaddr[2 * ri + s] + 64 * (HW_IW * ri + s) + 1024 * ph));
    };
    auto rdW = [&](int P, int s, int r, int f) {       // f = 2 p + j
      return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + (P ? pofs : 0) + waddr + (((3 * r + s) * 2) * 64 + 16 * f) * 16));
    };
    auto mfma_step = [&](auto pc, auto lc) {
      constexpr int P = decltype(pc)::value, L = decltype(lc)::value;
      constexpr int ri = (L >> 1) % SPS, ph = L & 1;
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int m = ri - r;
        if (m >= 0 && m < MT) {
#pragma unroll
          for (int f = 0; f < 4; ++f) acc[m][ph][f] = H16<T>::mfma16(Wf[r][f], Xf[L % RING], acc[m][ph][f]);
        }
      }
    };
    // W fragments due at step L of the stage in buffer P (see above); steps 32 and 34 read the OTHER buffer (the next stage, shift 0)
    auto w_loads = [&](auto pc, auto lc) {
      constexpr int P = decltype(pc)::value, L = decltype(lc)::value;
      constexpr int s = L / GRP, j = L % GRP;
      if constexpr (j == 0) {
#pragma unroll
        for (int f = 0; f < 4; ++f) Wf[2][f] = rdW(P, s, 2, f);
      }
      if constexpr (j == 8 || j == 10) {
        constexpr int r = j == 8 ? 0 : 1;
        constexpr int Pn = s == 2 ? (P ^ 1) : P, sn = s == 2 ? 0 : s + 1;
#pragma unroll
        for (int f = 0; f < 4; ++f) Wf[r][f] = rdW(Pn, sn, r, f);
      }
    };
    auto head = [&](auto pc) {            // a conv's first stage: W rows 0 and 1 of shift 0 (row 2 comes with step 0), X(0) .. X(AHEAD - 1)
      constexpr int P = decltype(pc)::value;
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int f = 0; f < 4; ++f) Wf[r][f] = rdW(P, 0, r, f);
#pragma unroll
      for (int L = 0; L < AHEAD; ++L) Xf[L % RING] = rdX(P, L);
    };
    auto publish = [&]() {
      if (sig_pending) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(wg_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        sig_pending = false;
      }
    };
    auto stage_fn = [&](auto pc) {
      constexpr int P = decltype(pc)::value;
      using Pc = std::integral_constant<int, P>;
      pofs = G::STAGE4 * 16;
      asm volatile("" : "+s"(pofs));
      auto step = [&](auto lc) {
        constexpr int L = decltype(lc)::value;
        if constexpr (P == 0 && L == H16_CHAIN_SIG_STEP) publish();
        if constexpr (!(H16_ML_EXP & 2)) {
          if constexpr (L + AHEAD < STEPS) Xf[(L + AHEAD) % RING] = rdX(P, L + AHEAD);
          else Xf[(L + AHEAD) % RING] = rdX(P ^ 1, L + AHEAD - STEPS);         // (behind the barrier: the next stage's first steps)
          w_loads(Pc{}, lc);
        }
        __builtin_amdgcn_sched_barrier(0);
        mfma_step(Pc{}, lc);
        __builtin_amdgcn_sched_barrier(0);
      };
      [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step(std::integral_constant<int, Ls>{}), ...); }(std::make_integer_sequence<int, STEPS - DEFER>{});
      // every read of this stage's buffer has returned (behind the barrier the loaders overwrite it); behind the barrier the next stage has landed
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step(std::integral_constant<int, STEPS - DEFER + Ls>{}), ...); }(std::make_integer_sequence<int, DEFER>{});
    };
    // the accumulators start at the bias (lds_bias: written by loader wave 0 one barrier earlier): lane (n16, Gl) owns channels 32 p + 8 Gl + 4 j + reg
    auto acc_init = [&]() {
      int lo = lane;
      asm volatile("" : "+v"(lo));
      const f32x4* bl = reinterpret_cast<const f32x4*>(lds_bias + 8 * (lo >> 4));
      f32x4 bv[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) bv[f] = bl[8 * (f >> 1) + (f & 1)];          // 32 p + 4 j floats on
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int ph = 0; ph < 2; ++ph)
#pragma unroll
          for (int f = 0; f < 4; ++f) acc[m][ph][f] = bv[f];
    };
    for (int c = 0; c < nconv; ++c) {
      const srk_conv_args& a = A.c[c];
      const int nq = a.Cin >> 5;
      const h16_epi_pre ep = h16_epi_fetch(a);
      frag_addresses();
      if (c > 0) acc_init();           // (conv 0: its bias becomes visible behind the barrier below)
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      H16C_STAMP(0, c, 0);
      if (c == 0) acc_init();
      head(I0{});
      for (int q = 0; q < nq; q += 2) {
        stage_fn(I0{});
        stage_fn(I1{});
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      H16C_STAMP(0, c, 1);
      const int n_aux = (a.r1 ? 1 : 0) + (a.r2 ? 1 : 0) + (a.mask ? 1 : 0);
      if (n_aux == 0) h16_epilogue16<T, MT, 0, STORE_AUX>(ep, a, acc, n, oh0, ow0, wv, lane, tile, lds_signs, c);
      else if (n_aux == 1) h16_epilogue16<T, MT, 1, STORE_AUX>(ep, a, acc, n, oh0, ow0, wv, lane, tile, lds_signs, c);
      else if (n_aux == 2) h16_epilogue16<T, MT, 2, STORE_AUX>(ep, a, acc, n, oh0, ow0, wv, lane, tile, lds_signs, c);
      else h16_epilogue16<T, MT, 3, STORE_AUX>(ep, a, acc, n, oh0, ow0, wv, lane, tile, lds_signs, c);
      H16C_STAMP(0, c, 2);
      sig_pending = c + 1 < nconv;
    }
    return;
  }
  f32x16 acc[MT][2];
  constexpr int SPS = MT + 2, STEPS1 = 3 * SPS, STEPS = 2 * STEPS1;
  int aaddr[STEPS1];
  int baddr = 0;
  // (the fragment addresses are formed again for every conv, from a lane id the compiler cannot see through: kept live across the
  // epilogue they are what it spills, and it reloads them from scratch in the middle of the main loop)
  auto frag_addresses = [&]() {
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int hlo = lo >> 5, l32o = lo & 31;
#pragma unroll
    for (int i = 0; i < STEPS1; ++i) {
      const int s = i / SPS, ri = i % SPS;
      const int hp = (MT * wv + ri) * HW_IW + l32o + s;
      aaddr[i] = hp * 64 + ((hlo ^ ((hp >> 2) & 1)) + 2 * ((hp >> 3) & 1)) * 16;
    }
    baddr = (G::WBASE + hlo * 64 + l32o) * 16;
  };
  constexpr int DEFER = 4, AHEAD = 4;
  constexpr int RINGP = STEPS & 7;
  v8 Af[8], Bf[2][3][2];
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  const char* sm = reinterpret_cast<const char*>(smem);
  bool sig_pending = false;
  // (buffer 1 lies beyond the 64 KB a ds_read offset field reaches: its byte offset is added per read from an SGPR the compiler cannot
  // see through -- left to itself it hoists all 36 sums out of the stage loop and spills them)
  int pofs = 0;
  auto rdA = [&](int P, int L) {
    const int i = L % STEPS1, kk = L / STEPS1;
    return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + (P ? pofs : 0) + (kk ? (aaddr[i] ^ 32) : aaddr[i])));
  };
  auto rdB = [&](int P, int S, int r, int t) {
    const int kk = S / 3, s = S % 3;
    return __builtin_bit_cast(v8, *reinterpret_cast<const float4*>(sm + (P ? pofs : 0) + baddr + ((kk * 18 + (3 * r + s) * 2) * 64 + 32 * t) * 16));
  };
  auto mfma_step = [&](auto pc, auto lc) {
    constexpr int P = decltype(pc)::value, L = decltype(lc)::value;
    constexpr int S = L / SPS, ri = L % SPS;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int m = ri - r;
      if (m >= 0 && m < MT) {
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[m][t] = H16<T>::mfma(Af[(RINGP * P + L) & 7], Bf[S & 1][r][t], acc[m][t]);
      }
    }
  };
  auto head = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int t = 0; t < 2; ++t) Bf[0][r][t] = rdB(P, 0, r, t);
#pragma unroll
    for (int L = 0; L < AHEAD; ++L) Af[(RINGP * P + L) & 7] = rdA(P, L);
  };
  // the previous conv of this tile: my stores are acknowledged (the MFMA waves issue no other vector-memory instruction); loader wave 0
  // counts the four waves and releases the flag
  auto publish = [&]() {
    if (sig_pending) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_fetch_add(wg_cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      sig_pending = false;
    }
  };
  auto stage_fn = [&](auto pc) {
    constexpr int P = decltype(pc)::value;
    using Pc = std::integral_constant<int, P>; using Pn = std::integral_constant<int, P ^ 1>;
    pofs = G::STAGE4 * 16;
    asm volatile("" : "+s"(pofs));
    auto step = [&](auto lc) {
      constexpr int L = decltype(lc)::value;
      constexpr int S = L / SPS, ri = L % SPS;
      if constexpr (P == 0 && L == H16_CHAIN_SIG_STEP) publish();
      if constexpr (L + AHEAD < STEPS) Af[(RINGP * P + L + AHEAD) & 7] = rdA(P, L + AHEAD);
      if constexpr (ri >= 1 && ri <= 3 && S < 5) {
#pragma unroll
        for (int t = 0; t < 2; ++t) Bf[(S + 1) & 1][ri - 1][t] = rdB(P, S + 1, ri - 1, t);
      }
      __builtin_amdgcn_sched_barrier(0);
      mfma_step(Pc{}, lc);
      __builtin_amdgcn_sched_barrier(0);
    };
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (step(std::integral_constant<int, Ls>{}), ...); }(std::make_integer_sequence<int, STEPS - DEFER>{});
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    head(Pn{});
    __builtin_amdgcn_sched_barrier(0);
    [&]<int... Ls>(std::integer_sequence<int, Ls...>) { (mfma_step(Pc{}, std::integral_constant<int, STEPS - DEFER + Ls>{}), ...); }(std::make_integer_sequence<int, DEFER>{});
    __builtin_amdgcn_sched_barrier(0);
  };

  float* ls = reinterpret_cast<float*>(smem + G::STAGE4) + wv * 2048;      // transposition scratch: buffer 1, which held the LAST stage
  for (int c = 0; c < nconv; ++c) {
    const srk_conv_args& a = A.c[c];
    const int nq = a.Cin >> 5;
    frag_addresses();
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (c > 0) my reads of the epilogue scratch
    __builtin_amdgcn_s_barrier();
    H16C_STAMP(0, c, 0);
    head(I0{});
    for (int q = 0; q < nq; q += 2) {
      stage_fn(I0{});
      stage_fn(I1{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    H16C_STAMP(0, c, 1);
    const int n_aux = (a.r1 ? 1 : 0) + (a.r2 ? 1 : 0) + (a.mask ? 1 : 0);
    if (n_aux == 0) h16_epilogue<T, MT, 0, false, STORE_AUX>(a, acc, ls, n, oh0, ow0, 0, wv, lane, tile);
    else if (n_aux == 1) h16_epilogue<T, MT, 1, false, STORE_AUX>(a, acc, ls, n, oh0, ow0, 0, wv, lane, tile);
    else if (n_aux == 2) h16_epilogue<T, MT, 2, false, STORE_AUX>(a, acc, ls, n, oh0, ow0, 0, wv, lane, tile);
    else h16_epilogue<T, MT, 3, false, STORE_AUX>(a, acc, ls, n, oh0, ow0, 0, wv, lane, tile);
    H16C_STAMP(0, c, 2);
    sig_pending = c + 1 < nconv;
  }
}

// ------------------------------------------------------------------------------------------ weight packing (formats 7 / 8)
// dst[q16][tap][h][Mp64][8] of T, k = 16 q + 8 h + e.  Work item = one (q, tap, h, m): 8 consecutive k, one 16-byte store.
// Position m of a 32-group holds output channel h16_chan_of_pos(m) (round 4): the chain kernel's 16x16x32 MFMAs take the WEIGHTS as
// their row operand, 16 consecutive positions per MFMA, and a lane then holds rows 4 G .. 4 G + 3 (G = lane / 16) of two MFMAs j = 0, 1:
// with position 16 j + 4 a + b <-> channel 8 a + 4 j + b those are the 8 CONSECUTIVE channels 8 G .. 8 G + 7 of one pixel -- a 16-byte
// store straight from the accumulators, no transposition through the LDS.  The 32x32x16 kernels read the same image; their lane l32 of
// half t simply owns channel 32 t + h16_chan_of_pos(l32) (one index in h16_epilogue).
template <typename T>
__global__ void pack_h16_kernel(const srk_pack_entry* __restrict__ tab, int n, long total) {
  typedef typename H16<T>::v8 v8;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  int lo_ = 0, hi_ = n - 1;
  while (lo_ < hi_) {
    const int mid = (lo_ + hi_ + 1) >> 1;
    if (tab[mid].elem_begin <= gid) lo_ = mid; else hi_ = mid - 1;
  }
  const srk_pack_entry e = tab[lo_];
  long t = gid - e.elem_begin;
  const int Mp = (e.M + 63) & ~63;
  const int mpos = (int)(t % Mp); t /= Mp;
  const int m = (mpos & ~31) | h16_chan_of_pos(mpos & 31);          // the output channel this position holds
  const int h = (int)(t & 1); t >>= 1;
  const int tap = (int)(t % 9); t /= 9;
  const int q = (e.k_off >> 4) + (int)t;
  const int Cps = e.src_cout >> 2;
  h16_f32x8 v;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = 16 * q + 8 * h + j;
    const int kr = k - e.k_off;
    float val = 0.f;
    if (m < e.M && kr >= 0 && kr < e.k_len) {
      if (!e.transpose) {
        int o = m;
        if (e.ps) o = 4 * (m % Cps) + m / Cps;
        val = e.src[((long)o * e.src_cin + e.c_begin + kr) * 9 + tap];
      } else {
        int o = kr;
        if (e.ps) o = 4 * (kr % Cps) + kr / Cps;
        val = e.src[((long)o * e.src_cin + e.c_begin + m) * 9 + (8 - tap)];
      }
      val *= e.scale;
    }
    v[j] = val;
  }
  const v8 hv = __builtin_convertvector(v, v8);
  float4* d = reinterpret_cast<float4*>(e.dst) + (((long)q * 9 + tap) * 2 + h) * Mp + mpos;
  *d = __builtin_bit_cast(float4, hv);
}

int g_h16_mt = -1;

#define H16_LAUNCH(T, MODE, MT)                                                                                        \
  do {                                                                                                                 \
    if (a.flags & SRK_CONV_OUT_F32) hipLaunchKernelGGL((conv3x3_h16_kernel<T, MODE, MT, true>), grid, dim3(H16_THREADS), 0, st, a); \
    else hipLaunchKernelGGL((conv3x3_h16_kernel<T, MODE, MT, false>), grid, dim3(H16_THREADS), 0, st, a);                      \
  } while (0)

#define H16S_LAUNCH(T, MODE)                                                                                           \
  do {                                                                                                                 \
    if (a.flags & SRK_CONV_OUT_F32) hipLaunchKernelGGL((conv3x3_h16s_kernel<T, MODE, true>), grid, dim3(256), 0, st, a); \
    else hipLaunchKernelGGL((conv3x3_h16s_kernel<T, MODE, false>), grid, dim3(256), 0, st, a);                         \
  } while (0)

int launch_h16_any(const srk_conv_args& a, int mt, hipStream_t st) {
  const bool un = a.in_mode == SRK_IN_UNSHUFFLE;
  if (mt == 1) {          // the shared-CU form: 8-row tiles, two workgroups per CU
    dim3 grid((unsigned)(a.N * srk_div_up(a.OH, 8) * srk_div_up(a.OW, HW_TW)), (unsigned)(srk_round_up(a.Cout, 64) / 64));
    if (a.wp_format == 7) { if (un) H16S_LAUNCH(_Float16, 1); else H16S_LAUNCH(_Float16, 0); }
    else { if (un) H16S_LAUNCH(__bf16, 1); else H16S_LAUNCH(__bf16, 0); }
    SRK_CHECK_LAUNCH();
    return SRK_OK;
  }
  const int tilesW = srk_div_up(a.OW, HW_TW), tilesH = srk_div_up(a.OH, 4 * mt);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)(srk_round_up(a.Cout, 64) / 64));
  if (a.wp_format == 7) {
    if (mt == 4) { if (un) H16_LAUNCH(_Float16, 1, 4); else H16_LAUNCH(_Float16, 0, 4); }
    else { if (un) H16_LAUNCH(_Float16, 1, 2); else H16_LAUNCH(_Float16, 0, 2); }
  } else {
    if (mt == 4) { if (un) H16_LAUNCH(__bf16, 1, 4); else H16_LAUNCH(__bf16, 0, 4); }
    else { if (un) H16_LAUNCH(__bf16, 1, 2); else H16_LAUNCH(__bf16, 0, 2); }
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
#undef H16_LAUNCH
#undef H16S_LAUNCH

}  // namespace

#ifdef SRK_STAMP
extern "C" int srk_debug_set_h16_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_h16_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -5;
}
#endif

extern "C" int srk_debug_set_h16_mt(int mt) { g_h16_mt = (mt == 1 || mt == 2 || mt == 4) ? mt : 0; return SRK_OK; }

// Form of the kernel: 16-row tiles (MT = 4: fewer halo rows, weight reads and barriers per MFMA; one workgroup per CU) when they
// fill the chip (>= 200 workgroups), else the shared-CU form (8-row tiles, two workgroups per CU; at the c4 trunk it runs a
// dense block in 165 us against 157-170 of MT = 4 and 198 of the one-workgroup-per-CU 8-row form MT = 2, which stays selectable).
// SRK_H16_MT = 1 | 2 | 4 / srk_debug_set_h16_mt force one (A/B, tests).
int srk_conv_h16_mt(const srk_conv_args& a) {
  if (g_h16_mt < 0) { const char* e = getenv("SRK_H16_MT"); g_h16_mt = e ? atoi(e) : 0; }
  if (g_h16_mt == 1 || g_h16_mt == 2 || g_h16_mt == 4) return g_h16_mt;
  const long wg16 = (long)a.N * srk_div_up(a.OH, 16) * srk_div_up(a.OW, HW_TW) * (srk_round_up(a.Cout, 64) / 64);
  return wg16 >= 200 ? 4 : 1;
}

int srk_conv_h16_check(const srk_conv_args& a) {
  if (!a.x || !a.y || !a.wp || a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.Cin <= 0 || a.Cout <= 0) return SRK_ERR_BAD_ARG;
  if (a.stride != 1 || (a.in_mode != SRK_IN_PLAIN && a.in_mode != SRK_IN_UNSHUFFLE) || (a.Cin % 32) || a.in_slope != 1.f) return SRK_ERR_UNSUPPORTED;
  if (!(a.slope >= 0.f && a.slope <= 1.f)) return SRK_ERR_UNSUPPORTED;       // LeakyReLU as max(t, slope * t)
  if (a.OH != a.H || a.OW != a.W) return SRK_ERR_BAD_ARG;
  if (a.in_mode == SRK_IN_UNSHUFFLE && ((a.Cin & 3) || ((a.Cin >> 2) % 32))) return SRK_ERR_UNSUPPORTED;
  if ((a.x_ldc % 8) || (a.x_coff % 8) || (((uintptr_t)a.x | (uintptr_t)a.wp) & 15)) return SRK_ERR_ALIGNMENT;
  const bool f32o = a.flags & SRK_CONV_OUT_F32;
  if (a.flags & (SRK_CONV_WRITE_SIGNS | SRK_CONV_MASK_SIGNS)) {        // sign bits (srk_conv_args.signs)
    if ((a.flags & SRK_CONV_WRITE_SIGNS) && (a.flags & SRK_CONV_MASK_SIGNS)) return SRK_ERR_UNSUPPORTED;
    if (f32o || a.ps_out || !a.signs || ((a.flags & SRK_CONV_MASK_SIGNS) && a.mask)) return SRK_ERR_UNSUPPORTED;
    if (((uintptr_t)a.signs) & 15) return SRK_ERR_ALIGNMENT;
  }
  if (f32o) {
    if (a.r1 || a.r2 || a.mask || a.ps_out) return SRK_ERR_UNSUPPORTED;
    if (((uintptr_t)a.y) & 3) return SRK_ERR_ALIGNMENT;
  } else {
    if (a.Cout % 8) return SRK_ERR_UNSUPPORTED;
    if (a.ps_out && ((a.Cout & 3) || ((a.Cout >> 2) % 8))) return SRK_ERR_UNSUPPORTED;
    if ((a.y_ldc % 8) || (a.y_coff % 8) || (((uintptr_t)a.y) & 15)) return SRK_ERR_ALIGNMENT;
    if (a.bias && (((uintptr_t)a.bias) & 15)) return SRK_ERR_ALIGNMENT;
    if (a.r1 && ((a.r1_ldc % 8) || (a.r1_coff % 8) || (((uintptr_t)a.r1) & 15))) return SRK_ERR_ALIGNMENT;
    if (a.r2 && ((a.r2_ldc % 8) || (a.r2_coff % 8) || (((uintptr_t)a.r2) & 15))) return SRK_ERR_ALIGNMENT;
    if (a.mask && ((a.m_ldc % 8) || (a.m_coff % 8) || (((uintptr_t)a.mask) & 15))) return SRK_ERR_ALIGNMENT;
  }
  // one image of every tensor is addressed through a 32-bit buffer resource
  if ((long)a.H * a.W * a.x_ldc * 2 * (a.in_mode == SRK_IN_UNSHUFFLE ? 4 : 1) > 0x7fffffffL) return SRK_ERR_UNSUPPORTED;
  {
    const long px = (long)a.OH * a.OW * (a.ps_out ? 4 : 1);
    long ld = (long)a.y_ldc * (f32o ? 2 : 1);
    if (a.r1 && a.r1_ldc > ld) ld = a.r1_ldc;
    if (a.r2 && a.r2_ldc > ld) ld = a.r2_ldc;
    if (a.mask && a.m_ldc > ld) ld = a.m_ldc;
    if (px * ld * 2 > 0x7fffffffL) return SRK_ERR_UNSUPPORTED;
  }
  return SRK_OK;
}

// ------------------------------------------------------------------------------------------ the chain form: host side
namespace {
int g_h16_chain = -1;          // 0: never, 1: where the 16-row form would run (default), 2: wherever the sequence is eligible (tests)
int h16_chain_mode() {
  if (g_h16_chain < 0) { const char* e = getenv("SRK_H16_CHAIN"); g_h16_chain = e ? atoi(e) : 1; }
  return g_h16_chain;
}
// Is args[0..n) a sequence the chain kernel runs?  (srk_chain_pattern_ok: the dense-block pattern; here: the 16-bit formats, one tile per CU at most)
bool h16_chain_eligible(const srk_conv_args* args, int n, int mode) {
  if (mode <= 0 || n < 2 || n > SRK_CHAIN_MAX) return false;
  const srk_conv_args& f = args[0];
  if (f.wp_format != 7 && f.wp_format != 8) return false;
  if (mode == 1 && srk_conv_h16_mt(f) != 4) return false;
  const long tiles = (long)f.N * srk_div_up(f.H, 16) * srk_div_up(f.W, HW_TW);
  const int cus = srk_chain_cus();
  if (cus <= 0 || tiles > cus || tiles > SRK_CHAIN_FLAGS) return false;
  for (int c = 0; c < n; ++c) {
    const srk_conv_args& a = args[c];
    if (a.wp_format != f.wp_format || (a.flags & SRK_CONV_OUT_F32) || srk_conv_h16_check(a) != SRK_OK) return false;
  }
  return srk_chain_pattern_ok(args, n, 2);
}
}  // namespace

extern "C" int srk_debug_set_h16_chain(int mode) { g_h16_chain = (mode >= 0 && mode <= 2) ? mode : 1; return SRK_OK; }
// MFMA shape of the chain kernel: 1 (default, SRK_H16_CHAIN_M16) = 16x16x32, weights as the row operand, epilogue without LDS; 0 = 32x32x16
static int g_h16_chain_m16 = -1;
static bool h16_chain_m16() {
  if (g_h16_chain_m16 < 0) { const char* e = getenv("SRK_H16_CHAIN_M16"); g_h16_chain_m16 = e ? (atoi(e) != 0) : 1; }
  return g_h16_chain_m16 != 0;
}
extern "C" int srk_debug_set_h16_chain_m16(int on) { g_h16_chain_m16 = on ? 1 : 0; return SRK_OK; }

// 1: the sequence goes out as ONE chain launch; 0: not eligible (the caller launches the convs one by one)
int srk_conv_h16_chain_would(const srk_conv_args* args, int n) {
  return (h16_chain_eligible(args, n, h16_chain_mode()) && !srk_chain_resting(false)) ? 1 : 0;
}

// 1: launched as one chain kernel, 0: not eligible (nothing launched), < 0: error
int srk_launch_conv_h16_chain(const srk_conv_args* args, int n, hipStream_t st) {
  if (!h16_chain_eligible(args, n, h16_chain_mode())) return 0;
  if (srk_chain_resting(true)) return 0;
  srk_chain_args A;
  const srk_conv_args& f = args[0];
  const dim3 grid((unsigned)(f.N * srk_div_up(f.H, 16) * srk_div_up(f.W, HW_TW)));
  const int rc = srk_chain_begin(st, n, (int)grid.x, 0, &A);
  if (rc != 1) return rc;
  for (int c = 0; c < n; ++c) A.c[c] = args[c];
  if (h16_chain_m16()) {
    if (f.wp_format == 7) hipLaunchKernelGGL((conv3x3_h16_chain_kernel<_Float16, true>), grid, dim3(CH_THREADS), 0, st, A);
    else hipLaunchKernelGGL((conv3x3_h16_chain_kernel<__bf16, true>), grid, dim3(CH_THREADS), 0, st, A);
  } else {
    if (f.wp_format == 7) hipLaunchKernelGGL((conv3x3_h16_chain_kernel<_Float16, false>), grid, dim3(CH_THREADS), 0, st, A);
    else hipLaunchKernelGGL((conv3x3_h16_chain_kernel<__bf16, false>), grid, dim3(CH_THREADS), 0, st, A);
  }
  const bool ok = hipGetLastError() == hipSuccess;
  const int rc2 = srk_chain_end(st, ok);
  return ok ? (rc2 ? rc2 : 1) : SRK_ERR_LAUNCH;
}

int srk_conv_h16_chain_name(const srk_conv_args* args, int n, char* buf, size_t len) {
  (void)n;
  snprintf(buf, len, "conv3x3_h16_chain_kernel<%s, %s>", args[0].wp_format == 7 ? "_Float16" : "__bf16", h16_chain_m16() ? "true" : "false");
  return SRK_OK;
}
// 1: the chain kernel runs its 16x16x32 form (whose sign-bit layout is its own: srk_conv3x3_seq_signs_tag)
int srk_conv_h16_chain_m16() { return h16_chain_m16() ? 1 : 0; }

// bytes of the sign-bit buffer of this launch: 16 bytes per lane of its (grid.x x grid.y) four-wave workgroups; 0 = not supported
size_t srk_conv_h16_signs_bytes(const srk_conv_args& a) {
  srk_conv_args b = a;
  b.flags &= ~(SRK_CONV_WRITE_SIGNS | SRK_CONV_MASK_SIGNS);
  if ((b.flags & SRK_CONV_OUT_F32) || b.ps_out || srk_conv_h16_check(b) != SRK_OK) return 0;
  const int mt = srk_conv_h16_mt(a);
  const size_t tiles = (size_t)a.N * srk_div_up(a.OH, mt == 1 ? 8 : 4 * mt) * srk_div_up(a.OW, HW_TW) * (srk_round_up(a.Cout, 64) / 64);
  return tiles * 4 * 64 * 16;
}

int srk_launch_conv_h16(const srk_conv_args& a, hipStream_t st) {
  const int rc = srk_conv_h16_check(a);
  if (rc) return rc;
  const int mt = srk_conv_h16_mt(a);
  return launch_h16_any(a, mt, st);
}

int srk_conv_h16_name(const srk_conv_args& a, char* buf, size_t len) {
  if (srk_conv_h16_mt(a) == 1)
    snprintf(buf, len, "conv3x3_h16s_kernel<%s, %d, %s>", a.wp_format == 7 ? "_Float16" : "__bf16", a.in_mode, (a.flags & SRK_CONV_OUT_F32) ? "true" : "false");
  else
    snprintf(buf, len, "conv3x3_h16_kernel<%s, %d, %d, %s>", a.wp_format == 7 ? "_Float16" : "__bf16", a.in_mode, srk_conv_h16_mt(a),
             (a.flags & SRK_CONV_OUT_F32) ? "true" : "false");
  return SRK_OK;
}

int srk_launch_pack_h16(const srk_pack_entry* dev, int n, int64_t total, int fmt, hipStream_t st) {
  if (fmt == 7) hipLaunchKernelGGL(pack_h16_kernel<_Float16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dev, n, (long)total);
  else hipLaunchKernelGGL(pack_h16_kernel<__bf16>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dev, n, (long)total);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
