// srk_conv.hip -- fused 3x3 / pad-1 convolution for gfx950 (MI355X), fp32 in / fp32 accumulate.
//
// Implicit GEMM on the exact-fp32 matrix cores (v_mfma_f32_32x32x2_f32):
//   M = output pixels, N = output channels, K = 9 * Cin.
// One workgroup (4 waves) owns an 8x16 tile of output pixels x BN output channels.  Per 8-channel
// K-chunk it stages the (8*S+2+..)x(16*S+..) input halo ONCE into LDS ([halo pixel][8 ch]) and the
// 9 tap weight slices ([tap][k-half][cout][4]); all nine taps then read the same halo through shifted
// LDS addresses, so HBM/L2 sees each input element once per workgroup instead of nine times.
// Each wave owns 2 output rows (32 pixels = one MFMA M-tile) x BN/32 accumulator tiles.  A/B fragments
// are ds_read_b128 (4 consecutive k per lane -> 4 MFMAs per read).  Double-buffered LDS, global loads
// for chunk q+1 are issued before the MFMAs of chunk q and written to LDS after them (one barrier per
// chunk).  The epilogue fuses bias, residual adds (x2), LeakyReLU, LeakyReLU-backward masking, the
// channel-slice store of the concat-free dense block and PixelShuffle(2).
//
// Four kernels share that structure and one epilogue (srk_epilogue.h); srk_conv3x3() picks by srk_conv_args.wp_format and geometry:
//   conv3x3_f32_kernel        direct, register-staged: stride 2, zero-upsample, small / unaligned channel counts
//   conv3x3_f32_lw_kernel     direct, a fifth wave stages global->LDS (stride 1, 64-channel tiles)
//   conv3x3_f32_wino_kernel   Winograd F(2,3) along W (wp_format 3): 2/3 of the MFMAs, 16x16 tiles, loader wave
//   conv3x3_f32_wino4_kernel  Winograd F(4,3) along W (wp_format 5): 1/2 of the MFMAs, 32x16 tiles, DMA staging
//
// Mirrors: nn.Conv2d/LeakyReLU/cat/mul+add/PixelShuffle of /root/reference/models.py:19-21,36-41,53,
// 63,67,86-90,97-99,126,142-145,168 (forward) and their autograd data-gradients.
#include "srk_internal.h"
#include <type_traits>
#include "srk_epilogue.h"
#include <stdio.h>
#include <stdlib.h>

#ifdef SRK_STAMP
// diagnostic build only: per-workgroup phase stamps (s_memrealtime, 100 MHz) into a side buffer
__device__ unsigned long long* g_srk_stamps = nullptr;
extern "C" int srk_debug_set_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_srk_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -5;
}
#define SRK_STAMP_AT(k) do { if (threadIdx.x == 0 && g_srk_stamps) g_srk_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SRK_CLOCK_AT(k) do { if (threadIdx.x == 0 && g_srk_stamps) g_srk_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memtime(); } while (0)
#define SRK_SEG_BEGIN() unsigned long long seg_t = __builtin_amdgcn_s_memtime(); unsigned long long seg_sum[6] = {0,0,0,0,0,0}
#define SRK_SEG_RESET() do { seg_t = __builtin_amdgcn_s_memtime(); } while (0)
#define SRK_SEG(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_sum[k] += t_ - seg_t; seg_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define SRK_SEG_END() do { if ((threadIdx.x == 0 || threadIdx.x == 256) && g_srk_stamps) for (int k_ = 0; k_ < 6; ++k_) g_srk_stamps[(threadIdx.x ? 4096 * 16 : 0) + (blockIdx.x + gridDim.x * blockIdx.y) * 16 + 8 + k_] = seg_sum[k_]; } while (0)
#else
#define SRK_SEG_BEGIN() do { } while (0)
#define SRK_SEG(k) do { } while (0)
#define SRK_SEG_RESET() do { } while (0)
#define SRK_SEG_END() do { } while (0)
#define SRK_CLOCK_AT(k) do { } while (0)
#define SRK_STAMP_AT(k) do { } while (0)
#endif

namespace {

template <int S, int MT>
struct Geo {
  static constexpr int TH = SRK_TH * MT;            // output rows per workgroup tile (8 or 16)
  static constexpr int IH = (TH - 1) * S + 3;
  static constexpr int IW = (SRK_TW - 1) * S + 3;
  static constexpr int NHP = IH * IW;               // halo pixels
  static constexpr int NX4 = 2 * NHP;               // float4 per chunk (8 ch per pixel)
  static constexpr int NXS = (NX4 + SRK_THREADS - 1) / SRK_THREADS;  // slots per thread
};

// one 16-byte-per-lane global -> LDS DMA piece (buffer_load_dwordx4 ... lds); lds_dst is wave-uniform
__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float4* lds_dst, unsigned voffset, unsigned soffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, soffset, 0, 0);
}

// (32-channel output tiles -- the discriminator layers -- are HBM-latency-bound with 2-4 K chunks of work per workgroup: ask for
// four waves per SIMD (<= 128 registers) instead of the two the epilogue's prefetch depth would otherwise cost)
#ifndef SRK_CONV_BN32_WAVES
#define SRK_CONV_BN32_WAVES 4
#endif
template <int BN, int S, int MODE, bool VEC, int MT, bool DMA>
__global__ __launch_bounds__(SRK_THREADS, ((BN == 32 && S == 1) ? SRK_CONV_BN32_WAVES : 1)) void conv3x3_f32_kernel(const srk_conv_args a) {
  using G = Geo<S, MT>;
  constexpr int NW4 = 18 * BN;                                  // weight float4 per chunk
  constexpr int NWS = (NW4 + SRK_THREADS - 1) / SRK_THREADS;
  constexpr int NXS = G::NXS;
  constexpr int NTN = BN / 32;                                  // accumulator tiles per wave

  __shared__ float4 smem[2 * (G::NX4 + NW4)];
  constexpr int BUF4 = G::NX4 + NW4;   // buffer b: xs at smem + b*BUF4, ws right after it

  SRK_STAMP_AT(0);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int hl = lane >> 5;     // k-half supplied by this lane
  const int l32 = lane & 31;

  const int tilesW = (a.OW + SRK_TW - 1) / SRK_TW;
  const int tilesH = (a.OH + G::TH - 1) / G::TH;
  int bid = blockIdx.x;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * G::TH, ow0 = tx * SRK_TW;
  const int n0 = blockIdx.y * BN;
  const int CoutP = (a.Cout + 31) & ~31;
  const int nq = (a.Cin + 7) >> 3;
  const int Cps_in = a.Cin >> 2;   // only used by MODE 1

  // ---- per-thread staging plan (pixel part is chunk-invariant)
  long xoff[NXS];
  bool xin[NXS];
  int xhalf[NXS];
  const int ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;
#pragma unroll
  for (int u = 0; u < NXS; ++u) {
    const int idx = tid + u * SRK_THREADS;
    const int hp = idx >> 1, half = idx & 1;
    const int hy = hp / G::IW, hx = hp - hy * G::IW;
    const int ih = ih0 + hy, iw = iw0 + hx;
    bool inb = idx < G::NX4;
    long off;
    if (MODE == SRK_IN_ZERO_UPSAMPLE) {
      inb = inb && ih >= 0 && iw >= 0 && !(ih & 1) && !(iw & 1) && (ih >> 1) < a.H && (iw >> 1) < a.W;
      off = ((long)(n * a.H + (ih >> 1)) * a.W + (iw >> 1)) * a.x_ldc + a.x_coff + 4 * half;
    } else if (MODE == SRK_IN_UNSHUFFLE) {
      inb = inb && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
      off = ((long)(n * 2 * a.H + 2 * ih) * (2 * a.W) + 2 * iw) * a.x_ldc + a.x_coff + 4 * half;
    } else {
      inb = inb && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
      off = ((long)(n * a.H + ih) * a.W + iw) * a.x_ldc + a.x_coff + 4 * half;
    }
    xoff[u] = off; xin[u] = inb; xhalf[u] = half;
  }
  // VEC path: branch-free staging through buffer descriptors (out-of-range lanes read 0).  The descriptor
  // covers this image only (so byte offsets fit 32 bits); the chunk's channel offset goes in soffset.
  constexpr unsigned OOB = 0x80000000u;
  unsigned xvo[NXS], wvo[NWS];
  long img_elems = (long)a.H * a.W * a.x_ldc;
  if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
  const float* ximg = a.x + (long)n * img_elems;
#pragma unroll
  for (int u = 0; u < NXS; ++u) xvo[u] = xin[u] ? (unsigned)((xoff[u] - (long)n * img_elems) * 4) : OOB;
#pragma unroll
  for (int v = 0; v < NWS; ++v) {
    const int idx = tid + v * SRK_THREADS;
    const int th = idx / BN, co = idx - th * BN;
    wvo[v] = (idx < NW4 && n0 + co < CoutP) ? (unsigned)((th * CoutP + n0 + co) * 16) : OOB;
  }
  const unsigned xbytes = (unsigned)(img_elems * 4 > 0x7fffffffL ? 0x7fffffffL : img_elems * 4);
  const unsigned wbytes = (unsigned)((long)nq * 18 * CoutP * 16);
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);

  float4 xr[NXS];
  float4 wr[NWS];

  constexpr int NSLOT = NXS + NWS;              // staging work items per chunk, one 16-byte load/store each
  // slot i < NXS: input halo float4 #i of this thread; else weight float4 #(i - NXS)
  auto load_slot = [&](int i, int q) {
    if (i < NXS) {
      const int u = i;
      long cadd = 8 * q;
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c8 = 8 * q;
        const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
        cadd = (long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c;
      }
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (VEC) {
        const f32x4 t4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xvo[u], (unsigned)(cadd * 4), 0));
        v = make_float4(t4[0], t4[1], t4[2], t4[3]);
      } else if (xin[u]) {
        const float* p = a.x + xoff[u] + cadd;
        const int c = 8 * q + 4 * xhalf[u];
        if (c + 0 < a.Cin) v.x = p[0];
        if (c + 1 < a.Cin) v.y = p[1];
        if (c + 2 < a.Cin) v.z = p[2];
        if (c + 3 < a.Cin) v.w = p[3];
      }
      xr[u] = v;
    } else {
      const int v = i - NXS;
      const f32x4 t4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo[v], (unsigned)(q * 18 * CoutP * 16), 0));
      wr[v] = make_float4(t4[0], t4[1], t4[2], t4[3]);
    }
  };
  auto store_slot = [&](int i, int b) {
    if (i < NXS) {
      const int u = i;
      const int idx = tid + u * SRK_THREADS;
      float4 v = xr[u];
      if (a.in_slope != 1.f) {
        v.x = v.x > 0.f ? v.x : v.x * a.in_slope; v.y = v.y > 0.f ? v.y : v.y * a.in_slope;
        v.z = v.z > 0.f ? v.z : v.z * a.in_slope; v.w = v.w > 0.f ? v.w : v.w * a.in_slope;
      }
      if (idx < G::NX4) smem[b * BUF4 + idx] = v;
    } else {
      const int v = i - NXS;
      const int idx = tid + v * SRK_THREADS;
      if (idx < NW4) smem[b * BUF4 + G::NX4 + idx] = wr[v];
    }
  };
  // DMA variant: the same slot goes global -> LDS directly (buffer_load ... lds: LDS address = wave-uniform base
  // + lane*16, which is exactly the lane-linear [idx] layout used here; out-of-range lanes deliver 0).
  auto dma_slot = [&](int i, int q, int b) {
    if (i < NXS) {
      const int u = i;
      long cadd = 8 * q;
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c8 = 8 * q;
        const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
        cadd = (long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c;
      }
      const int idx0 = wv * 64 + u * SRK_THREADS;                 // lane 0's float4 index
      if (idx0 + lane < G::NX4)
        dma16(xrsrc, smem + b * BUF4 + idx0, xvo[u], (unsigned)(cadd * 4));
    } else {
      const int v = i - NXS;
      const int idx0 = wv * 64 + v * SRK_THREADS;
      if (idx0 + lane < NW4)
        dma16(wrsrc, smem + b * BUF4 + G::NX4 + idx0, wvo[v], (unsigned)(q * 18 * CoutP * 16));
    }
  };
  auto load_chunk = [&](int q) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) load_slot(i, q);
  };
  auto store_chunk = [&](int b) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) store_slot(i, b);
  };

  f32x16 acc[MT][NTN];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

  // A-fragment base: this lane's output pixel inside the tile.  Wave wv owns rows 2wv, 2wv+1 of every
  // 8-row group m (pixel i of the 32x32 MFMA tile = row 2wv + (i>>4) + 8m, column i & 15).
  const int apy = 2 * wv + (l32 >> 4), apx = l32 & 15;
  const int abase = (apy * S) * G::IW + apx * S;
  constexpr int AM = 8 * S * G::IW;          // halo-pixel offset between the wave's two M tiles

  if constexpr (DMA) {
#pragma unroll
    for (int i = 0; i < NSLOT; ++i) dma_slot(i, 0, 0);
    SRK_STAMP_AT(1);
  } else {
    load_chunk(0);
    SRK_STAMP_AT(1);
    store_chunk(0);
  }
  __syncthreads();
  SRK_STAMP_AT(2);
  SRK_CLOCK_AT(5);

  // Fragment registers: fr[p] holds (A per M tile, B per N tile) of one tap; tap+1 is always in flight while
  // tap's MFMAs issue.  The last tap of a chunk is issued AFTER the chunk's barrier, behind the first fragment
  // reads of the next chunk, so neither the barrier nor the LDS latency behind it drains the matrix pipe.
  float4 av[2][MT], bv[2][NTN];
  auto ld_frag = [&](int p, int b, int tap) {
    const int r = tap / 3, s = tap - 3 * r;
    const float4* xb = smem + b * BUF4 + abase * 2 + hl;
    const float4* wb = smem + b * BUF4 + G::NX4 + hl * BN + l32;
#pragma unroll
    for (int m = 0; m < MT; ++m) av[p][m] = xb[(m * AM + r * G::IW + s) * 2];
#pragma unroll
    for (int t = 0; t < NTN; ++t) bv[p][t] = wb[tap * 2 * BN + t * 32];
  };
  // One k-step (4 accumulators x 1 MFMA each) of tap fragments fr[p]; k-step outermost so that consecutive
  // MFMAs hit different accumulators.
  auto mfma_kstep = [&](int p, int e) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int t = 0; t < NTN; ++t) {
        const float ae = e == 0 ? av[p][m].x : e == 1 ? av[p][m].y : e == 2 ? av[p][m].z : av[p][m].w;
        const float be = e == 0 ? bv[p][t].x : e == 1 ? bv[p][t].y : e == 2 ? bv[p][t].z : bv[p][t].w;
        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc[m][t], 0, 0, 0);
      }
  };

  // Staging is spread over the MFMA stream, ONE 16-byte global load (taps 0..) or LDS store (taps 4..) per
  // k-step: issued as a burst, the 4 waves' 1-KB requests queue in the CU's memory pipe (~16 clk each) and
  // stall the in-order waves for ~1000 clk per chunk; issued one at a time they hide behind the MFMAs.
  constexpr int ST0 = 4;                          // first tap that carries LDS stores
  static_assert(NSLOT <= 16, "staging slots must fit taps 0..3 / 4..7");
  ld_frag(0, 0, 0);
  SRK_SEG_BEGIN();
  for (int q = 0; q < nq; ++q) {
    const int b = q & 1;
    const bool more = q + 1 < nq;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      ld_frag((tap + 1) & 1, b, tap + 1);
      __builtin_amdgcn_sched_barrier(0);         // keep the next tap's ds_reads AHEAD of this tap's MFMAs
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        mfma_kstep(tap & 1, e);
        const int li = tap * 4 + e, si = (tap - ST0) * 4 + e;
        if constexpr (DMA) {
          if (more && li < NSLOT) { __builtin_amdgcn_sched_barrier(0); dma_slot(li, q + 1, b ^ 1); __builtin_amdgcn_sched_barrier(0); }
        } else {
          if (more && tap < ST0 && li < NSLOT) { __builtin_amdgcn_sched_barrier(0); load_slot(li, q + 1); __builtin_amdgcn_sched_barrier(0); }
          if (more && tap >= ST0 && si < NSLOT) { __builtin_amdgcn_sched_barrier(0); store_slot(si, b ^ 1); __builtin_amdgcn_sched_barrier(0); }
        }
      }
    }
    SRK_SEG(3);
    __syncthreads();                             // buffer b fully consumed (tap 8 is in registers), b^1 visible
    SRK_SEG(4);
    if (more) ld_frag(1, b ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int e = 0; e < 4; ++e) mfma_kstep(0, e);  // tap 8
    if (more) {
#pragma unroll
      for (int m = 0; m < MT; ++m) av[0][m] = av[1][m];
#pragma unroll
      for (int t = 0; t < NTN; ++t) bv[0][t] = bv[1][t];
    }
    SRK_SEG(5);
  }

  SRK_SEG_END();
  SRK_CLOCK_AT(6);
  SRK_STAMP_AT(3);
  conv_epilogue<BN, MT>(a, acc, smem, n, oh0, ow0, n0, wv, lane);
  SRK_STAMP_AT(4);
}

// ---------------------------------------------------------------------------------------------------------
// Loader-wave variant (stride 1, 16x16 tile, 16-byte-addressable input): 5 waves per workgroup.  Waves 0-3 run
// ONLY the MFMA stream (fragment reads + MFMAs + epilogue); wave 4 does all global->LDS staging of the next
// K-chunk.  Measured motivation (tools/stamp_conv.py): each buffer_load / ds_write_b128 costs the wave that issues
// it ~65 cycles of its in-order issue stream, 16 of them per chunk put the MFMA waves at 73.6 cycles/MFMA instead
// of the pipe's 64; moved to a fifth wave that cost overlaps the matrix pipe instead of stalling it.
template <int BN, int MODE>
__global__ __launch_bounds__(320) void conv3x3_f32_lw_kernel(const srk_conv_args a) {
  constexpr int S = 1, MT = 2;
  using G = Geo<S, MT>;
  constexpr int NW4 = 18 * BN;
  constexpr int NTN = BN / 32;
  constexpr int BUF4 = G::NX4 + NW4;
  constexpr int NXL = (G::NX4 + 63) / 64;          // loader slots (one float4 per lane each): input halo
  constexpr int NWL = (NW4 + 63) / 64;             //                                          weights
  constexpr int NL = NXL + NWL;
  constexpr int LB = 8;                            // loader batch: loads in flight before their ds_writes
  __shared__ float4 smem[2 * BUF4];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + SRK_TW - 1) / SRK_TW, tilesH = (a.OH + G::TH - 1) / G::TH;
  int bid = blockIdx.x;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * G::TH, ow0 = tx * SRK_TW, n0 = blockIdx.y * BN;
  const int CoutP = (a.Cout + 31) & ~31;
  const int nq = (a.Cin + 7) >> 3;

  if (wv == 4) {
    // ------------------------------------------------------------------ loader wave
    const int Cps_in = a.Cin >> 2;
    constexpr unsigned OOB = 0x80000000u;
    long img_elems = (long)a.H * a.W * a.x_ldc;
    if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
    const float* ximg = a.x + (long)n * img_elems;
    const unsigned xbytes = (unsigned)(img_elems * 4 > 0x7fffffffL ? 0x7fffffffL : img_elems * 4);
    const unsigned wbytes = (unsigned)((long)nq * 18 * CoutP * 16);
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);
    unsigned vo[NL];
    const int ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int idx = lane + i * 64;
      const int hp = idx >> 1, half = idx & 1;
      const int hy = hp / G::IW, hx = hp - hy * G::IW;
      const int ih = ih0 + hy, iw = iw0 + hx;
      const bool inb = idx < G::NX4 && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
      long off;
      if (MODE == SRK_IN_UNSHUFFLE) off = ((long)(2 * ih) * (2 * a.W) + 2 * iw) * a.x_ldc + a.x_coff + 4 * half;
      else off = ((long)ih * a.W + iw) * a.x_ldc + a.x_coff + 4 * half;
      vo[i] = inb ? (unsigned)(off * 4) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = lane + i * 64;
      const int th = idx / BN, co = idx - th * BN;
      vo[NXL + i] = (idx < NW4 && n0 + co < CoutP) ? (unsigned)((th * CoutP + n0 + co) * 16) : OOB;
    }
    const float in_slope = a.in_slope;
    auto stage = [&](int q, int b) {
      unsigned xso = (unsigned)(8 * q * 4);
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c8 = 8 * q;
        const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
        xso = (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 4);
      }
      const unsigned wso = (unsigned)(q * 18 * CoutP * 16);
      float4* dst = smem + b * BUF4;
#pragma unroll
      for (int i0 = 0; i0 < NL; i0 += LB) {
        f32x4 r[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {
          const int i = i0 + j;
          if (i < NL)
            r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i < NXL ? xrsrc : wrsrc, vo[i], i < NXL ? xso : wso, 0));
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
          const int i = i0 + j;
          if (i >= NL) continue;
          float4 v = make_float4(r[j][0], r[j][1], r[j][2], r[j][3]);
          if (i < NXL) {
            if (in_slope != 1.f) {
              v.x = v.x > 0.f ? v.x : v.x * in_slope; v.y = v.y > 0.f ? v.y : v.y * in_slope;
              v.z = v.z > 0.f ? v.z : v.z * in_slope; v.w = v.w > 0.f ? v.w : v.w * in_slope;
            }
            const int idx = lane + i * 64;
            if (idx < G::NX4) dst[idx] = v;
          } else {
            const int idx = lane + (i - NXL) * 64;
            if (idx < NW4) dst[G::NX4 + idx] = v;
          }
        }
      }
    };
    stage(0, 0);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
      if (q + 1 < nq) stage(q + 1, (q & 1) ^ 1);
      __syncthreads();
    }
    return;
  }

  // ---------------------------------------------------------------------- MFMA waves
  f32x16 acc[MT][NTN];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int t = 0; t < NTN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;
  const int apy = 2 * wv + (l32 >> 4), apx = l32 & 15;
  const int abase = apy * G::IW + apx;
  constexpr int AM = 8 * G::IW;
  float4 av[2][MT], bv[2][NTN];
  auto ld_frag = [&](int p, int b, int tap) {
    const int r = tap / 3, s = tap - 3 * r;
    const float4* xb = smem + b * BUF4 + abase * 2 + hl;
    const float4* wb = smem + b * BUF4 + G::NX4 + hl * BN + l32;
#pragma unroll
    for (int m = 0; m < MT; ++m) av[p][m] = xb[(m * AM + r * G::IW + s) * 2];
#pragma unroll
    for (int t = 0; t < NTN; ++t) bv[p][t] = wb[tap * 2 * BN + t * 32];
  };
  auto mfma_tap = [&](int p) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
          const float ae = e == 0 ? av[p][m].x : e == 1 ? av[p][m].y : e == 2 ? av[p][m].z : av[p][m].w;
          const float be = e == 0 ? bv[p][t].x : e == 1 ? bv[p][t].y : e == 2 ? bv[p][t].z : bv[p][t].w;
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc[m][t], 0, 0, 0);
        }
  };
  __syncthreads();                               // chunk 0 staged by the loader
  ld_frag(0, 0, 0);
  for (int q = 0; q < nq; ++q) {
    const int b = q & 1;
    const bool more = q + 1 < nq;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      ld_frag((tap + 1) & 1, b, tap + 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_tap(tap & 1);
    }
    __syncthreads();                             // buffer b consumed (tap 8 is in registers); b^1 staged
    if (more) ld_frag(1, b ^ 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_tap(0);                                 // tap 8
    if (more) {
#pragma unroll
      for (int m = 0; m < MT; ++m) av[0][m] = av[1][m];
#pragma unroll
      for (int t = 0; t < NTN; ++t) bv[0][t] = bv[1][t];
    }
  }
  conv_epilogue<BN, MT>(a, acc, smem, n, oh0, ow0, n0, wv, lane);
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(2,3) along W ("wino", wp_format 3): the same 16x16 x 64-channel workgroup tile and the same staged
// halo as the loader-wave kernel, but each output ROW PAIR-OF-COLUMNS (2c, 2c+1) is computed from four products instead
// of six:   m0 = (d0-d2) u0, m1 = (d1+d2) u1, m2 = (d2-d1) u2, m3 = (d1-d3) u3,   y0 = m0+m1+m2, y1 = m1-m2-m3
// with u = G w (u0 = w0, u1 = (w0+w1+w2)/2, u2 = (w0-w1+w2)/2, u3 = w2) precomputed by the weight packing (12 "taps"
// = 3 rows x 4 positions instead of 9).  2/3 of the MFMAs of the direct kernel for the same result up to fp32 rounding
// (all arithmetic stays fp32; the transform adds ~1 ulp per operand).
//   * the INPUT transform costs no LDS and no extra wave: a lane reads the four raw pixels (row+r, 2c..2c+3) as float4
//     (4 channels each) and forms the four transformed A operands with 16 VALU ops, in the shadow of the MFMAs;
//   * the OUTPUT transform is register-local: the four position accumulators of a wave hold the same (tile, channel)
//     in the same register of the same lane, and the M index -> (row, column pair) map is chosen so that y0/y1 land
//     exactly in the accumulator layout conv_epilogue expects (pure register renaming, no shuffles);
//   * 9 waves: wave 8 stages global->LDS (halo + 12 weight slices per 8-channel chunk), waves 0-7 = 4 row groups x 2
//     halves of the 64 output channels, four 32x32 accumulators each.
// M tile of row group g (32 "column pairs"): index i -> rows q = i>>3 in {2g, 2g+1, 8+2g, 9+2g}, pair c = cmap(i&7).
__device__ __forceinline__ constexpr int wino_cmap(int u) { return (u & 1) | (((u >> 2) & 1) << 1) | (((u >> 1) & 1) << 2); }

template <int MODE>
__global__ __launch_bounds__(576) void conv3x3_f32_wino_kernel(const srk_conv_args a) {
  constexpr int BN = 64;
  using G = Geo<1, 2>;
  constexpr int NW4 = 24 * BN;                     // 12 taps x 2 k-halves x BN float4 per chunk
  constexpr int BUF4 = G::NX4 + NW4;
  constexpr int NXL = (G::NX4 + 63) / 64;
  constexpr int NWL = (NW4 + 63) / 64;
  constexpr int NL = NXL + NWL;
  constexpr int LB = 8;
  __shared__ float4 smem[2 * BUF4];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + SRK_TW - 1) / SRK_TW, tilesH = (a.OH + G::TH - 1) / G::TH;
  int bid = blockIdx.x;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * G::TH, ow0 = tx * SRK_TW, n0 = blockIdx.y * BN;
  const int CoutP = (a.Cout + 31) & ~31;
  const int nq = (a.Cin + 7) >> 3;

  SRK_STAMP_AT(0);
  if (wv == 8) {
    // ------------------------------------------------------------------ loader wave (as in the lw kernel, 24 weight slices)
    const int Cps_in = a.Cin >> 2;
    constexpr unsigned OOB = 0x80000000u;
    long img_elems = (long)a.H * a.W * a.x_ldc;
    if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
    const float* ximg = a.x + (long)n * img_elems;
    const unsigned xbytes = (unsigned)(img_elems * 4 > 0x7fffffffL ? 0x7fffffffL : img_elems * 4);
    const unsigned wbytes = (unsigned)((long)nq * 24 * CoutP * 16);
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);
    unsigned vo[NL];
    const int ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
    for (int i = 0; i < NXL; ++i) {
      const int idx = lane + i * 64;
      const int hp = idx >> 1, half = idx & 1;
      const int hy = hp / G::IW, hx = hp - hy * G::IW;
      const int ih = ih0 + hy, iw = iw0 + hx;
      const bool inb = idx < G::NX4 && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
      long off;
      if (MODE == SRK_IN_UNSHUFFLE) off = ((long)(2 * ih) * (2 * a.W) + 2 * iw) * a.x_ldc + a.x_coff + 4 * half;
      else off = ((long)ih * a.W + iw) * a.x_ldc + a.x_coff + 4 * half;
      vo[i] = inb ? (unsigned)(off * 4) : OOB;
    }
#pragma unroll
    for (int i = 0; i < NWL; ++i) {
      const int idx = lane + i * 64;
      const int th = idx / BN, co = idx - th * BN;
      vo[NXL + i] = (idx < NW4 && n0 + co < CoutP) ? (unsigned)((th * CoutP + n0 + co) * 16) : OOB;
    }
    const float in_slope = a.in_slope;
    auto stage = [&](int q, int b) {
      unsigned xso = (unsigned)(8 * q * 4);
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c8 = 8 * q;
        const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
        xso = (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 4);
      }
      const unsigned wso = (unsigned)(q * 24 * CoutP * 16);
      float4* dst = smem + b * BUF4;
#pragma unroll
      for (int i0 = 0; i0 < NL; i0 += LB) {
        f32x4 r[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {
          const int i = i0 + j;
          if (i < NL)
            r[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(i < NXL ? xrsrc : wrsrc, vo[i], i < NXL ? xso : wso, 0));
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
          const int i = i0 + j;
          if (i >= NL) continue;
          float4 v = make_float4(r[j][0], r[j][1], r[j][2], r[j][3]);
          if (i < NXL) {
            if (in_slope != 1.f) {
              v.x = v.x > 0.f ? v.x : v.x * in_slope; v.y = v.y > 0.f ? v.y : v.y * in_slope;
              v.z = v.z > 0.f ? v.z : v.z * in_slope; v.w = v.w > 0.f ? v.w : v.w * in_slope;
            }
            const int idx = lane + i * 64;
            if (idx < G::NX4) dst[idx] = v;
          } else {
            const int idx = lane + (i - NXL) * 64;
            if (idx < NW4) dst[G::NX4 + idx] = v;
          }
        }
      }
    };
    stage(0, 0);
    __syncthreads();
    for (int q = 0; q < nq; ++q) {
      if (q + 1 < nq) stage(q + 1, (q & 1) ^ 1);
      __syncthreads();
    }
    return;
  }

  // ---------------------------------------------------------------------- MFMA waves
  const int wg = wv & 3, nh = wv >> 2;             // row group, output-channel half
  f32x16 acc[4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  const int tq = l32 >> 3, tc = wino_cmap(l32 & 7);
  const int trow = 8 * (tq >> 1) + 2 * wg + (tq & 1);
  const int abase = (trow * G::IW + 2 * tc) * 2 + hl;             // float4 index of raw pixel d0 for tap row 0
  const int wbase = G::NX4 + hl * BN + 32 * nh + l32;
  // B fragments are double-buffered by compile-time parity (row steps alternate 0,1,0 | 1,0,1 over a PAIR of chunks), so
  // every ds_read lands in the register the MFMAs will use -- no copies in the loop.
  f32x4 dn[4], V[2][4], Bv[2][4];
  auto ld_row = [&](int b, int r, int par) {
    const f32x4* xb = reinterpret_cast<const f32x4*>(smem + b * BUF4) + abase + r * G::IW * 2;
    const f32x4* wb = reinterpret_cast<const f32x4*>(smem + b * BUF4) + wbase + (4 * r) * 2 * BN;
#pragma unroll
    for (int j = 0; j < 4; ++j) dn[j] = xb[2 * j];
#pragma unroll
    for (int p = 0; p < 4; ++p) Bv[par][p] = wb[p * 2 * BN];
  };
  auto transform = [&](int par) {
    V[par][0] = dn[0] - dn[2]; V[par][1] = dn[1] + dn[2]; V[par][2] = dn[2] - dn[1]; V[par][3] = dn[1] - dn[3];
  };
  auto mfma_row = [&](int par) {
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int p = 0; p < 4; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[par][p][e], Bv[par][p][e], acc[p], 0, 0, 0);
  };
  // one chunk whose row 0 operands sit in parity P0 (already loaded and transformed); leaves the next chunk's row 0 in
  // parity P0 ^ 1
  auto chunk = [&](int q, auto P0c) {
    constexpr int P0 = decltype(P0c)::value;
    const int b = q & 1;
    const bool more = q + 1 < nq;
    ld_row(b, 1, P0 ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(P0);
    __builtin_amdgcn_sched_barrier(0);
    transform(P0 ^ 1);
    ld_row(b, 2, P0);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(P0 ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    transform(P0);
    __syncthreads();                             // buffer b consumed (row 2 is in registers); b^1 staged
    if (more) ld_row(b ^ 1, 0, P0 ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    mfma_row(P0);                                // tap row 2
    __builtin_amdgcn_sched_barrier(0);
    if (more) transform(P0 ^ 1);
  };
  SRK_STAMP_AT(1);
  __syncthreads();                               // chunk 0 staged by the loader
  SRK_STAMP_AT(2);
  SRK_CLOCK_AT(5);
  ld_row(0, 0, 0);
  transform(0);
  int q = 0;
  for (; q + 1 < nq; q += 2) {
    chunk(q, std::integral_constant<int, 0>{});
    chunk(q + 1, std::integral_constant<int, 1>{});
  }
  if (q < nq) chunk(q, std::integral_constant<int, 0>{});
  SRK_CLOCK_AT(6);
  SRK_STAMP_AT(3);
  // output transform + register renaming into conv_epilogue's layout: source register 4*tq + s holds column pair
  // c = cmap(4*hl + s) of row tq; destination tile m = tq>>1, register 4*(2*(tq&1) + (s>>1)) + 2*(s&1) + e.
  f32x16 out[2][1];
#pragma unroll
  for (int tq2 = 0; tq2 < 4; ++tq2)
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
      const int src = 4 * tq2 + sx;
      const float m0 = acc[0][src], m1 = acc[1][src], m2 = acc[2][src], m3 = acc[3][src];
      const int dst = 4 * (2 * (tq2 & 1) + (sx >> 1)) + 2 * (sx & 1);
      out[tq2 >> 1][0][dst] = (m0 + m1) + m2;
      out[tq2 >> 1][0][dst + 1] = (m1 - m2) - m3;
    }
  conv_epilogue<32, 2>(a, out, smem, n, oh0, ow0, n0 + 32 * nh, wg, lane, wv);
  SRK_STAMP_AT(4);
}

// ---------------------------------------------------------------------------------------------------------
// Winograd F(4,3) along W ("wino4", wp_format 5): four output columns from six products per kernel row (direct: 12,
// F(2,3): 8), i.e. half the MFMAs of the direct kernel.  Transforms (Lavin & Gray):
//   v = B^T d :  v0 = 4d0-5d2+d4, v1 = (d3+d4)-4(d1+d2), v2 = (d4-d3)+4(d1-d2), v3 = (d4-d2)+2(d3-d1), v4 = (d4-d2)-2(d3-d1),
//                v5 = 4d1-5d3+d5
//   u = G w   :  w0/4, -(w0+w1+w2)/6, -(w0-w1+w2)/6, w0/24+w1/12+w2/6, w0/24-w1/12+w2/6, w2      (in the weight packing)
//   y = A^T m :  y0 = m0+m1+m2+m3+m4, y1 = (m1-m2)+2(m3-m4), y2 = (m1+m2)+4(m3+m4), y3 = (m1-m2)+8(m3-m4)+m5
// fp32 throughout; the larger constants cost about 1.5 decimal digits against the direct kernel (2e-6 vs 1e-7 relative on
// a K = 2000 dot product), still 2-3 orders inside the 1e-4 / 1e-3 parity bars.
// Workgroup tile 32 rows x 16 columns x 64 channels, 8 waves = 4 row groups x 2 channel halves, six 32x32 accumulators per
// wave; an M tile is 8 rows x 4 column quads, mapped so that the output transform is again a pure register renaming into
// conv_epilogue's layout (source register 4q+s -> tile q, registers 4s..4s+3).  No loader wave: 8 waves x 2 per SIMD leave
// 256 VGPRs each, and the 56 KB per 8-channel chunk (34x18 halo + 18 weight slices) go global -> LDS by DMA from the MFMA
// waves themselves (7 instructions per wave and chunk), double-buffered, one barrier per chunk.  Needs in_slope == 1
// (the DMA cannot apply the input LeakyReLU).
// NH = channel halves (32 output channels each) per workgroup.  NH = 2 (default): the 8-wave, 64-channel workgroup described
// above, one per CU.  NH = 1 (SRK_WINO4_NH=1): a 4-wave, 32-channel workgroup (one wave per SIMD, 76 KB of LDS, <= 256 registers):
// TWO independent workgroups share a CU, each SIMD hosts one wave of each, so the two waves of a SIMD no longer meet at one
// barrier (with NH = 2 the older wave wins the matrix-pipe arbitration and waits ~2.6 k of 11.6 k cycles at the chunk barrier for
// its partner) and the halo is staged twice per CU (78 instead of 56 KB of DMA per chunk).  Measured (stamps + GAN step, same
// box): both workgroups are resident and finish 89-107 us apart instead of together, but the launch as a whole takes the same
// time (138.3 vs 138.0 us per launch in the step) -- the matrix pipe is shared the same way either way.  Kept as the form a
// two-stream schedule (two dependency chains per CU) would need; not the default.
template <int MODE, int NH>
__device__ __forceinline__ void wino4_body(const srk_conv_args& a) {
  constexpr int BN = 32 * NH, TH = 32, IH = TH + 2, IW = SRK_TW + 2;
  constexpr int NWV = 4 * NH;                      // waves per workgroup
  constexpr int NX4 = IH * IW * 2;                 // 1224 float4: [halo pixel][k-half]
  constexpr int NXI = (NX4 + 63) / 64;             // 20 wave-wide DMA instructions of halo (the last one: 8 live lanes)
  constexpr int NXP = NXI * 64;                    // halo region padded to whole instructions (1280 float4)
  constexpr int NW4 = 36 * BN;                     // 18 taps x 2 k-halves x BN
  constexpr int NWI = NW4 / 64;                    // 36 (NH = 2) / 18 (NH = 1) instructions of weights
  constexpr int BUF4 = NXP + NW4;                  // 3584 float4 = 57,344 B (NH = 2) / 2432 float4 = 38,912 B (NH = 1)
  __shared__ float4 smem[2 * BUF4];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + SRK_TW - 1) / SRK_TW, tilesH = (a.OH + TH - 1) / TH;
  // workgroup ids are dealt round-robin to the 8 XCDs (one L2 each): give every XCD a CONTIGUOUS range of tiles so that
  // neighbouring tiles (shared halo rows / columns) and the images of one sample meet in the same L2 (measured, batch 32:
  // HBM fetch 180 MB instead of ~300 MB for the Cin=320 conv = 1.07x its input, and 1-2 % less time; the same remap made the
  // 16x16-tile kernels 3-5 % SLOWER at batch 16 and is not applied there)
  int bid = blockIdx.x;
  {
    const int T = gridDim.x;
    if ((T & 7) == 0) bid = (bid & 7) * (T >> 3) + (bid >> 3);
  }
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * TH, ow0 = tx * SRK_TW, n0 = blockIdx.y * BN;
  const int CoutP = (a.Cout + 31) & ~31;
  const int nq = (a.Cin + 7) >> 3;
  SRK_STAMP_AT(0);

  // ---- DMA plan.  Per chunk NXI = 20 halo + NWI weight instructions (1 KB each), dealt to the waves in turn: piece j of a wave
  // has a COMPILE-TIME kind so that issuing it is an m0 update + one buffer_load...lds: j < NXJ -> halo instruction wv + NWV j,
  // else weight instruction wv + NWV (j - NXJ); the last piece of either kind exists for the low waves only.
  constexpr unsigned OOB = 0x80000000u;
  constexpr int NXJ = (NXI + NWV - 1) / NWV;       // halo pieces per wave: 3 (NH = 2) / 5 (NH = 1)
  constexpr int NWJ = (NWI + NWV - 1) / NWV;       // weight pieces per wave: 5 / 5
  constexpr int NPW = NXJ + NWJ;                   // 8 / 10
  const int Cps_in = a.Cin >> 2;
  long img_elems = (long)a.H * a.W * a.x_ldc;
  if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
  const float* ximg = a.x + (long)n * img_elems;
  const unsigned xbytes = (unsigned)(img_elems * 4 > 0x7fffffffL ? 0x7fffffffL : img_elems * 4);
  const unsigned wbytes = (unsigned)((long)nq * 36 * CoutP * 16);
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);
  unsigned vo[NXJ + 1];
  {
    const int ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
    for (int j = 0; j < NXJ; ++j) {
      const int idx = (wv + NWV * j) * 64 + lane;
      unsigned v = OOB;
      if (idx < NX4) {
        const int hp = idx >> 1, half = idx & 1;
        const int hy = hp / IW, hx = hp - hy * IW;
        const int ih = ih0 + hy, iw = iw0 + hx;
        long off;
        if (MODE == SRK_IN_UNSHUFFLE) off = ((long)(2 * ih) * (2 * a.W) + 2 * iw) * a.x_ldc + a.x_coff + 4 * half;
        else off = ((long)ih * a.W + iw) * a.x_ldc + a.x_coff + 4 * half;
        if (ih >= 0 && iw >= 0 && ih < a.H && iw < a.W) v = (unsigned)(off * 4);
      }
      vo[j] = v;
    }
    // a weight instruction covers 64 / BN packed rows (th) of BN output channels: the lane part of the offset is the same for
    // every piece, the first row of the instruction goes into the scalar offset
    const int wrow = lane / BN, wch = lane - wrow * BN;
    vo[NXJ] = (n0 + wch < CoutP) ? (unsigned)((wrow * CoutP + n0 + wch) * 16) : OOB;
  }
  auto piece = [&](int q, int b, auto jc) {
    constexpr int j = decltype(jc)::value;
    static_assert(j < NPW, "piece index");
    if (q >= nq) return;
    float4* base = smem + b * BUF4;
    if (j < NXJ) {
      const int i = wv + NWV * j;
      if (NXJ * NWV > NXI && j == NXJ - 1 && i >= NXI) return;
      unsigned xso = (unsigned)(8 * q * 4);
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c8 = 8 * q;
        const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
        xso = (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 4);
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, (__attribute__((address_space(3))) void*)(base + i * 64), 16, vo[j < NXJ ? j : 0], xso, 0, 0);
    } else {
      const int i = wv + NWV * (j - NXJ);
      if (NWJ * NWV > NWI && j == NPW - 1 && i >= NWI) return;
      const unsigned wso = (unsigned)((q * 36 + i * (64 / BN)) * CoutP * 16);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (__attribute__((address_space(3))) void*)(base + NXP + i * 64), 16, vo[NXJ], wso, 0, 0);
    }
  };
  static_assert(NXI == 20 && (NWI == 36 || NWI == 18), "piece schedule assumes 20 halo + 36 / 18 weight DMA instructions");

  // ---- MFMA role
  const int wg = wv & 3, nh = wv >> 2;             // row group, output-channel half
  f32x16 acc[6];
#pragma unroll
  for (int p = 0; p < 6; ++p)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
  // M index i = l32 -> column quad tcol = ((i>>2)&1) + 2*(i&1), row = 8*(i>>3) + 2*wg + ((i>>1)&1)
  const int tcol = ((l32 >> 2) & 1) + 2 * (l32 & 1);
  const int trow = 8 * (l32 >> 3) + 2 * wg + ((l32 >> 1) & 1);
  const int abase = (trow * IW + 4 * tcol) * 2 + hl;
  const int wbase = NXP + hl * BN + 32 * nh + l32;
  f32x4 dn[6], V[2][6], Bv[2][6];
  auto ld_d = [&](int b, int r) {
    const f32x4* xb = reinterpret_cast<const f32x4*>(smem + b * BUF4) + abase + r * IW * 2;
#pragma unroll
    for (int j = 0; j < 6; ++j) dn[j] = xb[2 * j];
  };
  auto ld_w = [&](int b, int r, int par) {
    const f32x4* wb = reinterpret_cast<const f32x4*>(smem + b * BUF4) + wbase + (6 * r) * 2 * BN;
#pragma unroll
    for (int p = 0; p < 6; ++p) Bv[par][p] = wb[p * 2 * BN];
  };
  auto ld_row = [&](int b, int r, int par) { ld_d(b, r); ld_w(b, r, par); };
  auto transform = [&](int par) {
    const f32x4 t1 = dn[1] + dn[2], t2 = dn[4] + dn[3], t3 = dn[1] - dn[2], t4 = dn[4] - dn[3];
    const f32x4 t5 = dn[4] - dn[2], t6 = dn[3] - dn[1];
    V[par][0] = 4.f * dn[0] - 5.f * dn[2] + dn[4];
    V[par][1] = t2 - 4.f * t1;
    V[par][2] = t4 + 4.f * t3;
    V[par][3] = t5 + 2.f * t6;
    V[par][4] = t5 - 2.f * t6;
    V[par][5] = 4.f * dn[1] - 5.f * dn[3] + dn[5];
  };
  // 24 MFMAs of one kernel row; DMA pieces J0 .. J0 + NJ - 1 of chunk dq into buffer db are issued between the k-step groups
  auto mfma_row = [&](int par, int dq, int db, auto j0c, auto njc) {
    constexpr int J0 = decltype(j0c)::value, NJ = decltype(njc)::value;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
#pragma unroll
      for (int p = 0; p < 6; ++p) acc[p] = __builtin_amdgcn_mfma_f32_32x32x2f32(V[par][p][e], Bv[par][p][e], acc[p], 0, 0, 0);
      if (e < NJ) {
        __builtin_amdgcn_sched_barrier(0);
        if (e == 0) piece(dq, db, std::integral_constant<int, J0>{});
        if (e == 1) piece(dq, db, std::integral_constant<int, (NJ > 1 ? J0 + 1 : J0)>{});
        if (e == 2) piece(dq, db, std::integral_constant<int, (NJ > 2 ? J0 + 2 : J0)>{});
        if (e == 3) piece(dq, db, std::integral_constant<int, (NJ > 3 ? J0 + 3 : J0)>{});
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  SRK_SEG_BEGIN();
  // pieces 0-1 of a chunk go behind the previous chunk's last row, the other NPW - 2 behind rows 0 and 1, half each
  constexpr int NJR = (NPW - 2) / 2;               // 3 (NH = 2) / 4 (NH = 1)
  using I0 = std::integral_constant<int, 0>; using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, NJR>;
  using I5 = std::integral_constant<int, 2 + NJR>;
  // chunk q sits in buffer b.  Chunk q+1 streams into b^1: its pieces 0-1 were issued behind the previous chunk's last
  // row, 2-7 go behind rows 0 and 1 here; after the barrier b is free and chunk q+2's pieces 0-1 go behind row 2.
  //
  // The two waves of a SIMD (wv and wv + 4: same rows, the two channel halves) leave every barrier together.  If both ran the
  // same order "24 MFMAs, then the next row's transform", their transforms (~60 VALU instructions each) would coincide and
  // the matrix pipe would idle three times per chunk.  So the second wave (ROLE_B) runs every (MFMA row, transform) pair the
  // other way round: it transforms while its partner issues MFMAs and the reverse.  Same instructions, same registers; the
  // offset is in the program order, so barriers do not undo it.
  auto chunk = [&](int q, auto P0c, auto rolec) {
    constexpr int P0 = decltype(P0c)::value;
    constexpr bool ROLE_B = decltype(rolec)::value;
    const int b = q & 1;
    const bool more = q + 1 < nq;
    if constexpr (!ROLE_B) {
      ld_row(b, 1, P0 ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(0);
      mfma_row(P0, q + 1, b ^ 1, I2{}, I3{});
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(1);
      transform(P0 ^ 1);
      SRK_SEG(2);
      ld_row(b, 2, P0);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(0);
      mfma_row(P0 ^ 1, q + 1, b ^ 1, I5{}, I3{});
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(1);
      transform(P0);
      SRK_SEG(2);
      __syncthreads();                             // buffer b consumed (row 2 in registers); chunk q+1 landed in b^1
      SRK_SEG(3);
      if (more) ld_row(b ^ 1, 0, P0 ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(0);
      mfma_row(P0, q + 2, b, I0{}, I2{});          // tap row 2
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(1);
      if (more) transform(P0 ^ 1);
      SRK_SEG(2);
    } else {
      // raw row -> transform -> only then that row's weights (their registers and the raw row are never live together)
      ld_d(b, 1);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(0);
      transform(P0 ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(2);
      ld_w(b, 1, P0 ^ 1);
      mfma_row(P0, q + 1, b ^ 1, I2{}, I3{});
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(1);
      ld_d(b, 2);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(0);
      transform(P0);
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(2);
      ld_w(b, 2, P0);
      mfma_row(P0 ^ 1, q + 1, b ^ 1, I5{}, I3{});
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(1);
      __syncthreads();
      SRK_SEG(3);
      if (more) {
        ld_d(b ^ 1, 0);
        __builtin_amdgcn_sched_barrier(0);
        SRK_SEG(0);
        transform(P0 ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        SRK_SEG(2);
        ld_w(b ^ 1, 0, P0 ^ 1);
      }
      mfma_row(P0, q + 2, b, I0{}, I2{});
      __builtin_amdgcn_sched_barrier(0);
      SRK_SEG(1);
    }
  };
  piece(0, 0, std::integral_constant<int, 0>{}); piece(0, 0, std::integral_constant<int, 1>{});
  piece(0, 0, std::integral_constant<int, 2>{}); piece(0, 0, std::integral_constant<int, 3>{});
  piece(0, 0, std::integral_constant<int, 4>{}); piece(0, 0, std::integral_constant<int, 5>{});
  piece(0, 0, std::integral_constant<int, 6>{}); piece(0, 0, std::integral_constant<int, 7>{});
  if constexpr (NPW > 8) { piece(0, 0, std::integral_constant<int, (NPW > 8 ? 8 : 0)>{}); piece(0, 0, std::integral_constant<int, (NPW > 9 ? 9 : 0)>{}); }
  piece(1, 1, std::integral_constant<int, 0>{});
  piece(1, 1, std::integral_constant<int, 1>{});
  SRK_STAMP_AT(1);
  __syncthreads();
  SRK_STAMP_AT(2);
  SRK_CLOCK_AT(5);
  ld_row(0, 0, 0);
  transform(0);
  SRK_SEG_RESET();
#define SRK_W4_LOOP(ROLE)                                                   \
  {                                                                         \
    int q = 0;                                                              \
    for (; q + 1 < nq; q += 2) {                                            \
      chunk(q, std::integral_constant<int, 0>{}, ROLE{});                   \
      chunk(q + 1, std::integral_constant<int, 1>{}, ROLE{});               \
    }                                                                       \
    if (q < nq) chunk(q, std::integral_constant<int, 0>{}, ROLE{});         \
  }
#ifdef SRK_WINO4_NO_ROLES
  SRK_W4_LOOP(std::false_type)
#else
  if constexpr (NH == 1) SRK_W4_LOOP(std::false_type)        // (the two waves of a SIMD belong to different workgroups: no roles)
  else if (nh == 0) SRK_W4_LOOP(std::false_type) else SRK_W4_LOOP(std::true_type)
#endif
#undef SRK_W4_LOOP
  SRK_SEG_END();
  SRK_CLOCK_AT(6);
  SRK_STAMP_AT(3);
  // output transform: source register 4q+s -> epilogue tile q, registers 4s .. 4s+3 (columns 4*tcol .. +3)
  f32x16 out[4][1];
#pragma unroll
  for (int tq = 0; tq < 4; ++tq)
#pragma unroll
    for (int sx = 0; sx < 4; ++sx) {
      const int src = 4 * tq + sx;
      const float m0 = acc[0][src], m1 = acc[1][src], m2 = acc[2][src], m3 = acc[3][src], m4 = acc[4][src], m5 = acc[5][src];
      const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
      out[tq][0][4 * sx + 0] = (m0 + s12) + s34;
      out[tq][0][4 * sx + 1] = d12 + 2.f * d34;
      out[tq][0][4 * sx + 2] = s12 + 4.f * s34;
      out[tq][0][4 * sx + 3] = (d12 + 8.f * d34) + m5;
    }
  __builtin_amdgcn_sched_barrier(0);     // keep the epilogue's 32 prefetch loads (128 VGPRs) behind the output transform
  conv_epilogue<32, 4, false, 16>(a, out, smem, n, oh0, ow0, n0 + 32 * nh, wg, lane, wv);
  SRK_STAMP_AT(4);
}

// The two launchable forms of the body above (thin kernels: the launch bounds differ, and a __global__ template over NH lost its
// host-side launch stubs in this clang without any diagnostic).
template <int MODE>
__global__ __launch_bounds__(512) void conv3x3_f32_wino4_kernel(const srk_conv_args a) { wino4_body<MODE, 2>(a); }
template <int MODE>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_wino4h_kernel(const srk_conv_args a) { wino4_body<MODE, 1>(a); }   // 2 waves per SIMD = two workgroups per CU: at most 256 registers

template <int BN, int S, int MODE, bool VEC, int MT, bool DMA>
int launch_k(const srk_conv_args& a, hipStream_t st) {
  const int tilesW = srk_div_up(a.OW, SRK_TW), tilesH = srk_div_up(a.OH, SRK_TH * MT);
  const int CoutP = srk_round_up(a.Cout, 32);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)srk_div_up(CoutP, BN));
  hipLaunchKernelGGL((conv3x3_f32_kernel<BN, S, MODE, VEC, MT, DMA>), grid, dim3(SRK_THREADS), 0, st, a);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

// The global->LDS DMA variant (DMA = true) is correct but measured SLOWER than register staging on gfx950
// (80 vs 73.6 cycles per MFMA at one workgroup per CU: each buffer_load...lds piece costs the issuing MFMA wave
// more issue time than a buffer_load + ds_write_b128 pair), so it is not dispatched.
template <int MODE>
int launch_wino(const srk_conv_args& a, hipStream_t st) {
  const int tilesW = srk_div_up(a.OW, SRK_TW), tilesH = srk_div_up(a.OH, SRK_TH * 2);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)(srk_round_up(a.Cout, 64) / 64));
  hipLaunchKernelGGL((conv3x3_f32_wino_kernel<MODE>), grid, dim3(576), 0, st, a);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

static int g_wino4_nh = -1;   // SRK_WINO4_NH = 1 | 2: channel halves per workgroup of the F(4,3) kernel (A/B measurements)

template <int MODE>
int launch_wino4(const srk_conv_args& a, hipStream_t st) {
  const int tilesW = srk_div_up(a.OW, SRK_TW), tilesH = srk_div_up(a.OH, 32);
  if (g_wino4_nh < 0) { const char* e = getenv("SRK_WINO4_NH"); g_wino4_nh = e ? atoi(e) : 2; }
  if (g_wino4_nh == 1) {
    dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)(srk_round_up(a.Cout, 32) / 32));
    hipLaunchKernelGGL((conv3x3_f32_wino4h_kernel<MODE>), grid, dim3(256), 0, st, a);
  } else {
    dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)(srk_round_up(a.Cout, 64) / 64));
    hipLaunchKernelGGL((conv3x3_f32_wino4_kernel<MODE>), grid, dim3(512), 0, st, a);
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

template <int BN, int MODE>
int launch_lw(const srk_conv_args& a, hipStream_t st) {
  const int tilesW = srk_div_up(a.OW, SRK_TW), tilesH = srk_div_up(a.OH, SRK_TH * 2);
  const int CoutP = srk_round_up(a.Cout, 32);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)srk_div_up(CoutP, BN));
  hipLaunchKernelGGL((conv3x3_f32_lw_kernel<BN, MODE>), grid, dim3(320), 0, st, a);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

static int g_use_lw = -1;   // SRK_CONV_LW=0 disables the loader-wave kernels (A/B measurements)

template <int BN, int S, int MODE, bool VEC, int MT>
int launch(const srk_conv_args& a, hipStream_t st) {
  if constexpr (S == 1 && VEC && MT == 2 && MODE != SRK_IN_ZERO_UPSAMPLE) {
    // SRK_CONV_LW: 0 = never, 1 = 64-channel tiles only (default), 2 = also 32-channel tiles with Cout >= 16
    if (g_use_lw < 0) { const char* e = getenv("SRK_CONV_LW"); g_use_lw = e ? atoi(e) : 1; }
    if (g_use_lw >= 1 && BN == 64) return launch_lw<BN, MODE>(a, st);
    if (g_use_lw >= 2 && BN == 32 && a.Cout >= 16) return launch_lw<BN, MODE>(a, st);
  }
  return launch_k<BN, S, MODE, VEC, MT, false>(a, st);
}

// Tile choice: 16x16 pixels (two M tiles per wave: twice the MFMA work per LDS fragment and per barrier)
// whenever the image is tall enough that the 16-row tile wastes no more rows than the 8-row one.
template <int S, int MODE, bool VEC>
int launch_bn(const srk_conv_args& a, hipStream_t st) {
  if constexpr (S == 1) {
    const bool tall = srk_round_up(a.OH, 16) == srk_round_up(a.OH, 8);
    if (a.Cout > 32) return tall ? launch<64, S, MODE, VEC, 2>(a, st) : launch<64, S, MODE, VEC, 1>(a, st);
    return tall ? launch<32, S, MODE, VEC, 2>(a, st) : launch<32, S, MODE, VEC, 1>(a, st);
  }
  return launch<32, S, MODE, VEC, 1>(a, st);
}

}  // namespace

int srk_launch_conv_bf16x3(const srk_conv_args& a, hipStream_t st);
int srk_launch_conv_wino42(const srk_conv_args& a, hipStream_t st);      // srk_conv_w42.hip
int srk_conv_wino42_nmt(const srk_conv_args& a);
int srk_launch_conv_h16(const srk_conv_args& a, hipStream_t st);         // srk_conv_h16.hip (wp_format 7 / 8)
int srk_conv_h16_name(const srk_conv_args& a, char* buf, size_t len);
size_t srk_conv_h16_signs_bytes(const srk_conv_args& a);
int srk_conv_h16_mt(const srk_conv_args& a);
int srk_launch_conv_h16_chain(const srk_conv_args* args, int n, hipStream_t st);   // 1: launched as one chain kernel, 0: not eligible, < 0: error
unsigned srk_chain_fault();          // (srk_chain.h) != 0 while a chain launch's time-out is pending
int srk_conv_h16_chain_would(const srk_conv_args* args, int n);
int srk_conv_h16_chain_m16();
int srk_conv_h16_chain_name(const srk_conv_args* args, int n, char* buf, size_t len);
int srk_launch_conv_w42_chain(const srk_conv_args* args, int n, hipStream_t st);   // the same for the fp32 F(2x4,3x3) kernel (wp_format 6)
int srk_conv_w42_chain_would(const srk_conv_args* args, int n);
int srk_conv_w42_chain_name(const srk_conv_args* args, int n, char* buf, size_t len);
size_t srk_conv_w42_chain_signs_bytes(const srk_conv_args* args, int n);
int srk_conv_small_kind(const srk_conv_args& a);                          // srk_conv_small.hip
int srk_launch_conv_small(const srk_conv_args& a, int kind, hipStream_t st);

extern "C" int srk_conv3x3(const srk_conv_args* pa, void* stream) {
  if (!pa) return SRK_ERR_BAD_ARG;
  const srk_conv_args& a = *pa;
  hipStream_t st = (hipStream_t)stream;
  // the kernels address one input image through a 32-bit buffer resource (out-of-range lanes read 0)
  if (a.H > 0 && a.W > 0 && a.x_ldc > 0 &&
      (long)a.H * a.W * a.x_ldc * 4 * (a.in_mode == SRK_IN_UNSHUFFLE ? 4 : 1) > 0x7fffffffL) return SRK_ERR_UNSUPPORTED;
  // the epilogue addresses one output image (and the same image of r1 / r2 / mask) through a 32-bit buffer resource
  if (a.OH > 0 && a.OW > 0) {
    const long px = (long)a.OH * a.OW * (a.ps_out ? 4 : 1);
    int ld = a.y_ldc;
    if (a.r1 && a.r1_ldc > ld) ld = a.r1_ldc;
    if (a.r2 && a.r2_ldc > ld) ld = a.r2_ldc;
    if (a.mask && a.m_ldc > ld) ld = a.m_ldc;
    if (px * ld * 4 > 0x7fffffffL) return SRK_ERR_UNSUPPORTED;
  }
  if (a.wp_format == 7 || a.wp_format == 8) return srk_launch_conv_h16(a, st);     // 16-bit activation storage: its own checks
  if (a.flags & (SRK_CONV_WRITE_SIGNS | SRK_CONV_MASK_SIGNS)) return SRK_ERR_UNSUPPORTED;      // (sign bits: the 16-bit kernels only)
  if (a.wp_format == 1 || a.wp_format == 2) {
    if (!a.x || !a.y || !a.wp || a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0) return SRK_ERR_BAD_ARG;
    if (!srk_conv3x3_bf16x3_supported(pa) || (((uintptr_t)a.wp & 15) != 0)) return SRK_ERR_UNSUPPORTED;
    if (a.ps_out && (a.Cout & 3)) return SRK_ERR_BAD_ARG;
    return srk_launch_conv_bf16x3(a, st);
  }
  if (a.wp_format == 6) {
    // 2-D Winograd F(2x4, 3x3) fragments (fmt 6): same eligibility as format 5
    if (!a.x || !a.y || !a.wp || a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.Cin <= 0 || a.Cout <= 0) return SRK_ERR_BAD_ARG;
    if (a.stride != 1 || (a.in_mode != SRK_IN_PLAIN && a.in_mode != SRK_IN_UNSHUFFLE) || (a.Cout % 64) || (a.Cin % 8) || a.in_slope != 1.f) return SRK_ERR_UNSUPPORTED;
    if (a.in_mode == SRK_IN_UNSHUFFLE && ((a.Cin & 3) || ((a.Cin >> 2) % 8))) return SRK_ERR_UNSUPPORTED;
    if ((a.x_ldc % 4) || (a.x_coff % 4) || (((uintptr_t)a.x | (uintptr_t)a.wp) & 15)) return SRK_ERR_ALIGNMENT;
    if (a.ps_out && (a.Cout & 3)) return SRK_ERR_BAD_ARG;
    return srk_launch_conv_wino42(a, st);
  }
  if (a.wp_format == 5) {
    // Winograd F(4,3)-along-W fragments (fmt 5): as format 3, plus no input activation (staged by DMA)
    if (!a.x || !a.y || !a.wp || a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.Cin <= 0 || a.Cout <= 0) return SRK_ERR_BAD_ARG;
    if (a.stride != 1 || (a.in_mode != SRK_IN_PLAIN && a.in_mode != SRK_IN_UNSHUFFLE) || (a.Cout % 64) || (a.Cin % 8) || a.in_slope != 1.f) return SRK_ERR_UNSUPPORTED;
    if (a.in_mode == SRK_IN_UNSHUFFLE && ((a.Cin & 3) || ((a.Cin >> 2) % 8))) return SRK_ERR_UNSUPPORTED;
    if ((a.x_ldc % 4) || (a.x_coff % 4) || (((uintptr_t)a.x | (uintptr_t)a.wp) & 15)) return SRK_ERR_ALIGNMENT;
    if (a.ps_out && (a.Cout & 3)) return SRK_ERR_BAD_ARG;
    return a.in_mode == SRK_IN_PLAIN ? launch_wino4<SRK_IN_PLAIN>(a, st) : launch_wino4<SRK_IN_UNSHUFFLE>(a, st);
  }
  if (a.wp_format == 3) {
    // Winograd F(2,3)-along-W fragments (srk_pack_weights with fmt 3): stride 1, 8-channel chunks, 64-channel output tiles
    if (!a.x || !a.y || !a.wp || a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.Cin <= 0 || a.Cout <= 0) return SRK_ERR_BAD_ARG;
    if (a.stride != 1 || (a.in_mode != SRK_IN_PLAIN && a.in_mode != SRK_IN_UNSHUFFLE) || (a.Cout % 64) || (a.Cin % 8)) return SRK_ERR_UNSUPPORTED;
    if (a.in_mode == SRK_IN_UNSHUFFLE && ((a.Cin & 3) || ((a.Cin >> 2) % 8))) return SRK_ERR_UNSUPPORTED;
    if ((a.x_ldc % 4) || (a.x_coff % 4) || (((uintptr_t)a.x | (uintptr_t)a.wp) & 15)) return SRK_ERR_ALIGNMENT;
    if (a.ps_out && (a.Cout & 3)) return SRK_ERR_BAD_ARG;
    return a.in_mode == SRK_IN_PLAIN ? launch_wino<SRK_IN_PLAIN>(a, st) : launch_wino<SRK_IN_UNSHUFFLE>(a, st);
  }
  if (a.wp_format != 0) return SRK_ERR_UNSUPPORTED;
  if (!a.x || !a.y || !a.wp) return SRK_ERR_BAD_ARG;
  if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.Cin <= 0 || a.Cout <= 0) return SRK_ERR_BAD_ARG;
  if (a.stride != 1 && a.stride != 2) return SRK_ERR_UNSUPPORTED;
  if (a.stride == 2 && a.in_mode != SRK_IN_PLAIN) return SRK_ERR_UNSUPPORTED;
  if (a.ps_out && (a.Cout & 3)) return SRK_ERR_BAD_ARG;
  // vector (16-byte) loads need 8-channel chunks that never straddle the view and 16-B alignment
  bool vec = (a.Cin % 8 == 0) && (a.x_ldc % 4 == 0) && (a.x_coff % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
  if (a.in_mode == SRK_IN_UNSHUFFLE) {
    if ((a.Cin & 3) || ((a.Cin >> 2) % 8)) return SRK_ERR_UNSUPPORTED;
    if (!vec) return SRK_ERR_ALIGNMENT;
  }
  if (((uintptr_t)a.wp & 15) != 0) return SRK_ERR_ALIGNMENT;
  if (const int small = srk_conv_small_kind(a)) return srk_launch_conv_small(a, small, st);   // <= 4 channels on one side: HBM-bound kernels
  if (a.stride == 2) return vec ? launch_bn<2, SRK_IN_PLAIN, true>(a, st) : launch_bn<2, SRK_IN_PLAIN, false>(a, st);
  switch (a.in_mode) {
    case SRK_IN_PLAIN:
      return vec ? launch_bn<1, SRK_IN_PLAIN, true>(a, st) : launch_bn<1, SRK_IN_PLAIN, false>(a, st);
    case SRK_IN_UNSHUFFLE:
      return launch_bn<1, SRK_IN_UNSHUFFLE, true>(a, st);
    case SRK_IN_ZERO_UPSAMPLE:
      return vec ? launch_bn<1, SRK_IN_ZERO_UPSAMPLE, true>(a, st) : launch_bn<1, SRK_IN_ZERO_UPSAMPLE, false>(a, st);
    default:
      return SRK_ERR_UNSUPPORTED;
  }
}

extern "C" int srk_conv3x3_seq(const srk_conv_args* args, int n, void* stream) {
  if (!args || n <= 0) return SRK_ERR_BAD_ARG;
  // a chain launch that gave up (srk_chain.h): nothing more goes out until the caller has recovered, so that the iteration is repeated
  // rather than continued on activations that were never computed
  if (srk_chain_fault()) return SRK_ERR_CHAIN_TIMEOUT;
  if (n >= 2) {
    // a dense block's sequence as ONE persistent launch (the chain forms: srk_chain.h; 16-bit storage and the fp32 F(2x4,3x3) kernel)
    const int fmt = args[0].wp_format;
    const int rc = (fmt == 7 || fmt == 8) ? srk_launch_conv_h16_chain(args, n, (hipStream_t)stream)
                 : (fmt == 6 ? srk_launch_conv_w42_chain(args, n, (hipStream_t)stream) : 0);
    if (rc < 0) return rc;
    if (rc == 1) return SRK_OK;
  }
  for (int i = 0; i < n; ++i) {
    const int rc = srk_conv3x3(args + i, stream);
    if (rc) return rc;
  }
  return SRK_OK;
}

extern "C" size_t srk_conv3x3_signs_bytes(const srk_conv_args* pa) {
  if (!pa || (pa->wp_format != 7 && pa->wp_format != 8)) return 0;
  return srk_conv_h16_signs_bytes(*pa);
}

// sign bits for a whole srk_conv3x3_seq call: bytes of ONE conv's buffer, 0 if this sequence's launches do not offer them (the fp32
// F(2x4,3x3) kernel has them in its chain form only: a sequence that would go conv by conv has none)
extern "C" size_t srk_conv3x3_seq_signs_bytes(const srk_conv_args* args, int n) {
  if (!args || n <= 0) return 0;
  const int fmt = args[0].wp_format;
  if (fmt == 7 || fmt == 8) {
    size_t b = srk_conv_h16_signs_bytes(args[0]);
    for (int i = 1; i < n && b; ++i) if (srk_conv_h16_signs_bytes(args[i]) != b) b = 0;
    // (a chain kernel indexes the buffer by its 16-row tiles; where the one-conv launches of this geometry would run 8-row tiles the buffer
    // they ask for is larger, never smaller: 2 x the tiles x the same bytes per tile)
    return b;
  }
  if (fmt == 6 && n >= 2) return srk_conv_w42_chain_signs_bytes(args, n);
  return 0;
}

// Layout tag of those sign bits: rows of the workgroup tile that writes / reads them (8 or 16) | wp_format << 8; 0 = no sign bits.  A
// forward sequence and the data-gradient sequence that reads its bits may be dispatched differently (a chain kernel works on 16-row tiles
// wherever it runs, the one-conv kernels on 8 or 16 by launch size; debug switches; the chain forms resting after a time-out): the caller
// uses the bits only when both sequences report the SAME tag, else it passes the mask tensors.
extern "C" int srk_conv3x3_seq_signs_tag(const srk_conv_args* args, int n) {
  if (!args || n <= 0 || srk_conv3x3_seq_signs_bytes(args, n) == 0) return 0;
  const int fmt = args[0].wp_format;
  if (fmt == 7 || fmt == 8) {
    int rows;
    if (n >= 2 && srk_conv_h16_chain_would(args, n) == 1) rows = 16 | (srk_conv_h16_chain_m16() ? 0x40 : 0);      // (the 16x16x32 form: a layout of its own)
    else { const int mt = srk_conv_h16_mt(args[0]); rows = mt == 1 ? 8 : 4 * mt; }
    return rows | (fmt << 8);
  }
  if (fmt == 6) return (16 * srk_conv_wino42_nmt(args[0])) | (fmt << 8);
  return 0;
}

// Name of the ONE kernel srk_conv3x3_seq would launch for the whole sequence (the chain form), or "" when it launches the convolutions
// one by one (then srk_conv3x3_kernel_name names each).
extern "C" int srk_conv3x3_seq_kernel_name(const srk_conv_args* args, int n, char* buf, size_t len) {
  if (!args || n <= 0 || !buf || len < 8) return SRK_ERR_BAD_ARG;
  buf[0] = 0;
  if (n < 2) return SRK_OK;
  const int fmt = args[0].wp_format;
  if ((fmt == 7 || fmt == 8) && srk_conv_h16_chain_would(args, n) == 1) return srk_conv_h16_chain_name(args, n, buf, len);
  if (fmt == 6 && srk_conv_w42_chain_would(args, n) == 1) return srk_conv_w42_chain_name(args, n, buf, len);
  return SRK_OK;
}

// Name (as rocprofv3 prints it) of the kernel srk_conv3x3 dispatches to for these arguments; launches nothing.  The
// measurement harness labels its per-launch event times with it, so the dispatch rules live in this file only.
extern "C" int srk_conv3x3_kernel_name(const srk_conv_args* pa, char* buf, size_t len) {
  if (!pa || !buf || len < 8) return SRK_ERR_BAD_ARG;
  const srk_conv_args& a = *pa;
  if (a.wp_format == 7 || a.wp_format == 8) return srk_conv_h16_name(a, buf, len);
  if (a.wp_format == 1 || a.wp_format == 2) { snprintf(buf, len, "conv3x3_bf16x3_kernel<%d, %d>", a.in_mode, a.wp_format == 1 ? 3 : 1); return SRK_OK; }
  if (a.wp_format == 5) {
    if (g_wino4_nh < 0) { const char* e = getenv("SRK_WINO4_NH"); g_wino4_nh = e ? atoi(e) : 2; }
    snprintf(buf, len, g_wino4_nh == 1 ? "conv3x3_f32_wino4h_kernel<%d>" : "conv3x3_f32_wino4_kernel<%d>", a.in_mode);
    return SRK_OK;
  }
  if (a.wp_format == 6) { snprintf(buf, len, "conv3x3_f32_wino42_kernel<%d, %d>", a.in_mode, srk_conv_wino42_nmt(a)); return SRK_OK; }
  if (a.wp_format == 3) { snprintf(buf, len, "conv3x3_f32_wino_kernel<%d>", a.in_mode); return SRK_OK; }
  if (a.wp_format != 0) return SRK_ERR_UNSUPPORTED;
  if (const int small = srk_conv_small_kind(a)) {
    if (small == 1) snprintf(buf, len, "conv3x3_cin_small_kernel<%d>", a.Cin); else snprintf(buf, len, "conv3x3_cout_small_kernel<%d>", a.Cout);
    return SRK_OK;
  }
  const bool vec = (a.Cin % 8 == 0) && (a.x_ldc % 4 == 0) && (a.x_coff % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
  const int bn = (a.stride == 1 && a.Cout > 32) ? 64 : 32;
  const int mt = (a.stride == 1 && srk_round_up(a.OH, 16) == srk_round_up(a.OH, 8)) ? 2 : 1;
  if (g_use_lw < 0) { const char* e = getenv("SRK_CONV_LW"); g_use_lw = e ? atoi(e) : 1; }
  if (a.stride == 1 && vec && mt == 2 && a.in_mode != SRK_IN_ZERO_UPSAMPLE && ((g_use_lw >= 1 && bn == 64) || (g_use_lw >= 2 && bn == 32 && a.Cout >= 16))) {
    snprintf(buf, len, "conv3x3_f32_lw_kernel<%d, %d>", bn, a.in_mode);
    return SRK_OK;
  }
  snprintf(buf, len, "conv3x3_f32_kernel<%d, %d, %d, %s, %d, false>", bn, a.stride, a.in_mode, vec ? "true" : "false", mt);
  return SRK_OK;
}

#ifdef SRK_STAMP
// diagnostic build only: what the runtime says about residency of the two F(4,3) kernel forms
extern "C" int srk_debug_occupancy(int* out4) {
  int n = 0;
  hipDeviceProp_t pr;
  if (hipGetDeviceProperties(&pr, 0) != hipSuccess) return -5;
  out4[2] = (int)pr.maxSharedMemoryPerMultiProcessor; out4[3] = (int)pr.sharedMemPerBlock;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_f32_wino4h_kernel<SRK_IN_PLAIN>, 256, 0) != hipSuccess) return -5;
  out4[0] = n;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, conv3x3_f32_wino4_kernel<SRK_IN_PLAIN>, 512, 0) != hipSuccess) return -5;
  out4[1] = n;
  return 0;
}
#endif
