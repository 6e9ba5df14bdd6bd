// srk_conv_bf16x3.hip -- the same fused 3x3 convolution on the bf16 matrix cores with fp32-grade operands.
//
// The fp32 matrix pipe of gfx950 runs at 1/16 of the bf16 rate.  Here every fp32 operand is split into two bf16
// values, x = hi + lo (hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits), and each product is three
// v_mfma_f32_32x32x16_bf16 accumulated in fp32:  a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  (the dropped lo*lo term and
// the split residuals are <= 2^-16 relative per product).  3 MFMAs at 16x the rate = 5.3x the fp32-MFMA ceiling.
// Activations stay fp32 in HBM: the split happens while the halo tile is staged into LDS.  Weights are split
// once per step by srk_pack_weights (format 1).  This mode is OPT-IN (srk_conv_args.wp_format = 1): results differ
// from the exact-fp32 kernels at the 1e-5 level, inside the 1e-3 parity bar but not bit-faithful fp32.
//
// Workgroup = 7 waves: waves 0-3 only read fragments and issue MFMAs; waves 4-6 stage the next 16-channel chunk
// (global fp32 -> split -> LDS, plus the pre-split weight slices).  Tile = 8 rows x 32 columns of output pixels x 64
// output channels; a 32x32 MFMA tile is ONE image row (32 consecutive pixels -> conflict-free ds_read_b128).
// LDS per buffer: 4 planes [hi|lo][k-half][340 halo px][8 ch bf16] = 21.8 KB + weights [tap][hi|lo][k-half][64][8] =
// 36.9 KB; double-buffered (117 KB, one workgroup per CU).
#include "srk_internal.h"
#include "srk_epilogue.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int BT_H = 8, BT_W = 32;               // output tile
constexpr int BIH = BT_H + 2, BIW = BT_W + 2;    // halo
constexpr int BNHP = BIH * BIW;                  // 340 halo pixels
constexpr int XP4 = 4 * BNHP;                    // float4-sized (16 B) units of the x planes per buffer
constexpr int WP4 = 9 * 2 * 2 * 64;              // 16-B units of the weights per buffer (2304)
constexpr int BBUF4 = XP4 + WP4;                 // 3664 units = 58,624 B
constexpr int NLOAD = 3;                         // loader waves (4 measured identical: the main loop is not loader-bound)
constexpr int BTHREADS = 64 * (4 + NLOAD);

// TERMS = 3: split-bf16 (hi*hi + hi*lo + lo*hi);  TERMS = 1: plain bf16 operands (hi*hi only, the lo planes are never
// written or read) -- mixed precision with fp32 accumulate / fp32 master weights (BASELINE config 4).
template <int MODE, int TERMS>
__global__ __launch_bounds__(BTHREADS) void conv3x3_bf16x3_kernel(const srk_conv_args a) {
  constexpr int BN = 64, MT = 2, NTN = 2;
  __shared__ float4 smem[2 * BBUF4];

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + BT_W - 1) / BT_W, tilesH = (a.OH + BT_H - 1) / BT_H;
  int bid = blockIdx.x;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * BT_H, ow0 = tx * BT_W, n0 = blockIdx.y * BN;
  const int CoutP = (a.Cout + 31) & ~31;
  const int nq = (a.Cin + 15) >> 4;

  if (wv >= 4) {
    // ------------------------------------------------------------------ loader waves
    const int lw = wv - 4;
    const int Cps_in = a.Cin >> 2;
    constexpr unsigned OOB = 0x80000000u;
    long img_elems = (long)a.H * a.W * a.x_ldc;
    if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
    const float* ximg = a.x + (long)n * img_elems;
    const unsigned xbytes = (unsigned)(img_elems * 4 > 0x7fffffffL ? 0x7fffffffL : img_elems * 4);
    const unsigned wbytes = (unsigned)((long)nq * 9 * 4 * CoutP * 16);
    __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, xbytes, 0x00020000);
    __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);
    // x slots: (halo pixel, k-half) pairs, 2*BNHP = 680, dealt lane + 64*(lw + NLOAD*i)
    constexpr int NXI = (2 * BNHP + 64 * NLOAD - 1) / (64 * NLOAD);      // 4
    constexpr int NWI = WP4 / (64 * NLOAD);                             // 12
    static_assert(WP4 % (64 * NLOAD) == 0, "weight pieces must divide evenly over the loader lanes");
    unsigned xvo[NXI];
    int xdst[NXI];
    const int ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int slot = lane + 64 * (lw + NLOAD * i);
      const int hp = slot >> 1, h = slot & 1;
      const int hy = hp / BIW, hx = hp - hy * BIW;
      const int ih = ih0 + hy, iw = iw0 + hx;
      const bool ok = slot < 2 * BNHP;
      const bool inb = ok && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W;
      long off;
      if (MODE == SRK_IN_UNSHUFFLE) off = ((long)(2 * ih) * (2 * a.W) + 2 * iw) * a.x_ldc + a.x_coff + 8 * h;
      else off = ((long)ih * a.W + iw) * a.x_ldc + a.x_coff + 8 * h;
      xvo[i] = inb ? (unsigned)(off * 4) : OOB;
      xdst[i] = ok ? (h * BNHP + hp) : -1;           // hi plane index; lo plane is + 2*BNHP
    }
    unsigned wvo[NWI];
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
      const int idx = lane + 64 * (lw + NLOAD * i);              // [tap][part][h][64 co]
      const int th = idx >> 6, co = idx & 63;
      wvo[i] = (n0 + co < CoutP) ? (unsigned)((th * CoutP + n0 + co) * 16) : OOB;
    }
    const float in_slope = a.in_slope;
    auto stage = [&](int q, int b) {
      unsigned xso = (unsigned)(16 * q * 4);
      if (MODE == SRK_IN_UNSHUFFLE) {
        const int c16 = 16 * q;
        const int ij = c16 / Cps_in, c = c16 - ij * Cps_in;
        xso = (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 4);
      }
      const unsigned wso = (unsigned)(q * 9 * 4 * CoutP * 16);
      float4* dst = smem + b * BBUF4;
      f32x4 xa[NXI], xb[NXI], wr[NWI];
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        xa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xvo[i], xso, 0));
        xb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xvo[i] == OOB ? OOB : xvo[i] + 16, xso, 0));
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int part = ((lane + 64 * (lw + NLOAD * i)) >> 7) & 1;       // [tap][part][h][64]: bit 7 of the piece index
        if (TERMS == 3 || part == 0) wr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvo[i], wso, 0));
      }
#pragma unroll
      for (int i = 0; i < NXI; ++i) {
        f32x8 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = xa[i][j]; v[4 + j] = xb[i][j]; }
        if (in_slope != 1.f) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v[j] = v[j] > 0.f ? v[j] : v[j] * in_slope;
        }
        const bf16x8 hi = __builtin_convertvector(v, bf16x8);
        if (xdst[i] >= 0) dst[xdst[i]] = __builtin_bit_cast(float4, hi);
        if constexpr (TERMS == 3) {
          const f32x8 hf = __builtin_convertvector(hi, f32x8);
          const bf16x8 lo = __builtin_convertvector(v - hf, bf16x8);
          if (xdst[i] >= 0) dst[2 * BNHP + xdst[i]] = __builtin_bit_cast(float4, lo);
        }
      }
#pragma unroll
      for (int i = 0; i < NWI; ++i) {
        const int idx = lane + 64 * (lw + NLOAD * i);
        if (TERMS == 3 || ((idx >> 7) & 1) == 0) dst[XP4 + idx] = make_float4(wr[i][0], wr[i][1], wr[i][2], wr[i][3]);
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

  // fragment addresses: x plane [part][h][halo px], element = 8 bf16 (16 B); weights [tap][part][h][co]
  const int abase = hl * BNHP + wv * BIW + l32;          // + m*4*BIW + r*BIW + s  (+ 2*BNHP for the lo plane)
  const int wbase = XP4 + hl * 64 + l32;                 // + (tap*2 + part)*128 + 32t
  bf16x8 ah[2][MT], al[2][MT], bh[2][NTN], bl[2][NTN];
  auto ld_frag = [&](int p, int b, int tap) {
    const int r = tap / 3, s = tap - 3 * r;
    const float4* base = smem + b * BBUF4;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      ah[p][m] = __builtin_bit_cast(bf16x8, base[abase + (4 * m + r) * BIW + s]);
      if constexpr (TERMS == 3) al[p][m] = __builtin_bit_cast(bf16x8, base[2 * BNHP + abase + (4 * m + r) * BIW + s]);
    }
#pragma unroll
    for (int t = 0; t < NTN; ++t) {
      bh[p][t] = __builtin_bit_cast(bf16x8, base[wbase + (tap * 2 + 0) * 128 + 32 * t]);
      if constexpr (TERMS == 3) bl[p][t] = __builtin_bit_cast(bf16x8, base[wbase + (tap * 2 + 1) * 128 + 32 * t]);
    }
  };
  auto mfma_tap = [&](int p) {
    // product term outermost: consecutive MFMAs hit different accumulators
#pragma unroll
    for (int term = 0; term < TERMS; ++term)
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NTN; ++t) {
          const bf16x8 av = term == 2 ? al[p][m] : ah[p][m];
          const bf16x8 bv = term == 1 ? bl[p][t] : bh[p][t];
          acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[m][t], 0, 0, 0);
        }
  };
  __syncthreads();                               // chunk 0 staged by the loaders
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
      for (int m = 0; m < MT; ++m) { ah[0][m] = ah[1][m]; if constexpr (TERMS == 3) al[0][m] = al[1][m]; }
#pragma unroll
      for (int t = 0; t < NTN; ++t) { bh[0][t] = bh[1][t]; if constexpr (TERMS == 3) bl[0][t] = bl[1][t]; }
    }
  }
  conv_epilogue<BN, MT, true>(a, acc, smem, n, oh0, ow0, n0, wv, lane);
}

// ------------------------------------------------------------------------------------------ weight packing (format 1)
// dst[q16][tap][part][h][Mp][8] bf16, k = 16q + 8h + e; part 0 = hi, 1 = lo.  Same byte size as format 0.
// work item = one (q, tap, h, m): 8 consecutive k -> writes the hi piece and the lo piece (16 B each).
__global__ void pack_bf16x3_kernel(const srk_pack_entry* __restrict__ tab, int n, long total) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  int lo_ = 0, hi_ = n - 1;
  while (lo_ < hi_) {
    const int mid = (lo_ + hi_ + 1) >> 1;
    if (tab[mid].elem_begin <= gid) lo_ = mid; else hi_ = mid - 1;
  }
  const srk_pack_entry e = tab[lo_];
  long t = gid - e.elem_begin;
  const int Mp = (e.M + 31) & ~31;
  const int m = (int)(t % Mp); t /= Mp;
  const int h = (int)(t & 1); t >>= 1;
  const int tap = (int)(t % 9); t /= 9;
  const int q = (e.k_off >> 4) + (int)t;
  const int Cps = e.src_cout >> 2;
  f32x8 v;
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
  const bf16x8 hi = __builtin_convertvector(v, bf16x8);
  const f32x8 hf = __builtin_convertvector(hi, f32x8);
  const bf16x8 lo = __builtin_convertvector(v - hf, bf16x8);
  float4* d = reinterpret_cast<float4*>(e.dst) + (((long)q * 9 + tap) * 2) * 2 * Mp;
  d[(0 * 2 + h) * Mp + m] = __builtin_bit_cast(float4, hi);
  d[(1 * 2 + h) * Mp + m] = __builtin_bit_cast(float4, lo);
}

}  // namespace

extern "C" int srk_conv3x3_bf16x3_supported(const srk_conv_args* pa) {
  if (!pa) return 0;
  const srk_conv_args& a = *pa;
  const bool vec = (a.x_ldc % 4 == 0) && (a.x_coff % 4 == 0) && (((uintptr_t)a.x & 15) == 0);
  if (a.stride != 1 || !vec || (a.Cin % 16) || a.Cout < 16) return 0;
  if (a.in_mode == SRK_IN_ZERO_UPSAMPLE) return 0;
  if (a.in_mode == SRK_IN_UNSHUFFLE && ((a.Cin & 3) || ((a.Cin >> 2) % 16))) return 0;
  return 1;
}

int srk_launch_conv_bf16x3(const srk_conv_args& a, hipStream_t st) {
  const int tilesW = srk_div_up(a.OW, BT_W), tilesH = srk_div_up(a.OH, BT_H);
  const int CoutP = srk_round_up(a.Cout, 32);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)srk_div_up(CoutP, 64));
  if (a.wp_format == 2) {
    if (a.in_mode == SRK_IN_UNSHUFFLE) hipLaunchKernelGGL((conv3x3_bf16x3_kernel<SRK_IN_UNSHUFFLE, 1>), grid, dim3(BTHREADS), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_bf16x3_kernel<SRK_IN_PLAIN, 1>), grid, dim3(BTHREADS), 0, st, a);
  } else {
    if (a.in_mode == SRK_IN_UNSHUFFLE) hipLaunchKernelGGL((conv3x3_bf16x3_kernel<SRK_IN_UNSHUFFLE, 3>), grid, dim3(BTHREADS), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_bf16x3_kernel<SRK_IN_PLAIN, 3>), grid, dim3(BTHREADS), 0, st, a);
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

int srk_launch_pack_bf16x3(const srk_pack_entry* dev, int n, int64_t total, hipStream_t st) {
  hipLaunchKernelGGL(pack_bf16x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dev, n, (long)total);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
