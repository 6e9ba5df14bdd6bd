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
// Mirrors: nn.Conv2d/LeakyReLU/cat/mul+add/PixelShuffle of /root/reference/models.py:19-21,36-41,53,
// 63,67,86-90,97-99,126,142-145,168 (forward) and their autograd data-gradients.
#include "srk_internal.h"

namespace {

template <int S>
struct Geo {
  static constexpr int IH = (SRK_TH - 1) * S + 3;
  static constexpr int IW = (SRK_TW - 1) * S + 3;
  static constexpr int NHP = IH * IW;               // halo pixels
  static constexpr int NX4 = 2 * NHP;               // float4 per chunk (8 ch per pixel)
  static constexpr int NXS = (NX4 + SRK_THREADS - 1) / SRK_THREADS;  // slots per thread
};

template <int BN, int S, int MODE, bool VEC>
__global__ __launch_bounds__(SRK_THREADS) void conv3x3_f32_kernel(const srk_conv_args a) {
  using G = Geo<S>;
  constexpr int NW4 = 18 * BN;                                  // weight float4 per chunk
  constexpr int NWS = (NW4 + SRK_THREADS - 1) / SRK_THREADS;
  constexpr int NXS = G::NXS;
  constexpr int NTN = BN / 32;                                  // accumulator tiles per wave

  __shared__ float4 smem[2 * (G::NX4 + NW4)];
  constexpr int BUF4 = G::NX4 + NW4;   // buffer b: xs at smem + b*BUF4, ws right after it

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wv = tid >> 6;
  const int hl = lane >> 5;     // k-half supplied by this lane
  const int l32 = lane & 31;

  const int tilesW = (a.OW + SRK_TW - 1) / SRK_TW;
  const int tilesH = (a.OH + SRK_TH - 1) / SRK_TH;
  int bid = blockIdx.x;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * SRK_TH, ow0 = tx * SRK_TW;
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

  float4 xr[NXS];
  float4 wr[NWS];

  auto load_chunk = [&](int q) {
    long cadd = 8 * q;
    if (MODE == SRK_IN_UNSHUFFLE) {
      const int c8 = 8 * q;
      const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
      cadd = (long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c;
    }
#pragma unroll
    for (int u = 0; u < NXS; ++u) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (xin[u]) {
        const float* p = a.x + xoff[u] + cadd;
        if (VEC) {
          v = *reinterpret_cast<const float4*>(p);
        } else {
          const int c = 8 * q + 4 * xhalf[u];
          if (c + 0 < a.Cin) v.x = p[0];
          if (c + 1 < a.Cin) v.y = p[1];
          if (c + 2 < a.Cin) v.z = p[2];
          if (c + 3 < a.Cin) v.w = p[3];
        }
      }
      if (a.in_slope != 1.f) {
        v.x = v.x > 0.f ? v.x : v.x * a.in_slope; v.y = v.y > 0.f ? v.y : v.y * a.in_slope;
        v.z = v.z > 0.f ? v.z : v.z * a.in_slope; v.w = v.w > 0.f ? v.w : v.w * a.in_slope;
      }
      xr[u] = v;
    }
    const float4* wq = reinterpret_cast<const float4*>(a.wp) + (long)q * 18 * CoutP;
#pragma unroll
    for (int v = 0; v < NWS; ++v) {
      const int idx = tid + v * SRK_THREADS;
      const int th = idx / BN, co = idx - th * BN;
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < NW4 && n0 + co < CoutP) w4 = wq[th * CoutP + n0 + co];
      wr[v] = w4;
    }
  };
  auto store_chunk = [&](int b) {
#pragma unroll
    for (int u = 0; u < NXS; ++u) {
      const int idx = tid + u * SRK_THREADS;
      if (idx < G::NX4) smem[b * BUF4 + idx] = xr[u];
    }
#pragma unroll
    for (int v = 0; v < NWS; ++v) {
      const int idx = tid + v * SRK_THREADS;
      if (idx < NW4) smem[b * BUF4 + G::NX4 + idx] = wr[v];
    }
  };

  f32x16 acc[NTN];
#pragma unroll
  for (int t = 0; t < NTN; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // A-fragment base: this lane's output pixel inside the tile (wave wv owns rows 2wv, 2wv+1)
  const int apy = 2 * wv + (l32 >> 4), apx = l32 & 15;
  const int abase = (apy * S) * G::IW + apx * S;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();

  for (int q = 0; q < nq; ++q) {
    const int b = q & 1;
    if (q + 1 < nq) load_chunk(q + 1);
    const float4* xb = smem + b * BUF4;
    const float4* wb = xb + G::NX4;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int r = tap / 3, s = tap - 3 * r;
      const float4 av = xb[(abase + r * G::IW + s) * 2 + hl];
#pragma unroll
      for (int t = 0; t < NTN; ++t) {
        const float4 bv = wb[(tap * 2 + hl) * BN + t * 32 + l32];
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[t], 0, 0, 0);
      }
    }
    if (q + 1 < nq) store_chunk(b ^ 1);
    __syncthreads();
  }

  // ---- epilogue.  acc[t][reg]: pixel i = (reg&3) + 8*(reg>>2) + 4*hl, channel = n0 + 32t + l32
  const int Cps_out = a.Cout >> 2;
#pragma unroll
  for (int t = 0; t < NTN; ++t) {
    const int co = n0 + t * 32 + l32;
    if (co >= a.Cout) continue;
    const float bz = a.bias ? a.bias[co] : 0.f;
    int ch = co, pi = 0, pj = 0;
    if (a.ps_out) { const int ij = co / Cps_out; ch = co - ij * Cps_out; pi = ij >> 1; pj = ij & 1; }
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
      const int oh = oh0 + 2 * wv + (i >> 4), ow = ow0 + (i & 15);
      if (oh >= a.OH || ow >= a.OW) continue;
      long pix;
      if (a.ps_out) pix = ((long)(n * 2 * a.OH) + 2 * oh + pi) * (2 * a.OW) + 2 * ow + pj;
      else pix = ((long)n * a.OH + oh) * a.OW + ow;
      float v = a.alpha * (acc[t][reg] + bz);
      if (a.r1) v += a.beta1 * a.r1[pix * a.r1_ldc + a.r1_coff + ch];
      if (a.r2) v += a.beta2 * a.r2[pix * a.r2_ldc + a.r2_coff + ch];
      v = v > 0.f ? v : v * a.slope;
      if (a.mask) v *= (a.mask[pix * a.m_ldc + a.m_coff + ch] > 0.f ? 1.f : a.mask_slope);
      a.y[pix * a.y_ldc + a.y_coff + ch] = v;
    }
  }
}

template <int BN, int S, int MODE, bool VEC>
int launch(const srk_conv_args& a, hipStream_t st) {
  const int tilesW = srk_div_up(a.OW, SRK_TW), tilesH = srk_div_up(a.OH, SRK_TH);
  const int CoutP = srk_round_up(a.Cout, 32);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)srk_div_up(CoutP, BN));
  hipLaunchKernelGGL((conv3x3_f32_kernel<BN, S, MODE, VEC>), grid, dim3(SRK_THREADS), 0, st, a);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

template <int S, int MODE, bool VEC>
int launch_bn(const srk_conv_args& a, hipStream_t st) {
  if constexpr (S == 1) {
    if (a.Cout > 32) return launch<64, S, MODE, VEC>(a, st);
  }
  return launch<32, S, MODE, VEC>(a, st);
}

}  // namespace

extern "C" int srk_conv3x3(const srk_conv_args* pa, void* stream) {
  if (!pa) return SRK_ERR_BAD_ARG;
  const srk_conv_args& a = *pa;
  hipStream_t st = (hipStream_t)stream;
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
