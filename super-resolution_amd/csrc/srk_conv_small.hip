// srk_conv_small.hip -- 3x3 / pad-1 / stride-1 convolutions with a SMALL channel count on one side, for gfx950, fp32.
//
// The layers at the two ends of the networks have almost no arithmetic per byte: the discriminator's first conv
// (image -> 16 channels at 256x256: 288 FLOP per 68 B), its data gradient (16 -> image), the generator's conv1 (image -> F) and
// its tail conv3.2 (F -> image) with their data gradients.  They are bound by HBM, and on the MFMA kernels of srk_conv.hip --
// whose tiles are built for 8-channel K chunks and 32 / 64 output channels -- they ran at 0.8-1.1 TB/s.  The two kernels here
// are plain VALU kernels shaped for the memory system instead:
//   conv3x3_cin_small_kernel   Cin <= 4,  Cout % 4 == 0: a lane owns 4 output channels of one pixel (16-byte stores, a wave
//                              writes 1 KB contiguous), the whole 3x3xCin weight slice of its channel quad sits in registers
//   conv3x3_cout_small_kernel  Cout <= 4, Cin % 4 == 0: a lane owns one output pixel, the input tile goes through LDS in
//                              16-channel chunks (each input byte is fetched once per 16x16 tile), weights come from the
//                              scalar cache (they are wave-uniform)
// Both read the direct-format packed weights (wp_format 0) that srk_pack_weights already produces and implement the same
// fused epilogue as conv_epilogue (bias, alpha, r1 / r2 residuals, LeakyReLU, LeakyReLU' mask) and the same input options
// (channel-slice views, input LeakyReLU), so srk_conv3x3 can route to them by shape alone.
//
// Mirrors nn.Conv2d at /root/reference/models.py:63 (conv1), :99 (conv3[2]), :142 (first discriminator conv), :168 (last
// discriminator conv) and their autograd data gradients.
#include "srk_internal.h"
#include <stdio.h>
#include <stdlib.h>

namespace {

constexpr int ST = 16;                 // output tile 16 x 16
constexpr int SI = ST + 2;             // input tile with halo

// packed fp32 fragments (fmt 0): float4 index ((q8 * 9 + tap) * 2 + h) * Mp + m holds input channels 8 q8 + 4 h + {0..3} of
// output channel m (srk_misc.hip pack_item)
__device__ __forceinline__ const float4* wp4_of(const srk_conv_args& a) { return reinterpret_cast<const float4*>(a.wp); }

template <int CIN>
__global__ __launch_bounds__(256) void conv3x3_cin_small_kernel(const srk_conv_args a, int ntiles) {
  __shared__ float xin[2][SI * SI * CIN];
  const int tid = threadIdx.x;
  const int Q = a.Cout >> 2;                        // channel quads per pixel (host: power of two, <= 64)
  const int q = tid & (Q - 1), pl0 = tid / Q, ppp = 256 / Q;
  const int tilesW = (a.OW + ST - 1) / ST, tilesH = (a.OH + ST - 1) / ST;
  const int Mp = (a.Cout + 31) & ~31;

  // this lane's weights: 9 taps x CIN inputs x 4 outputs, loaded once per workgroup (the workgroup walks over many tiles:
  // per tile they would be 2.3x the bytes of the tile's output)
  float4 w[9][CIN];
  {
    const float4* wp = wp4_of(a);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      float4 t[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = wp[(tap * 2) * Mp + 4 * q + j];
#pragma unroll
      for (int c = 0; c < CIN; ++c) {
        const float* f0 = reinterpret_cast<const float*>(&t[0]);
        const float* f1 = reinterpret_cast<const float*>(&t[1]);
        const float* f2 = reinterpret_cast<const float*>(&t[2]);
        const float* f3 = reinterpret_cast<const float*>(&t[3]);
        w[tap][c] = make_float4(f0[c], f1[c], f2[c], f3[c]);
      }
    }
  }
  float4 bq = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.bias) bq = *reinterpret_cast<const float4*>(a.bias + 4 * q);

  // XCD-contiguous tile ranges (round 4): workgroup b runs on XCD b & 7, and each XCD has its own L2.  With tile = b, b + grid, ... the two
  // 16-pixel tiles that share a 128-byte line of an image row, and the tiles that share halo rows, landed on different XCDs: every L2
  // fetched the same lines again (PMC, Cin = 1 at 256 x 256: 67.8 MB fetched for 8.4 MB of input).  Now XCD x walks tiles
  // [x T8, (x + 1) T8) with its gridDim / 8 workgroups.
  int buf = 0;
  const bool xcd = (gridDim.x & 7) == 0;
  const int T8 = (ntiles + 7) >> 3;
  const int t_first = xcd ? (int)(blockIdx.x & 7) * T8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int t_step = xcd ? (int)(gridDim.x >> 3) : (int)gridDim.x;
  const int t_end = xcd ? (((int)(blockIdx.x & 7) + 1) * T8 < ntiles ? ((int)(blockIdx.x & 7) + 1) * T8 : ntiles) : ntiles;
  for (int tile = t_first; tile < t_end; tile += t_step, buf ^= 1) {
    int bid = tile;
    const int tx = bid % tilesW; bid /= tilesW;
    const int ty = bid % tilesH; bid /= tilesH;
    const int n = bid;
    const int oh0 = ty * ST, ow0 = tx * ST;
    // input tile (zero padded, input LeakyReLU applied); two LDS buffers: one barrier per tile
    const float* ximg = a.x + (long)n * a.H * a.W * a.x_ldc + a.x_coff;
    float* xs = xin[buf];
    for (int t = tid; t < SI * SI * CIN; t += 256) {
      const int c = t % CIN, hp = t / CIN;
      const int hy = hp / SI, hx = hp - hy * SI;
      const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
      float v = 0.f;
      if (ih >= 0 && iw >= 0 && ih < a.H && iw < a.W) v = ximg[((long)ih * a.W + iw) * a.x_ldc + c];
      xs[t] = v > 0.f ? v : v * a.in_slope;
    }
    __syncthreads();

    for (int pl = pl0; pl < ST * ST; pl += ppp) {
      const int r = pl >> 4, c = pl & 15;
      const int oh = oh0 + r, ow = ow0 + c;
      if (oh >= a.OH || ow >= a.OW) continue;
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const float* xp = xs + ((r + dy) * SI + c + dx) * CIN;
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const float xv = xp[ci];
            const float4 wv = w[dy * 3 + dx][ci];
            acc.x += xv * wv.x; acc.y += xv * wv.y; acc.z += xv * wv.z; acc.w += xv * wv.w;
          }
        }
      const long pix = ((long)n * a.OH + oh) * a.OW + ow;
      float4 o = make_float4((acc.x + bq.x) * a.alpha, (acc.y + bq.y) * a.alpha, (acc.z + bq.z) * a.alpha, (acc.w + bq.w) * a.alpha);
      if (a.r1) { const float4 rr = *reinterpret_cast<const float4*>(a.r1 + pix * a.r1_ldc + a.r1_coff + 4 * q);
                  o.x += a.beta1 * rr.x; o.y += a.beta1 * rr.y; o.z += a.beta1 * rr.z; o.w += a.beta1 * rr.w; }
      if (a.r2) { const float4 rr = *reinterpret_cast<const float4*>(a.r2 + pix * a.r2_ldc + a.r2_coff + 4 * q);
                  o.x += a.beta2 * rr.x; o.y += a.beta2 * rr.y; o.z += a.beta2 * rr.z; o.w += a.beta2 * rr.w; }
      o.x = o.x > 0.f ? o.x : o.x * a.slope; o.y = o.y > 0.f ? o.y : o.y * a.slope;
      o.z = o.z > 0.f ? o.z : o.z * a.slope; o.w = o.w > 0.f ? o.w : o.w * a.slope;
      if (a.mask) { const float4 m = *reinterpret_cast<const float4*>(a.mask + pix * a.m_ldc + a.m_coff + 4 * q);
                    o.x *= (m.x > 0.f ? 1.f : a.mask_slope); o.y *= (m.y > 0.f ? 1.f : a.mask_slope);
                    o.z *= (m.z > 0.f ? 1.f : a.mask_slope); o.w *= (m.w > 0.f ? 1.f : a.mask_slope); }
      *reinterpret_cast<float4*>(a.y + pix * a.y_ldc + a.y_coff + 4 * q) = o;
    }
  }
}

constexpr int CK = 16;                 // input channels per LDS chunk
constexpr int CKP = CK + 4;            // padded pixel stride (floats): the 16-byte reads of a wave hit distinct bank groups

// Cout <= 4.  Lanes = (pixel, channel quad of the current 16-channel chunk): a lane multiplies ITS four input channels of the
// nine taps with its own slice of the weights (registers: no per-product weight fetch), partial sums of the four quads of a
// pixel are added with two lane shuffles at the end.  The input tile goes through LDS once per 16x16 tile and chunk.
template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_cout_small_kernel(const srk_conv_args a, int ntiles) {
  __shared__ __attribute__((aligned(16))) float xin[SI * SI * CKP];
  const int tid = threadIdx.x;
  const int qd = tid & 3, p0 = tid >> 2;             // pixel p0 + 64 * pass of the tile
  const int tilesW = (a.OW + ST - 1) / ST, tilesH = (a.OH + ST - 1) / ST;
  int bid = blockIdx.x;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * ST, ow0 = tx * ST;
  const int Mp = (a.Cout + 31) & ~31;
  const float4* wp = wp4_of(a);
  const float* ximg = a.x + (long)n * a.H * a.W * a.x_ldc + a.x_coff;
  const int nck = (a.Cin + CK - 1) / CK;

  float acc[4][COUT];
#pragma unroll
  for (int ps = 0; ps < 4; ++ps)
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[ps][o] = 0.f;

  for (int ck = 0; ck < nck; ++ck) {
    if (ck) __syncthreads();                         // previous chunk fully read
    for (int t = tid; t < SI * SI * (CK / 4); t += 256) {
      const int sq = t & 3, hp = t >> 2;
      const int hy = hp / SI, hx = hp - hy * SI;
      const int ih = oh0 - 1 + hy, iw = ow0 - 1 + hx;
      const int ch = ck * CK + 4 * sq;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ih >= 0 && iw >= 0 && ih < a.H && iw < a.W && ch < a.Cin) v = *reinterpret_cast<const float4*>(ximg + ((long)ih * a.W + iw) * a.x_ldc + ch);
      v.x = v.x > 0.f ? v.x : v.x * a.in_slope; v.y = v.y > 0.f ? v.y : v.y * a.in_slope;
      v.z = v.z > 0.f ? v.z : v.z * a.in_slope; v.w = v.w > 0.f ? v.w : v.w * a.in_slope;
      *reinterpret_cast<float4*>(xin + hp * CKP + 4 * sq) = v;
    }
    // this lane's weights for the chunk: input channels ck*16 + 4 qd .. + 3 of every tap and output
    float4 w[9][COUT];
    {
      const int q8 = 2 * ck + (qd >> 1), h = qd & 1;
      const bool live = ck * CK + 4 * qd < a.Cin;   // (the packed buffer ends with the last 8-channel group of Cin)
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int o = 0; o < COUT; ++o) w[tap][o] = live ? wp[((q8 * 9 + tap) * 2 + h) * Mp + o] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int pl = p0 + 64 * ps, r = pl >> 4, c = pl & 15;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float4 xv = *reinterpret_cast<const float4*>(xin + ((r + tap / 3) * SI + c + tap % 3) * CKP + 4 * qd);
#pragma unroll
        for (int o = 0; o < COUT; ++o) acc[ps][o] += xv.x * w[tap][o].x + xv.y * w[tap][o].y + xv.z * w[tap][o].z + xv.w * w[tap][o].w;
      }
    }
  }
#pragma unroll
  for (int ps = 0; ps < 4; ++ps) {
    const int pl = p0 + 64 * ps, r = pl >> 4, c = pl & 15;
    const int oh = oh0 + r, ow = ow0 + c;
    const long pix = ((long)n * a.OH + oh) * a.OW + ow;
#pragma unroll
    for (int o = 0; o < COUT; ++o) {
      float v = acc[ps][o];
      v += __shfl_xor(v, 1);
      v += __shfl_xor(v, 2);
      if (qd == (o & 3) && oh < a.OH && ow < a.OW && o < a.Cout) {      // the four lanes of a pixel share the output channels
        v = (v + (a.bias ? a.bias[o] : 0.f)) * a.alpha;
        if (a.r1) v += a.beta1 * a.r1[pix * a.r1_ldc + a.r1_coff + o];
        if (a.r2) v += a.beta2 * a.r2[pix * a.r2_ldc + a.r2_coff + o];
        v = v > 0.f ? v : v * a.slope;
        if (a.mask) v *= (a.mask[pix * a.m_ldc + a.m_coff + o] > 0.f ? 1.f : a.mask_slope);
        a.y[pix * a.y_ldc + a.y_coff + o] = v;
      }
    }
  }
}

bool aligned16(const float* p, int ldc, int coff) { return !p || (((ldc | coff) & 3) == 0 && (((uintptr_t)p) & 15) == 0); }

}  // namespace

// Which of the kernels above serves a direct-format (wp_format 0), stride-1, plain-input convolution with <= 4 channels on one
// side: 1 = conv3x3_cin_small_kernel<Cin>, 2 = conv3x3_cout_small_kernel<Cout>, 0 = neither (the MFMA kernels run it).
static int g_small_mode = -1;          // 0 = off, 1 = when the tiles fill the chip (default), 2 = always (tests)
extern "C" int srk_debug_set_conv_small(int mode) { g_small_mode = mode; return SRK_OK; }

int srk_conv_small_kind(const srk_conv_args& a) {
  if (g_small_mode < 0) { const char* e = getenv("SRK_CONV_SMALL"); g_small_mode = e ? atoi(e) : 1; }
  if (!g_small_mode) return 0;
  if (a.wp_format != 0 || a.stride != 1 || a.in_mode != SRK_IN_PLAIN || a.ps_out) return 0;
  if ((((uintptr_t)a.wp) & 15) != 0) return 0;
  // one 256-thread block per 16x16 tile: worth it when the tiles fill the chip (small images stay on the MFMA kernels)
  if (g_small_mode == 1 && (long)a.N * srk_div_up(a.OH, ST) * srk_div_up(a.OW, ST) < 512) return 0;
  if (a.Cin <= 4 && (a.Cout & 3) == 0 && a.Cout >= 4 && a.Cout <= 256 && (a.Cout & (a.Cout - 1)) == 0 &&
      aligned16(a.y, a.y_ldc, a.y_coff) && aligned16(a.r1, a.r1_ldc, a.r1_coff) && aligned16(a.r2, a.r2_ldc, a.r2_coff) &&
      aligned16(a.mask, a.m_ldc, a.m_coff) && aligned16(a.bias, 0, 0)) return 1;
  if (a.Cout <= 4 && (a.Cin & 3) == 0 && aligned16(a.x, a.x_ldc, a.x_coff)) return 2;
  return 0;
}

int srk_launch_conv_small(const srk_conv_args& a, int kind, hipStream_t st) {
  const int tiles = a.N * srk_div_up(a.OH, ST) * srk_div_up(a.OW, ST);
  const int grid1 = tiles < 256 * 8 ? tiles : 256 * 8, grid2 = tiles;   // cin_small is persistent: its workgroups walk over tiles
  if (kind == 1) {
    switch (a.Cin) {
      case 1: hipLaunchKernelGGL(conv3x3_cin_small_kernel<1>, dim3(grid1), dim3(256), 0, st, a, tiles); break;
      case 2: hipLaunchKernelGGL(conv3x3_cin_small_kernel<2>, dim3(grid1), dim3(256), 0, st, a, tiles); break;
      case 3: hipLaunchKernelGGL(conv3x3_cin_small_kernel<3>, dim3(grid1), dim3(256), 0, st, a, tiles); break;
      default: hipLaunchKernelGGL(conv3x3_cin_small_kernel<4>, dim3(grid1), dim3(256), 0, st, a, tiles); break;
    }
  } else {
    switch (a.Cout) {
      case 1: hipLaunchKernelGGL(conv3x3_cout_small_kernel<1>, dim3(grid2), dim3(256), 0, st, a, tiles); break;
      case 2: hipLaunchKernelGGL(conv3x3_cout_small_kernel<2>, dim3(grid2), dim3(256), 0, st, a, tiles); break;
      case 3: hipLaunchKernelGGL(conv3x3_cout_small_kernel<3>, dim3(grid2), dim3(256), 0, st, a, tiles); break;
      default: hipLaunchKernelGGL(conv3x3_cout_small_kernel<4>, dim3(grid2), dim3(256), 0, st, a, tiles); break;
    }
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
