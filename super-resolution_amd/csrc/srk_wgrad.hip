// srk_wgrad.hip -- weight/bias gradient of the 3x3 / pad-1 convolution for gfx950, fp32.
//
//   dW[o][c][r][s] = scale * sum_{n,oh,ow} DY[n,oh,ow,o] * X[n, S*oh+r-1, S*ow+s-1, c]
//
// GEMM view: M = 64 output channels, N = 64 input channels (x 9 taps), K = pixels.  A workgroup owns one
// (64 cout) x (64 cin) x 9-tap block of dW and a contiguous range of pixel tiles; wave (a,b) keeps the nine
// 32x32 tiles dW[32a.., 32b.., tap] in 144 accumulator registers for the whole range, so the pixel
// reduction never leaves registers inside a workgroup.  Per pixel tile the DY tile [px][64] and the X halo
// [(rows+2)x(cols+2)][64] are staged once in LDS; the nine taps are nine shifted ds_read_b32 streams of
// the same halo (v_mfma_f32_32x32x2_f32: lane (i, h) supplies pixel 2kk+h).  Partial blocks of the P
// pixel-splits go to a caller workspace and are summed in fixed order by wgrad_reduce (deterministic,
// no float atomics), which also writes the canonical OIHW layout and undoes the PixelShuffle packing.
//
// Mirrors the autograd weight/bias gradient of nn.Conv2d at /root/reference/models.py:19,63,67,87,97,99,
// 142,144,168.
#include "srk_internal.h"

namespace {

constexpr int WTW = 16;
template <int S> struct WGeo {
  static constexpr int TH = (S == 1) ? 4 : 2;
  static constexpr int TP = TH * WTW;              // pixels per tile
  static constexpr int IH = (TH - 1) * S + 3;
  static constexpr int IW = (WTW - 1) * S + 3;
  static constexpr int NHP = IH * IW;
};

struct WPlan { int P, tpb, tilesH, tilesW, total_tiles, nCy, nCz; };

template <int S>
WPlan make_plan(const srk_wgrad_args& a) {
  using G = WGeo<S>;
  WPlan p;
  p.tilesW = srk_div_up(a.OW, WTW);
  p.tilesH = srk_div_up(a.OH, G::TH);
  p.total_tiles = a.N * p.tilesH * p.tilesW;
  p.nCy = srk_div_up(a.Cin, 64);
  p.nCz = srk_div_up(a.Cout, 64);
  int target = 512 / (p.nCy * p.nCz);
  if (target < 1) target = 1;
  int P = p.total_tiles < target ? p.total_tiles : target;
  p.tpb = srk_div_up(p.total_tiles, P);
  p.P = srk_div_up(p.total_tiles, p.tpb);
  return p;
}

template <int S, int DYMODE, bool VEC>
__global__ __launch_bounds__(SRK_THREADS, 2) void wgrad_f32_kernel(const srk_wgrad_args a, const WPlan pl, float* part, float* pbias) {
  using G = WGeo<S>;
  __shared__ float smem[G::TP * 64 + G::NHP * 64];
  float* dys = smem;                 // [TP][64]
  float* xs = smem + G::TP * 64;     // [NHP][64]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int hl = lane >> 5, l32 = lane & 31;
  const int wa = wv & 1, wb = wv >> 1;
  const int p = blockIdx.x, cy = blockIdx.y, cz = blockIdx.z;
  const int cin0 = cy * 64, cout0 = cz * 64;
  const bool active = (cout0 + 32 * wa < a.Cout) && (cin0 + 32 * wb < a.Cin);
  const bool do_bias = (pbias != nullptr) && cy == 0 && wb == 0 && (cout0 + 32 * wa < a.Cout);
  const int Cps = a.Cout >> 2;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int t_begin = p * pl.tpb;
  int t_end = t_begin + pl.tpb;
  if (t_end > pl.total_tiles) t_end = pl.total_tiles;

  for (int tile = t_begin; tile < t_end; ++tile) {
    int tt = tile;
    const int tx = tt % pl.tilesW; tt /= pl.tilesW;
    const int ty = tt % pl.tilesH; tt /= pl.tilesH;
    const int n = tt;
    const int oh0 = ty * G::TH, ow0 = tx * WTW;
    const int ih0 = oh0 * S - 1, iw0 = ow0 * S - 1;

    // ---- stage DY tile
    for (int idx = tid; idx < G::TP * 16; idx += SRK_THREADS) {
      const int px = idx >> 4, c4 = idx & 15;
      const int oh = oh0 + px / WTW, ow = ow0 + px % WTW;
      const int co = cout0 + 4 * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (oh < a.OH && ow < a.OW && co < a.Cout) {
        const float* src;
        if (DYMODE == SRK_IN_UNSHUFFLE) {
          const int ij = co / Cps, c = co - ij * Cps;
          src = a.dy + ((long)(n * 2 * a.OH + 2 * oh + (ij >> 1)) * (2 * a.OW) + 2 * ow + (ij & 1)) * a.dy_ldc + a.dy_coff + c;
        } else {
          src = a.dy + ((long)(n * a.OH + oh) * a.OW + ow) * a.dy_ldc + a.dy_coff + co;
        }
        if (VEC) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (co + 1 < a.Cout) v.y = src[1];
          if (co + 2 < a.Cout) v.z = src[2];
          if (co + 3 < a.Cout) v.w = src[3];
        }
      }
      reinterpret_cast<float4*>(dys)[idx] = v;
    }
    // ---- stage X halo
    for (int idx = tid; idx < G::NHP * 16; idx += SRK_THREADS) {
      const int hp = idx >> 4, c4 = idx & 15;
      const int hy = hp / G::IW, hx = hp - hy * G::IW;
      const int ih = ih0 + hy, iw = iw0 + hx;
      const int ci = cin0 + 4 * c4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ih >= 0 && iw >= 0 && ih < a.H && iw < a.W && ci < a.Cin) {
        const float* src = a.x + ((long)(n * a.H + ih) * a.W + iw) * a.x_ldc + a.x_coff + ci;
        if (VEC) {
          v = *reinterpret_cast<const float4*>(src);
        } else {
          v.x = src[0];
          if (ci + 1 < a.Cin) v.y = src[1];
          if (ci + 2 < a.Cin) v.z = src[2];
          if (ci + 3 < a.Cin) v.w = src[3];
        }
      }
      if (a.in_slope != 1.f) {
        v.x = v.x > 0.f ? v.x : v.x * a.in_slope; v.y = v.y > 0.f ? v.y : v.y * a.in_slope;
        v.z = v.z > 0.f ? v.z : v.z * a.in_slope; v.w = v.w > 0.f ? v.w : v.w * a.in_slope;
      }
      reinterpret_cast<float4*>(xs)[idx] = v;
    }
    __syncthreads();

    if (active) {
#pragma unroll 2
      for (int kk = 0; kk < G::TP / 2; ++kk) {
        const int px = 2 * kk + hl;
        const int py = px / WTW, pxx = px % WTW;
        const float av = dys[px * 64 + 32 * wa + l32];
        if (do_bias) bsum += av;
        const float* xrow = xs + ((py * S) * G::IW + pxx * S) * 64 + 32 * wb + l32;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const int r = tap / 3, s = tap - 3 * r;
          const float bv = xrow[(r * G::IW + s) * 64];
          acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // ---- write partial block: part[p][cz][cy][tap][64 cout][64 cin]
  if (active) {
    float* dst = part + (((long)p * pl.nCz + cz) * pl.nCy + cy) * (9 * 64 * 64);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
        dst[(tap * 64 + 32 * wa + i) * 64 + 32 * wb + l32] = acc[tap][reg];
      }
  }
  if (do_bias) {
    const float tot = bsum + __shfl_xor(bsum, 32);
    if (hl == 0) pbias[((long)p * pl.nCz + cz) * 64 + 32 * wa + l32] = tot;
  }
}

// one thread per (o, c, tap); sums the P partial blocks in fixed order
__global__ void wgrad_reduce_kernel(const srk_wgrad_args a, const WPlan pl, const float* part, const float* pbias) {
  const long total = (long)pl.nCz * pl.nCy * 9 * 64 * 64;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < total) {
    long t = gid;
    const int cl = t & 63; t >>= 6;
    const int ol = t & 63; t >>= 6;
    const int tap = t % 9; t /= 9;
    const int cy = t % pl.nCy; t /= pl.nCy;
    const int cz = (int)t;
    const int o = cz * 64 + ol, c = cy * 64 + cl;
    if (o < a.Cout && c < a.Cin) {
      const long stride = (long)pl.nCz * pl.nCy * 9 * 64 * 64;
      const float* src = part + (((long)cz * pl.nCy + cy) * 9 + tap) * 4096 + ol * 64 + cl;
      float s = 0.f;
      for (int p = 0; p < pl.P; ++p) s += src[p * stride];
      int os = o;
      if (a.dy_mode == SRK_IN_UNSHUFFLE) { const int Cps = a.Cout >> 2; os = 4 * (o % Cps) + o / Cps; }
      float* d = a.dw + ((long)os * a.Cin + c) * 9 + tap;
      *d = a.accumulate ? (*d + a.scale * s) : a.scale * s;
    }
  }
  if (a.db && gid < a.Cout) {
    const int o = (int)gid;
    float s = 0.f;
    for (int p = 0; p < pl.P; ++p) s += pbias[(long)p * pl.nCz * 64 + o];
    int os = o;
    if (a.dy_mode == SRK_IN_UNSHUFFLE) { const int Cps = a.Cout >> 2; os = 4 * (o % Cps) + o / Cps; }
    a.db[os] = a.accumulate ? (a.db[os] + a.scale * s) : a.scale * s;
  }
}

size_t ws_bytes(const WPlan& pl) {
  return ((size_t)pl.P * pl.nCz * pl.nCy * 9 * 64 * 64 + (size_t)pl.P * pl.nCz * 64) * sizeof(float);
}

template <int S, int DYMODE, bool VEC>
int launch(const srk_wgrad_args& a, hipStream_t st) {
  const WPlan pl = make_plan<S>(a);
  if (!a.workspace || a.workspace_bytes < ws_bytes(pl)) return SRK_ERR_WORKSPACE;
  float* part = (float*)a.workspace;
  float* pbias = part + (size_t)pl.P * pl.nCz * pl.nCy * 9 * 64 * 64;
  dim3 grid(pl.P, pl.nCy, pl.nCz);
  hipLaunchKernelGGL((wgrad_f32_kernel<S, DYMODE, VEC>), grid, dim3(SRK_THREADS), 0, st, a, pl, part, a.db ? pbias : nullptr);
  SRK_CHECK_LAUNCH();
  const long total = (long)pl.nCz * pl.nCy * 9 * 64 * 64;
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a, pl, part, pbias);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

int validate(const srk_wgrad_args& a) {
  if (!a.x || !a.dy || !a.dw) return SRK_ERR_BAD_ARG;
  if (a.N <= 0 || a.H <= 0 || a.W <= 0 || a.OH <= 0 || a.OW <= 0 || a.Cin <= 0 || a.Cout <= 0) return SRK_ERR_BAD_ARG;
  if (a.stride != 1 && a.stride != 2) return SRK_ERR_UNSUPPORTED;
  if (a.dy_mode != SRK_IN_PLAIN && a.dy_mode != SRK_IN_UNSHUFFLE) return SRK_ERR_UNSUPPORTED;
  if (a.dy_mode == SRK_IN_UNSHUFFLE && (a.stride != 1 || (a.Cout & 3) || ((a.Cout >> 2) & 3))) return SRK_ERR_UNSUPPORTED;
  return SRK_OK;
}

bool is_vec(const srk_wgrad_args& a) {
  return (a.Cin % 4 == 0) && (a.Cout % 4 == 0) && (a.x_ldc % 4 == 0) && (a.x_coff % 4 == 0) && (a.dy_ldc % 4 == 0) &&
         (a.dy_coff % 4 == 0) && (((uintptr_t)a.x & 15) == 0) && (((uintptr_t)a.dy & 15) == 0);
}

}  // namespace

extern "C" int srk_conv3x3_wgrad_workspace(const srk_wgrad_args* pa, size_t* bytes) {
  if (!pa || !bytes) return SRK_ERR_BAD_ARG;
  int rc = validate(*pa);
  if (rc) return rc;
  *bytes = pa->stride == 1 ? ws_bytes(make_plan<1>(*pa)) : ws_bytes(make_plan<2>(*pa));
  return SRK_OK;
}

extern "C" int srk_conv3x3_wgrad(const srk_wgrad_args* pa, void* stream) {
  if (!pa) return SRK_ERR_BAD_ARG;
  const srk_wgrad_args& a = *pa;
  int rc = validate(a);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  const bool vec = is_vec(a);
  if (a.dy_mode == SRK_IN_UNSHUFFLE) {
    if (!vec) return SRK_ERR_ALIGNMENT;
    return launch<1, SRK_IN_UNSHUFFLE, true>(a, st);
  }
  if (a.stride == 1) return vec ? launch<1, SRK_IN_PLAIN, true>(a, st) : launch<1, SRK_IN_PLAIN, false>(a, st);
  return vec ? launch<2, SRK_IN_PLAIN, true>(a, st) : launch<2, SRK_IN_PLAIN, false>(a, st);
}
