// srk_misc.hip -- bandwidth-bound helpers for gfx950: weight packing, standalone PixelShuffle,
// NCHW<->NHWC boundary copies, SumPool2d.  All are one-pass, coalesced on the side that matters.
#include "srk_internal.h"
#include <string.h>

namespace {

// ---------------------------------------------------------------- weight packing
// work item = one float4 of dst: (q_local, tap, h, m) -> 4 consecutive k.
// fmt 3 ("wino"): 12 taps = 3 kernel rows x 4 Winograd F(2,3) positions along the kernel's column axis,
//   u0 = w0, u1 = (w0 + w1 + w2)/2, u2 = (w0 - w1 + w2)/2, u3 = w2   (w_s = the row's three column taps)
// The nine taps of (dst output m, dst input k = k_off + kr) in CONV-WINDOW order (w[3u + v] multiplies the input pixel at offset
// (u - 1, v - 1) of the output pixel), for the three kinds of entry:
//   transpose 0 (forward)        W[m'][c_begin + kr][u][v]
//   transpose 1 (data gradient)  W[kr'][c_begin + m][2 - u][2 - v]                  (input / output swapped, taps flipped)
//   transpose 2 (data gradient of a STRIDE-2 conv as a stride-1 conv on dy whose 4 x Cin outputs are PixelShuffled into dx:
//                output m = (2a + b) Cin + c is dx channel c at pixel parity (a, b); dx[2i+a][2j+b] only sees dy[i + di][j + dj] with
//                di, dj in {0, 1}: a = 0 -> kernel row 1 at di = 0; a = 1 -> row 2 at di = 0 and row 0 at di = 1; same for columns)
//                W[kr][c_begin + c][r(a, u)][s(b, v)]  or 0 where the window position carries no tap
// x' = ps ? 4 (x % (Cout_src / 4)) + x / (Cout_src / 4) : x  maps packed PixelShuffle order to OIHW rows (transpose 0 / 1).
__device__ __forceinline__ void load_w9(const srk_pack_entry& e, int m, int kr, float (&w)[9]) {
  const int Cps = e.src_cout >> 2;
  if (e.transpose == 2) {
    const int cin = e.M >> 2;                      // dx channels
    const int g = m / cin, c = m - g * cin, pa = g >> 1, pb = g & 1;
    const float* w9 = e.src + ((long)kr * e.src_cin + e.c_begin + c) * 9;
#pragma unroll
    for (int u = 0; u < 3; ++u)
#pragma unroll
      for (int v = 0; v < 3; ++v) {
        const int r = pa == 0 ? (u == 1 ? 1 : -1) : (u == 1 ? 2 : (u == 2 ? 0 : -1));
        const int sc = pb == 0 ? (v == 1 ? 1 : -1) : (v == 1 ? 2 : (v == 2 ? 0 : -1));
        w[3 * u + v] = (r >= 0 && sc >= 0) ? w9[3 * r + sc] : 0.f;
      }
    return;
  }
  if (!e.transpose) {
    int o = m;
    if (e.ps) o = 4 * (m % Cps) + m / Cps;
    const float* w9 = e.src + ((long)o * e.src_cin + e.c_begin + kr) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = w9[t];
  } else {
    int o = kr;
    if (e.ps) o = 4 * (kr % Cps) + kr / Cps;
    const float* w9 = e.src + ((long)o * e.src_cin + e.c_begin + m) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = w9[8 - t];
  }
}

// fmt 6 (F(2x4, 3x3)): work item = (m, h, row position rp, q_local) -> the six column positions of 4 consecutive k: the nine taps
// of a channel pair are read once per row position instead of once per transformed tap.
__device__ __forceinline__ void pack_item6(const srk_pack_entry& e, long t) {
  const int Mp = (e.M + 31) & ~31;
  const int m = (int)(t % Mp); t /= Mp;
  const int h = (int)(t & 1); t >>= 1;
  const int rp = (int)(t & 3); t >>= 2;
  const int q = (e.k_off >> 3) + (int)t;
  float v[6][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 8 * q + 4 * h + j, kr = k - e.k_off;
    float g[3] = {0.f, 0.f, 0.f};
    if (m < e.M && kr >= 0 && kr < e.k_len) {
      float w[9];
      load_w9(e, m, kr, w);
#pragma unroll
      for (int b = 0; b < 3; ++b) {
        const float c0 = w[b], c1 = w[3 + b], c2 = w[6 + b];       // column tap b of window rows 0, 1, 2
        g[b] = e.scale * (rp == 0 ? c0 : rp == 1 ? 0.5f * ((c0 + c1) + c2) : rp == 2 ? 0.5f * ((c0 - c1) + c2) : c2);
      }
    }
    const float w0 = g[0], w1 = g[1], w2 = g[2];
    v[0][j] = 0.25f * w0;
    v[1][j] = (-1.f / 6.f) * ((w0 + w1) + w2);
    v[2][j] = (-1.f / 6.f) * ((w0 - w1) + w2);
    v[3][j] = (w0 * (1.f / 24.f) + w1 * (1.f / 12.f)) + w2 * (1.f / 6.f);
    v[4][j] = (w0 * (1.f / 24.f) - w1 * (1.f / 12.f)) + w2 * (1.f / 6.f);
    v[5][j] = w2;
  }
  // [q][channel pair][tap = 6 rp + p][h][Mp] float2
  float2* d2 = reinterpret_cast<float2*>(e.dst);
#pragma unroll
  for (int p6 = 0; p6 < 6; ++p6) {
    const int tap = 6 * rp + p6;
    d2[((((long)q * 2 + 0) * 24 + tap) * 2 + h) * Mp + m] = make_float2(v[p6][0], v[p6][1]);
    d2[((((long)q * 2 + 1) * 24 + tap) * 2 + h) * Mp + m] = make_float2(v[p6][2], v[p6][3]);
  }
}

__device__ __forceinline__ void pack_item(const srk_pack_entry& e, long t) {
  const int Mp = (e.M + 31) & ~31;
  const int m = (int)(t % Mp); t /= Mp;
  const int h = (int)(t & 1); t >>= 1;
  const int ntap = e.fmt == 3 ? 12 : (e.fmt == 5 ? 18 : 9);
  const int tap = (int)(t % ntap); t /= ntap;
  const int ql = (int)t;                      // chunk index relative to k_off/8
  const int q = (e.k_off >> 3) + ql;
  float v[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = 8 * q + 4 * h + j;          // dst k
    const int kr = k - e.k_off;               // k relative to this entry
    float val = 0.f;
    if (m < e.M && kr >= 0 && kr < e.k_len) {
      float w[9];
      load_w9(e, m, kr, w);
      if (e.fmt == 5) {
        // F(4,3): 18 taps = 3 window rows x 6 positions, u = G w
        const int r = tap / 6, p = tap - 6 * r;
        const float w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        val = p == 0 ? 0.25f * w0
            : p == 1 ? (-1.f / 6.f) * ((w0 + w1) + w2)
            : p == 2 ? (-1.f / 6.f) * ((w0 - w1) + w2)
            : p == 3 ? (w0 * (1.f / 24.f) + w1 * (1.f / 12.f)) + w2 * (1.f / 6.f)
            : p == 4 ? (w0 * (1.f / 24.f) - w1 * (1.f / 12.f)) + w2 * (1.f / 6.f)
            : w2;
      } else if (e.fmt == 3) {
        const int r = tap >> 2, p = tap & 3;
        const float w0 = w[3 * r], w1 = w[3 * r + 1], w2 = w[3 * r + 2];
        val = p == 0 ? w0 : p == 1 ? 0.5f * ((w0 + w1) + w2) : p == 2 ? 0.5f * ((w0 - w1) + w2) : w2;
      } else {
        val = w[tap];
      }
      val *= e.scale;
    }
    v[j] = val;
  }
  float4* d = reinterpret_cast<float4*>(e.dst) + (((long)q * ntap + tap) * 2 + h) * Mp + m;
  *d = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ void pack_kernel(const srk_pack_entry* __restrict__ tab, int n, long total) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  // binary search entry with elem_begin <= gid
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].elem_begin <= gid) lo = mid; else hi = mid - 1;
  }
  const srk_pack_entry e = tab[lo];
  if (e.fmt == 6) pack_item6(e, gid - e.elem_begin); else pack_item(e, gid - e.elem_begin);
}

// one entry passed by value (the flat entry points below: no device-side table to upload)
__global__ void pack_one_kernel(const srk_pack_entry e, long total) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid < total) { if (e.fmt == 6) pack_item6(e, gid); else pack_item(e, gid); }
}

// bias[o] -> packed PixelShuffle order p = (2i+j)*Cout/4 + c  <-  o = 4c + 2i + j
__global__ void permute_bias_kernel(const float* __restrict__ b, float* __restrict__ bp, int Cout) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= Cout) return;
  const int Cps = Cout >> 2;
  bp[p] = b[4 * (p % Cps) + p / Cps];
}

// ---------------------------------------------------------------- pixel shuffle (standalone)
// x [N,H,W,4C] (channel = 4c + 2i + j)  ->  y [N,2H,2W,C]  (models.py:89; index map pinned bit-exact by G1).
// One thread per (input pixel, c): the four values of a 2x2 output patch are ONE aligned float4 of x (channels 4c .. 4c+3);
// consecutive lanes take consecutive c, so the read is a fully coalesced 16 B per lane and each of the four scatter stores
// writes 4 B per lane into one contiguous run of an output pixel (256 B per wave at C = 64).  The backward is the mirror:
// four gathered dwords, one 16-byte store.
__global__ void pixel_shuffle_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int N, int H, int W, int C) {
  const long total = (long)N * H * W * C;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    long t = gid;
    const int c = (int)(t % C); t /= C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const long n = t;
    const float4 v = *reinterpret_cast<const float4*>(x + gid * 4);
    float* o = y + (((n * 2 * H + 2 * h) * (2 * W)) + 2 * w) * C + c;
    const long row = (long)2 * W * C;
    o[0] = v.x; o[C] = v.y; o[row] = v.z; o[row + C] = v.w;
  }
}
__global__ void pixel_shuffle_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int N, int H, int W, int C) {
  const long total = (long)N * H * W * C;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    long t = gid;
    const int c = (int)(t % C); t /= C;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const long n = t;
    const float* o = dy + (((n * 2 * H + 2 * h) * (2 * W)) + 2 * w) * C + c;
    const long row = (long)2 * W * C;
    *reinterpret_cast<float4*>(dx + gid * 4) = make_float4(o[0], o[C], o[row], o[row + C]);
  }
}

// ---------------------------------------------------------------- layout
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int ldc, int coff, int N, int C, int H, int W) {
  const long total = (long)N * C * H * W;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    long t = gid;   // iterate in NHWC order of the destination
    const int c = (int)(t % C); t /= C;
    const long pix = t % ((long)H * W); t /= ((long)H * W);
    const int n = (int)t;
    y[((long)n * H * W + pix) * ldc + coff + c] = x[((long)n * C + c) * H * W + pix];
  }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, int ldc, int coff, float* __restrict__ y, int N, int C, int H, int W) {
  const long total = (long)N * C * H * W;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    long t = gid;   // NCHW order of the destination
    const long pix = t % ((long)H * W); t /= ((long)H * W);
    const int c = (int)(t % C); t /= C;
    const int n = (int)t;
    y[gid] = x[((long)n * H * W + pix) * ldc + coff + c];
  }
}

// ---------------------------------------------------------------- sum pool (planes = N*C)
__global__ void sum_pool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int NC, int H, int W, int k) {
  const int OH = H / k, OW = W / k;
  const long total = (long)NC * OH * OW;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    long t = gid;
    const int ow = (int)(t % OW); t /= OW;
    const int oh = (int)(t % OH); t /= OH;
    const float* p = x + ((long)t * H + oh * k) * W + ow * k;
    float s = 0.f;
    for (int i = 0; i < k; ++i)
      for (int j = 0; j < k; ++j) s += p[i * W + j];
    // k*k * avg_pool (models.py:305): avg = s / (k*k), times k*k
    y[gid] = (float)(k * k) * (s / (float)(k * k));
  }
}
__global__ void sum_pool_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int NC, int H, int W, int k) {
  const int OH = H / k, OW = W / k;
  const long total = (long)NC * H * W;
  for (long gid = (long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long)gridDim.x * blockDim.x) {
    long t = gid;
    const int w = (int)(t % W); t /= W;
    const int h = (int)(t % H); t /= H;
    const int oh = h / k, ow = w / k;
    dx[gid] = (oh < OH && ow < OW) ? dy[((long)t * OH + oh) * OW + ow] : 0.f;
  }
}

inline unsigned grid_for(long total, int block = 256) {
  long g = (total + block - 1) / block;
  if (g > 256L * 16) g = 256L * 16;
  if (g < 1) g = 1;
  return (unsigned)g;
}

}  // namespace

extern "C" size_t srk_packed_floats(int K, int M) {
  return (size_t)srk_div_up(K, 16) * 2 * 9 * 2 * srk_round_up(M, 32) * 4;
}

extern "C" size_t srk_packed_floats_wino(int K, int M) {
  return (size_t)srk_div_up(K, 16) * 2 * 12 * 2 * srk_round_up(M, 32) * 4;
}

extern "C" size_t srk_packed_floats_wino4(int K, int M) {
  return (size_t)srk_div_up(K, 16) * 2 * 18 * 2 * srk_round_up(M, 32) * 4;
}

extern "C" size_t srk_packed_floats_wino42(int K, int M) {
  // + one channel pair (24 positions x 2 k-halves x Mp float2) of padding: the wino42 main loop prefetches pair 0 of chunk
  // q + 1 unconditionally (branch-free), i.e. of chunk nq behind the last one, through the SCALAR offset of its buffer loads,
  // which the range check of a raw buffer does not cover.  The values are never used; the bytes must be mapped.
  return (size_t)srk_div_up(K, 16) * 2 * 24 * 2 * srk_round_up(M, 32) * 4 + (size_t)24 * 2 * srk_round_up(M, 32) * 2;
}

extern "C" size_t srk_packed_floats_h16(int K, int M) {
  return (size_t)srk_div_up(K, 16) * 9 * 2 * srk_round_up(M, 64) * 4;     // 16 bytes per (k-half, tap, output) slot
}

extern "C" int srk_pack_plan(srk_pack_entry* e, int n, int64_t* total) {
  if (!e || !total || n <= 0) return SRK_ERR_BAD_ARG;
  int64_t acc = 0;
  for (int i = 0; i < n; ++i) {
    if (!e[i].src || !e[i].dst || e[i].M <= 0 || e[i].k_len <= 0 || (e[i].k_off & 7)) return SRK_ERR_BAD_ARG;
    const bool f16b = e[i].fmt == 1 || e[i].fmt == 7 || e[i].fmt == 8;       // 16-channel chunks of 16-bit elements
    if (e[i].transpose < 0 || e[i].transpose > 2 || (e[i].transpose == 2 && ((e[i].M & 3) || e[i].ps || f16b))) return SRK_ERR_BAD_ARG;
    if (e[i].ps && (e[i].src_cout & 3)) return SRK_ERR_BAD_ARG;
    if (e[i].fmt != e[0].fmt || (e[i].fmt != 0 && e[i].fmt != 1 && e[i].fmt != 3 && e[i].fmt != 5 && e[i].fmt != 6 && e[i].fmt != 7 && e[i].fmt != 8)) return SRK_ERR_BAD_ARG;
    if (f16b && (e[i].k_off & 15)) return SRK_ERR_BAD_ARG;
    e[i].elem_begin = acc;
    // chunks covered: the last entry of a dst owns the zero-filled tail of the final chunk
    const int ck = f16b ? 16 : 8;
    int k_end = e[i].k_off + e[i].k_len;
    int nq = srk_div_up(k_end, ck) - e[i].k_off / ck;
    // fmt 0 with K not a multiple of 16: also zero the second half of the last 16-chunk (buffers are sized for 16)
    if (e[i].fmt == 7 || e[i].fmt == 8) { acc += (int64_t)nq * 9 * 2 * srk_round_up(e[i].M, 64); continue; }
    acc += (int64_t)nq * (e[i].fmt == 3 ? 12 : (e[i].fmt == 5 ? 18 : (e[i].fmt == 6 ? 4 : 9))) * 2 * srk_round_up(e[i].M, 32);   // (fmt 6: one item per row position)
  }
  *total = acc;
  return SRK_OK;
}

int srk_launch_pack_h16(const srk_pack_entry* dev, int n, int64_t total, int fmt, hipStream_t st);
extern "C" int srk_pack_weights_h16(const srk_pack_entry* dev, int n, int64_t total, int fmt, void* stream) {
  if (!dev || n <= 0 || total <= 0 || (fmt != 7 && fmt != 8)) return SRK_ERR_BAD_ARG;
  return srk_launch_pack_h16(dev, n, total, fmt, (hipStream_t)stream);
}

extern "C" int srk_pack_weights(const srk_pack_entry* dev, int n, int64_t total, void* stream) {
  if (!dev || n <= 0 || total <= 0) return SRK_ERR_BAD_ARG;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dev, n, (long)total);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

int srk_launch_pack_bf16x3(const srk_pack_entry* dev, int n, int64_t total, hipStream_t st);
extern "C" int srk_pack_weights_bf16x3(const srk_pack_entry* dev, int n, int64_t total, void* stream) {
  if (!dev || n <= 0 || total <= 0) return SRK_ERR_BAD_ARG;
  return srk_launch_pack_bf16x3(dev, n, total, (hipStream_t)stream);
}

extern "C" int srk_pixel_shuffle_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream) {
  if (!x || !y || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SRK_ERR_BAD_ARG;
  if (((uintptr_t)x) & 15) return SRK_ERR_ALIGNMENT;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(pixel_shuffle_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, N, H, W, C);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_pixel_shuffle_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream) {
  if (!dy || !dx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return SRK_ERR_BAD_ARG;
  if (((uintptr_t)dx) & 15) return SRK_ERR_ALIGNMENT;
  const long total = (long)N * H * W * C;
  hipLaunchKernelGGL(pixel_shuffle_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, N, H, W, C);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_nchw_to_nhwc(const float* x, float* y, int y_ldc, int y_coff, int N, int C, int H, int W, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || y_ldc < C) return SRK_ERR_BAD_ARG;
  const long total = (long)N * C * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, y_ldc, y_coff, N, C, H, W);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_nhwc_to_nchw(const float* x, int x_ldc, int x_coff, float* y, int N, int C, int H, int W, void* stream) {
  if (!x || !y || N <= 0 || C <= 0 || H <= 0 || W <= 0 || x_ldc < C) return SRK_ERR_BAD_ARG;
  const long total = (long)N * C * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, x_ldc, x_coff, y, N, C, H, W);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_sum_pool_fwd(const float* x, float* y, int NC, int H, int W, int k, void* stream) {
  if (!x || !y || NC <= 0 || H <= 0 || W <= 0 || k <= 0 || H < k || W < k) return SRK_ERR_BAD_ARG;
  const long total = (long)NC * (H / k) * (W / k);
  hipLaunchKernelGGL(sum_pool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, NC, H, W, k);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_sum_pool_bwd(const float* dy, float* dx, int NC, int H, int W, int k, void* stream) {
  if (!dy || !dx || NC <= 0 || H <= 0 || W <= 0 || k <= 0 || H < k || W < k) return SRK_ERR_BAD_ARG;
  const long total = (long)NC * H * W;
  hipLaunchKernelGGL(sum_pool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, dy, dx, NC, H, W, k);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

// ---------------------------------------------------------------- flat (self-contained) entry points
// Canonical OIHW weights in, packed into the caller's workspace on the same stream, then the fused conv kernel.
namespace {
size_t flat_ws_floats(int K, int M) { return srk_packed_floats(K, M) + (size_t)srk_round_up(M, 32); }

int pack_one(const float* w, float* dst, int Cout, int Cin, bool transpose, bool ps, hipStream_t st) {
  srk_pack_entry e;
  memset(&e, 0, sizeof(e));
  e.src = w; e.dst = dst; e.src_cout = Cout; e.src_cin = Cin; e.transpose = transpose ? 1 : 0; e.c_begin = 0;
  e.M = transpose ? Cin : Cout; e.k_off = 0; e.k_len = transpose ? Cout : Cin; e.K_total = e.k_len;
  e.ps = ps ? 1 : 0; e.scale = 1.f; e.fmt = 0;
  int64_t total = 0;
  int rc = srk_pack_plan(&e, 1, &total);
  if (rc) return rc;
  hipLaunchKernelGGL(pack_one_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, e, (long)total);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
}  // namespace

extern "C" int srk_workspace_bytes(int op, int N, int H, int W, int Cin, int Cout, int dtype, size_t* out) {
  if (!out || N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SRK_ERR_BAD_ARG;
  if (dtype != 0) return SRK_ERR_UNSUPPORTED;
  switch (op) {
    case SRK_OP_CONV_FWD: *out = flat_ws_floats(Cin, Cout) * sizeof(float); return SRK_OK;
    case SRK_OP_CONV_DGRAD: *out = flat_ws_floats(Cout, Cin) * sizeof(float); return SRK_OK;
    case SRK_OP_CONV_WGRAD: {
      // upper bound over stride 1 / 2 and both dy modes for this geometry
      size_t best = 0;
      for (int stride = 1; stride <= 2; ++stride) {
        srk_wgrad_args a;
        memset(&a, 0, sizeof(a));
        a.N = N; a.H = H; a.W = W; a.OH = srk_div_up(H, stride); a.OW = srk_div_up(W, stride); a.Cin = Cin; a.Cout = Cout;
        a.stride = stride; a.x = a.dy = (const float*)16; a.dw = (float*)16; a.x_ldc = Cin; a.dy_ldc = Cout; a.in_slope = 1.f;
        size_t b = 0;
        int rc = srk_conv3x3_wgrad_workspace(&a, &b);
        if (rc) return rc;
        if (b > best) best = b;
      }
      *out = best;
      return SRK_OK;
    }
    default: return SRK_ERR_BAD_ARG;
  }
}

extern "C" int srk_conv3x3_fwd(const void* x, int ldc_in, int c_in_off, int Cin, const void* w, const void* bias, void* y,
                               int ldc_out, int c_out_off, int Cout, int N, int H, int W, int stride, float lrelu_slope,
                               const void* residual, float res_scale, int pixel_shuffle_r, int dtype, void* workspace,
                               size_t ws_bytes, void* stream) {
  if (!x || !w || !y || !workspace) return SRK_ERR_BAD_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SRK_ERR_BAD_ARG;
  if (dtype != 0 || (stride != 1 && stride != 2) || (pixel_shuffle_r != 0 && pixel_shuffle_r != 2)) return SRK_ERR_UNSUPPORTED;
  if (pixel_shuffle_r == 2 && ((Cout & 3) || stride != 1 || residual)) return SRK_ERR_UNSUPPORTED;
  if (residual && lrelu_slope != 1.f) return SRK_ERR_UNSUPPORTED;      // models.py:40: the residual conv is linear
  if (ws_bytes < flat_ws_floats(Cin, Cout) * sizeof(float)) return SRK_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* wp = (float*)workspace;
  float* bp = wp + srk_packed_floats(Cin, Cout);
  int rc = pack_one((const float*)w, wp, Cout, Cin, false, pixel_shuffle_r == 2, st);
  if (rc) return rc;
  const float* b = (const float*)bias;
  if (b && pixel_shuffle_r == 2) {
    hipLaunchKernelGGL(permute_bias_kernel, dim3(srk_div_up(Cout, 256)), dim3(256), 0, st, b, bp, Cout);
    SRK_CHECK_LAUNCH();
    b = bp;
  }
  srk_conv_args a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.H = H; a.W = W; a.OH = srk_div_up(H, stride); a.OW = srk_div_up(W, stride); a.Cin = Cin; a.Cout = Cout;
  a.stride = stride; a.in_mode = SRK_IN_PLAIN; a.ps_out = pixel_shuffle_r == 2;
  a.x = (const float*)x; a.x_ldc = ldc_in; a.x_coff = c_in_off; a.in_slope = 1.f;
  a.wp = wp; a.bias = b;
  a.y = (float*)y; a.y_ldc = ldc_out; a.y_coff = c_out_off;
  a.alpha = residual ? res_scale : 1.f;
  if (residual) { a.r1 = (const float*)residual; a.r1_ldc = Cout; a.r1_coff = 0; a.beta1 = 1.f; }
  a.slope = lrelu_slope; a.mask_slope = 1.f;
  return srk_conv3x3(&a, stream);
}

extern "C" int srk_conv3x3_dgrad(const void* dy, int ldc_dy, int c_dy_off, int Cout, const void* w, void* dx, int ldc_dx,
                                 int c_dx_off, int Cin, int N, int H, int W, int stride, int pixel_shuffle_r, int dtype,
                                 void* workspace, size_t ws_bytes, void* stream) {
  if (!dy || !w || !dx || !workspace) return SRK_ERR_BAD_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SRK_ERR_BAD_ARG;
  if (dtype != 0 || (stride != 1 && stride != 2) || (pixel_shuffle_r != 0 && pixel_shuffle_r != 2)) return SRK_ERR_UNSUPPORTED;
  if (pixel_shuffle_r == 2 && ((Cout & 3) || stride != 1)) return SRK_ERR_UNSUPPORTED;
  if (ws_bytes < flat_ws_floats(Cout, Cin) * sizeof(float)) return SRK_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* wpt = (float*)workspace;
  int rc = pack_one((const float*)w, wpt, Cout, Cin, true, pixel_shuffle_r == 2, st);
  if (rc) return rc;
  srk_conv_args a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.Cin = Cout; a.Cout = Cin; a.stride = 1;
  if (stride == 2) { a.H = srk_div_up(H, 2); a.W = srk_div_up(W, 2); a.in_mode = SRK_IN_ZERO_UPSAMPLE; }
  else { a.H = H; a.W = W; a.in_mode = pixel_shuffle_r == 2 ? SRK_IN_UNSHUFFLE : SRK_IN_PLAIN; }
  a.OH = H; a.OW = W;
  a.x = (const float*)dy; a.x_ldc = ldc_dy; a.x_coff = c_dy_off; a.in_slope = 1.f;
  a.wp = wpt; a.y = (float*)dx; a.y_ldc = ldc_dx; a.y_coff = c_dx_off;
  a.alpha = 1.f; a.slope = 1.f; a.mask_slope = 1.f;
  return srk_conv3x3(&a, stream);
}

extern "C" int srk_conv3x3_wgrad_flat(const void* x, int ldc_in, int c_in_off, int Cin, const void* dy, int ldc_dy, int c_dy_off, int Cout,
                                      void* dw, void* dbias, int N, int H, int W, int stride, float scale, int accumulate,
                                      int pixel_shuffle_r, int dtype, void* workspace, size_t ws_bytes, void* stream) {
  if (!x || !dy || !dw || !workspace) return SRK_ERR_BAD_ARG;
  if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return SRK_ERR_BAD_ARG;
  if (dtype != 0 || (stride != 1 && stride != 2) || (pixel_shuffle_r != 0 && pixel_shuffle_r != 2)) return SRK_ERR_UNSUPPORTED;
  if (pixel_shuffle_r == 2 && stride != 1) return SRK_ERR_UNSUPPORTED;
  srk_wgrad_args a;
  memset(&a, 0, sizeof(a));
  a.N = N; a.H = H; a.W = W; a.OH = srk_div_up(H, stride); a.OW = srk_div_up(W, stride); a.Cin = Cin; a.Cout = Cout; a.stride = stride;
  a.dy_mode = pixel_shuffle_r == 2 ? SRK_IN_UNSHUFFLE : SRK_IN_PLAIN;
  a.x = (const float*)x; a.x_ldc = ldc_in; a.x_coff = c_in_off; a.in_slope = 1.f;
  a.dy = (const float*)dy; a.dy_ldc = ldc_dy; a.dy_coff = c_dy_off;
  a.dw = (float*)dw; a.db = (float*)dbias; a.scale = scale; a.accumulate = accumulate;
  a.workspace = workspace; a.workspace_bytes = ws_bytes;
  size_t need = 0;
  int rc = srk_conv3x3_wgrad_workspace(&a, &need);
  if (rc) return rc;
  if (ws_bytes < need) return SRK_ERR_WORKSPACE;
  return srk_conv3x3_wgrad(&a, stream);
}

extern "C" const char* srk_strerror(int s) {
  switch (s) {
    case SRK_OK: return "ok";
    case SRK_ERR_BAD_ARG: return "bad argument (null pointer or non-positive dimension)";
    case SRK_ERR_UNSUPPORTED: return "unsupported mode combination";
    case SRK_ERR_ALIGNMENT: return "view is not 16-byte aligned for the vector path";
    case SRK_ERR_WORKSPACE: return "workspace missing or too small";
    case SRK_ERR_LAUNCH: return "kernel launch failed";
    case SRK_ERR_CHAIN_TIMEOUT: return "a conv3x3 chain launch gave up (grid not resident within its bound, or a flag wait timed out): call srk_chain_recover() and repeat the iteration";
    default: return "unknown srk status";
  }
}
extern "C" int srk_version(void) { return SRK_VERSION; }
