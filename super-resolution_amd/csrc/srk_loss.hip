// srk_loss.hip -- the generator's optional physics loss heads (esrgan.py:522-547) as fused, bandwidth-bound kernels
// for gfx950.  The reference builds each head from 3-6 HR-sized ATen ops (sigmoid, sub, abs, split/cat, boolean
// indexing, sum ...), i.e. 3-6 passes over N x C x 256 x 256 tensors plus their autograd mirrors; here every head is
// ONE read pass forward (reduced on the fly) and ONE read+write pass backward.
//
//   soft count   softgreater(x, val, sigma).sum(1).sum(1).sum(1)            utils.py:259-261, esrgan.py:523
//   mask L1      L1Loss(nnz_mask(a), nnz_mask(b))                            utils.py:271-272, esrgan.py:527-529
//   hitogram     get_hitogram(t, factor, threshold, sig)                     utils.py:264-268, esrgan.py:544-545
//   soft hist    DiffableHistogram(binedges)(x[x > 0])                       models.py:308-342, esrgan.py:533-536
//   sigmoid      softgreater / nnz_mask as stand-alone elementwise functions utils.py:259-261,271-272
//
// Reductions are two-stage and fixed-order (per-workgroup partials in a caller workspace, then one workgroup sums
// them): deterministic, no float atomics.  All tensors are dense fp32; n / per-image counts are element counts.
#include "srk_internal.h"

namespace {

constexpr int LT = 256;            // threads per workgroup
constexpr int MAX_PART = 1024;     // stage-1 workgroups per output row

// v_exp_f32 + v_rcp_f32: ~1e-6 relative error in the transition region, exact 0 / 1 in saturation (the kernels are
// issue-bound on this expression, not on HBM, once more than a few sigmoids are evaluated per element)
__device__ __forceinline__ float sigmoidf(float z) { return __builtin_amdgcn_rcpf(1.f + __expf(-z)); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s);
  return v;
}

// block-wide sum of one value per thread, result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// ---------------------------------------------------------------- elementwise sigmoid(scale * x + shift)
// All elementwise loops below run on float4 when `vec` (16-byte aligned bases, count % 4 == 0): n4 = count / 4.
__global__ void sigmoid_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float scale, float shift, int vec) {
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    float4* y4 = reinterpret_cast<float4*>(y);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * LT) {
      const float4 v = x4[i];
      y4[i] = make_float4(sigmoidf(scale * v.x + shift), sigmoidf(scale * v.y + shift), sigmoidf(scale * v.z + shift), sigmoidf(scale * v.w + shift));
    }
    return;
  }
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) y[i] = sigmoidf(scale * x[i] + shift);
}
// out = g * lrelu_s'(x): the scaling the gradient penalty's double backward applies between two conv nodes (ops.py ConvDgrad.backward;
// torch.where(x > 0, g, g * s) is three passes and a bool tensor, this is one: 2 reads + 1 write)
__global__ void lrelu_grad_mul_kernel(const float* __restrict__ x, const float* __restrict__ g, float* __restrict__ out, long n, float slope, int vec) {
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(x);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* o4 = reinterpret_cast<float4*>(out);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * LT) {
      const float4 a = x4[i], b = g4[i];
      o4[i] = make_float4(a.x > 0.f ? b.x : b.x * slope, a.y > 0.f ? b.y : b.y * slope, a.z > 0.f ? b.z : b.z * slope, a.w > 0.f ? b.w : b.w * slope);
    }
    return;
  }
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) out[i] = x[i] > 0.f ? g[i] : g[i] * slope;
}

__global__ void sigmoid_bwd_kernel(const float* __restrict__ y, const float* __restrict__ gy, float* __restrict__ dx, long n, float scale, int vec) {
  if (vec) {
    const float4* y4 = reinterpret_cast<const float4*>(y);
    const float4* g4 = reinterpret_cast<const float4*>(gy);
    float4* d4 = reinterpret_cast<float4*>(dx);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * LT) {
      const float4 s = y4[i], g = g4[i];
      d4[i] = make_float4(g.x * scale * s.x * (1.f - s.x), g.y * scale * s.y * (1.f - s.y), g.z * scale * s.z * (1.f - s.z), g.w * scale * s.w * (1.f - s.w));
    }
    return;
  }
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) {
    const float s = y[i];
    dx[i] = gy[i] * scale * s * (1.f - s);
  }
}

// ---------------------------------------------------------------- soft count: out[b] = sum_i sigmoid(sigma*(x[b,i]-val))
// grid (P, B); stage 2: grid B.  hard != 0: out[b] = count(x[b,i] > val)  (the target of esrgan.py:524)
__global__ void soft_count_part_kernel(const float* __restrict__ x, float* __restrict__ part, long per, float sigma, float val, int hard, int vec) {
  __shared__ float red[4];
  const float* xb = x + (long)blockIdx.y * per;
  float acc = 0.f;
  auto one = [&](float v) { return hard ? (v > val ? 1.f : 0.f) : sigmoidf(sigma * (v - val)); };
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(xb);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (per >> 2); i += (long)gridDim.x * LT) {
      const float4 v = x4[i];
      acc += (one(v.x) + one(v.y)) + (one(v.z) + one(v.w));
    }
  } else {
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < per; i += (long)gridDim.x * LT) acc += one(xb[i]);
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) part[(long)blockIdx.y * gridDim.x + blockIdx.x] = tot;
}
// sums `np` partials in fixed order; grid (rows, K): out[row*K + k] = scale * sum_p part[(row*np + p)*K + k]
__global__ void sum_partials_kernel(const float* __restrict__ part, float* __restrict__ out, int np, int K, float scale) {
  __shared__ float red[4];
  const int row = blockIdx.x, k = blockIdx.y;
  float acc = 0.f;
  for (int p = threadIdx.x; p < np; p += LT) acc += part[((long)row * np + p) * K + k];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) out[(long)row * K + k] = scale * tot;
}
// grid (blocks per image, B)
__global__ void soft_count_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gout, float* __restrict__ dx, long per,
                                      float sigma, float val, int vec) {
  const float g = gout[blockIdx.y] * sigma;
  const float* xb = x + (long)blockIdx.y * per;
  float* db = dx + (long)blockIdx.y * per;
  auto one = [&](float v) { const float s = sigmoidf(sigma * (v - val)); return g * s * (1.f - s); };
  if (vec) {
    const float4* x4 = reinterpret_cast<const float4*>(xb);
    float4* d4 = reinterpret_cast<float4*>(db);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (per >> 2); i += (long)gridDim.x * LT) {
      const float4 v = x4[i];
      d4[i] = make_float4(one(v.x), one(v.y), one(v.z), one(v.w));
    }
    return;
  }
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < per; i += (long)gridDim.x * LT) db[i] = one(xb[i]);
}

// ---------------------------------------------------------------- mask L1: mean |sigmoid(sigma a) - sigmoid(sigma b)|
__global__ void mask_l1_part_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ part, long n, float sigma, int vec) {
  __shared__ float red[4];
  float acc = 0.f;
  auto one = [&](float u, float v) { return fabsf(sigmoidf(sigma * u) - sigmoidf(sigma * v)); };
  if (vec) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * LT) {
      const float4 u = a4[i], v = b4[i];
      acc += (one(u.x, v.x) + one(u.y, v.y)) + (one(u.z, v.z) + one(u.w, v.w));
    }
  } else {
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) acc += one(a[i], b[i]);
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot;
}
// d/da: gout/n * sign(sa - sb) * sigma * sa (1 - sa)     (torch's L1Loss backward uses sign(), 0 at equality)
__global__ void mask_l1_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ gout,
                                   float* __restrict__ da, long n, float sigma, int vec) {
  const float g = gout[0] / (float)n * sigma;
  auto one = [&](float u, float v) {
    const float sa = sigmoidf(sigma * u), sb = sigmoidf(sigma * v);
    const float d = sa - sb;
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    return g * sg * sa * (1.f - sa);
  };
  if (vec) {
    const float4* a4 = reinterpret_cast<const float4*>(a);
    const float4* b4 = reinterpret_cast<const float4*>(b);
    float4* d4 = reinterpret_cast<float4*>(da);
    for (long i = (long)blockIdx.x * LT + threadIdx.x; i < (n >> 2); i += (long)gridDim.x * LT) {
      const float4 u = a4[i], v = b4[i];
      d4[i] = make_float4(one(u.x, v.x), one(u.y, v.y), one(u.z, v.z), one(u.w, v.w));
    }
    return;
  }
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) da[i] = one(a[i], b[i]);
}

// ---------------------------------------------------------------- hitogram
// t [BC][H][W] -> out[i*F + j] = mean over (bc, p, q) of  sigmoid(sig * (t[bc, F p + i, F q + j] - thr))   (sig <= 0: t itself)
// One workgroup per group of F rows (grid-stride over bc * H/F groups); thread tx owns columns tx, tx+256, ... so its
// column phase j = tx % F is fixed (256 % F == 0) and row phase i is the unrolled loop index: F register accumulators.
template <int F>
__global__ void hitogram_part_kernel(const float* __restrict__ t, float* __restrict__ part, int BC, int H, int W, float thr, float sig) {
  __shared__ float red[4][F * F];
  float acc[F];
#pragma unroll
  for (int i = 0; i < F; ++i) acc[i] = 0.f;
  const int groups = BC * (H / F);
  for (int g = blockIdx.x; g < groups; g += gridDim.x) {
    const int bc = g / (H / F), p = g - bc * (H / F);
    const float* base = t + ((long)bc * H + (long)p * F) * W;
    for (int w = threadIdx.x; w < W; w += LT) {
#pragma unroll
      for (int i = 0; i < F; ++i) {
        const float v = base[(long)i * W + w];
        acc[i] += sig > 0.f ? sigmoidf(sig * (v - thr)) : v;
      }
    }
  }
  // lanes with equal (lane % F) hold the same column phase: xor-shuffle over offsets >= F
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int i = 0; i < F; ++i) {
    float v = acc[i];
#pragma unroll
    for (int s = 32; s >= F; s >>= 1) v += __shfl_xor(v, s);
    if (lane < F) red[wv][i * F + lane] = v;
  }
  __syncthreads();
  if (threadIdx.x < F * F) {
    const int k = threadIdx.x;
    part[(long)blockIdx.x * (F * F) + k] = (red[0][k] + red[1][k]) + (red[2][k] + red[3][k]);
  }
}
// one workgroup per image row (grid-stride over BC*H rows): the row phase h % F is workgroup-uniform
__global__ void hitogram_bwd_kernel(const float* __restrict__ t, const float* __restrict__ gout, float* __restrict__ dt, int rows, int H, int W,
                                    int F, float thr, float sig, float inv_count) {
  for (int r = blockIdx.x; r < rows; r += gridDim.x) {
    const int h = r % H;
    const float* grow = gout + (h % F) * F;
    const float* trow = t + (long)r * W;
    float* drow = dt + (long)r * W;
    for (int w = threadIdx.x; w < W; w += LT) {
      const float g = grow[w % F] * inv_count;
      if (sig > 0.f) {
        const float s = sigmoidf(sig * (trow[w] - thr));
        drow[w] = g * sig * s * (1.f - s);
      } else {
        drow[w] = g;
      }
    }
  }
}

// ---------------------------------------------------------------- soft histogram of the positive entries
// out[k] = sum_{x_i > 0} sigmoid(sigma (x_i - c_k + d_k/2)) - sigmoid(sigma (x_i - c_k - d_k/2)),  k < K <= KMAX
template <int KMAX>
__global__ void soft_hist_part_kernel(const float* __restrict__ x, long n, const float* __restrict__ centers, const float* __restrict__ delta,
                                      int K, float sigma, int positive_only, float* __restrict__ part) {
  __shared__ float red[4];
  __shared__ float cs[KMAX], ds[KMAX];
  if (threadIdx.x < KMAX) {
    cs[threadIdx.x] = threadIdx.x < K ? centers[threadIdx.x] : 0.f;
    ds[threadIdx.x] = threadIdx.x < K ? 0.5f * delta[threadIdx.x] : 0.f;
  }
  __syncthreads();
  float acc[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) acc[k] = 0.f;
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) {
    const float v = x[i];
    if (!positive_only || v > 0.f) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        if (k < K) {
          const float u = v - cs[k];
          acc[k] += sigmoidf(sigma * (u + ds[k])) - sigmoidf(sigma * (u - ds[k]));
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      const float tot = block_sum(acc[k], red);
      if (threadIdx.x == 0) part[(long)blockIdx.x * K + k] = tot;
    }
  }
}
__global__ void soft_hist_bwd_kernel(const float* __restrict__ x, long n, const float* __restrict__ centers, const float* __restrict__ delta,
                                     int K, float sigma, int positive_only, const float* __restrict__ gout, float* __restrict__ dx) {
  extern __shared__ float sh[];      // [K] centers, [K] half widths, [K] gout
  float* cs = sh; float* ds = sh + K; float* gs = sh + 2 * K;
  for (int k = threadIdx.x; k < K; k += LT) { cs[k] = centers[k]; ds[k] = 0.5f * delta[k]; gs[k] = gout[k]; }
  __syncthreads();
  for (long i = (long)blockIdx.x * LT + threadIdx.x; i < n; i += (long)gridDim.x * LT) {
    const float v = x[i];
    float d = 0.f;
    if (!positive_only || v > 0.f) {
      for (int k = 0; k < K; ++k) {
        const float u = v - cs[k];
        const float s1 = sigmoidf(sigma * (u + ds[k])), s2 = sigmoidf(sigma * (u - ds[k]));
        d += gs[k] * sigma * (s1 * (1.f - s1) - s2 * (1.f - s2));
      }
    }
    dx[i] = d;
  }
}

// ---------------------------------------------------------------- sparse jet decode (datasets.py:136-145 `extract`)
// rows [B][row_stride]: interleaved (pos_0, E_0, pos_1, E_1, ...), L pairs per event, zero-padded.  The reference walks
// the pairs in order, adds E_i to pixel (eta = pos % etaBins, phi = pos // etaBins) and stops at the first E_i == 0.
// One workgroup per event: the pair list is staged in LDS; entry i OWNS its pixel if no earlier valid entry hits it and
// then sums all later hits in list order -- the reference's summation order exactly, no atomics, bit-identical.
// A ThresholdImageCutter (datasets.py:170-175) is folded into the store (thr < 0: none).
__global__ void jet_extract_kernel(const float* __restrict__ rows, int L, int row_stride, int etaBins, int phiBins, float thr,
                                   float* __restrict__ out) {
  extern __shared__ float sh[];                 // [L] pixel index (as int bits), [L] energy
  int* pix = reinterpret_cast<int*>(sh);
  float* en = sh + L;
  __shared__ int first_zero;
  const int b = blockIdx.x;
  const float* row = rows + (long)b * row_stride;
  float* img = out + (long)b * etaBins * phiBins;
  if (threadIdx.x == 0) first_zero = L;
  for (int i = threadIdx.x; i < etaBins * phiBins; i += LT) img[i] = 0.f;
  __syncthreads();
  for (int i = threadIdx.x; i < L; i += LT) {
    const float pos = row[2 * i], e = row[2 * i + 1];
    const int ip = (int)pos;
    const int phi = ip / etaBins, eta = ip - phi * etaBins;
    pix[i] = (ip >= 0 && phi < phiBins) ? eta * phiBins + phi : -1;
    en[i] = e;
    if (e == 0.f) atomicMin(&first_zero, i);
  }
  __syncthreads();
  const int n = first_zero;                     // entries [0, n) are added (entry n adds 0 and ends the walk)
  for (int i = threadIdx.x; i < n; i += LT) {
    const int p = pix[i];
    if (p < 0) continue;
    bool owner = true;
    for (int j = 0; j < i; ++j) owner = owner && (pix[j] != p);
    if (!owner) continue;
    float acc = 0.f;
    for (int j = i; j < n; ++j) acc += (pix[j] == p) ? en[j] : 0.f;
    img[p] = (thr < 0.f || acc > thr) ? acc : 0.f;
  }
}

int parts_for(long n) {
  long p = (n + (long)LT * 8 - 1) / ((long)LT * 8);      // >= 8 elements per thread
  if (p < 1) p = 1;
  if (p > MAX_PART) p = MAX_PART;
  return (int)p;
}
bool vec_ok(const void* a, const void* b, const void* c, long n) {
  return (n % 4 == 0) && (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) == 0;
}
unsigned ew_grid(long n) {
  long g = (n + LT - 1) / LT;
  if (g > 8192) g = 8192;
  return (unsigned)(g < 1 ? 1 : g);
}

}  // namespace

extern "C" int srk_jet_extract(const float* rows, int B, int L, int row_stride, int etaBins, int phiBins, float threshold, float* out,
                               void* stream) {
  if (!rows || !out || B <= 0 || L <= 0 || etaBins <= 0 || phiBins <= 0 || row_stride < 2 * L) return SRK_ERR_BAD_ARG;
  if (L > 4096) return SRK_ERR_UNSUPPORTED;          // LDS staging of the pair list (32 KB); jets have O(100) constituents
  hipLaunchKernelGGL(jet_extract_kernel, dim3(B), dim3(LT), (size_t)L * 8, (hipStream_t)stream, rows, L, row_stride, etaBins, phiBins,
                     threshold, out);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_loss_workspace_bytes(size_t* out) {
  if (!out) return SRK_ERR_BAD_ARG;
  *out = (size_t)MAX_PART * 64 * sizeof(float);          // any head: <= MAX_PART partial rows of <= 64 values
  return SRK_OK;
}

extern "C" int srk_sigmoid_fwd(const float* x, float* y, long n, float scale, float shift, void* stream) {
  if (!x || !y || n <= 0) return SRK_ERR_BAD_ARG;
  const int vec = vec_ok(x, y, nullptr, n);
  hipLaunchKernelGGL(sigmoid_fwd_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(LT), 0, (hipStream_t)stream, x, y, n, scale, shift, vec);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_lrelu_grad_mul(const float* x, const float* g, float* out, long n, float slope, void* stream) {
  if (!x || !g || !out || n <= 0) return SRK_ERR_BAD_ARG;
  const int vec = vec_ok(x, g, out, n);
  hipLaunchKernelGGL(lrelu_grad_mul_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(LT), 0, (hipStream_t)stream, x, g, out, n, slope, vec);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_sigmoid_bwd(const float* y, const float* gy, float* dx, long n, float scale, void* stream) {
  if (!y || !gy || !dx || n <= 0) return SRK_ERR_BAD_ARG;
  const int vec = vec_ok(y, gy, dx, n);
  hipLaunchKernelGGL(sigmoid_bwd_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(LT), 0, (hipStream_t)stream, y, gy, dx, n, scale, vec);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_soft_count_fwd(const float* x, float* out, int B, long per_image, float sigma, float val, int hard, void* workspace,
                                  size_t ws_bytes, void* stream) {
  if (!x || !out || !workspace || B <= 0 || per_image <= 0) return SRK_ERR_BAD_ARG;
  int P = parts_for(per_image);
  while ((long)P * B > MAX_PART * 64 && P > 1) P >>= 1;
  if (ws_bytes < (size_t)P * B * sizeof(float)) return SRK_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  const int vec = vec_ok(x, nullptr, nullptr, per_image);
  hipLaunchKernelGGL(soft_count_part_kernel, dim3(P, B), dim3(LT), 0, st, x, (float*)workspace, per_image, sigma, val, hard, vec);
  SRK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(B, 1), dim3(LT), 0, st, (const float*)workspace, out, P, 1, 1.f);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_soft_count_bwd(const float* x, const float* gout, float* dx, int B, long per_image, float sigma, float val, void* stream) {
  if (!x || !gout || !dx || B <= 0 || per_image <= 0) return SRK_ERR_BAD_ARG;
  const int vec = vec_ok(x, dx, nullptr, per_image);
  const long work = vec ? per_image / 4 : per_image;
  long g = (work + LT - 1) / LT;
  if (g > 256) g = 256;
  hipLaunchKernelGGL(soft_count_bwd_kernel, dim3((unsigned)(g < 1 ? 1 : g), B), dim3(LT), 0, (hipStream_t)stream, x, gout, dx, per_image, sigma, val, vec);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_mask_l1_fwd(const float* a, const float* b, float* out, long n, float sigma, void* workspace, size_t ws_bytes, void* stream) {
  if (!a || !b || !out || !workspace || n <= 0) return SRK_ERR_BAD_ARG;
  const int P = parts_for(n);
  if (ws_bytes < (size_t)P * sizeof(float)) return SRK_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(mask_l1_part_kernel, dim3(P), dim3(LT), 0, st, a, b, (float*)workspace, n, sigma, (int)vec_ok(a, b, nullptr, n));
  SRK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1, 1), dim3(LT), 0, st, (const float*)workspace, out, P, 1, 1.f / (float)n);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_mask_l1_bwd(const float* a, const float* b, const float* gout, float* da, long n, float sigma, void* stream) {
  if (!a || !b || !gout || !da || n <= 0) return SRK_ERR_BAD_ARG;
  const int vec = vec_ok(a, b, da, n);
  hipLaunchKernelGGL(mask_l1_bwd_kernel, dim3(ew_grid(vec ? n / 4 : n)), dim3(LT), 0, (hipStream_t)stream, a, b, gout, da, n, sigma, vec);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_hitogram_fwd(const float* t, float* out, int BC, int H, int W, int factor, float thr, float sig, void* workspace,
                                size_t ws_bytes, void* stream) {
  if (!t || !out || !workspace || BC <= 0 || H <= 0 || W <= 0 || factor <= 0) return SRK_ERR_BAD_ARG;
  if ((H % factor) || (W % factor)) return SRK_ERR_BAD_ARG;          // torch.cat of ragged splits raises in the reference too
  if (factor != 1 && factor != 2 && factor != 4 && factor != 8) return SRK_ERR_UNSUPPORTED;
  const int groups = BC * (H / factor);
  const int P = groups < MAX_PART ? groups : MAX_PART;
  const int K = factor * factor;
  if (ws_bytes < (size_t)P * K * sizeof(float)) return SRK_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  switch (factor) {
    case 1: hipLaunchKernelGGL(hitogram_part_kernel<1>, dim3(P), dim3(LT), 0, st, t, part, BC, H, W, thr, sig); break;
    case 2: hipLaunchKernelGGL(hitogram_part_kernel<2>, dim3(P), dim3(LT), 0, st, t, part, BC, H, W, thr, sig); break;
    case 4: hipLaunchKernelGGL(hitogram_part_kernel<4>, dim3(P), dim3(LT), 0, st, t, part, BC, H, W, thr, sig); break;
    default: hipLaunchKernelGGL(hitogram_part_kernel<8>, dim3(P), dim3(LT), 0, st, t, part, BC, H, W, thr, sig); break;
  }
  SRK_CHECK_LAUNCH();
  const float inv = 1.f / ((float)BC * (float)(H / factor) * (float)(W / factor));
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1, K), dim3(LT), 0, st, (const float*)part, out, P, K, inv);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_hitogram_bwd(const float* t, const float* gout, float* dt, int BC, int H, int W, int factor, float thr, float sig, void* stream) {
  if (!t || !gout || !dt || BC <= 0 || H <= 0 || W <= 0 || factor <= 0 || (H % factor) || (W % factor)) return SRK_ERR_BAD_ARG;
  const int rows = BC * H;
  const float inv = 1.f / ((float)BC * (float)(H / factor) * (float)(W / factor));
  hipLaunchKernelGGL(hitogram_bwd_kernel, dim3(rows < 8192 ? rows : 8192), dim3(LT), 0, (hipStream_t)stream, t, gout, dt, rows, H, W, factor, thr, sig, inv);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_soft_hist_fwd(const float* x, long n, const float* centers, const float* delta, int K, float sigma, int positive_only,
                                 float* out, void* workspace, size_t ws_bytes, void* stream) {
  if (!x || !centers || !delta || !out || !workspace || n <= 0 || K <= 0) return SRK_ERR_BAD_ARG;
  if (K > 64) return SRK_ERR_UNSUPPORTED;
  const int P = parts_for(n);
  if (ws_bytes < (size_t)P * K * sizeof(float)) return SRK_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (K <= 16) hipLaunchKernelGGL(soft_hist_part_kernel<16>, dim3(P), dim3(LT), 0, st, x, n, centers, delta, K, sigma, positive_only, part);
  else hipLaunchKernelGGL(soft_hist_part_kernel<64>, dim3(P), dim3(LT), 0, st, x, n, centers, delta, K, sigma, positive_only, part);
  SRK_CHECK_LAUNCH();
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1, K), dim3(LT), 0, st, (const float*)part, out, P, K, 1.f);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_soft_hist_bwd(const float* x, long n, const float* centers, const float* delta, int K, float sigma, int positive_only,
                                 const float* gout, float* dx, void* stream) {
  if (!x || !centers || !delta || !gout || !dx || n <= 0 || K <= 0) return SRK_ERR_BAD_ARG;
  if (K > 64) return SRK_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(soft_hist_bwd_kernel, dim3(ew_grid(n)), dim3(LT), 3 * K * sizeof(float), (hipStream_t)stream, x, n, centers, delta, K,
                     sigma, positive_only, gout, dx);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
