// srk_optim.hip -- the Adam update of a whole parameter list in ONE launch (include/srk.h: srk_adam_step).
//
// The reference steps torch.optim.Adam three times per iteration (esrgan.py:299,305: generator, two discriminators).  ATen's fused
// multi-tensor kernel spends 1.5 ms on the generator's 702 tensors (38.5 M parameters: 1.08 GB of traffic at 0.7 TB/s, 20
// launches); this kernel walks a pointer table in 4096-element chunks with 16-byte accesses: HBM-bound, 28 bytes per parameter.
// Arithmetic as ATen's fused Adam (fused_adam_utils.cuh), in fp32 with the bias corrections formed in double from a device step
// counter: grad /= grad_scale (optional), L2 weight decay into the gradient, exp_avg by lerp, exp_avg_sq, param -= lr / bc1 *
// exp_avg / (sqrt(exp_avg_sq) / sqrt(bc2) + eps); skipped altogether when *found_inf != 0 (torch.amp.GradScaler).
// srk_adam_count_step is the counting half of a step and the place where ONE decision per step is taken: it folds the fault word of the
// chain kernels (srk_chain.h: a launch that gave up leaves garbage in the gradient buffers; the word is host memory the device reads, so
// no round trip however far the host has run ahead) into the skip word the update kernel then reads as its found_inf.
#include "srk_internal.h"
#include "srk_chain.h"
#include <math.h>

namespace {
constexpr int ADAM_CHUNK = 4096;

constexpr int ADAM_SMALL = 64;          // entries that travel in the kernel arguments (a discriminator has 8-14 tensors, with new gradient
struct adam_small_table { srk_adam_entry e[ADAM_SMALL]; };       // tensors every step: no host-to-device copy of a table, no sync)

__device__ __forceinline__ void adam_body(const srk_adam_entry* __restrict__ tab, int n, float lr, float beta1, float beta2, float eps,
                                          float weight_decay, const float* __restrict__ step, const float* __restrict__ grad_scale,
                                          const float* __restrict__ found_inf) {
  if (found_inf && *found_inf != 0.f) return;
  __shared__ float bc[2];
  if (threadIdx.x == 0) {
    const double st = (double)*step;
    bc[0] = (float)((double)lr / (1.0 - pow((double)beta1, st)));           // step size
    bc[1] = (float)sqrt(1.0 - pow((double)beta2, st));                       // sqrt of the second bias correction
  }
  const long chunk = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].chunk_begin <= chunk) lo = mid; else hi = mid - 1;
  }
  const srk_adam_entry e = tab[lo];
  __syncthreads();
  const float step_size = bc[0], bc2s = bc[1];
  const float inv_scale_div = grad_scale ? *grad_scale : 1.f;
  const float w1 = 1.f - beta1, w2 = 1.f - beta2;
  const long base = (chunk - e.chunk_begin) * ADAM_CHUNK;
  auto upd = [&](float& p, float g, float& m, float& v) {
    if (grad_scale) g = g / inv_scale_div;
    if (weight_decay != 0.f) g = __builtin_fmaf(weight_decay, p, g);
    m = __builtin_fmaf(w1, g - m, m);                                         // lerp(m, g, 1 - beta1), weight < 0.5
    v = __builtin_fmaf(w2 * g, g, beta2 * v);
    const float denom = sqrtf(v) / bc2s + eps;
    p -= step_size * m / denom;
  };
  const bool vec = ((((uintptr_t)e.p | (uintptr_t)e.g | (uintptr_t)e.m | (uintptr_t)e.v) & 15) == 0);
#pragma unroll
  for (int it = 0; it < ADAM_CHUNK / 1024; ++it) {
    const long i = base + it * 1024 + threadIdx.x * 4;
    if (i >= e.n) break;
    if (vec && i + 4 <= e.n) {
      f32x4 p = *reinterpret_cast<const f32x4*>(e.p + i), m = *reinterpret_cast<const f32x4*>(e.m + i), v = *reinterpret_cast<const f32x4*>(e.v + i);
      const f32x4 g = *reinterpret_cast<const f32x4*>(e.g + i);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float pk = p[k], mk = m[k], vk = v[k];
        upd(pk, g[k], mk, vk);
        p[k] = pk; m[k] = mk; v[k] = vk;
      }
      *reinterpret_cast<f32x4*>(e.p + i) = p; *reinterpret_cast<f32x4*>(e.m + i) = m; *reinterpret_cast<f32x4*>(e.v + i) = v;
    } else {
      for (long j = i; j < e.n && j < i + 4; ++j) upd(e.p[j], e.g[j], e.m[j], e.v[j]);
    }
  }
}
__global__ __launch_bounds__(256) void adam_kernel(const srk_adam_entry* __restrict__ tab, int n, float lr, float beta1, float beta2, float eps,
                                                    float weight_decay, const float* __restrict__ step, const float* __restrict__ grad_scale,
                                                    const float* __restrict__ found_inf) {
  adam_body(tab, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf);
}
__global__ __launch_bounds__(256) void adam_small_kernel(const adam_small_table T, int n, float lr, float beta1, float beta2, float eps,
                                                          float weight_decay, const float* __restrict__ step, const float* __restrict__ grad_scale,
                                                          const float* __restrict__ found_inf) {
  adam_body(T.e, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, found_inf);
}
__global__ void adam_count_kernel(float* step, const float* found_inf, float* skip_out, const unsigned* fault) {
  const bool skip = (found_inf && *found_inf != 0.f) || (fault && __hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u);
  if (skip_out) *skip_out = skip ? 1.f : 0.f;
  if (!skip) *step += 1.f;
}
}  // namespace

extern "C" int srk_adam_count_step(float* step, const float* found_inf, float* skip_out, void* stream) {
  if (!step) return SRK_ERR_BAD_ARG;
  if (srk_chain_fault()) return SRK_ERR_CHAIN_TIMEOUT;
  hipLaunchKernelGGL(adam_count_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step, found_inf, skip_out, srk_chain_fault_word());
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_adam_plan(srk_adam_entry* host_entries, int n, int64_t* total_chunks) {
  if (!host_entries || n <= 0 || !total_chunks) return SRK_ERR_BAD_ARG;
  int64_t c = 0;
  for (int i = 0; i < n; ++i) {
    if (!host_entries[i].p || !host_entries[i].g || !host_entries[i].m || !host_entries[i].v || host_entries[i].n <= 0) return SRK_ERR_BAD_ARG;
    host_entries[i].chunk_begin = c;
    c += (host_entries[i].n + ADAM_CHUNK - 1) / ADAM_CHUNK;
  }
  *total_chunks = c;
  return SRK_OK;
}

// the same for up to 64 tensors from a HOST table (planned by srk_adam_plan), which travels in the kernel arguments
extern "C" int srk_adam_step_small(const srk_adam_entry* host_entries, int n, int64_t total_chunks, float lr, float beta1, float beta2, float eps,
                                   float weight_decay, const float* step, const float* grad_scale, const float* found_inf, void* stream) {
  if (!host_entries || n <= 0 || n > ADAM_SMALL || total_chunks <= 0 || total_chunks > 0x7fffffffL || !step) return SRK_ERR_BAD_ARG;
  if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return SRK_ERR_BAD_ARG;
  if (1.f - beta1 >= 0.5f) return SRK_ERR_UNSUPPORTED;
  if (srk_chain_fault()) return SRK_ERR_CHAIN_TIMEOUT;
  adam_small_table T;
  for (int i = 0; i < n; ++i) T.e[i] = host_entries[i];
  for (int i = n; i < ADAM_SMALL; ++i) T.e[i] = host_entries[n - 1];
  hipLaunchKernelGGL(adam_small_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, T, n, lr, beta1, beta2, eps, weight_decay,
                     step, grad_scale, found_inf);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

extern "C" int srk_adam_step(const srk_adam_entry* device_entries, int n, int64_t total_chunks, float lr, float beta1, float beta2, float eps,
                             float weight_decay, const float* step, const float* grad_scale, const float* found_inf, void* stream) {
  if (!device_entries || n <= 0 || total_chunks <= 0 || total_chunks > 0x7fffffffL || !step) return SRK_ERR_BAD_ARG;
  if (!(beta1 >= 0.f && beta1 < 1.f) || !(beta2 >= 0.f && beta2 < 1.f)) return SRK_ERR_BAD_ARG;
  if (1.f - beta1 >= 0.5f) return SRK_ERR_UNSUPPORTED;          // (lerp's other branch: beta1 <= 0.5 is not Adam as anybody runs it)
  if (srk_chain_fault()) return SRK_ERR_CHAIN_TIMEOUT;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, device_entries, n, lr, beta1, beta2, eps,
                     weight_decay, step, grad_scale, found_inf);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
