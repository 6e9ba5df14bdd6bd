// Microbenchmark: cycles per v_mfma_f32_32x32x2_f32 for a single wave per SIMD (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC, bool LDS>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float4 lds[2048];
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f + 1.f;
  if (LDS) for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float4(a, b, a, b);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    float4 v = make_float4(a, b, a, b);
    if (LDS) v = lds[(threadIdx.x + it * 64) & 2047];
#pragma unroll
    for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(v.x + u, v.y, acc[i], 0, 0, 0);
      }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc; hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  unsigned long long h[1024];
  const int iters = 2000;
#define RUN(NACC, LDS, blocks) { k<NACC, LDS><<<blocks, 256>>>(out, cyc, iters); hipDeviceSynchronize(); \
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); hipEventRecord(e0); k<NACC, LDS><<<blocks, 256>>>(out, cyc, iters); hipEventRecord(e1); hipDeviceSynchronize(); \
    float ms; hipEventElapsedTime(&ms, e0, e1); hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost); double s = 0; for (int i = 0; i < blocks; ++i) s += h[i]; \
    double per = s / blocks / (iters * 16.0); double tf = blocks * 4.0 * iters * 16.0 * 4096 / (ms * 1e-3) / 1e12; \
    printf("NACC=%d LDS=%d blocks=%4d: %.2f cycles/MFMA, %.3f ms, %.1f TF/s, clock %.2f GHz\n", NACC, LDS, blocks, per, ms, tf, s / blocks / (ms * 1e-3) / 1e9); }
  RUN(1, false, 256) RUN(2, false, 256) RUN(4, false, 256) RUN(4, true, 256) RUN(4, false, 512) RUN(4, true, 512) RUN(4, false, 1)
  return 0;
}
