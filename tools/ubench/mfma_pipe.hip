// Microbenchmark: software-pipelined ds_read_b128 fragments + 16 MFMAs per "tap" (mimics the conv main loop).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int VARIANT>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ float4 lds[3600];
  f32x16 acc[2][2];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i >> 1][i & 1][r] = 0.f;
  for (int i = threadIdx.x; i < 3600; i += 256) lds[i] = make_float4(i * 1e-3f, 1.f, 0.5f, 0.25f);
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hl = lane >> 5, l32 = lane & 31;
  const int abase = ((2 * wv + (l32 >> 4)) * 18 + (l32 & 15)) * 2 + hl;
  const float4* xb = lds + abase;
  const float4* wb = lds + 648 + hl * 64 + l32;
  float4 av[2][2], bv[2][2];
  av[0][0] = xb[0]; av[0][1] = xb[288]; bv[0][0] = wb[0]; bv[0][1] = wb[32];
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int cur = tap & 1, nxt = cur ^ 1;
      const int r = (tap + 1) / 3, s = (tap + 1) % 3;
      if (VARIANT != 2) {
        av[nxt][0] = xb[(r * 18 + s) * 2]; av[nxt][1] = xb[(144 + r * 18 + s) * 2];
        bv[nxt][0] = wb[(tap + 1) * 128]; bv[nxt][1] = wb[(tap + 1) * 128 + 32];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const float ae = e == 0 ? av[cur][m].x : e == 1 ? av[cur][m].y : e == 2 ? av[cur][m].z : av[cur][m].w;
            const float be = e == 0 ? bv[cur][t].x : e == 1 ? bv[cur][t].y : e == 2 ? bv[cur][t].z : bv[cur][t].w;
            acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(ae, be, acc[m][t], 0, 0, 0);
          }
      if (VARIANT == 1) __builtin_amdgcn_sched_barrier(0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i >> 1][i & 1][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int V> void run(float* out, unsigned long long* cyc, int blocks) {
  const int iters = 500; unsigned long long h[1024];
  k<V><<<blocks, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0);
  k<V><<<blocks, 256>>>(out, cyc, iters); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  printf("variant %d blocks=%4d: %.2f cycles/MFMA, clock %.2f GHz\n", V, blocks, s / blocks / (iters * 128.0), s / blocks / (ms * 1e-3) / 1e9);
}
int main() {
  float* out; unsigned long long* cyc; (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&cyc, 1024 * 8);
  run<0>(out, cyc, 256); run<1>(out, cyc, 256); run<2>(out, cyc, 256); run<0>(out, cyc, 512); run<0>(out, cyc, 1);
  return 0;
}
