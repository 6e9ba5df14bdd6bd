// Microbenchmark: what PACKED fp32 vector instructions cost beside fp32 MFMAs with ONE wave per SIMD (wino22 / wino42 kernel shape).
// A k-step is 16 back-to-back v_mfma_f32_32x32x2_f32 (1024 cycles of pipe time); ONE gap (behind MFMA 7) holds N vector instructions
// of one kind.  lds_beside_mfma.hip found 64 + 14 + 4 n cycles per gap for n v_add_f32; here: does a v_pk_add_f32 / v_pk_fma_f32 /
// v_pk_mul_f32 (two results per lane) cost 4 or 8?
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_beside_mfma.hip -o tools/ubench/pk_beside_mfma && tools/ubench/pk_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// KIND 0: v_add_f32   1: v_pk_add_f32   2: v_fma_f32   3: v_pk_fma_f32   4: v_pk_mul_f32   5: v_pk_add_f32 with op_sel_hi (neg mods)
template <int KIND, int N>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = lane * 0.5f, b = 1.f + wv;
  f32x2 x[4], y[4], z[4];
  for (int i = 0; i < 4; ++i) { x[i] = f32x2{1.f + lane, 2.f + i}; y[i] = f32x2{0.5f * i, 0.25f}; z[i] = f32x2{0.f, 0.f}; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      if (i == 7) {
#pragma unroll
        for (int n = 0; n < N; ++n) {
          const int r = n & 3;
          if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(z[r][0]) : "v"(x[r][0]), "v"(y[r][1]));
          if (KIND == 1) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(z[r]) : "v"(x[r]), "v"(y[r]));
          if (KIND == 2) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(z[r][0]) : "v"(x[r][0]), "v"(y[r][1]));
          if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(z[r]) : "v"(x[r]), "v"(y[r]));
          if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(z[r]) : "v"(x[r]), "v"(y[r]));
          if (KIND == 5) asm volatile("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(z[r]) : "v"(x[r]), "v"(y[r]));
        }
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 4; ++i) s += z[i][0] + z[i][1];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int N> void run(float* out, unsigned long long* cyc, const char* what) {
  const int iters = 2000, blocks = 256; unsigned long long h[256];
  auto kern = k<KIND, N>;
  kern<<<blocks, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  kern<<<blocks, 256>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  const double per = s / blocks / iters;
  printf("%-28s %2d in one gap: %7.1f cycles per 16 MFMAs (+%6.1f over 1024; %5.2f per instruction behind the first 14)\n", what, N, per, per - 1024.0,
         N ? (per - 1024.0 - 14.0) / N : 0.0);
}
int main() {
  float* out; unsigned long long* cyc; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<0, 0>(out, cyc, "bare MFMAs");
  run<0, 8>(out, cyc, "v_add_f32");   run<0, 16>(out, cyc, "v_add_f32");
  run<1, 8>(out, cyc, "v_pk_add_f32"); run<1, 16>(out, cyc, "v_pk_add_f32");
  run<5, 8>(out, cyc, "v_pk_add_f32 neg");
  run<2, 8>(out, cyc, "v_fma_f32");   run<2, 16>(out, cyc, "v_fma_f32");
  run<3, 8>(out, cyc, "v_pk_fma_f32"); run<3, 16>(out, cyc, "v_pk_fma_f32");
  run<4, 8>(out, cyc, "v_pk_mul_f32");
  return 0;
}
