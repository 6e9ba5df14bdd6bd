// Microbenchmark: what does it cost a wave to ISSUE global -> LDS DMA instructions (buffer_load_dwordx4 ... lds) on gfx950,
// compared with a plain buffer_load_dwordx4 (+ ds_write_b128)?  One wave per CU issues NI instructions per iteration and waits
// for them (vmcnt(0)); the issue span is measured with s_memtime before the wait.
//   mode 0: plain buffer loads into registers + ds_write_b128
//   mode 1: DMA, a different LDS base (m0) per instruction
//   mode 2: DMA, one LDS base, instruction immediates 0 / 1024 / 2048 / 3072 (groups of four)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NI = 8;
template <int MODE>
__global__ __launch_bounds__(64) void k(const float* src, float* out, unsigned long long* cyc, int iters) {
  __shared__ float4 lds[NI * 64 * 2];
  const int lane = threadIdx.x;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src) + (size_t)blockIdx.x * 65536, 0, 1u << 30, 0x00020000);
  unsigned long long issue = 0, total = 0;
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    const unsigned base = (unsigned)(((it * 7) & 15) * 16384) + lane * 16;
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (MODE == 0) {
      f32x4 v[NI];
#pragma unroll
      for (int i = 0; i < NI; ++i) v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, base + i * 1024, 0, 0));
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int i = 0; i < NI; ++i) lds[i * 64 + lane] = make_float4(v[i][0], v[i][1], v[i][2], v[i][3]);
      issue += t1 - t0;
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < NI; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + i * 64), 16, base + i * 1024, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      issue += __builtin_amdgcn_s_memtime() - t0;
    } else {
#pragma unroll
      for (int i = 0; i < NI; i += 4) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + i * 64), 16, base + i * 1024, 0, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + i * 64), 16, base + i * 1024, 0, 1024, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + i * 64), 16, base + i * 1024, 0, 2048, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(lds + i * 64), 16, base + i * 1024, 0, 3072, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      issue += __builtin_amdgcn_s_memtime() - t0;
    }
    __syncthreads();
    total += __builtin_amdgcn_s_memtime() - t0;
    acc += reinterpret_cast<float*>(lds)[lane + (it & 7) * 64];
    __syncthreads();
  }
  out[blockIdx.x * 64 + lane] = acc;
  if (lane == 0) { cyc[2 * blockIdx.x] = issue; cyc[2 * blockIdx.x + 1] = total; }
}
int main() {
  float *src, *out; unsigned long long* cyc;
  hipMalloc(&src, (size_t)256 * 65536 * 4 + (1 << 20)); hipMemset(src, 0, (size_t)256 * 65536 * 4 + (1 << 20));
  hipMalloc(&out, 256 * 64 * 4); hipMalloc(&cyc, 256 * 16);
  unsigned long long h[512];
  const int iters = 2000;
#define RUN(MODE, blocks) { k<MODE><<<blocks, 64>>>(src, out, cyc, iters); hipDeviceSynchronize(); k<MODE><<<blocks, 64>>>(src, out, cyc, iters); hipDeviceSynchronize(); \
    hipMemcpy(h, cyc, blocks * 16, hipMemcpyDeviceToHost); double a = 0, b = 0; for (int i = 0; i < blocks; ++i) { a += h[2 * i]; b += h[2 * i + 1]; } \
    printf("mode %d, %3d waves: issue %.0f cycles per instruction, issue+wait %.0f cycles per group of %d\n", MODE, blocks, a / blocks / iters / NI, b / blocks / iters, NI); }
  RUN(0, 1) RUN(1, 1) RUN(2, 1) RUN(0, 256) RUN(1, 256) RUN(2, 256)
  return 0;
}
