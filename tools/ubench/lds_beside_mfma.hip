// Microbenchmark: what an LDS read costs beside fp32 MFMAs with ONE wave per SIMD (the wino42 / wino22 kernel shape).
// A "k-step" is 16 back-to-back v_mfma_f32_32x32x2_f32 (1024 cycles of pipe time) with NR reads of one kind spread over the
// first slots and one s_waitcnt lgkmcnt(0) in slot 8.  Everything is volatile inline asm, so the order below is the order run.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_beside_mfma.hip -o /tmp/lds_beside_mfma && /tmp/lds_beside_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 0: none  1: ds_read_b32  2: ds_read2st64_b32  3: ds_read_b64  4: ds_read2_b64  5: ds_read_b128  6: ds_read2st64_b64
template <int KIND, int NR, int PER_SLOT>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ float lds[];                       // 160 KB requested at launch: one workgroup per CU
  for (int i = threadIdx.x; i < 40960; i += 256) lds[i] = i * 1e-3f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = lane * 0.5f, b = 1.f + wv;
  // per-lane byte address: conflict-free for every width (consecutive lanes, consecutive elements of the width read)
  const unsigned w = KIND == 1 || KIND == 2 ? 4 : (KIND == 3 || KIND == 4 || KIND == 6 ? 8 : 16);
  const unsigned addr = wv * 16384 + lane * w;
  float r1[16]; f32x2 r2[16]; f32x4 r4[16];
  for (int i = 0; i < 16; ++i) { r1[i] = 0.f; r2[i] = f32x2{0.f, 0.f}; r4[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    int issued = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
      for (int q = 0; q < PER_SLOT; ++q) {
        if (issued < NR) {
          const int n = issued;
          if (KIND == 1) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1[n]) : "v"(addr), "n"(256 * (n % 16)));
          if (KIND == 2) asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(r2[n]) : "v"(addr), "n"(2 * (n % 16)), "n"(2 * (n % 16) + 1));
          if (KIND == 3) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2[n]) : "v"(addr), "n"(512 * (n % 16)));
          if (KIND == 4) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r4[n]) : "v"(addr), "n"(128 * (n % 2)), "n"(128 * (n % 2) + 64));
          if (KIND == 5) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4[n]) : "v"(addr), "n"(1024 * (n % 16)));
          if (KIND == 6) asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r4[n]) : "v"(addr), "n"(2 * (n % 16)), "n"(2 * (n % 16) + 1));
          ++issued;
        }
      }
      if (i == 8) asm volatile("s_waitcnt lgkmcnt(0)");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 16; ++i) s += r1[i] + r2[i][0] + r2[i][1] + r4[i][0] + r4[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}


// VALU fillers: one asm block per MFMA slot (nothing the compiler could pad), N adds per slot.
// SRC 0: adds read registers no MFMA touches; 1: adds read the MFMA's own A / B operands; 2: 4 dependent chains
template <int N, int SRC>
__global__ __launch_bounds__(256) void kv(float* out, unsigned long long* cyc, int iters) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = lane * 0.5f, b = 1.f + wv, c = 2.f + lane, d = 3.f + wv;
  float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
#define ADD4(x, y) "v_add_f32 %1, " x ", " y "\n v_sub_f32 %2, " x ", " y "\n v_add_f32 %3, " y ", " x "\n v_sub_f32 %4, " y ", " x "\n"
#define DEP4 "v_add_f32 %1, %1, %7\n v_add_f32 %2, %2, %7\n v_add_f32 %3, %3, %7\n v_add_f32 %4, %4, %7\n"
      if (N == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 4 && SRC == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" ADD4("%7", "%8") : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 8 && SRC == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" ADD4("%7", "%8") ADD4("%7", "%8") : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 12 && SRC == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" ADD4("%7", "%8") ADD4("%7", "%8") ADD4("%7", "%8") : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 16 && SRC == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" ADD4("%7", "%8") ADD4("%7", "%8") ADD4("%7", "%8") ADD4("%7", "%8") : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 4 && SRC == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" ADD4("%5", "%6") : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 8 && SRC == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" ADD4("%5", "%6") ADD4("%5", "%6") : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 4 && SRC == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" DEP4 : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
      if (N == 8 && SRC == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %5, %6, %0\n" DEP4 DEP4 : "+a"(acc[i]), "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3) : "v"(a), "v"(b), "v"(c), "v"(d));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = r0 + r1 + r2 + r3;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int N, int SRC> void runv(float* out, unsigned long long* cyc, const char* what) {
  const int iters = 2000, blocks = 256; unsigned long long h[256];
  auto kern = kv<N, SRC>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  kern<<<blocks, 256, 163840>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  kern<<<blocks, 256, 163840>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  const double per = s / blocks / iters;
  printf("%-60s %2d VALU per MFMA: %7.1f cycles per 16 MFMAs (+%6.1f over 1024; %5.2f per VALU)\n", what, N, per, per - 1024.0, N ? (per - 1024.0) / (16 * N) : 0.0);
}

// Vector-memory instructions beside the MFMAs: N per k-step (one per gap from gap 0 on), one s_waitcnt vmcnt(0) at the end.
// KIND 0: buffer_load_dword  1: dwordx2  2: dwordx4 (to registers; L2 hits: 64 KB walked in a ring)  3: buffer_load_dwordx4 ... lds
template <int KIND, int N>
__global__ __launch_bounds__(256) void km(float* out, unsigned long long* cyc, int iters, const float* src) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  f32x16 acc[16];
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float a = lane * 0.5f, b = 1.f + wv;
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const unsigned long long pa = (unsigned long long)src;
  i32x4 rsrc;                                          // raw buffer: base, no stride, 1 MB, the flags the library's DMA uses
  rsrc[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)pa); rsrc[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((pa >> 32) & 0xffff));
  rsrc[2] = 1 << 20; rsrc[3] = 0x00020000;
  unsigned voff = (blockIdx.x & 7) * 65536 + wv * 16384 + lane * 16;
  float r1[16]; f32x2 r2[16]; f32x4 r4[16];
  int sdummy = 0;
  for (int i = 0; i < 16; ++i) { r1[i] = 0.f; r2[i] = f32x2{0.f, 0.f}; r4[i] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  asm volatile("s_mov_b32 m0, %0" :: "s"(__builtin_amdgcn_readfirstlane(wv * 16384)));
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
      if (i < N) {
        if (KIND == 0) asm volatile("buffer_load_dword %0, %1, %2, 0 offen offset:%3" : "=v"(r1[i]) : "v"(voff), "s"(rsrc), "n"(1024 * (i % 4)));
        if (KIND == 1) asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 offen offset:%3" : "=v"(r2[i]) : "v"(voff), "s"(rsrc), "n"(1024 * (i % 4)));
        if (KIND == 2) asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen offset:%3" : "=v"(r4[i]) : "v"(voff), "s"(rsrc), "n"(1024 * (i % 4)));
        if (KIND == 3) asm volatile("buffer_load_dwordx4 %0, %1, 0 offen offset:%2 lds" :: "v"(voff), "s"(rsrc), "n"(1024 * (i % 4)) : "memory");
        if (KIND == 4) asm volatile("s_waitcnt vmcnt(15)");                                    // (always satisfied)
        if (KIND == 5) asm volatile("s_add_u32 %0, %0, 4" : "+s"(sdummy) :: "scc");
        if (KIND == 6) asm volatile("s_waitcnt vmcnt(15)\n s_add_u32 %1, %1, 4\n buffer_load_dwordx2 %0, %2, %3, 0 offen offset:%4" : "=v"(r2[i]), "+s"(sdummy) : "v"(voff), "s"(rsrc), "n"(1024 * (i % 4)) : "scc");
        if (KIND == 7) asm volatile("s_nop 0");
      }
      if (i == 15) asm volatile("s_waitcnt vmcnt(0)");
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = lds[threadIdx.x] + sdummy;
  for (int t = 0; t < 16; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
  for (int i = 0; i < 16; ++i) s += r1[i] + r2[i][0] + r2[i][1] + r4[i][0] + r4[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int KIND, int N> void runm(float* out, unsigned long long* cyc, const float* src, const char* what) {
  const int iters = 2000, blocks = 256; unsigned long long h[256];
  auto kern = km<KIND, N>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  kern<<<blocks, 256, 163840>>>(out, cyc, iters, src); (void)hipDeviceSynchronize();
  kern<<<blocks, 256, 163840>>>(out, cyc, iters, src); (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  const double per = s / blocks / iters;
  printf("%-44s %2d per 16 MFMAs: %7.1f cycles (+%6.1f over 1024; %5.1f per instruction)\n", what, N, per, per - 1024.0, N ? (per - 1024.0) / N : 0.0);
}

template <int KIND, int NR, int PER_SLOT> void run(float* out, unsigned long long* cyc, const char* what) {
  const int iters = 2000, blocks = 256; unsigned long long h[256];
  auto kern = k<KIND, NR, PER_SLOT>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  kern<<<blocks, 256, 163840>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  kern<<<blocks, 256, 163840>>>(out, cyc, iters); (void)hipDeviceSynchronize();
  (void)hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
  double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
  const double per = s / blocks / iters;
  printf("%-44s %2d reads, %d per slot: %7.1f cycles per k-step (+%6.1f over 1024; %5.1f per read, %5.2f per dword)\n", what, NR, PER_SLOT, per,
         per - 1024.0, NR ? (per - 1024.0) / NR : 0.0,
         NR ? (per - 1024.0) / NR / (KIND == 1 ? 1 : KIND == 2 || KIND == 3 ? 2 : 4) : 0.0);
}
int main() {
  float* out; unsigned long long* cyc; (void)hipMalloc(&out, 256 * 256 * 4); (void)hipMalloc(&cyc, 256 * 8);
  run<0, 0, 1>(out, cyc, "no reads");
  run<1, 14, 2>(out, cyc, "ds_read_b32");
  run<1, 14, 1>(out, cyc, "ds_read_b32");
  run<2, 7, 2>(out, cyc, "ds_read2st64_b32 (the wino22 reads)");
  run<2, 7, 1>(out, cyc, "ds_read2st64_b32");
  run<3, 7, 2>(out, cyc, "ds_read_b64");
  run<3, 7, 1>(out, cyc, "ds_read_b64");
  run<3, 14, 2>(out, cyc, "ds_read_b64");
  run<4, 7, 1>(out, cyc, "ds_read2_b64");
  run<6, 7, 1>(out, cyc, "ds_read2st64_b64");
  run<5, 4, 1>(out, cyc, "ds_read_b128");
  run<5, 7, 1>(out, cyc, "ds_read_b128");
  run<5, 14, 2>(out, cyc, "ds_read_b128");
  runv<0, 0>(out, cyc, "bare MFMAs");
  runv<4, 0>(out, cyc, "v_add/v_sub on registers no MFMA reads");
  runv<8, 0>(out, cyc, "v_add/v_sub on registers no MFMA reads");
  runv<12, 0>(out, cyc, "v_add/v_sub on registers no MFMA reads");
  runv<16, 0>(out, cyc, "v_add/v_sub on registers no MFMA reads");
  runv<4, 1>(out, cyc, "v_add/v_sub reading the MFMA's A / B registers");
  runv<8, 1>(out, cyc, "v_add/v_sub reading the MFMA's A / B registers");
  runv<4, 2>(out, cyc, "four dependent v_add chains");
  runv<8, 2>(out, cyc, "four dependent v_add chains");
  float* src; (void)hipMalloc(&src, 1 << 20); (void)hipMemset(src, 0, 1 << 20);
  runm<0, 4>(out, cyc, src, "buffer_load_dword");
  runm<1, 4>(out, cyc, src, "buffer_load_dwordx2");
  runm<1, 8>(out, cyc, src, "buffer_load_dwordx2");
  runm<2, 4>(out, cyc, src, "buffer_load_dwordx4");
  runm<2, 8>(out, cyc, src, "buffer_load_dwordx4");
  runm<3, 2>(out, cyc, src, "buffer_load_dwordx4 ... lds");
  runm<3, 4>(out, cyc, src, "buffer_load_dwordx4 ... lds");
  runm<4, 12>(out, cyc, src, "s_waitcnt vmcnt(15) (satisfied)");
  runm<5, 12>(out, cyc, src, "s_add_u32");
  runm<7, 12>(out, cyc, src, "s_nop 0");
  runm<6, 12>(out, cyc, src, "s_waitcnt + s_add_u32 + buffer_load_dwordx2 (a wino42 weight-load gap)");
  return 0;
}
