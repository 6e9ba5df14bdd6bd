// Verify ds_read_b64_tr_b16 semantics on gfx950 with integer-valued bf16/u16 data.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out) {
  __shared__ unsigned short lds[64 * 32];            // [row 64][col 32] u16, value = row*100 + col
  for (int i = threadIdx.x; i < 64 * 32; i += 64) lds[i] = (unsigned short)((i / 32) * 100 + (i % 32));
  __syncthreads();
  const int l = threadIdx.x, g = l >> 4, q = (l & 15) >> 2, p = l & 3;
  // group g reads rows r0..r0+3 (r0 = 8*(g>>1)), cols 16*(g&1) + 0..15; lane 4q+p supplies row q, cols 4p..4p+3
  const int r0 = 8 * (g >> 1);
  const unsigned short* addr = lds + (r0 + q) * 32 + 16 * (g & 1) + 4 * p;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; (void)hipMalloc(&d, 64 * 4 * 2);
  k<<<1, 64>>>(d); (void)hipDeviceSynchronize();
  unsigned short h[256]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15, r0 = 8 * (g >> 1), c = 16 * (g & 1) + i;
    for (int j = 0; j < 4; ++j) { const int exp = (r0 + j) * 100 + c; if (h[l * 4 + j] != exp) { if (bad < 8) printf("lane %d elem %d: got %d expected %d\n", l, j, h[l*4+j], exp); ++bad; } }
  }
  printf("tr16_b64 check: %s (%d mismatches)\n", bad ? "FAIL" : "OK: lane i of a 16-lane group gets column i of the 4 rows, row q in element q", bad);
  return bad != 0;
}
