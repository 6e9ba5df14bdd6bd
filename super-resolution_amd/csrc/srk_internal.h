// Internal helpers shared by the gfx950 kernels (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "srk.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SRK_TH 8          // output rows per workgroup tile
#define SRK_TW 16         // output cols per workgroup tile
#define SRK_THREADS 256   // 4 waves of 64

static inline int srk_round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int srk_div_up(int v, int m) { return (v + m - 1) / m; }

#define SRK_CHECK_LAUNCH()                                   \
  do {                                                       \
    hipError_t e__ = hipGetLastError();                      \
    if (e__ != hipSuccess) return SRK_ERR_LAUNCH;            \
  } while (0)

