// srk_chain.h -- what the CHAIN forms of the conv kernels share (srk_conv_h16.hip, srk_conv_w42.hip; host side in srk_chain.hip).
//
// A chain kernel runs a dense block's convolutions in ONE persistent launch: every workgroup keeps its tile for the whole sequence
// and hands the slice it has just written to its (up to eight) neighbouring tiles through a device-scope flag,
//     flags[tile] = epoch + k + 1   <=>   the tile's outputs of conv k of this launch have reached memory (write-through stores),
// instead of through a kernel boundary.  Rules every form keeps:
//   * at most one workgroup per CU (host-checked against the device) and at most ONE chain kernel in flight per device (the host
//     orders launches on different streams by an event): no workgroup waits for a tile that cannot become resident;
//   * a tile publishes conv k before it waits for anybody's conv k (no cycle), and every wait is bounded (SRK_CHAIN_WAIT_TICKS = 30 s: an all-reduce kernel holding CUs while a peer rank
//     is seconds late must not trip it): on a time-out the
//     kernel sets *err and goes on, so it always drains; the host turns the word into an error on its next call and switches the
//     chain forms off;
//   * a conv reads from its predecessor's output only through its LAST 64 input channels, fetched behind the wait; nobody reads a
//     slice before it is written, slices are whole 128-byte lines, and an L2 holds nothing from before the launch, so no XCD can
//     hold a stale copy of what it fetches behind the wait.
#pragma once
#include "srk_internal.h"

constexpr int SRK_CHAIN_MAX = 8;          // convolutions per launch
constexpr int SRK_CHAIN_FLAGS = 1024;     // tiles per launch (>= CUs of the device)
constexpr unsigned long long SRK_CHAIN_WAIT_TICKS = 3000000000ull;     // 30 s of the 100 MHz s_memrealtime counter

struct srk_chain_args {
  srk_conv_args c[SRK_CHAIN_MAX];
  int n;
  unsigned epoch;
  unsigned* flags;
  unsigned* err;
};

#if defined(__HIPCC__)
// The neighbour lane `lane` (0..8: dy = lane / 3 - 1, dx = lane % 3 - 1) of tile (n, ty, tx) watches; lanes that watch nothing read the
// tile's own flag and ignore it.
struct srk_chain_watch {
  const unsigned* fp;
  bool on;
};
__device__ __forceinline__ srk_chain_watch srk_chain_watch_of(const unsigned* flags, int lane, int n, int ty, int tx, int tilesH, int tilesW) {
  const int ndy = lane / 3 - 1, ndx = lane % 3 - 1;
  srk_chain_watch w;
  w.on = lane < 9 && lane != 4 && (unsigned)(ty + ndy) < (unsigned)tilesH && (unsigned)(tx + ndx) < (unsigned)tilesW;
  w.fp = flags + (w.on ? (n * tilesH + ty + ndy) * tilesW + tx + ndx : (n * tilesH + ty) * tilesW + tx);
  return w;
}
// wave-wide: returns once every watched flag has reached `target` (or after the time limit, with *err set)
__device__ __forceinline__ void srk_chain_wait(const srk_chain_watch& w, unsigned target, unsigned* err, int lane) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned v = w.on ? __hip_atomic_load(w.fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
    if (__all((int)(v - target) >= 0)) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > SRK_CHAIN_WAIT_TICKS) {
      if (lane == 0) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      break;
    }
    __builtin_amdgcn_s_sleep(4);
  }
}
// The same wait with SCALAR instructions only (a wave whose vector registers are all spoken for: the fp32 F(2x4,3x3) kernel).  The flag
// array is uncached device memory (hipDeviceMallocUncached: no L2 keeps a copy), the loads bypass the scalar cache (glc).  Returns
// false after the time limit; the caller reports that through *err where it has a vector register to spare.
__device__ __forceinline__ bool srk_chain_wait_scalar(const unsigned* flags, int n, int ty, int tx, int tilesH, int tilesW, unsigned target) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool ok = true;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    if (k == 4) continue;
    const int ny = ty + k / 3 - 1, nx = tx + k % 3 - 1;
    if ((unsigned)ny >= (unsigned)tilesH || (unsigned)nx >= (unsigned)tilesW) continue;      // (uniform)
    const unsigned* fp = flags + (n * tilesH + ny) * tilesW + nx;
    for (;;) {
      unsigned v;
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(fp) : "memory");
      if ((int)(v - target) >= 0) break;
      if (__builtin_amdgcn_s_memrealtime() - t0 > SRK_CHAIN_WAIT_TICKS) { ok = false; break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  return ok;
}
#endif

// ---- host side (srk_chain.hip)
int srk_chain_cus();                       // CUs of the current device, 0 if the chain forms cannot be used on it
bool srk_chain_flags_uncached();           // the flag array is uncached memory (srk_chain_wait_scalar may be used)
// Claims the device for one chain launch on `st`: 1 = go (A->epoch / flags / err filled for n convs, `st` ordered behind the previous chain
// launch; call srk_chain_end afterwards), 0 = not now (stream capture, forms switched off), < 0 = error (a time-out of an earlier launch)
int srk_chain_begin(hipStream_t st, int n, srk_chain_args* A);
int srk_chain_end(hipStream_t st, bool launched);
// do channels [ca, ca + na) of view (pa, lda) and [cb, cb + nb) of (pb, ldb) share memory?  (px pixels per tensor, esz bytes per element)
bool srk_chain_views_overlap(const void* pa, int lda, int ca, int na, const void* pb, int ldb, int cb, int nb, long px, int esz);
// the dense-block pattern over args[0..n): one geometry, <= 64 outputs, whole 64-channel slices on 128-byte lines, every conv >= 1 with
// >= 128 inputs of which only the last 64 may come from its predecessor's output, no conv writing what it reads
bool srk_chain_pattern_ok(const srk_conv_args* args, int n, int esz);
