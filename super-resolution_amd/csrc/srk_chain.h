// srk_chain.h -- what the CHAIN forms of the conv kernels share (srk_conv_h16.hip, srk_conv_w42.hip; host side in srk_chain.hip).
//
// A chain kernel runs a dense block's convolutions in ONE persistent launch: every workgroup keeps its tile for the whole sequence
// and hands the slice it has just written to its (up to eight) neighbouring tiles through a device-scope flag,
//     flags[tile] = epoch + k + 1   <=>   the tile's outputs of conv k of this launch have reached memory (write-through stores),
// instead of through a kernel boundary.  Rules every form keeps:
//   * at most one workgroup per CU (host-checked against the device) and at most ONE chain kernel in flight per device (the host
//     orders launches on different streams by an event): no workgroup waits for a tile that cannot become resident;
//   * a tile publishes conv k before it waits for anybody's conv k (no cycle), and EVERY WAIT IS BOUNDED by SRK_CHAIN_WAIT_MS (round 4:
//     50 ms by default -- nothing of this process holds a CU for that long --, 30 s for a data-parallel job, whose collectives' kernels
//     hold CUs while a peer rank is late; round 3: a fixed 30 s).  A tile whose neighbour has not published in time -- it is not
//     resident: a foreign process on the GPU, a kernel of another stream holding CUs beyond the bound -- sets *err = 1, poisons a
//     device word and DRAINS: it skips every later wait, keeps computing and publishing, so nobody hangs on it and the launch ends in
//     its usual time; what it writes is garbage in activation buffers the repeated iteration overwrites.  Every waiting tile looks
//     at the poison word now and then and drains too; launches already queued find it at their first wait.
//     (Tried first, and measured: a CENSUS -- every workgroup counts itself in, all wait until the whole grid is resident.  At kernel
//     entry it cost the fp32 kernels nothing by itself, but the exit it guarded made the compiler's code for the K loops 20-90 % slower
//     (profiles/r04_ab_w42_census_bisect.txt); in front of the first flag wait it was free in isolation and cost the training step 33 %
//     (114 -> 152 ms per iteration): the generator's backward runs each chain launch BESIDE the previous block's weight gradient,
//     whose workgroups hand their CUs over one by one -- tiles that wait only for their NEIGHBOURS start on the CUs that are free,
//     tiles that wait for the whole grid idle until the last one is.  Residency of the whole grid is not what a tile needs.)
//   * *err != 0 makes srk_adam_step skip its update ON THE DEVICE (the word is host memory the device reads: no round trip), and the
//     next srk_conv3x3_seq / srk_adam_step call on the host returns SRK_ERR_CHAIN_TIMEOUT until srk_chain_recover() has been called:
//     weights and optimizer state never see gradients of a launch that timed out (train.Stepper re-runs the iteration);
//   * a conv reads from its predecessor's output only through its LAST 64 input channels, fetched behind the wait; nobody reads a
//     slice before it is written, slices are whole 128-byte lines, and an L2 holds nothing from before the launch, so no XCD can
//     hold a stale copy of what it fetches behind the wait;
//   * the epoch is 32-bit and flags are compared as differences; the host zeroes the flags (on the launching stream) before the epoch
//     passes 2^30 (srk_chain_epoch_plan), so no stale flag can ever compare as "ready" after a wrap.
#pragma once
#include "srk_internal.h"

constexpr int SRK_CHAIN_MAX = 8;          // convolutions per launch
constexpr int SRK_CHAIN_FLAGS = 1024;     // tiles per launch (>= CUs of the device)

constexpr unsigned SRK_CHAIN_WRAP = 1u << 30;           // the epoch is zeroed by the host (with the flags) before it passes this

struct srk_chain_args {
  srk_conv_args c[SRK_CHAIN_MAX];
  int n;
  unsigned epoch;
  unsigned* flags;
  unsigned* err;               // host memory: 1 = a wait of some launch ran into its bound
  unsigned* poison;            // device word (uncached, behind the flags): != 0 = some tile has given up; everybody drains
  unsigned wait_ticks;         // bound of every wait (100 MHz ticks)
  unsigned skew_ticks;         // start skew: workgroup b begins (b >> 3) % skew_groups * skew_ticks late (srk_chain_skew), 0 = none
  unsigned skew_groups;
};

#if defined(__HIPCC__)
// ---- start skew.  Left alone, all tiles of a chain launch run in lockstep: every CU reaches a conv's epilogue within a microsecond of the
// others, 16.8 MB (16-bit kernel; fp32: 33.5 MB) of write-through stores hit the memory system at once, and every wave sits in its store
// instructions until HBM has taken the burst (stamps: 4.7 us per conv with NOTHING but eight fused multiply-adds and a store per item
// left in the epilogue) while the matrix pipes idle.  Tiles that start a little apart stay apart (a tile that is ahead of its neighbours
// does not wait for them until the last two stages of its next conv: slack of (nq - 2) stages), so their bursts land beside other
// tiles' main loops.  Costs the skew once per launch.
__device__ __forceinline__ void srk_chain_skew(const srk_chain_args& A) {
  if (A.skew_ticks) {
    const unsigned d = ((blockIdx.x >> 3) % A.skew_groups) * A.skew_ticks;       // (blockIdx.x >> 3: the workgroup's index inside its XCD)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)d) __builtin_amdgcn_s_sleep(4);
  }
}

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
// ONE lane: this tile gives up (a wait ran into its bound): everybody drains, the host learns of it
__device__ __forceinline__ void srk_chain_give_up(const srk_chain_args& A) {
  __hip_atomic_store(A.poison, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(A.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// wave-wide: true once every watched flag has reached `target`; false if the bound ran out (then this tile has given up) or somebody
// else has given up: the caller drains -- no more waiting in this launch
__device__ __forceinline__ bool srk_chain_wait(const srk_chain_watch& w, unsigned target, const srk_chain_args& A, int lane) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned spins = 0;; ++spins) {
    const unsigned v = w.on ? __hip_atomic_load(w.fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
    if (__all((int)(v - target) >= 0)) return true;
    if ((spins & 15u) == 15u) {
      if (__hip_atomic_load(A.poison, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return false;
      if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)A.wait_ticks) {
        if (lane == 0) srk_chain_give_up(A);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(4);
  }
}
// The same wait with SCALAR instructions only (a wave whose vector registers are all spoken for: the fp32 F(2x4,3x3) kernel).  Flags and
// poison word are uncached device memory (hipDeviceMallocUncached: no L2 keeps a copy), the loads bypass the scalar cache (glc).
// Returns 0 = go, 1 = the bound ran out (the caller calls srk_chain_give_up where it has a vector register to spare), 2 = somebody else
// has given up; != 0: the caller drains.
__device__ __forceinline__ int srk_chain_wait_scalar(const unsigned* flags, const unsigned* poison, unsigned wait_ticks, int n, int ty, int tx,
                                                     int tilesH, int tilesW, unsigned target) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  int rc = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    if (k == 4) continue;
    const int ny = ty + k / 3 - 1, nx = tx + k % 3 - 1;
    if ((unsigned)ny >= (unsigned)tilesH || (unsigned)nx >= (unsigned)tilesW) continue;      // (uniform)
    const unsigned* fp = flags + (n * tilesH + ny) * tilesW + nx;
    for (unsigned spins = 0; rc == 0; ++spins) {
      unsigned v;
      asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(fp) : "memory");
      if ((int)(v - target) >= 0) break;
      if ((spins & 15u) == 15u) {
        unsigned pz;
        asm volatile("s_load_dword %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=s"(pz) : "s"(poison) : "memory");
        if (pz != 0u) rc = 2;
        else if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)wait_ticks) rc = 1;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
  return rc;
}
#endif

// ---- host side (srk_chain.hip)
int srk_chain_cus();                       // CUs of the current device, 0 if the chain forms cannot be used on it
bool srk_chain_flags_uncached();           // the flag array is uncached memory (srk_chain_wait_scalar may be used)
// Claims the device for one chain launch of `tiles` workgroups on `st`: 1 = go (A->epoch / flags / err / poison / wait bound filled for n convs,
// `st` ordered behind the previous chain launch; call srk_chain_end afterwards), 0 = not now (stream capture, forms switched off or backing
// off after a time-out), < 0 = error (SRK_ERR_CHAIN_TIMEOUT: an earlier launch timed out and srk_chain_recover has not been called)
// kind: 0 = the 16-bit kernel, 1 = the fp32 F(2x4,3x3) kernel (start skew: SRK_H16_ / SRK_W42_CHAIN_SKEW_NS, _GROUPS; srk_debug_chain_skew)
int srk_chain_begin(hipStream_t st, int n, int tiles, int kind, srk_chain_args* A);
int srk_chain_end(hipStream_t st, bool launched);
// true while the chain forms rest after a recovered time-out (sequences go conv by conv); tick: this is a launch attempt, count it off
bool srk_chain_resting(bool tick);
// != 0 while a time-out is pending; a relaxed read of host memory
unsigned srk_chain_fault();
// the word itself (host memory, device-readable) for kernels that must not act on a faulted launch's results (srk_adam_step); nullptr if
// no chain kernel has run on this device
const unsigned* srk_chain_fault_word();
// do channels [ca, ca + na) of view (pa, lda) and [cb, cb + nb) of (pb, ldb) share memory?  (px pixels per tensor, esz bytes per element)
bool srk_chain_views_overlap(const void* pa, int lda, int ca, int na, const void* pb, int ldb, int cb, int nb, long px, int esz);
// the dense-block pattern over args[0..n): one geometry, <= 64 outputs, whole 64-channel slices on 128-byte lines, every conv >= 1 with
// >= 128 inputs of which only the last 64 may come from its predecessor's output, no conv writing what it or a LATER conv reads with
// plain loads, no auxiliary view (r1 / r2 / mask / signs) overlapping an output of the sequence
bool srk_chain_pattern_ok(const srk_conv_args* args, int n, int esz);
