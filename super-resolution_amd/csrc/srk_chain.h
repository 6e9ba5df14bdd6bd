// srk_chain.h -- what the CHAIN forms of the conv kernels share (srk_conv_h16.hip, srk_conv_w42.hip; host side in srk_chain.hip).
//
// A chain kernel runs a dense block's convolutions in ONE persistent launch: every workgroup keeps its tile for the whole sequence
// and hands the slice it has just written to its (up to eight) neighbouring tiles through a device-scope flag,
//     flags[tile] = epoch + k + 1   <=>   the tile's outputs of conv k of this launch have reached memory (write-through stores),
// instead of through a kernel boundary.  Rules every form keeps:
//   * at most one workgroup per CU (host-checked against the device) and at most ONE chain kernel in flight per device (the host
//     orders launches on different streams by an event): no workgroup waits for a tile that cannot become resident;
//   * CENSUS (round 4): before a workgroup stores anything or waits for anybody, every workgroup of the launch has counted itself in
//     (srk_chain_census_*: one atomic add per workgroup on a device word, then a bounded wait -- SRK_CHAIN_ENTRY_MS, 50 ms -- until
//     the count has reached the launch's target).  If the grid is not resident by then (a foreign process on the GPU, a kernel of
//     another stream holding CUs for longer than that) the launch gives up BEFORE any arithmetic has consumed unpublished data and
//     before any store: the first workgroup to time out ORs a poison bit into the count word; the last arrival, which would have
//     published "go", finds the bit and publishes nothing, so all workgroups reach the same verdict; *err = 1, every workgroup returns;
//   * a tile publishes conv k before it waits for anybody's conv k (no cycle), and every flag wait is bounded as well
//     (SRK_CHAIN_WAIT_TICKS = 30 s; with the census in front it can only trip on a fault): it sets *err = 2 and goes on, so the
//     kernel always drains;
//   * *err != 0 makes srk_adam_step skip its update ON THE DEVICE (the word is host memory the device reads: no round trip), and the
//     next srk_conv3x3_seq / srk_adam_step call on the host returns SRK_ERR_CHAIN_TIMEOUT until srk_chain_recover() has been called:
//     weights and optimizer state never see gradients of a launch that timed out (train.Stepper re-runs the iteration);
//   * a conv reads from its predecessor's output only through its LAST 64 input channels, fetched behind the wait; nobody reads a
//     slice before it is written, slices are whole 128-byte lines, and an L2 holds nothing from before the launch, so no XCD can
//     hold a stale copy of what it fetches behind the wait;
//   * epoch and census count are 32-bit and compared as differences; the host zeroes flags and count (on the launching stream) before
//     either passes 2^30 (srk_chain_epoch_plan), so no stale flag can ever compare as "ready" after a wrap.
#pragma once
#include "srk_internal.h"

constexpr int SRK_CHAIN_MAX = 8;          // convolutions per launch
constexpr int SRK_CHAIN_FLAGS = 1024;     // tiles per launch (>= CUs of the device)
constexpr unsigned long long SRK_CHAIN_WAIT_TICKS = 3000000000ull;     // 30 s of the 100 MHz s_memrealtime counter

constexpr unsigned SRK_CHAIN_POISON = 0x80000000u;     // bit of the census word: a launch has given up (stays until srk_chain_recover)
constexpr unsigned SRK_CHAIN_WRAP = 1u << 30;           // epoch / census count are zeroed by the host before they pass this

struct srk_chain_args {
  srk_conv_args c[SRK_CHAIN_MAX];
  int n;
  unsigned epoch;
  unsigned* flags;
  unsigned* err;
  unsigned* arrive;            // census word (device, uncached): workgroups of all chain launches so far | SRK_CHAIN_POISON
  unsigned arrive_target;      // its value once every workgroup of THIS launch has counted itself in
  unsigned entry_ticks;        // bound of the census wait (100 MHz ticks)
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

// ---- census: is every workgroup of this launch resident?  Called by ONE lane of the workgroup.
// Two words, 64 bytes apart: `arrive` is only ever touched by atomics (one add per workgroup; an OR when somebody gives up), `go`
// (= arrive + 16) is stored ONCE per launch, by the last workgroup to arrive, and is what everybody polls.  (First version, one word:
// 256 workgroups polling the word the late arrivals' atomic adds were queued on -- the polls starved the adds, and the census of the
// fp32 kernels took 84-242 us instead of ~2: profiles/r04_ab_w42_census_first_version.txt.)
// step 1, at kernel entry: count me in.  0 = an earlier launch has poisoned the word (give up), 1 = counted, 2 = counted as the LAST one
// (the whole grid is resident; `go` has been published).
__device__ __forceinline__ int srk_chain_census_arrive(const srk_chain_args& A) {
  const unsigned old = __hip_atomic_fetch_add(A.arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (old & SRK_CHAIN_POISON) return 0;
  if (old + 1u == A.arrive_target) {
    __hip_atomic_store(A.arrive + 16, A.arrive_target, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 2;
  }
  return 1;
}
// step 2, before the workgroup's first store / first flag wait: true = all resident, go; false = give up (nothing touched so far).
// The verdict is the same for every workgroup: whoever times out ORs the poison bit into `arrive`; the OR returns the count as it was --
// short of the target: the last arrival will find the bit and publish nothing, so nobody ever goes; at the target: everybody had arrived
// (the bit came too late to matter) and `go` is on its way, keep waiting for it.
__device__ __forceinline__ bool srk_chain_census_wait(const srk_chain_args& A, int arrived) {
  if (arrived == 0) return false;
  if (arrived == 2) return true;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  bool timed_out = false;
  for (unsigned spins = 0;; ++spins) {
    if (__hip_atomic_load(A.arrive + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.arrive_target) return true;
    if ((spins & 63u) == 63u && (__hip_atomic_load(A.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & SRK_CHAIN_POISON)) {
      // somebody gave up: unless everybody had arrived before that (then `go` is coming), so do I
      if ((__hip_atomic_load(A.arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~SRK_CHAIN_POISON) != A.arrive_target) return false;
    }
    if (!timed_out && __builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)A.entry_ticks) {
      timed_out = true;
      const unsigned old = __hip_atomic_fetch_or(A.arrive, SRK_CHAIN_POISON, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if ((old & ~SRK_CHAIN_POISON) != A.arrive_target) {
        __hip_atomic_store(A.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return false;
      }
    }
    __builtin_amdgcn_s_sleep(16);
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
// wave-wide: returns once every watched flag has reached `target` (or after the time limit, with *err set)
__device__ __forceinline__ void srk_chain_wait(const srk_chain_watch& w, unsigned target, unsigned* err, int lane) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    const unsigned v = w.on ? __hip_atomic_load(w.fp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : target;
    if (__all((int)(v - target) >= 0)) break;
    if (__builtin_amdgcn_s_memrealtime() - t0 > SRK_CHAIN_WAIT_TICKS) {
      if (lane == 0) __hip_atomic_store(err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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
// Claims the device for one chain launch of `tiles` workgroups on `st`: 1 = go (A->epoch / flags / err / census fields filled for n convs,
// `st` ordered behind the previous chain launch; call srk_chain_end afterwards), 0 = not now (stream capture, forms switched off or backing
// off after a time-out), < 0 = error (SRK_ERR_CHAIN_TIMEOUT: an earlier launch timed out and srk_chain_recover has not been called)
int srk_chain_begin(hipStream_t st, int n, int tiles, srk_chain_args* A);
// start skew of a chain kernel kind ("H16" / "W42"): SRK_<kind>_CHAIN_SKEW_NS (per phase) and SRK_<kind>_CHAIN_SKEW_GROUPS (phases)
void srk_chain_skew_of(const char* kind, unsigned dflt_ns, unsigned dflt_groups, srk_chain_args* A);
int srk_chain_end(hipStream_t st, bool launched);
// true while the chain forms rest after a recovered time-out (sequences go conv by conv); tick: this is a launch attempt, count it off
bool srk_chain_resting(bool tick);
// != 0 while a time-out is pending (1: census, 2: flag wait); a relaxed read of host memory
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
