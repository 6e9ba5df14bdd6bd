// srk_chain.hip -- host side of the chain kernels (srk_chain.h): the per-device flag array, the time-out word, the launch epoch and
// the one-chain-kernel-in-flight rule.
#include "srk_chain.h"
#include <mutex>
#include <stdio.h>
#include <stdlib.h>

namespace {
constexpr int CHAIN_WORDS = SRK_CHAIN_FLAGS + 16;        // flags + the poison word (its own 64 bytes), one uncached block
struct ChainDev {
  unsigned* flags = nullptr;   // one word per tile (device), then the poison word at [SRK_CHAIN_FLAGS]
  unsigned* err = nullptr;     // pinned host word a kernel writes when a wait ran into its bound
  unsigned epoch = 0;
  unsigned wait_ticks = 5000000;   // bound of every wait of a launch (100 MHz ticks): SRK_CHAIN_WAIT_MS, srk_chain_set_wait_us
  int cus = 0;
  hipEvent_t ev = nullptr;     // end of the newest chain launch, once a second stream has shown up
  hipStream_t last = nullptr;
  bool used = false, multi = false, dead = false, uncached = false, fault_reported = false;
  int strikes = 0;             // time-outs recovered from so far
  long off_calls = 0;          // srk_chain_begin calls that still answer "not now" (back-off after a recovered time-out)
  unsigned long long launches = 0, resets = 0;
  unsigned skew_ns[2] = {0, 0}, skew_groups[2] = {1, 1};      // start skew per kernel kind (0: 16-bit, 1: fp32): srk_chain_skew
  bool skew_read = false;
};
ChainDev g_dev[16];
std::mutex g_mu;

unsigned wait_ticks_from_env() {
  const char* e = getenv("SRK_CHAIN_WAIT_MS");          // bound of every wait of a chain launch
  double ms = e ? atof(e) : 50.0;
  if (!(ms >= 0.01)) ms = 0.01;
  if (ms > 40000.0) ms = 40000.0;
  return (unsigned)(ms * 100000.0);
}

ChainDev* chain_dev() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  ChainDev& D = g_dev[dev];
  if (!D.flags && !D.dead) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, dev) != hipSuccess) { D.dead = true; return nullptr; }
    D.cus = pr.multiProcessorCount;
    // uncached device memory: no L2 keeps a copy of a flag, so that even a scalar load (which cannot ask for device scope) sees a store
    // from another XCD; plain device memory if the runtime refuses (then only the kernels that poll with sc1 vector loads chain)
    D.uncached = hipExtMallocWithFlags((void**)&D.flags, CHAIN_WORDS * sizeof(unsigned), hipDeviceMallocUncached) == hipSuccess;
    if (!D.uncached) { (void)hipGetLastError(); D.flags = nullptr; }
    if ((!D.uncached && hipMalloc((void**)&D.flags, CHAIN_WORDS * sizeof(unsigned)) != hipSuccess) ||
        hipMemset(D.flags, 0, CHAIN_WORDS * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void**)&D.err, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      D.dead = true; D.flags = nullptr;
      return nullptr;
    }
    *D.err = 0;
    D.wait_ticks = wait_ticks_from_env();
  }
  return D.dead ? nullptr : &D;
}
}  // namespace

// What the host does to the epoch before a launch of n convs: *reset = 1 if it (and the flag array) must be zeroed first, so that it never
// passes SRK_CHAIN_WRAP.  Pure arithmetic (CPU-tested: tests/test_host_cpu.py); flags compare as (int)(flag - target) >= 0, which is only
// right while |flag - target| < 2^31: with every live value below 2^30 + 8 it always is.
extern "C" int srk_chain_epoch_plan(unsigned epoch, int n, unsigned* epoch_out, int* reset) {
  if (n <= 0 || n > SRK_CHAIN_MAX || !epoch_out || !reset) return SRK_ERR_BAD_ARG;
  *reset = epoch >= SRK_CHAIN_WRAP - (unsigned)n ? 1 : 0;
  *epoch_out = *reset ? 0u : epoch;
  return SRK_OK;
}

int srk_chain_cus() {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  return D ? D->cus : 0;
}

bool srk_chain_flags_uncached() {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  return D && D->uncached;
}

unsigned srk_chain_fault() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return 0;
  const ChainDev& D = g_dev[dev];
  return D.err ? *reinterpret_cast<volatile unsigned*>(D.err) : 0u;
}

const unsigned* srk_chain_fault_word() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  return g_dev[dev].err;
}

namespace {
void skew_from_env(ChainDev* D) {
  static const char* names[2] = {"H16", "W42"};
  for (int k = 0; k < 2; ++k) {
    char name[64];
    snprintf(name, sizeof name, "SRK_%s_CHAIN_SKEW_NS", names[k]);
    const char* e = getenv(name);
    unsigned ns = e ? (unsigned)atoi(e) : 0u;
    snprintf(name, sizeof name, "SRK_%s_CHAIN_SKEW_GROUPS", names[k]);
    e = getenv(name);
    unsigned g = e ? (unsigned)atoi(e) : 4u;
    if (g < 2 || g > 64 || ns > 1000000) { ns = 0; g = 1; }
    D->skew_ns[k] = ns; D->skew_groups[k] = g;
  }
  D->skew_read = true;
}
}  // namespace
// back-off after a recovered time-out: true while the chain forms rest (srk_chain_recover); `tick` counts one sequence call off
bool srk_chain_resting(bool tick) {
  std::lock_guard<std::mutex> lk(g_mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return false;
  ChainDev& D = g_dev[dev];
  if (D.off_calls <= 0) return false;
  if (tick) --D.off_calls;
  return true;
}

int srk_chain_begin(hipStream_t st, int n, int tiles, int kind, srk_chain_args* A) {
  g_mu.lock();
  ChainDev* D = chain_dev();
  if (!D) { g_mu.unlock(); return 0; }
  if (const unsigned code = *reinterpret_cast<volatile unsigned*>(D->err)) {
    // an earlier chain launch timed out.  Nothing more is launched as a chain until the caller has recovered (srk_chain_recover); the
    // word stays set, so that every srk_adam_step already queued or still to come skips its update.
    if (!D->fault_reported) {
      (void)code;
      fprintf(stderr, "libsrk: a conv3x3 chain launch gave up: a neighbouring tile did not publish within the bound (SRK_CHAIN_WAIT_MS; another "
                      "process or a long kernel holding CUs?).  Its results are invalid; optimizer steps are skipped until srk_chain_recover()\n");
      D->fault_reported = true;
    }
    g_mu.unlock();
    return SRK_ERR_CHAIN_TIMEOUT;
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {      // (the epoch would be baked into the graph)
    (void)hipGetLastError();
    g_mu.unlock();
    return 0;
  }
  // at most ONE chain kernel in flight per device: two of them, each holding part of the CUs and waiting for tiles that cannot become
  // resident, would wait for each other (their census waits would time out).  One stream orders its launches by itself; from the first
  // launch on a second stream on, every chain launch is followed by an event the next one (on whatever stream) waits for.  The event of
  // a launch is recorded at ITS end (srk_chain_end), on the stream it went to: no stream handle is kept.
  if (D->used && (D->multi || D->last != st)) {      // (once a second stream has been seen: always -- a stream handle may be reused)
    if (!D->multi) {
      // the first change of stream: the earlier launches left no event behind.  They went to ONE stream, which may be gone by now; the
      // device-wide join below happens once per process.
      if (!D->ev && hipEventCreateWithFlags(&D->ev, hipEventDisableTiming) != hipSuccess) { g_mu.unlock(); return SRK_ERR_LAUNCH; }
      if (hipDeviceSynchronize() != hipSuccess) { g_mu.unlock(); return SRK_ERR_LAUNCH; }
      D->multi = true;
    } else if (hipStreamWaitEvent(st, D->ev, 0) != hipSuccess) { g_mu.unlock(); return SRK_ERR_LAUNCH; }
  }
  int reset = 0;
  srk_chain_epoch_plan(D->epoch, n, &D->epoch, &reset);
  if (reset) {
    if (hipMemsetAsync(D->flags, 0, CHAIN_WORDS * sizeof(unsigned), st) != hipSuccess) { g_mu.unlock(); return SRK_ERR_LAUNCH; }
    ++D->resets;
  }
  (void)tiles;
  A->n = n; A->epoch = D->epoch; A->flags = D->flags; A->err = D->err;
  A->poison = D->flags + SRK_CHAIN_FLAGS; A->wait_ticks = D->wait_ticks;
  if (!D->skew_read) skew_from_env(D);
  A->skew_ticks = D->skew_ns[kind & 1] / 10; A->skew_groups = D->skew_groups[kind & 1];
  D->epoch += (unsigned)n;
  return 1;
}

// test aid / experiment: start skew of a chain kernel kind (0: 16-bit, 1: fp32 F(2x4,3x3)): workgroup b starts (b >> 3) % groups * ns late
extern "C" int srk_debug_chain_skew(int kind, unsigned ns, unsigned groups) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D || kind < 0 || kind > 1) return SRK_ERR_UNSUPPORTED;
  if (!D->skew_read) skew_from_env(D);
  if (groups < 2 || groups > 64 || ns > 1000000) { ns = 0; groups = 1; }
  D->skew_ns[kind] = ns; D->skew_groups[kind] = groups;
  return SRK_OK;
}

int srk_chain_end(hipStream_t st, bool launched) {
  int rc = SRK_OK;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && launched) {
    ChainDev& D = g_dev[dev];
    D.used = true; D.last = st; ++D.launches;
    if (D.multi && hipEventRecord(D.ev, st) != hipSuccess) rc = SRK_ERR_LAUNCH;
  }
  g_mu.unlock();
  return rc;
}

// After SRK_ERR_CHAIN_TIMEOUT: waits for the device (every optimizer step queued so far has skipped itself by then), clears the fault,
// zeroes flags / census / epoch and lets the chain forms rest for a while (64, 512, 4096, ... srk_conv3x3_seq calls: a GPU that is shared
// keeps timing out, one that was only busy once gets its fast path back).  Returns the fault code that was pending (0: none) or < 0.
extern "C" int srk_chain_recover(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D) return 0;
  if (hipDeviceSynchronize() != hipSuccess) return SRK_ERR_LAUNCH;
  const unsigned code = *reinterpret_cast<volatile unsigned*>(D->err);
  if (hipMemset(D->flags, 0, CHAIN_WORDS * sizeof(unsigned)) != hipSuccess) return SRK_ERR_LAUNCH;
  *D->err = 0;
  D->epoch = 0; D->fault_reported = false;
  if (code) {
    const int k = D->strikes < 5 ? D->strikes : 5;
    D->off_calls = 64L << (3 * k);
    ++D->strikes;
  }
  return (int)code;
}

// state of the chain forms on the current device (tests, tools): launches so far, wrap resets, recovered time-outs, calls left in the back-off
extern "C" int srk_chain_stats(unsigned long long* launches, unsigned long long* resets, int* strikes, long* off_calls) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D) return SRK_ERR_UNSUPPORTED;
  if (launches) *launches = D->launches;
  if (resets) *resets = D->resets;
  if (strikes) *strikes = D->strikes;
  if (off_calls) *off_calls = D->off_calls;
  return SRK_OK;
}

// ---- test aids
namespace {
// One workgroup = 4 waves (one per SIMD) with 64 KB of LDS spinning for `ticks` of the 100 MHz counter: beside it a CU has room neither for
// a 16-bit chain workgroup (152 KB of LDS) nor for an fp32 one (all 512 registers of every SIMD) -- what a collective's kernel does to the
// chain forms while it waits for its peers.
__global__ __launch_bounds__(256) void hold_cus_kernel(unsigned long long ticks, unsigned* sink) {
  __shared__ unsigned pad[16384];
  pad[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (sink && pad[(threadIdx.x * 7) & 255] == 0xffffffffu) *sink = 1;          // (keeps the LDS allocation alive)
}
// Fills the WHOLE LDS of every CU with a NaN bit pattern (the LDS is not cleared between kernels): a kernel that reads LDS bytes it never
// wrote -- padding slots of a DMA tile buffer, operands formed behind the last tile -- and lets them reach a result shows up as NaN in the next
// launch instead of passing or failing with whatever the previous kernel left behind.  1024 workgroups of 160 KB: at least one lands on every CU.
__global__ __launch_bounds__(256) void poison_lds_kernel(unsigned* sink) {
  __shared__ unsigned all[40960];
  for (int i = threadIdx.x; i < 40960; i += 256) all[i] = 0x7fc00000u + (unsigned)(i & 0xffff);
  __syncthreads();
  if (sink && all[(threadIdx.x * 37) % 40960] == 0u) *sink = 1;               // (keeps the stores alive)
}
}  // namespace
extern "C" int srk_debug_poison_lds(void* stream) {
  hipLaunchKernelGGL(poison_lds_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (unsigned*)nullptr);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
extern "C" int srk_debug_hold_cus(int workgroups, int usec, void* stream) {
  if (workgroups <= 0 || workgroups > 4096 || usec <= 0 || usec > 5000000) return SRK_ERR_BAD_ARG;
  hipLaunchKernelGGL(hold_cus_kernel, dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream, (unsigned long long)usec * 100ull, (unsigned*)nullptr);
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

// epoch / census count as if `epoch` convs had been chained (to cross SRK_CHAIN_WRAP within a test); clears the back-off when off == 0
extern "C" int srk_debug_chain_set(unsigned epoch, long off_calls) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D) return SRK_ERR_UNSUPPORTED;
  if (hipDeviceSynchronize() != hipSuccess) return SRK_ERR_LAUNCH;
  // (flags older than the new epoch: whatever they hold compares as "not yet" or is overwritten; zero them to keep the invariant simple)
  if (hipMemset(D->flags, 0, CHAIN_WORDS * sizeof(unsigned)) != hipSuccess) return SRK_ERR_LAUNCH;
  D->epoch = epoch; D->off_calls = off_calls;
  if (off_calls == 0) D->strikes = 0;
  return SRK_OK;
}
// bound of every wait of a chain launch in microseconds (0: back to SRK_CHAIN_WAIT_MS / 50 ms); srk.h
extern "C" int srk_chain_set_wait_us(unsigned us) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D) return SRK_ERR_UNSUPPORTED;
  D->wait_ticks = us == 0 ? wait_ticks_from_env() : (us > 40000000u ? 4000000000u : us * 100u);
  return SRK_OK;
}
// the same from the DEVICE side of `stream` (a memset of the word, in stream order): the host's own checks have passed by the time it lands
extern "C" int srk_debug_chain_inject_fault_async(void* stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D) return SRK_ERR_UNSUPPORTED;
  return hipMemsetAsync(D->err, 1, sizeof(unsigned), (hipStream_t)stream) == hipSuccess ? SRK_OK : SRK_ERR_LAUNCH;
}
// pretends a launch timed out (host write of the fault word: what a kernel's *err = code does)
extern "C" int srk_debug_chain_inject_fault(unsigned code) {
  std::lock_guard<std::mutex> lk(g_mu);
  ChainDev* D = chain_dev();
  if (!D) return SRK_ERR_UNSUPPORTED;
  *D->err = code;
  return SRK_OK;
}

bool srk_chain_views_overlap(const void* pa, int lda, int ca, int na, const void* pb, int ldb, int cb, int nb, long px, int esz) {
  if (na <= 0 || nb <= 0) return false;
  if (pa == pb && lda == ldb) return ca < cb + nb && cb < ca + na;
  const uintptr_t e = (uintptr_t)esz;
  const uintptr_t a0 = (uintptr_t)pa + e * (uintptr_t)ca, a1 = (uintptr_t)pa + e * ((uintptr_t)(px - 1) * lda + ca + na);
  const uintptr_t b0 = (uintptr_t)pb + e * (uintptr_t)cb, b1 = (uintptr_t)pb + e * ((uintptr_t)(px - 1) * ldb + cb + nb);
  return a0 < b1 && b0 < a1;
}

bool srk_chain_pattern_ok(const srk_conv_args* args, int n, int esz) {
  if (n < 2 || n > SRK_CHAIN_MAX) return false;
  const srk_conv_args& f = args[0];
  const long px = (long)f.N * f.H * f.W;
  const int line = 128 / esz;                        // elements per 128-byte line
  for (int c = 0; c < n; ++c) {
    const srk_conv_args& a = args[c];
    if (!a.x || !a.y || !a.wp) return false;
    if (a.in_mode != SRK_IN_PLAIN || a.ps_out || a.stride != 1 || a.in_slope != 1.f) return false;
    if (a.N != f.N || a.H != f.H || a.W != f.W || a.OH != f.H || a.OW != f.W || a.Cout > 64 || (a.Cout % 8) || (a.Cin % 64)) return false;
    if (c > 0 && a.Cin < 128) return false;
    // a slice is a whole number of lines on a line boundary: what a neighbour writes never shares a line with what was read before
    if ((a.x_ldc % line) || (a.x_coff % line) || (a.y_ldc % line) || (a.y_coff % line) || (((uintptr_t)a.x | (uintptr_t)a.y) & 127)) return false;
  }
  // Who may touch what.  Conv c fetches its input channels [0, Cin - 64) with PLAIN loads at any time from its start on (the chain kernels
  // stream the old slices in while neighbours are still at the previous conv) and its last 64 behind the wait for conv c - 1.  What conv
  // j <= c - 2 wrote is final by then for this tile and its halo (the tile waited for its neighbours' conv j in link j + 1) and was never
  // read before it was written, so no cache holds an older copy.  Hence:
  //   * no input channel of conv c may be written by conv c itself or by a LATER conv (it could land while a slow neighbour still reads);
  //   * conv c - 1 may write into the last 64 input channels only (that is the hand-over);
  //   * the auxiliary views of any conv (r1, r2, mask: plain loads in the epilogue) are not written by any conv of the sequence;
  //   * no two convs write the same memory; a sign-bit buffer belongs to one conv.
  for (int c = 0; c < n; ++c) {
    const srk_conv_args& a = args[c];
    for (int j = 0; j < n; ++j) {
      const srk_conv_args& w = args[j];
      if (j >= c && srk_chain_views_overlap(w.y, w.y_ldc, w.y_coff, w.Cout, a.x, a.x_ldc, a.x_coff, a.Cin, px, esz)) return false;
      if (j == c - 1 && srk_chain_views_overlap(w.y, w.y_ldc, w.y_coff, w.Cout, a.x, a.x_ldc, a.x_coff, a.Cin - 64, px, esz)) return false;
      if (a.r1 && srk_chain_views_overlap(w.y, w.y_ldc, w.y_coff, w.Cout, a.r1, a.r1_ldc, a.r1_coff, a.Cout, px, esz)) return false;
      if (a.r2 && srk_chain_views_overlap(w.y, w.y_ldc, w.y_coff, w.Cout, a.r2, a.r2_ldc, a.r2_coff, a.Cout, px, esz)) return false;
      if (a.mask && srk_chain_views_overlap(w.y, w.y_ldc, w.y_coff, w.Cout, a.mask, a.m_ldc, a.m_coff, a.Cout, px, esz)) return false;
      if (j > c && srk_chain_views_overlap(w.y, w.y_ldc, w.y_coff, w.Cout, a.y, a.y_ldc, a.y_coff, a.Cout, px, esz)) return false;
      if (j != c && a.signs && a.signs == w.signs) return false;
    }
  }
  return true;
}
