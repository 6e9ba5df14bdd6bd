// srk_chain.hip -- host side of the chain kernels (srk_chain.h): the per-device flag array, the time-out word, the launch epoch and
// the one-chain-kernel-in-flight rule.
#include "srk_chain.h"
#include <mutex>
#include <stdio.h>

namespace {
struct ChainDev {
  unsigned* flags = nullptr;   // one word per tile (device)
  unsigned* err = nullptr;     // pinned host word a kernel writes when a flag wait ran into its time limit
  unsigned epoch = 0;
  int cus = 0;
  hipEvent_t ev = nullptr;     // end of the newest chain launch, once a second stream has shown up
  hipStream_t last = nullptr;
  bool used = false, multi = false, dead = false, uncached = false;
};
ChainDev g_dev[16];
std::mutex g_mu;

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
    D.uncached = hipExtMallocWithFlags((void**)&D.flags, SRK_CHAIN_FLAGS * sizeof(unsigned), hipDeviceMallocUncached) == hipSuccess;
    if (!D.uncached) { (void)hipGetLastError(); D.flags = nullptr; }
    if ((!D.uncached && hipMalloc((void**)&D.flags, SRK_CHAIN_FLAGS * sizeof(unsigned)) != hipSuccess) ||
        hipMemset(D.flags, 0, SRK_CHAIN_FLAGS * sizeof(unsigned)) != hipSuccess ||
        hipHostMalloc((void**)&D.err, sizeof(unsigned), hipHostMallocMapped) != hipSuccess) {
      (void)hipGetLastError();
      D.dead = true; D.flags = nullptr;
      return nullptr;
    }
    *D.err = 0;
  }
  return D.dead ? nullptr : &D;
}
}  // namespace

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

int srk_chain_begin(hipStream_t st, int n, srk_chain_args* A) {
  g_mu.lock();
  ChainDev* D = chain_dev();
  if (!D) { g_mu.unlock(); return 0; }
  if (*reinterpret_cast<volatile unsigned*>(D->err)) {
    // a flag wait of an earlier chain launch ran into its time limit: that launch's results are not to be trusted
    fprintf(stderr, "libsrk: a conv3x3 chain launch timed out waiting for a neighbouring tile (results of that launch are invalid); "
                    "the chain forms are now off for this device (SRK_H16_CHAIN=0 SRK_W42_CHAIN=0 avoid them from the start)\n");
    *D->err = 0;
    D->dead = true;
    g_mu.unlock();
    return SRK_ERR_LAUNCH;
  }
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {      // (the epoch would be baked into the graph)
    (void)hipGetLastError();
    g_mu.unlock();
    return 0;
  }
  // at most ONE chain kernel in flight per device: two of them, each holding part of the CUs and waiting for tiles that cannot become
  // resident, would wait for each other.  One stream orders its launches by itself; from the first launch on a second stream on, every
  // chain launch is followed by an event the next one (on whatever stream) waits for.
  if (D->used && D->last != st) {
    bool ok = D->ev || hipEventCreateWithFlags(&D->ev, hipEventDisableTiming) == hipSuccess;
    if (ok && !D->multi) { ok = hipEventRecord(D->ev, D->last) == hipSuccess; D->multi = ok; }
    ok = ok && hipStreamWaitEvent(st, D->ev, 0) == hipSuccess;
    if (!ok) { g_mu.unlock(); return SRK_ERR_LAUNCH; }
  }
  A->n = n; A->epoch = D->epoch; A->flags = D->flags; A->err = D->err;
  D->epoch += (unsigned)n;
  return 1;
}

int srk_chain_end(hipStream_t st, bool launched) {
  int rc = SRK_OK;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 16 && launched) {
    ChainDev& D = g_dev[dev];
    D.used = true; D.last = st;
    if (D.multi && hipEventRecord(D.ev, st) != hipSuccess) rc = SRK_ERR_LAUNCH;
  }
  g_mu.unlock();
  return rc;
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
    if (srk_chain_views_overlap(a.y, a.y_ldc, a.y_coff, a.Cout, a.x, a.x_ldc, a.x_coff, a.Cin, px, esz)) return false;
    if (c > 0) {
      const srk_conv_args& p = args[c - 1];
      if (srk_chain_views_overlap(p.y, p.y_ldc, p.y_coff, p.Cout, a.x, a.x_ldc, a.x_coff, a.Cin - 64, px, esz)) return false;
    }
  }
  return true;
}
