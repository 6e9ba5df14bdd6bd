// srk_wgrad_h16.hip -- weight / bias gradient of the 3x3 convolution with 16-BIT ACTIVATION STORAGE (srk_wgrad_args.precision 3 =
// fp16, 4 = bf16) on v_mfma_f32_32x32x16_{f16,bf16}: x and dy are 16-bit NHWC views, dW / db come out in fp32 (BASELINE configs[4]).
//
// Same decomposition as the fp32 kernels (srk_wgrad.hip): workgroup = 4 waves = one 64 (cout) x 64 (cin) x 9 chunk of a batch of
// problems over a range of pixel tiles, wave (a, b) = nine 32 x 32 accumulator tiles (one per tap), partial blocks per pixel split
// summed by the deterministic reduction kernels.  K is the PIXEL index: a lane needs 8 consecutive pixels of one channel, while
// the tensors are [pixel][channel] -- the tiles stay [pixel][32 ch] in LDS (64-byte rows) and are read with the hardware transpose
// ds_read_b64_tr_b16.  With 16-bit storage the tiles go global -> LDS by DMA (buffer_load ... lds, 16 B = 8 channels per lane, four
// lanes per 64-byte pixel row), issued by the MFMA waves themselves one tile ahead: no loader waves, no conversion pass.
// Pixel tile = 8 rows x 16 columns: per k-step (one image row of the tile, 16 pixels) a dy fragment is read once per tile and an x
// fragment (halo row ri, column shift s) once per (ri, s), feeding the up to three taps (r, s) with ri = kk + r: 72 MFMAs per
// 16 + 60 transposed reads.
#include "srk_internal.h"
#include "srk_wgrad_internal.h"
#include <stdlib.h>

using namespace srkw;

typedef _Float16 wh_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 wh_f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 wh_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 wh_bf16x4 __attribute__((ext_vector_type(4)));
typedef short wh_s16x4 __attribute__((ext_vector_type(4)));

namespace {

template <typename T> struct WH;
template <> struct WH<_Float16> {
  typedef wh_f16x8 v8; typedef wh_f16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};
template <> struct WH<__bf16> {
  typedef wh_bf16x8 v8; typedef wh_bf16x4 v4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};

constexpr int HT_H = W16_TH, HT_P = HT_H * WTW, HT_IW = WTW + 2, HT_NHP = (HT_H + 2) * HT_IW;     // 128 px, 180 halo px
constexpr int HT_DY_SLOTS = 2 * HT_P * 4;                 // [half][128 px][4 x 8 ch]: 1024 16-byte slots = 16 pieces
constexpr int HT_X_SLOTS = 2 * HT_NHP * 4;                // [half][180 px][4 x 8 ch]: 1440 slots = 22.5 pieces
constexpr int HT_DY_PIECES = HT_DY_SLOTS / 64, HT_X_PIECES = (HT_X_SLOTS + 63) / 64;
constexpr int HT_STAGE4 = (HT_DY_PIECES + HT_X_PIECES) * 64;      // 2496 slots = 39,936 B per stage
constexpr int HT_XBASE = HT_DY_PIECES * 64;
constexpr int HT_NJD = HT_DY_PIECES / 4, HT_NJX = (HT_X_PIECES + 3) / 4;
constexpr unsigned HT_OOB = 0x80000000u;

// (a plain function: from inside the kernel template the host pass of hipcc 7.2 silently dropped the kernel's host stub)
__device__ __forceinline__ void wh_dma(__amdgpu_buffer_rsrc_t rs, float4* dst, unsigned vo) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)dst, 16, vo, 0, 0, 0);
}

template <typename T>
__device__ __forceinline__ typename WH<T>::v4 wh_tr_read(const char* p) {
  return __builtin_bit_cast(typename WH<T>::v4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) wh_s16x4*)p));
}

// LD (loader form): ONE workgroup per CU = four MFMA waves (one per SIMD) + four loader waves.  The ablation of the two-workgroups form
// (tools/debug/wgrad_h16_ablate.py) shows MFMAs, DMA issue and fragment reads overlapping only partly in waves that do all three; here
// the MFMA waves issue no vector-memory instruction, the loaders keep TWO tiles in flight (three stage buffers), one barrier per tile.
template <typename T, int DYMODE, bool LD>
__global__ __launch_bounds__(LD ? 512 : 256, 2) void wgrad_h16_kernel(const WBatch B, float* part, float* pbias) {
  typedef typename WH<T>::v8 v8;
  typedef typename WH<T>::v4 v4;
  __shared__ float4 smem[(LD ? 3 : 2) * HT_STAGE4];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wv = wv8 & 3;                 // MFMA wave (a, b) resp. the piece owner among the four waves that issue the DMA
  // XCD-aware id -> (pixel split p, chunk): all chunks of a split read the same x / dy tiles and must meet in ONE L2
  int p, chunk;
  {
    const int id = blockIdx.x, nc = B.n_chunks, pm = B.P & ~7;
    if (id < pm * nc) { const int s_ = id >> 3; p = (id & 7) + 8 * (s_ / nc); chunk = s_ - (s_ / nc) * nc; }
    else { const int r_ = id - pm * nc; p = pm + r_ / nc; chunk = r_ - (r_ / nc) * nc; }
  }
  const WProb& a = B.prob[B.c_prob[chunk]];
  const int cy = B.c_cy[chunk], cz = B.c_cz[chunk];
  const int cin0 = cy * 64, cout0 = cz * 64;
  const int Cps = a.Cout >> 2;
  const int t_begin = p * B.tpb;
  int t_end = t_begin + B.tpb;
  if (t_end > B.total_tiles) t_end = B.total_tiles;
  const T* xb = reinterpret_cast<const T*>(a.x);
  const T* dyb = reinterpret_cast<const T*>(a.dy);
  const int CinP = (a.Cin + 7) & ~7, CoutP = (a.Cout + 7) & ~7;     // channels the caller's views really hold (zero-padded to 8)

  // ---- DMA plan.  Slot = (half * NPX + px) * 4 + g: 8 channels [32 half + 8 g, +8) of pixel px.  Piece i = slots 64 i .. 64 i + 63.
  // per lane and piece: pixel coordinates relative to the tile and the channel offset; the rest is per tile
  int d_py[HT_NJD], d_px[HT_NJD], d_ch[HT_NJD];        // dy pieces of this wave: wv + 4 j
  int x_py[HT_NJX], x_px[HT_NJX], x_ch[HT_NJX];
#pragma unroll
  for (int j = 0; j < HT_NJD; ++j) {
    const int slot = (wv + 4 * j) * 64 + lane;
    const int g = slot & 3, px = (slot >> 2) % HT_P, half = (slot >> 2) / HT_P;
    d_py[j] = px / WTW; d_px[j] = px % WTW;
    const int co = cout0 + 32 * half + 8 * g;
    int ch = co;
    if (DYMODE == SRK_IN_UNSHUFFLE) { const int ij = co / Cps; ch = (co - ij * Cps) | (ij << 24); }    // (ij in the top byte)
    d_ch[j] = co < CoutP ? ch : -1;
  }
#pragma unroll
  for (int j = 0; j < HT_NJX; ++j) {
    const int slot = (wv + 4 * j) * 64 + lane;
    const int g = slot & 3, hp = (slot >> 2) % HT_NHP, half = (slot >> 2) / HT_NHP;
    x_py[j] = hp / HT_IW; x_px[j] = hp % HT_IW;
    const int ci = cin0 + 32 * half + 8 * g;
    x_ch[j] = (slot < HT_X_SLOTS && ci < CinP) ? ci : -1;
  }
  const long x_img = (long)B.H * B.W * a.x_ldc;
  const long dy_img = (long)B.OH * B.OW * a.dy_ldc * (DYMODE == SRK_IN_UNSHUFFLE ? 4 : 1);
  auto stage = [&](int tile, int b) {
    int tt = tile;
    const int tx = tt % B.tilesW; tt /= B.tilesW;
    const int ty = tt % B.tilesH; tt /= B.tilesH;
    const int n = tt;
    const int oh0 = ty * HT_H, ow0 = tx * WTW;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dyb + (long)n * dy_img), 0, (unsigned)(dy_img * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb + (long)n * x_img), 0, (unsigned)(x_img * 2), 0x00020000);
    float4* dst = smem + b * HT_STAGE4;
#pragma unroll
    for (int j = 0; j < HT_NJD; ++j) {
      const int oh = oh0 + d_py[j], ow = ow0 + d_px[j];
      unsigned vo = HT_OOB;
      if (d_ch[j] >= 0 && oh < B.OH && ow < B.OW) {
        if (DYMODE == SRK_IN_UNSHUFFLE) {
          const int ij = d_ch[j] >> 24, c = d_ch[j] & 0xffffff;
          vo = (unsigned)((((2 * oh + (ij >> 1)) * (2 * B.OW) + 2 * ow + (ij & 1)) * a.dy_ldc + a.dy_coff + c) * 2);
        } else {
          vo = (unsigned)(((oh * B.OW + ow) * a.dy_ldc + a.dy_coff + d_ch[j]) * 2);
        }
      }
      wh_dma(drs, dst + (wv + 4 * j) * 64, vo);
    }
#pragma unroll
    for (int j = 0; j < HT_NJX; ++j) {
      if (wv + 4 * j < HT_X_PIECES) {
        const int ih = oh0 - 1 + x_py[j], iw = ow0 - 1 + x_px[j];
        unsigned vo = HT_OOB;
        if (x_ch[j] >= 0 && ih >= 0 && iw >= 0 && ih < B.H && iw < B.W) vo = (unsigned)(((ih * B.W + iw) * a.x_ldc + a.x_coff + x_ch[j]) * 2);
        wh_dma(xrs, dst + HT_XBASE + (wv + 4 * j) * 64, vo);
      }
    }
  };

  // ---- MFMA side
  const int wa = wv & 1, wb = wv >> 1;
  const bool active = (cout0 + 32 * wa < a.Cout) && (cin0 + 32 * wb < a.Cin);
  const bool do_bias = (a.db != nullptr) && cy == 0 && wb == 0 && (cout0 + 32 * wa < a.Cout);
  f32x16 acc[9], accb;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) accb[r] = 0.f;
  v8 ones;
#pragma unroll
  for (int j = 0; j < 8; ++j) ones[j] = (T)1.0f;

  // transposed-read addressing (as wgrad_bf16x3_kernel): 16-lane group g = lane >> 4 reads 4 pixel rows x 16 channels; lane 4 q + pp of
  // the group supplies row q, channels 4 pp .. 4 pp + 3 and RECEIVES channel (lane & 15) of the group's 16, i.e. channel lane & 31.
  const int g = lane >> 4, h = g >> 1, q = (lane & 15) >> 2, pp = lane & 3;
  const int lane_col = (16 * (g & 1) + 4 * pp) * 2;
  const int a_lane = (wa * HT_P + 8 * h + q) * 64 + lane_col;                          // + (16 kk + 4 rd) * 64
  const int b_lane = HT_XBASE * 16 + (wb * HT_NHP + 8 * h + q) * 64 + lane_col;        // + (ri * IW + 4 rd + s) * 64

  if constexpr (LD) {
    if (wv8 >= 4) {
      // ---- loader waves: pieces per tile and wave 4 dy + 6 x (wave 3: 5 x); tiles it + 1, it + 2 in flight while tile it is computed
      if (t_begin < t_end) stage(t_begin, 0);
      if (t_begin + 1 < t_end) stage(t_begin + 1, 1);
      for (int tile = t_begin; tile < t_end; ++tile) {
        // (vector-memory operations retire in order: all but the pieces of the NEXT tile = this tile has landed)
        if (tile + 1 < t_end) { if (wv == 3) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // ... and the MFMA waves are done with tile - 1, whose buffer tile + 2 goes to
        if (tile + 2 < t_end) stage(tile + 2, (tile + 2 - t_begin) % 3);
      }
      return;
    }
  }
  if (!LD && t_begin < t_end) stage(t_begin, 0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = LD ? (tile - t_begin) % 3 : (tile - t_begin) & 1;
    if (!LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (my fragment reads of the previous tile have returned)
    __builtin_amdgcn_s_barrier();
#ifndef WH_NO_DMA       // (-DWH_NO_DMA / -DWH_NO_MFMA: timing-only ablation builds for tools/debug/wgrad_h16_ablate.py -- wrong results)
    if (!LD && tile + 1 < t_end) stage(tile + 1, b ^ 1);
#endif
    const char* buf = reinterpret_cast<const char*>(smem + b * HT_STAGE4);
    if (active) {
      // Fragments are read AHEAD by hand (sched_barrier between the slots): left to the compiler every x fragment was read right in
      // front of the MFMAs that use it (read, wait, 1-3 MFMAs: the LDS latency of 30 fragments per tile on top of 80 MFMAs).
      v8 af[HT_H];
      auto rdB = [&](int i) {                      // x fragment of step i = HT_IW-row ri, column shift s (i = s * (HT_H + 2) + ri)
        const int s = i / (HT_H + 2), ri = i % (HT_H + 2);
        const char* base = buf + b_lane + (ri * HT_IW + s) * 64;
        const v4 h0 = wh_tr_read<T>(base), h1 = wh_tr_read<T>(base + 4 * 64);
        v8 bf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bf[j] = h0[j]; bf[4 + j] = h1[j]; }
        return bf;
      };
      // (LD: one MFMA wave per SIMD -- nothing else covers the LDS latency: five steps ahead in a ring of eight, and the first x fragments
      // in front of all but the first two dy fragments, so that the first MFMAs wait for four reads, not for twenty-two)
      constexpr int NSTEP = 3 * (HT_H + 2), AHEAD = LD ? 5 : 3, RING = LD ? 8 : 4;
      v8 bring[RING];
      auto rdA = [&](int kk) {
        const v4 h0 = wh_tr_read<T>(buf + a_lane + (16 * kk) * 64), h1 = wh_tr_read<T>(buf + a_lane + (16 * kk + 4) * 64);
#pragma unroll
        for (int j = 0; j < 4; ++j) { af[kk][j] = h0[j]; af[kk][4 + j] = h1[j]; }
      };
      if constexpr (LD) {
        rdA(0); rdA(1);
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) bring[i] = rdB(i);
#pragma unroll
        for (int kk = 2; kk < HT_H; ++kk) rdA(kk);
      } else {
#pragma unroll
        for (int kk = 0; kk < HT_H; ++kk) rdA(kk);
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) bring[i] = rdB(i);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_bias) {
#pragma unroll
        for (int kk = 0; kk < HT_H; ++kk) accb = WH<T>::mfma(af[kk], ones, accb);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NSTEP; ++i) {
        const int s = i / (HT_H + 2), ri = i % (HT_H + 2);
        if (i + AHEAD < NSTEP) bring[(i + AHEAD) & (RING - 1)] = rdB(i + AHEAD);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int kk = ri - r;
#ifndef WH_NO_MFMA
          if (kk >= 0 && kk < HT_H) acc[3 * r + s] = WH<T>::mfma(af[kk], bring[i & (RING - 1)], acc[3 * r + s]);
#else
          if (kk >= 0 && kk < HT_H) { acc[3 * r + s][0] += (float)af[kk][0] + (float)bring[i & (RING - 1)][0]; }
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  const int hl = lane >> 5, l32 = lane & 31;
  if (active) {
    float* dst = part + ((size_t)p * B.n_chunks + chunk) * CHUNK_FLOATS;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
        dst[(tap * 64 + 32 * wa + i) * 64 + 32 * wb + l32] = acc[tap][reg];
      }
  }
  if (do_bias && l32 == 0) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
      pbias[((size_t)p * B.n_chunks + chunk) * 64 + 32 * wa + i] = accb[reg];
    }
  }
}

}  // namespace

int srk_launch_wgrad_h16(const WBatch& B, int precision, float* part, float* pbias, hipStream_t st) {
  dim3 grid((unsigned)(B.P * B.n_chunks));
  const bool un = B.dy_mode == SRK_IN_UNSHUFFLE;
  if (B.h16 == 2) {          // the loader form: one workgroup per CU (srk_wgrad.hip plans half as many pixel splits for it)
    if (precision == 3) {
      if (un) hipLaunchKernelGGL((wgrad_h16_kernel<_Float16, SRK_IN_UNSHUFFLE, true>), grid, dim3(512), 0, st, B, part, pbias);
      else hipLaunchKernelGGL((wgrad_h16_kernel<_Float16, SRK_IN_PLAIN, true>), grid, dim3(512), 0, st, B, part, pbias);
    } else {
      if (un) hipLaunchKernelGGL((wgrad_h16_kernel<__bf16, SRK_IN_UNSHUFFLE, true>), grid, dim3(512), 0, st, B, part, pbias);
      else hipLaunchKernelGGL((wgrad_h16_kernel<__bf16, SRK_IN_PLAIN, true>), grid, dim3(512), 0, st, B, part, pbias);
    }
  } else if (precision == 3) {
    if (un) hipLaunchKernelGGL((wgrad_h16_kernel<_Float16, SRK_IN_UNSHUFFLE, false>), grid, dim3(256), 0, st, B, part, pbias);
    else hipLaunchKernelGGL((wgrad_h16_kernel<_Float16, SRK_IN_PLAIN, false>), grid, dim3(256), 0, st, B, part, pbias);
  } else {
    if (un) hipLaunchKernelGGL((wgrad_h16_kernel<__bf16, SRK_IN_UNSHUFFLE, false>), grid, dim3(256), 0, st, B, part, pbias);
    else hipLaunchKernelGGL((wgrad_h16_kernel<__bf16, SRK_IN_PLAIN, false>), grid, dim3(256), 0, st, B, part, pbias);
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
