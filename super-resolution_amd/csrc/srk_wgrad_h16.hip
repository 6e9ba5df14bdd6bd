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

// tile geometry by its height TH (8: both forms; 16: the loader form on launches with enough tiles)
template <int TH> struct WHG {
  static constexpr int H = TH, P = TH * WTW, IW = WTW + 2, NHP = (TH + 2) * IW;        // 128 / 256 px, 180 / 324 halo px
  static constexpr int DY_SLOTS = 2 * P * 4;                 // [half][px][4 x 8 ch] 16-byte slots
  static constexpr int X_SLOTS = 2 * NHP * 4;
  static constexpr int DY_PIECES = DY_SLOTS / 64, X_PIECES = (X_SLOTS + 63) / 64;      // 16 + 22.5 / 32 + 40.5 one-KB pieces
  static constexpr int STAGE4 = (DY_PIECES + X_PIECES) * 64;                           // 39,936 / 74,752 B per stage
  static constexpr int XBASE = DY_PIECES * 64;
  static constexpr int NJD = DY_PIECES / 4, NJX = (X_PIECES + 3) / 4;
};
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
template <typename T, int DYMODE, bool LD, int TH>
__global__ __launch_bounds__(LD ? 512 : 256, 2) void wgrad_h16_kernel(const WBatch B, float* part, float* pbias) {
  typedef typename WH<T>::v8 v8;
  typedef typename WH<T>::v4 v4;
  typedef WHG<TH> G;
  constexpr int NSTAGE = (LD && TH == 8) ? 3 : 2;          // stage buffers (160 KB of LDS: three 8-row stages or two 16-row ones)
  __shared__ float4 smem[NSTAGE * G::STAGE4];
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
  int d_py[G::NJD], d_px[G::NJD], d_ch[G::NJD];        // dy pieces of this wave: wv + 4 j
  int x_py[G::NJX], x_px[G::NJX], x_ch[G::NJX];
#pragma unroll
  for (int j = 0; j < G::NJD; ++j) {
    const int slot = (wv + 4 * j) * 64 + lane;
    const int g = slot & 3, px = (slot >> 2) % G::P, half = (slot >> 2) / G::P;
    d_py[j] = px / WTW; d_px[j] = px % WTW;
    const int co = cout0 + 32 * half + 8 * g;
    int ch = co;
    if (DYMODE == SRK_IN_UNSHUFFLE) { const int ij = co / Cps; ch = (co - ij * Cps) | (ij << 24); }    // (ij in the top byte)
    d_ch[j] = co < CoutP ? ch : -1;
  }
#pragma unroll
  for (int j = 0; j < G::NJX; ++j) {
    const int slot = (wv + 4 * j) * 64 + lane;
    const int g = slot & 3, hp = (slot >> 2) % G::NHP, half = (slot >> 2) / G::NHP;
    x_py[j] = hp / G::IW; x_px[j] = hp % G::IW;
    const int ci = cin0 + 32 * half + 8 * g;
    x_ch[j] = (slot < G::X_SLOTS && ci < CinP) ? ci : -1;
  }
  const long x_img = (long)B.H * B.W * a.x_ldc;
  const long dy_img = (long)B.OH * B.OW * a.dy_ldc * (DYMODE == SRK_IN_UNSHUFFLE ? 4 : 1);
  auto stage = [&](int tile, int b) {
    int tt = tile;
    const int tx = tt % B.tilesW; tt /= B.tilesW;
    const int ty = tt % B.tilesH; tt /= B.tilesH;
    const int n = tt;
    const int oh0 = ty * G::H, ow0 = tx * WTW;
    const __amdgpu_buffer_rsrc_t drs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dyb + (long)n * dy_img), 0, (unsigned)(dy_img * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb + (long)n * x_img), 0, (unsigned)(x_img * 2), 0x00020000);
    float4* dst = smem + b * G::STAGE4;
#pragma unroll
    for (int j = 0; j < G::NJD; ++j) {
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
    for (int j = 0; j < G::NJX; ++j) {
      if (wv + 4 * j < G::X_PIECES) {
        const int ih = oh0 - 1 + x_py[j], iw = ow0 - 1 + x_px[j];
        unsigned vo = HT_OOB;
        if (x_ch[j] >= 0 && ih >= 0 && iw >= 0 && ih < B.H && iw < B.W) vo = (unsigned)(((ih * B.W + iw) * a.x_ldc + a.x_coff + x_ch[j]) * 2);
        wh_dma(xrs, dst + G::XBASE + (wv + 4 * j) * 64, vo);
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
  const int a_lane = (wa * G::P + 8 * h + q) * 64 + lane_col;                          // + (16 kk + 4 rd) * 64
  const int b_lane = G::XBASE * 16 + (wb * G::NHP + 8 * h + q) * 64 + lane_col;        // + (ri * IW + 4 rd + s) * 64

  if constexpr (LD) {
    if (wv8 >= 4) {
      // ---- loader waves: NSTAGE - 1 tiles in flight beside the one being computed.  Vector-memory operations retire in order: with at
      // most the pieces of the tiles issued AFTER this one outstanding, this one has landed (pieces per tile: NJD dy + NJX x, the
      // higher waves one x piece fewer when X_PIECES is not a multiple of four)
      constexpr int DEPTH = NSTAGE - 1;
      const int npc = G::NJD + G::NJX - ((wv + 4 * (G::NJX - 1) < G::X_PIECES) ? 0 : 1);       // my pieces per tile (wave-uniform)
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) if (t_begin + d < t_end) stage(t_begin + d, d);
      for (int tile = t_begin; tile < t_end; ++tile) {
        if (DEPTH == 2 && tile + 1 < t_end) {
          if (npc == G::NJD + G::NJX) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G::NJD + G::NJX) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(G::NJD + G::NJX - 1) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();              // ... and the MFMA waves are done with tile - 1, whose buffer tile + DEPTH goes to
        if (tile + DEPTH < t_end) stage(tile + DEPTH, (tile + DEPTH - t_begin) % NSTAGE);
      }
      return;
    }
  }
  if (!LD && t_begin < t_end) stage(t_begin, 0);
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int b = LD ? (tile - t_begin) % NSTAGE : (tile - t_begin) & 1;
    if (!LD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (my fragment reads of the previous tile have returned)
    __builtin_amdgcn_s_barrier();
#ifndef WH_NO_DMA       // (-DWH_NO_DMA / -DWH_NO_MFMA: timing-only ablation builds for tools/debug/wgrad_h16_ablate.py -- wrong results)
    if (!LD && tile + 1 < t_end) stage(tile + 1, b ^ 1);
#endif
    const char* buf = reinterpret_cast<const char*>(smem + b * G::STAGE4);
    if (active) {
      auto rdX = [&](int ri, int s) {              // x fragment: halo row ri, column shift s
        const char* base = buf + b_lane + (ri * G::IW + s) * 64;
        const v4 h0 = wh_tr_read<T>(base), h1 = wh_tr_read<T>(base + 4 * 64);
        v8 bf;
#pragma unroll
        for (int j = 0; j < 4; ++j) { bf[j] = h0[j]; bf[4 + j] = h1[j]; }
        return bf;
      };
      auto rdY = [&](int kk) {                     // dy fragment: tile row kk
        const v4 h0 = wh_tr_read<T>(buf + a_lane + (16 * kk) * 64), h1 = wh_tr_read<T>(buf + a_lane + (16 * kk + 4) * 64);
        v8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) { f[j] = h0[j]; f[4 + j] = h1[j]; }
        return f;
      };
      // Fragments are read AHEAD by hand (sched_barrier between the slots): left to the compiler every x fragment was read right in
      // front of the MFMAs that use it (read, wait, 1-3 MFMAs: the LDS latency of 30 fragments per tile on top of 80 MFMAs).
      if constexpr (LD) {
        // Halo row ri OUTER, column shift s inner: x fragment (ri, s) meets the dy rows kk = ri - r (r = 0..2), so only FOUR dy fragments
        // are live (rows ri + 1 .. ri - 2: a ring) instead of all TH, and they stream in one row ahead instead of in a burst at the
        // head of the tile.  One MFMA wave per SIMD: nothing else covers the LDS latency -- x fragments five steps ahead (ring of eight).
        constexpr int NR = G::H + 2, NSTEP = 3 * NR, AHEAD = 5, RING = 8;
        v8 ay[4], bx[RING];
        ay[0] = rdY(0);
#pragma unroll
        for (int j = 0; j < AHEAD; ++j) bx[j] = rdX(j / 3, j % 3);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < NSTEP; ++j) {
          const int ri = j / 3, s = j % 3;
          if (j + AHEAD < NSTEP) bx[(j + AHEAD) & (RING - 1)] = rdX((j + AHEAD) / 3, (j + AHEAD) % 3);
          if (s == 0 && ri + 1 < G::H) ay[(ri + 1) & 3] = rdY(ri + 1);
          __builtin_amdgcn_sched_barrier(0);
          if (s == 0 && ri < G::H) { if (do_bias) accb = WH<T>::mfma(ay[ri & 3], ones, accb); }
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int kk = ri - r;
#ifndef WH_NO_MFMA
            if (kk >= 0 && kk < G::H) acc[3 * r + s] = WH<T>::mfma(ay[kk & 3], bx[j & (RING - 1)], acc[3 * r + s]);
#else
            if (kk >= 0 && kk < G::H) { acc[3 * r + s][0] += (float)ay[kk & 3][0] + (float)bx[j & (RING - 1)][0]; }
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
        v8 af[G::H];
        constexpr int NSTEP = 3 * (G::H + 2), AHEAD = 3;      // step i = s * (H + 2) + ri
        v8 bring[4];
#pragma unroll
        for (int kk = 0; kk < G::H; ++kk) af[kk] = rdY(kk);
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) bring[i] = rdX(i % (G::H + 2), i / (G::H + 2));
        __builtin_amdgcn_sched_barrier(0);
        if (do_bias) {
#pragma unroll
          for (int kk = 0; kk < G::H; ++kk) accb = WH<T>::mfma(af[kk], ones, accb);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
          const int s = i / (G::H + 2), ri = i % (G::H + 2);
          if (i + AHEAD < NSTEP) bring[(i + AHEAD) & 3] = rdX((i + AHEAD) % (G::H + 2), (i + AHEAD) / (G::H + 2));
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const int kk = ri - r;
#ifndef WH_NO_MFMA
            if (kk >= 0 && kk < G::H) acc[3 * r + s] = WH<T>::mfma(af[kk], bring[i & 3], acc[3 * r + s]);
#else
            if (kk >= 0 && kk < G::H) { acc[3 * r + s][0] += (float)af[kk][0] + (float)bring[i & 3][0]; }
#endif
          }
          __builtin_amdgcn_sched_barrier(0);
        }
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
  // B.h16: 1 = two workgroups per CU; 2 = loader form (srk_wgrad.hip plans half as many pixel splits for it).  Both on 8-row tiles: the
  // loader form on 16-row tiles (two stage buffers, WHG<16>) measured the same 157 us per dense block
#define WH_LAUNCH(T, LD, TH, NT)                                                                                                       \
  do {                                                                                                                                \
    if (un) hipLaunchKernelGGL((wgrad_h16_kernel<T, SRK_IN_UNSHUFFLE, LD, TH>), grid, dim3(NT), 0, st, B, part, pbias);                  \
    else hipLaunchKernelGGL((wgrad_h16_kernel<T, SRK_IN_PLAIN, LD, TH>), grid, dim3(NT), 0, st, B, part, pbias);                         \
  } while (0)
  if (precision == 3) {
    if (B.h16 == 2) WH_LAUNCH(_Float16, true, 8, 512); else WH_LAUNCH(_Float16, false, 8, 256);
  } else {
    if (B.h16 == 2) WH_LAUNCH(__bf16, true, 8, 512); else WH_LAUNCH(__bf16, false, 8, 256);
  }
#undef WH_LAUNCH
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}
