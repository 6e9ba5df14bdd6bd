// srk_conv_w42.hip -- fused 3x3 convolution via the 2-D Winograd F(2x4, 3x3) ("wino42", wp_format 6) for gfx950.
//
// F(4,3) along the image row (as srk_conv.hip's wino4 kernel) TIMES F(2,3) along the column: an output patch of 2 rows x 4
// columns comes from 4 x 6 = 24 products per (input, output) channel pair instead of 72 -- a THIRD of the direct kernel's
// MFMAs (wino4: half).  fp32 throughout (v_mfma_f32_32x32x2_f32).
//   rows  (F(2,3)):  p0 = d0 - d2, p1 = d1 + d2, p2 = d2 - d1, p3 = d1 - d3;     y0 = m0 + m1 + m2, y1 = m1 - m2 - m3
//   cols  (F(4,3)):  v = B^T p, y = A^T m as in srk_conv.hip (Lavin & Gray);   u = G_h w G_w^T folded into the weight packing
//
// Shape of the kernel: the fp32 matrix pipe is slow (64 cycles per 32x32x2 MFMA), so ONE wave per SIMD has ~14 issue slots per
// MFMA and the whole 512-entry register file: workgroup = 4 waves = 32 x 16 output pixels (64 patches = two 32-patch M tiles)
// x 64 output channels x 24 positions = 96 accumulator tiles, 24 per wave (384 registers).  Wave w owns ROW POSITION p_w for
// both M tiles and both channel halves, so
//   * the input transform of a patch is computed once per workgroup (shared by the two channel halves),
//   * every transformed-weight fragment is needed by exactly one wave: it is loaded global -> registers (buffer_load_dwordx2,
//     all L2 hits) and never staged in LDS; only the 34 x 18 raw halo (20 KB per 8-channel chunk, 5 DMA pieces per wave) is,
//   * the four row positions of a patch meet only once, after the K loop, through LDS (y0 / y1 above), which also re-deals the
//     tiles so that every wave ends with the register layout conv_epilogue expects.
// Per 8-channel chunk and wave: 96 MFMAs, 72 packed VALU (transform: 18 per channel pair and M tile), 48 ds_read_b64, 24 buffer_load_dwordx2,
// 5 DMA pieces, 1 barrier.
#include "srk_internal.h"
#include "srk_epilogue.h"
#include "srk_chain.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#ifndef W42_CH_PREFETCH1
#define W42_CH_PREFETCH1 0       // 1: chain form of the 16-row kernel: prefetch across the link boundary (PF in wino42_body).  It has the registers
#endif                           // (no spill), so this is what the 32-row kernel could gain: block 276.0 -> 273.9 / 284.8 -> 277.7 us, step 60.9 -> 61.5 ms
#ifndef W42_CHAIN_SIGNS
#define W42_CHAIN_SIGNS 0        // 1: sign bits (srk_conv_args.signs) in the chain kernels.  Correct (bit-identical to the mask tensors), but with
#endif                           // them compiled in the 32-row chain kernel spills 131 registers around its exchange / epilogue: 455 -> 490 us per block
#ifndef W42_STORE_AUX
#define W42_STORE_AUX 0          // cache-policy bits of the one-conv kernels' 16-byte stores (A/B builds: 2 = nt, 16 = sc1)
#endif
#ifndef W42_LDSW
#define W42_LDSW 2       // the transformed weights reach the MFMAs 0: global -> registers, four phases ahead (rounds 2-3); 1: through the LDS (DMA three
#endif                   // channel pairs ahead into a ring behind the halo buffers); 2: through the LDS in the 16-row form only.  Same-box A/B of a dense
                         // block (tools/debug/ab_w42_ldsw.sh): 32-row form 445 (registers) vs 452 us (ring); 16-row form 276 vs 272 us.  The ring
                         // was built to test whether the weight loads cost LATENCY (the timing-only build without them runs 5 % / 19 %
                         // faster): they do not -- with every load three pairs (3-4 us) ahead and no vmcnt wait on a weight left, nothing changes.
#ifndef W42_XA
#define W42_XA 2         // the gap (relative to slot 6 of a group's second phase) that holds the input transform
#endif
#ifdef SRK_STAMP
__device__ unsigned long long* g_w42_stamps = nullptr;
#define W42_STAMP(k) do { if (threadIdx.x == 0 && g_w42_stamps) { g_w42_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); g_w42_stamps[(blockIdx.x + gridDim.x * blockIdx.y) * 16 + 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#define W42_SEG_BEGIN() unsigned long long seg_t = __builtin_amdgcn_s_memtime(); unsigned long long seg_sum[9] = {0,0,0,0,0,0,0,0,0}
#define W42_SEG_RESET() do { seg_t = __builtin_amdgcn_s_memtime(); } while (0)
#define W42_SEG(k) do { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg_sum[k] += t_ - seg_t; seg_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define W42_SEG_END() do { if (lane == 0 && g_w42_stamps) for (int k_ = 0; k_ < 9; ++k_) g_w42_stamps[(4096 + (blockIdx.x + gridDim.x * blockIdx.y) * 4 + wv) * 16 + k_] = seg_sum[k_]; } while (0)
#else
#define W42_STAMP(k) do { } while (0)
#define W42_SEG_BEGIN() do { } while (0)
#define W42_SEG_RESET() do { } while (0)
#define W42_SEG(k) do { } while (0)
#define W42_SEG_END() do { } while (0)
#endif

// LDS image of the raw halo, per k-half h (4 channels = one float4 per pixel):
//   slot(h, hy, hx) = 640 h + 18 hy + (hy >> 1) + hx          (34 rows x 18 pixels; one pad slot per row pair)
// A lane reads, for its patch (row pair k, column quad t), pixels (2k + d, 4t + j): slot = 37 k + 4 t + const, and the M-tile
// map below makes 37 k + 4 t distinct mod 16 over every 16-lane group of a ds_read_b128 and 2-uniform over the 32-lane groups
// of a ds_read_b64 (the minimum) -- conflict-free reads although patches step by two rows and four pixels.
// NMT = M tiles per workgroup: 2 = the 32 x 16-pixel form described above; 1 = a 16 x 16-pixel form (12 accumulator tiles per wave,
// half the MFMAs per weight load and per barrier) for launches whose 32-row tiles would leave CUs idle (batch 16 at 64 x 64).
// CHAIN: the body as one link of the chain kernel below (srk_chain.h): conv c of the launch A.  Its outputs are stored write-through
// and published through the tile's flag; (c > 0) the chunks of its last 64 input channels are fetched behind the wait for the
// neighbouring tiles' flags of conv c - 1.
// PF (chain form of the 16-row kernel, which has registers to spare -- 428 of 512): a link fetches its successor's first halo chunk and
// first weights in front of its own epilogue (carryP0 = the kernel's copy of those weights); the 32-row kernel cannot afford it.
// drain (CHAIN): a wait of this launch has run into its bound, here or in another tile (srk_chain.h) -- no more waiting in this or any later link.
template <int MODE, int NMT, bool CHAIN = false>
__device__ __forceinline__ void wino42_body(const srk_conv_args& a, const srk_chain_args* A = nullptr, int c = 0, f32x2 (*carryP0)[6][2] = nullptr,
                                            bool* drain = nullptr) {
  constexpr bool PF = CHAIN && NMT == 1 && W42_CH_PREFETCH1 != 0;
  constexpr int TH = 16 * NMT, IH = TH + 2, IW = SRK_TW + 2;
  constexpr int HS4 = NMT == 2 ? 640 : 384;         // slots per k-half: 37 per row pair (17 resp. 9 pairs), padded to whole instructions
  constexpr int BUF4 = 2 * HS4;                     // 1280 float4 = 20 KB (NMT 2) / 768 = 12 KB (NMT 1) per chunk
  constexpr int NPC = BUF4 / 256;                   // DMA pieces per wave and chunk: 5 / 3
  constexpr int EXSLOTS = 16 * NMT;                 // 4 KB slots of the exchange area (64 KB per M tile)
  // weight ring (W42_LDSW): behind the two halo buffers, per wave 4 slots of 6 KB = one (chunk, channel pair) of this wave's row position:
  // [c][k-half][64 output channels] float2 -- 96 KB.  NMT 2: inside the exchange area (idle during the K loop); NMT 1: the area grows to 120 KB.
  constexpr int WRING4 = 2 * BUF4, WSLOT4 = 384, WWAVE4 = 4 * WSLOT4;
  constexpr int SMEM4_EX = (EXSLOTS + 4) * 256;     // 144 KB / 80 KB: exchange + 4 x 4 KB epilogue scratch (>= 2 halo buffers)
  constexpr bool LDSW = W42_LDSW == 1 || (W42_LDSW == 2 && NMT == 1);
  constexpr int SMEM4 = (LDSW && WRING4 + 4 * WWAVE4 > SMEM4_EX) ? WRING4 + 4 * WWAVE4 : SMEM4_EX;
  __shared__ float4 smem[SMEM4];

  int tid_ = threadIdx.x;
  // (CHAIN: everything derived from the lane id is formed again for every conv -- hoisted out of the chain loop it stays live through
  // the K loop, where there is not one register to spare: the allocator then spills accumulator tiles inside the loop)
  if constexpr (CHAIN) asm volatile("" : "+v"(tid_));
  const int tid = tid_, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);      // = row position p of this wave
  const int hl = lane >> 5, l32 = lane & 31;
  const int tilesW = (a.OW + SRK_TW - 1) / SRK_TW, tilesH = (a.OH + TH - 1) / TH;
  int bid = blockIdx.x;
  {
    const int T = gridDim.x;                        // XCD-contiguous tile ranges (see wino4_body)
    if ((T & 7) == 0) bid = (bid & 7) * (T >> 3) + (bid >> 3);
  }
  const int tile = bid;
  const int tx = bid % tilesW; bid /= tilesW;
  const int ty = bid % tilesH; bid /= tilesH;
  const int n = bid;
  const int oh0 = ty * TH, ow0 = tx * SRK_TW, n0 = blockIdx.y * 64;
  const int CoutP = (a.Cout + 31) & ~31;
  const int nq = (a.Cin + 7) >> 3;
  W42_STAMP(0);
  // CHAIN: the first DMA of a chunk of the last 64 channels (chunks nq - 8 ..) is piece 0 of chunk nq - 8, issued in chunk nq - 10
  const int qwait = nq - 10;
  int wait_rc = 0;               // (CHAIN) 1: this wave's flag wait ran into its bound, 2: another tile has given up

  // ---- halo DMA plan: instruction i = wv + 4 j (j = 0..4) fills slots 64 i .. 64 i + 63
  constexpr unsigned OOB = 0x80000000u;
  const int Cps_in = a.Cin >> 2;
  long img_elems = (long)a.H * a.W * a.x_ldc;
  if (MODE == SRK_IN_UNSHUFFLE) img_elems *= 4;
  const float* ximg = a.x + (long)n * img_elems;
  const unsigned xbytes = (unsigned)(img_elems * 4 > 0x7fffffffL ? 0x7fffffffL : img_elems * 4);
  const unsigned wbytes = (unsigned)((long)nq * 96 * CoutP * 8);
  __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, xbytes, 0x00020000);
  __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, wbytes, 0x00020000);
  // weight fragments: [q][e-pair][p][c][h][CoutP] float2; the lane part of the byte offset:
  const unsigned vB = (unsigned)((hl * CoutP + n0 + l32) * 8);
  const unsigned sB_c = (unsigned)(2 * CoutP * 8);              // one column position
  const unsigned sB_ep = 24u * sB_c;                            // one channel pair of the chunk's k-halves
  f32x2 P0own[6][2], P1[6][2];       // weights of channels (0,1) resp. (2,3) of the chunk's k-halves: [c][nh]
  f32x2 (&P0)[6][2] = *[&]() { if constexpr (PF) return carryP0; else return &P0own; }();
  const bool fresh_start = !PF || c == 0;          // (PF, c > 0: the previous link has issued this conv's first weights and halo chunk)
  // the first weights need nothing but the lane id: they are on their way before the halo address arithmetic starts
  static_assert(!(PF && LDSW), "the prefetch across chain links fetches weights into registers");
  // W42_LDSW: DMA instruction cc of pair s (= 2 q + lp) copies rows (cc, k-half 0 / 1) of this wave's block -- 2 x 512 B of the 64 output
  // channels n0.. -- into ring slot s & 3; a lane reads its B operand (cc, nh) back as one ds_read_b64.
  const unsigned wvo = (unsigned)(((lane >> 5) * CoutP + n0) * 8 + (lane & 31) * 16);
  const float* wlds = reinterpret_cast<const float*>(smem + WRING4 + wv * WWAVE4) + hl * 128 + l32 * 2;
  auto wpiece = [&](int s_, auto slotc, int cc) {
    constexpr int SL = decltype(slotc)::value;
    // pairs past the last chunk (the ring runs three pairs ahead) go out through a descriptor of ZERO records, like the halo pieces past
    // the last chunk: the pair offset rides in the scalar offset, which the hardware's range check does not cover -- with the real
    // descriptor they would read up to four pairs (98 KB at 64 output channels) behind the packed weights
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wp), 0, s_ < 2 * nq ? wbytes : 0u, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(wr, (__attribute__((address_space(3))) void*)(smem + WRING4 + wv * WWAVE4 + SL * WSLOT4 + cc * 64), 16,
                                             wvo, (unsigned)s_ * sB_ep + (unsigned)wv * 6u * sB_c + (unsigned)cc * sB_c, 0, 0);
  };
  if (fresh_start) {
    if constexpr (LDSW) {
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) wpiece(0, std::integral_constant<int, 0>{}, cc);
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) wpiece(1, std::integral_constant<int, 1>{}, cc);
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) wpiece(2, std::integral_constant<int, 2>{}, cc);
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) wpiece(3, std::integral_constant<int, 3>{}, cc);
    } else {
      const unsigned so = (unsigned)wv * 6u * sB_c;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          P0[cc][nh] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, vB + nh * 256, so + cc * sB_c, 0));
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  unsigned vo[NPC];
  {
    const int ih0 = oh0 - 1, iw0 = ow0 - 1;
#pragma unroll
    for (int j = 0; j < NPC; ++j) {
      const int slot = (wv + 4 * j) * 64 + lane;
      const int half = slot >= HS4 ? 1 : 0, p = slot - half * HS4;
      const int R = p / 37, rem = p - R * 37;
      const int hy = 2 * R + (rem >= 18 ? 1 : 0), hx = rem >= 18 ? rem - 18 : rem;
      const int ih = ih0 + hy, iw = iw0 + hx;
      unsigned v = OOB;
      if (rem < 36 && hy < IH && ih >= 0 && iw >= 0 && ih < a.H && iw < a.W) {
        long off;
        if (MODE == SRK_IN_UNSHUFFLE) off = ((long)(2 * ih) * (2 * a.W) + 2 * iw) * a.x_ldc + a.x_coff + 4 * half;
        else off = ((long)ih * a.W + iw) * a.x_ldc + a.x_coff + 4 * half;
        v = (unsigned)(off * 4);
      }
      vo[j] = v;
    }
  }
  auto piece = [&](int q, int b, auto jc) {
    constexpr int j = decltype(jc)::value;
    unsigned xso = (unsigned)(8 * q * 4);
    if (MODE == SRK_IN_UNSHUFFLE) {
      const int c8 = 8 * q;
      const int ij = c8 / Cps_in, c = c8 - ij * Cps_in;
      xso = (unsigned)(((long)(ij >> 1) * (2 * a.W) * a.x_ldc + (long)(ij & 1) * a.x_ldc + c) * 4);
    }
    // past the last chunk the piece is issued through a descriptor of ZERO records (every lane out of range: nothing is read,
    // zeros land in a buffer nobody reads): a scalar select instead of a per-lane one -- VALU beside the MFMAs is never free
    if (q >= nq) xso = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ximg), 0, q < nq ? xbytes : 0u, 0x00020000);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + b * BUF4 + (wv + 4 * j) * 64), 16,
                                             vo[j], xso, 0, 0);
  };

  // ---- operand addressing
  // M index i = l32 = 8 q + 4 hb + s  ->  column quad t = hb + 2 (s & 1), row pair k = 8 mt + q + 4 (s >> 1)
  const int mq = l32 >> 3, hb = (l32 >> 2) & 1, ms = l32 & 3;
  const int tcol = hb + 2 * (ms & 1), krow = mq + 4 * (ms >> 1);
  // raw rows of this wave's row position: p0 = d0 - d2, p1 = d1 + d2, p2 = d2 - d1, p3 = d1 - d3
  const int da = wv == 0 ? 0 : (wv == 2 ? 2 : 1), db = wv == 3 ? 3 : (wv == 2 ? 1 : 2);
  const float sgn = wv == 1 ? 1.f : -1.f;
  const int slot0 = hl * HS4 + 37 * krow + 4 * tcol;
  const float* ldsA = reinterpret_cast<const float*>(smem) + (slot0 + 18 * da + (da >> 1)) * 4;
  const float* ldsB = reinterpret_cast<const float*>(smem) + (slot0 + 18 * db + (db >> 1)) * 4;

  f32x16 acc[12 * NMT];
  auto mfma = [&](int t, float va, float vb) {       // t is a constant after unrolling: one of the two statements survives
    // HAZARD: a VALU result needs two wait states before an MFMA may read it, and the compiler's hazard recogniser does not look
    // into inline assembly.  The schedule below forms every operand at least one MFMA slot ahead of its use and the main loop is
    // branch-free (one basic block: nothing is sunk next to its use); tools/check_w42_hazards.py verifies the generated code.
    // (an s_nop 1 inside the asm costs 12-14 cycles per MFMA: the slots are issue-bound)
    if (t < 16) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(va), "v"(vb));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(va), "v"(vb));
  };

  // Transformed operands of a GROUP (channel pair, M tile): VV[c] = column position c of (first, second) channel of the pair.  The
  // transform runs on PACKED fp32 instructions, both channels of the pair at once (tools/ubench/pk_beside_mfma.hip: beside fp32 MFMAs a
  // v_pk_fma_f32 costs what a v_fma_f32 does): 18 instructions per group instead of 18 per phase, and only every second phase has a
  // vector-ALU gap at all.  The raw pairs arrive as ds_read_b64 = aligned register pairs already; constants ride in SGPR pairs.
  f32x2 VVa[6], VVb[6];
  f32x2 ra[6], rb[6];                // raw pixels of the current group: rows a / b of this wave's row position, a channel pair each
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using IN = std::integral_constant<int, -1>;
  using Yes = std::true_type; using No = std::false_type;

  // Input transform of a group: row combination d = ra + sgn rb, then B^T d of F(4,3) in 12 operations
  // (4, 0, -5, 0, 1, 0 | 0, -4, -4, 1, 1, 0 | 0, 4, -4, -1, 1, 0 | 0, -2, -1, 2, 1, 0 | 0, 2, -1, -2, 1, 0 | 0, 4, 0, -5, 0, 1): 18 packed
  // instructions for BOTH channels of the pair (the same fused operations, in the same order, as the scalar form: same bits).
  const f32x2 sgn2 = {sgn, sgn};
  auto xform6 = [&](f32x2 (&VN)[6]) {
    const f32x2 kM4 = {-4.f, -4.f}, k4 = {4.f, 4.f}, kM5 = {-5.f, -5.f}, k2 = {2.f, 2.f}, kM2 = {-2.f, -2.f};
    f32x2 d[6], t1, t2, t3, t4, u0, u5;
#define W42_PKFMA(r, k, a, b) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(k), "v"(a), "v"(b))
#define W42_PKADD(r, a, b) asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b))
#define W42_PKSUB(r, a, b) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b))
#pragma unroll
    for (int j = 0; j < 6; ++j) W42_PKFMA(d[j], sgn2, rb[j], ra[j]);
    W42_PKFMA(t1, kM4, d[2], d[4]); W42_PKFMA(t2, kM4, d[1], d[3]);
    W42_PKSUB(t3, d[4], d[2]); W42_PKSUB(t4, d[3], d[1]);
    W42_PKFMA(u0, kM5, d[2], d[4]); W42_PKFMA(VN[0], k4, d[0], u0);
    W42_PKADD(VN[1], t1, t2); W42_PKSUB(VN[2], t1, t2);
    W42_PKFMA(VN[3], k2, t4, t3); W42_PKFMA(VN[4], kM2, t4, t3);
    W42_PKFMA(u5, kM5, d[3], d[5]); W42_PKFMA(VN[5], k4, d[1], u5);
  };

  // One phase = 12 MFMAs of (channel E of the k-halves, M tile MT) on V / B.  The phases of a chunk come in four GROUPS
  // (channel pair, M tile): the raw pixels of a group are read ONCE as channel pairs (ds_read_b64: half the LDS instructions of
  // dword reads and a 2-way instead of a 4-way bank conflict -- with dword reads the LDS cost 13 cycles of issue each, 1300 per
  // chunk).  In the shadow of the MFMAs, placed by hand (one sched_barrier per MFMA):
  //   first phase of a group   no vector-ALU work at all (MFMAs on the pair's first channel, V[.][0])
  //   second phase of a group  (MFMAs on V[.][1]) slots 0-5: the NEXT group's raw reads (buffer NB, M tile NMT, pair NP; two
  //                            ds_read_b64 each), slot 6 + W42_XA: its packed transform (xform6: both channels) into VN
  //   every slot  (LB) one weight load (dwordx2: a channel pair) of pair lp of chunk lq into BN
  //   slot 11     (DJ >= 0) halo piece DJ of chunk dq into buffer DB: one gather per phase, so that a piece never queues behind
  //               the previous one in the address unit (five in a row cost ~180 cycles each)
  auto phase = [&](auto mtc, auto ec, const f32x2 (&V)[6], const f32x2 (&B)[6][2], auto secondc, auto nbc, auto nmtc, auto npc,
                   f32x2 (&VN)[6], auto lbc, int lq, int lp, f32x2 (&BN)[6][2], auto djc, int dq, auto dbc,
                   auto lsc, auto wsc, int ws) {
    constexpr int MT = decltype(mtc)::value, E = decltype(ec)::value & 1;
    constexpr bool SECOND = decltype(secondc)::value;
    constexpr int off = (decltype(nbc)::value * BUF4 + decltype(nmtc)::value * 296) * 4 + 2 * decltype(npc)::value;
    constexpr bool LB = decltype(lbc)::value;
    constexpr int DJ = decltype(djc)::value, DB = decltype(dbc)::value;
    constexpr int X0 = 6;                                           // transform (second phase of a group only): first slot
    const unsigned so = (unsigned)(2 * lq + lp) * sB_ep + (unsigned)wv * 6u * sB_c;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int c = i >> 1, nh = i & 1;
      mfma((NMT * nh + MT) * 6 + c, V[c][E], B[c][nh][E]);
#ifndef W42_NO_LB      // (-DW42_NO_LB / -DW42_NO_DMA: timing-only ablation builds of tools/debug/run_var.sh -- wrong results)
      if constexpr (LDSW) {
        if (LB) BN[c][nh] = *reinterpret_cast<const f32x2*>(wlds + decltype(lsc)::value * (WSLOT4 * 4) + c * 256 + nh * 64);
        constexpr int WS = decltype(wsc)::value;
        if (WS >= 0 && i < 6) wpiece(ws, std::integral_constant<int, (WS >= 0 ? WS : 0)>{}, i);
      } else {
        if (LB) BN[c][nh] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wrsrc, vB + nh * 256, so + c * sB_c, 0));
      }
#endif
#ifndef W42_NO_DMA
      // (behind the phase's weight loads: vector memory retires in order, and issued in the transform gap, ahead of them, the
      // gather stands between every later weight load and its use -- measured +5 us per launch)
      if (DJ >= 0 && i == 11) piece(dq, DB, std::integral_constant<int, (DJ >= 0 ? DJ : 0)>{});
#endif
      if (SECOND && i < 6) {
        ra[i] = *reinterpret_cast<const f32x2*>(ldsA + off + 4 * i);
        rb[i] = *reinterpret_cast<const f32x2*>(ldsB + off + 4 * i);
      }
      // The transform's VALU work sits in ONE gap per phase, not spread over six: the fp32 MFMA and the vector ALU do not overlap
      // within a wave (tools/ubench/lds_beside_mfma.hip) and every gap that holds any VALU pays a fixed restart on top.  Pure
      // arithmetic floats freely between the (volatile) MFMAs when instructions are selected, so the inputs pass through an
      // empty volatile asm at the head of the gap and the results through one at its end.
      if (SECOND && i == X0 + W42_XA) {
        asm volatile("" : "+v"(ra[0]), "+v"(ra[1]), "+v"(ra[2]), "+v"(ra[3]), "+v"(ra[4]), "+v"(ra[5]),
                          "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(rb[4]), "+v"(rb[5]));
        xform6(VN);
        asm volatile("" : "+v"(VN[0]), "+v"(VN[1]), "+v"(VN[2]), "+v"(VN[3]), "+v"(VN[4]), "+v"(VN[5]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // The stage barrier's counted wait: vector-memory operations retire in order, and behind the last DMA piece of chunk q + 1 a wave
  // has issued exactly the weight loads of ONE phase (6 column positions x 2 channel halves, the LB phase (2,0) resp. (2)), so
  // "all but the W42_LB_PER_PHASE youngest" = every DMA piece of the next chunk has landed.  The literal in the asm below is this
  // constant (static_assert), and tools/check_w42_hazards.py checks the same statement in the disassembly of the shipped object.
  // W42_LDSW: behind the last halo piece of chunk q + 1 a wave has issued exactly the SIX weight DMA instructions of pair 2 q + 5 (phase
  // (3,0) resp. (2)): "all but the 6 youngest".  Everything a phase reads from the weight ring was issued before those pieces, i.e. has
  // landed by the PREVIOUS chunk's wait at the latest -- the weights need no wait of their own.
  constexpr int W42_LB_PER_PHASE = LDSW ? 6 : 6 * 2;
  // LDSW: a BARE barrier.  __syncthreads() carries a workgroup-scope fence, and with LDS-DMA instructions in flight (the weight ring's
  // six youngest) the compiler satisfies that fence with s_waitcnt vmcnt(0): the stage would wait for weights it needs two chunks later.
  // What the barrier orders here is LDS traffic of one CU, which the counted wait + s_barrier order by themselves; the empty asm
  // statements keep the compiler from moving LDS reads across it.
  static_assert(W42_LB_PER_PHASE == (LDSW ? 6 : 12), "the literals of the s_waitcnt vmcnt(6 | 12) in front of the stage barrier");
  auto stage_wait_and_barrier = [&]() {
    if constexpr (LDSW) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      __syncthreads();
    }
  };
  W42_SEG_BEGIN();
  // Chunk q sits in buffer b:  (e, mt) = (0,0) (1,0) | (0,1) (1,1) | (2,0) (3,0) | (2,1) | barrier | (3,1)
  //   weights: P1 <- pair 1 of chunk q during (0,0), P0 <- pair 0 of chunk q + 1 during (2,0): four phases ahead of their use
  //   halo:    pieces 1-4 of chunk q + 1 -> b ^ 1 behind the first four phases; at the barrier chunk q + 1 has landed and every
  //            wave has taken its last raw read of b; piece 0 of chunk q + 2 -> b behind (3,1), which also reads the first group
  //            of chunk q + 1 and forms its first V.
  auto chunk = [&](int q, auto bc) {
    constexpr int b = decltype(bc)::value;
    using Bc = std::integral_constant<int, b>; using Bn = std::integral_constant<int, b ^ 1>;
    // weight ring slots (pair s lives in slot s & 3, and 2 q & 3 = 2 b): P1 <- pair 2 q + 1, P0 <- pair 2 q + 2; filled: pairs 2 q + 4, 2 q + 5
    using S1 = std::integral_constant<int, 2 * b + 1>; using S2 = std::integral_constant<int, (2 * b + 2) & 3>;
    using F0 = std::integral_constant<int, 2 * b>; using F1 = std::integral_constant<int, 2 * b + 1>;
    if constexpr (NMT == 1) {
      // one M tile: (e) = (0) (1) | (2) | barrier | (3); pieces 1, 2 of chunk q + 1 behind (0), (1); piece 0 of chunk q + 2 behind (3)
      phase(I0{}, I0{}, VVa, P0, No{}, Bc{}, I0{}, I0{}, VVb, Yes{}, q, 1, P1, I1{}, q + 1, Bn{}, S1{}, IN{}, 0);
      phase(I0{}, I1{}, VVa, P0, Yes{}, Bc{}, I0{}, I1{}, VVb, No{}, 0, 0, P1, I2{}, q + 1, Bn{}, I0{}, F0{}, 2 * q + 4);
      phase(I0{}, I2{}, VVb, P1, No{}, Bc{}, I0{}, I0{}, VVa, Yes{}, q + 1, 0, P0, IN{}, 0, Bc{}, S2{}, F1{}, 2 * q + 5);
      stage_wait_and_barrier();
      phase(I0{}, I3{}, VVb, P1, Yes{}, Bn{}, I0{}, I0{}, VVa, No{}, 0, 0, P0, I0{}, q + 2, Bc{}, I0{}, IN{}, 0);
    } else {
    phase(I0{}, I0{}, VVa, P0, No{}, Bc{}, I0{}, I0{}, VVb, Yes{}, q, 1, P1, I1{}, q + 1, Bn{}, S1{}, IN{}, 0);
    W42_SEG(0);
    phase(I0{}, I1{}, VVa, P0, Yes{}, Bc{}, I1{}, I0{}, VVb, No{}, 0, 0, P1, I2{}, q + 1, Bn{}, I0{}, F0{}, 2 * q + 4);
    W42_SEG(1);
    phase(I1{}, I0{}, VVb, P0, No{}, Bc{}, I0{}, I0{}, VVa, No{}, 0, 0, P1, I3{}, q + 1, Bn{}, I0{}, IN{}, 0);
    W42_SEG(2);
    phase(I1{}, I1{}, VVb, P0, Yes{}, Bc{}, I0{}, I1{}, VVa, No{}, 0, 0, P1, I4{}, q + 1, Bn{}, I0{}, IN{}, 0);
    W42_SEG(3);
    phase(I0{}, I2{}, VVa, P1, No{}, Bc{}, I0{}, I0{}, VVb, Yes{}, q + 1, 0, P0, IN{}, 0, Bc{}, S2{}, IN{}, 0);   // (behind the last chunk: pair 0 of chunk nq is fetched from the padding srk_packed_floats_wino42 reserves and never used)
    W42_SEG(4);
    phase(I0{}, I3{}, VVa, P1, Yes{}, Bc{}, I1{}, I1{}, VVb, No{}, 0, 0, P0, IN{}, 0, Bc{}, I0{}, F1{}, 2 * q + 5);
    W42_SEG(5);
    phase(I1{}, I2{}, VVb, P1, No{}, Bc{}, I0{}, I0{}, VVa, No{}, 0, 0, P0, IN{}, 0, Bc{}, I0{}, IN{}, 0);
    W42_SEG(6);
    // vector-memory operations retire in order: all but the 12 weight loads of (2,0) = every DMA piece of chunk q + 1
    stage_wait_and_barrier();
    W42_SEG(7);
    phase(I1{}, I3{}, VVb, P1, Yes{}, Bn{}, I0{}, I0{}, VVa, No{}, 0, 0, P0, I0{}, q + 2, Bc{}, I0{}, IN{}, 0);   // (behind the last chunk: a V nobody uses, no branch)
    W42_SEG(8);
    }
  };

  if (fresh_start) {
    piece(0, 0, I0{}); piece(0, 0, I1{}); piece(0, 0, I2{});
    if constexpr (NPC == 5) { piece(0, 0, I3{}); piece(0, 0, I4{}); }
  }
  W42_STAMP(1);
  __builtin_amdgcn_sched_barrier(0);          // the accumulators are cleared while the first chunks are in flight
#pragma unroll
  for (int t = 0; t < 12 * NMT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PF) {
    // c > 0: the chunk and the weights were issued in front of the previous epilogue, which has issued its 16 stores (and its residual /
    // mask loads) behind them.  Vector-memory operations retire in order: with at most 16 outstanding everything older has landed -- no
    // waiting for the write-through acknowledgements here.  (A bare barrier: __syncthreads()' release fence would wait for them after all.)
    static_assert(2 * (2 * NMT) * 4 == 16 || !PF, "conv_epilogue_vec issues NI = NTN x MT x 4 stores");
    if (fresh_start) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  W42_STAMP(2);
  if constexpr (LDSW) {
#pragma unroll
    for (int cc = 0; cc < 6; ++cc)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) P0[cc][nh] = *reinterpret_cast<const f32x2*>(wlds + cc * 256 + nh * 64);
  }
  {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      ra[j] = *reinterpret_cast<const f32x2*>(ldsA + 4 * j);
      rb[j] = *reinterpret_cast<const f32x2*>(ldsB + 4 * j);
    }
    xform6(VVa);
  }
  piece(1, 1, I0{});           // (the state every chunk starts in: piece 0 of the next chunk in flight)
  W42_SEG_RESET();
  {
    int q = 0;
    if constexpr (CHAIN) {
      // Two copies of the loop with the wait BETWEEN them: a conditional wait inside the loop costs spills and reloads around it in
      // every iteration.  Scalar instructions only -- there is no vector register to spare (nq is a multiple of 8, >= 16 behind the
      // first conv: host-checked).
      for (; q < qwait; q += 2) {
        chunk(q, I0{});
        chunk(q + 1, I1{});
      }
      if (c > 0 && !*drain) wait_rc = srk_chain_wait_scalar(A->flags, A->poison, A->wait_ticks, n, ty, tx, tilesH, tilesW, A->epoch + (unsigned)c);
    }
    for (; q + 1 < nq; q += 2) {
      chunk(q, I0{});
      chunk(q + 1, I1{});
    }
    if constexpr (!CHAIN) { if (q < nq) chunk(q, I0{}); }      // (CHAIN: nq is even)
  }
  W42_SEG_END();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // trailing (empty) DMA pieces and weight loads
  __syncthreads();                                        // the exchange below reuses the halo buffers
  // inline-assembly MFMAs are invisible to the compiler's hazard recogniser: the last ones must have left the pipe before VALU reads
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 7" ::: "memory");
  W42_STAMP(3);

  // ---- output transform + exchange.  Wave p holds m_p[mt][nh][c][reg]; reg = 4 q + s of lane half hb is patch
  // (k = 8 mt + q + 4 (s >> 1), t = hb + 2 (s & 1)).  Column transform in registers (6 -> 4), then the four row positions are
  // combined through LDS: wave f collects registers 4 f .. 4 f + 3 of every wave, which are exactly the rows 8 m + 2 f, + 1
  // (m = 2 mt + (s >> 1)) of conv_epilogue's wave f.  Two passes (one per channel half): 128 KB of LDS each.
  f32x16 out[2 * NMT][2];
  f32x4* ex = reinterpret_cast<f32x4*>(smem);
#pragma unroll
  for (int nh = 0; nh < 2; ++nh) {
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int t0 = (NMT * nh + mt) * 6;
        const float m0 = acc[t0 + 0][r], m1 = acc[t0 + 1][r], m2 = acc[t0 + 2][r], m3 = acc[t0 + 3][r], m4 = acc[t0 + 4][r], m5 = acc[t0 + 5][r];
        const float s12 = m1 + m2, d12 = m1 - m2, s34 = m3 + m4, d34 = m3 - m4;
        f32x4 z;
        z[0] = (m0 + s12) + s34;
        z[1] = d12 + 2.f * d34;
        z[2] = s12 + 4.f * s34;
        z[3] = (d12 + 8.f * d34) + m5;
        const int f = r >> 2, s = r & 3;
        ex[(((wv * 4 + f) * NMT + mt) * 4 + s) * 64 + lane] = z;
      }
    __syncthreads();
#pragma unroll
    for (int mt = 0; mt < NMT; ++mt)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const f32x4 z0 = ex[(((0 * 4 + wv) * NMT + mt) * 4 + s) * 64 + lane], z1 = ex[(((1 * 4 + wv) * NMT + mt) * 4 + s) * 64 + lane];
        const f32x4 z2 = ex[(((2 * 4 + wv) * NMT + mt) * 4 + s) * 64 + lane], z3 = ex[(((3 * 4 + wv) * NMT + mt) * 4 + s) * 64 + lane];
        const f32x4 y0 = (z0 + z1) + z2, y1 = (z1 - z2) - z3;
        const int m = 2 * mt + (s >> 1), sx = s & 1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          out[m][nh][4 * sx + c] = y0[c];
          out[m][nh][4 * (sx + 2) + c] = y1[c];
        }
      }
    if (nh == 0) __syncthreads();
  }
  W42_STAMP(4);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (PF) {
    if (c + 1 < A->n) {
      // the next conv's first halo chunk (channels 0..7 of the same input view: an old slice) and first weights, beside this epilogue.
      // The barrier: every wave has taken its last read of the exchange area, which the chunk overwrites.
      __builtin_amdgcn_s_barrier();
      piece(0, 0, I0{}); piece(0, 0, I1{}); piece(0, 0, I2{});
      const srk_conv_args& an = A->c[c + 1];
      const __amdgpu_buffer_rsrc_t wn = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(an.wp), 0, (unsigned)((long)((an.Cin + 7) >> 3) * 96 * CoutP * 8), 0x00020000);
      const unsigned so = (unsigned)wv * 6u * sB_c;
#pragma unroll
      for (int cc = 0; cc < 6; ++cc)
#pragma unroll
        for (int nh = 0; nh < 2; ++nh)
          P0[cc][nh] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(wn, vB + nh * 256, so + cc * sB_c, 0));
    }
  }
  // Sign bits (srk_conv_args.signs, conv_epilogue's SIGNS): not in the one-conv kernels (they change the register allocation of the K
  // loop: 4-68 spills, depending on the instantiation), and in the chain kernels only as a build option (W42_CHAIN_SIGNS above).
  int tile_e = 0;
  if constexpr (CHAIN && W42_CHAIN_SIGNS) {
    tile_e = blockIdx.x;
    const int Tg = gridDim.x;
    if ((Tg & 7) == 0) tile_e = (tile_e & 7) * (Tg >> 3) + (tile_e >> 3);
  }
  conv_epilogue<64, 2 * NMT, false, 16, 8, CHAIN ? 16 : W42_STORE_AUX, CHAIN && W42_CHAIN_SIGNS>(a, out, smem, n, oh0, ow0, n0, wv, lane, EXSLOTS + wv, tile_e);     // (16 = sc1: write-through)
  W42_STAMP(5);
  if constexpr (CHAIN) {
    // publish this conv of the tile: every wave has seen its (write-through) stores acknowledged by memory; the barrier also ends this
    // conv's use of the LDS before the next one's first DMA.
    // (Tried, same-box A/B on the dense block at 32 x 64 x 64, forward / data gradient; five launches 477 / 491 us, this form 455 / 470:
    //  the next conv's first halo chunk issued in front of this epilogue 480 / 496, + its first weights 508 / 522, both behind the
    //  epilogue 481 / 495, the flag raised inside the next conv's K loop instead of here 465 / 480 -- whatever stays live across the
    //  conv boundary costs this kernel more in spills than the overlap returns.)
    if (wait_rc == 1 && lane == 0) srk_chain_give_up(*A);          // (every wave polls for itself: whichever of them ran out of time says so)
    if (wait_rc != 0) *drain = true;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (tid == 0 && c + 1 < A->n) __hip_atomic_store(A->flags + tile, A->epoch + (unsigned)c + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

template <int MODE, int NMT>
__global__ __launch_bounds__(256) void conv3x3_f32_wino42_kernel(const srk_conv_args a) { wino42_body<MODE, NMT>(a); }

// The chain form: a dense block's convolutions (forward or data gradient) in ONE persistent launch, one workgroup per tile for all of
// them (srk_chain.h).  What it saves against five launches: the kernel boundaries and their tails -- workgroups finish a 40-chunk conv
// up to 9 us apart, and a launch waits for its slowest tile where a chain link waits for its neighbours only.  Same arithmetic in the
// same order: bit-identical results.
template <int NMT>
__global__ __launch_bounds__(256) void conv3x3_f32_wino42_chain_kernel(const srk_chain_args A) {
  f32x2 P0[6][2];          // (16-row form: the next conv's first weights, fetched by the previous link)
  srk_chain_skew(A);
  bool drain = false;
  for (int c = 0; c < A.n; ++c) wino42_body<SRK_IN_PLAIN, NMT, true>(A.c[c], &A, c, &P0, &drain);
}

}  // namespace

// M tiles per workgroup: the 32-row form when its workgroups fill the chip (>= 200) and pad the image no more than 16-row tiles
// would, else the 16-row form.  SRK_WINO42_NMT = 1 | 2 / srk_debug_set_wino42_nmt force one (A/B measurements, tests).
static int g_w42_nmt = -1;
extern "C" int srk_debug_set_wino42_nmt(int nmt) { g_w42_nmt = (nmt == 1 || nmt == 2) ? nmt : 0; return SRK_OK; }

int srk_conv_wino42_nmt(const srk_conv_args& a) {
  if (g_w42_nmt < 0) { const char* e = getenv("SRK_WINO42_NMT"); g_w42_nmt = e ? atoi(e) : 0; }
  if (g_w42_nmt == 1 || g_w42_nmt == 2) return g_w42_nmt;
  const int tilesW = srk_div_up(a.OW, SRK_TW), cb = srk_round_up(a.Cout, 64) / 64;
  const bool fills = (long)a.N * srk_div_up(a.OH, 32) * tilesW * cb >= 200;
  const bool tall = srk_round_up(a.OH, 32) == srk_round_up(a.OH, 16);
  return (fills && tall) ? 2 : 1;
}

int srk_launch_conv_wino42(const srk_conv_args& a, hipStream_t st) {
  const int nmt = srk_conv_wino42_nmt(a);
  const int tilesW = srk_div_up(a.OW, SRK_TW), tilesH = srk_div_up(a.OH, 16 * nmt);
  dim3 grid((unsigned)(a.N * tilesH * tilesW), (unsigned)(srk_round_up(a.Cout, 64) / 64));
  if (nmt == 2) {
    if (a.in_mode == SRK_IN_PLAIN) hipLaunchKernelGGL((conv3x3_f32_wino42_kernel<SRK_IN_PLAIN, 2>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_f32_wino42_kernel<SRK_IN_UNSHUFFLE, 2>), grid, dim3(256), 0, st, a);
  } else {
    if (a.in_mode == SRK_IN_PLAIN) hipLaunchKernelGGL((conv3x3_f32_wino42_kernel<SRK_IN_PLAIN, 1>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv3x3_f32_wino42_kernel<SRK_IN_UNSHUFFLE, 1>), grid, dim3(256), 0, st, a);
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

// ---- the chain form: host side
static int g_w42_chain = -1;      // 0: never, 1 (default): wherever the sequence is eligible
extern "C" int srk_debug_set_w42_chain(int mode) { g_w42_chain = (mode == 0 || mode == 1) ? mode : 1; return SRK_OK; }

static bool w42_chain_eligible(const srk_conv_args* args, int n) {
  if (g_w42_chain < 0) { const char* e = getenv("SRK_W42_CHAIN"); g_w42_chain = e ? atoi(e) : 1; }
  if (g_w42_chain <= 0 || n < 2 || n > SRK_CHAIN_MAX || args[0].wp_format != 6) return false;
  const srk_conv_args& f = args[0];
  const int nmt = srk_conv_wino42_nmt(f);
  const long tiles = (long)f.N * srk_div_up(f.H, 16 * nmt) * srk_div_up(f.W, SRK_TW);
  const int cus = srk_chain_cus();
  if (cus <= 0 || tiles > cus || tiles > SRK_CHAIN_FLAGS || !srk_chain_flags_uncached()) return false;      // (the kernel polls with scalar loads)
  for (int c = 0; c < n; ++c) {
    const srk_conv_args& a = args[c];
    if (a.wp_format != 6 || a.Cout != 64 || srk_conv_wino42_nmt(a) != nmt || (((uintptr_t)a.wp) & 15)) return false;
    if (a.x != f.x || a.x_ldc != f.x_ldc || a.x_coff != f.x_coff) return false;      // one input view (a link may prefetch its successor's first chunk)
    // the epilogue's 16-byte path (the one that stores write-through): srk_epilogue.h `vec_out`
    if (a.bias && (((uintptr_t)a.bias) & 15)) return false;
    if (a.r1 && ((a.r1_ldc | a.r1_coff) & 3 || (((uintptr_t)a.r1) & 15))) return false;
    if (a.r2 && ((a.r2_ldc | a.r2_coff) & 3 || (((uintptr_t)a.r2) & 15))) return false;
    if (a.mask && ((a.m_ldc | a.m_coff) & 3 || (((uintptr_t)a.mask) & 15))) return false;
    if ((long)a.H * a.W * a.x_ldc * 4 > 0x7fffffffL) return false;
    if (a.flags & (SRK_CONV_WRITE_SIGNS | SRK_CONV_MASK_SIGNS)) {        // sign bits (srk_conv_args.signs): the chain kernels have them
      if (!W42_CHAIN_SIGNS || !a.signs || (((uintptr_t)a.signs) & 15)) return false;
      if ((a.flags & SRK_CONV_WRITE_SIGNS) && (a.flags & SRK_CONV_MASK_SIGNS)) return false;
      if ((a.flags & SRK_CONV_MASK_SIGNS) && a.mask) return false;
    }
  }
  return srk_chain_pattern_ok(args, n, 4);
}

int srk_conv_w42_chain_would(const srk_conv_args* args, int n) { return (w42_chain_eligible(args, n) && !srk_chain_resting(false)) ? 1 : 0; }

// bytes of one conv's sign-bit buffer when the sequence goes out as a chain kernel (which has sign bits); 0 otherwise
size_t srk_conv_w42_chain_signs_bytes(const srk_conv_args* args, int n) {
  if (!W42_CHAIN_SIGNS || !w42_chain_eligible(args, n)) return 0;
  const srk_conv_args& f = args[0];
  const size_t tiles = (size_t)f.N * srk_div_up(f.H, 16 * srk_conv_wino42_nmt(f)) * srk_div_up(f.W, SRK_TW);
  return tiles * 4 * 64 * 16;
}

int srk_conv_w42_chain_name(const srk_conv_args* args, int n, char* buf, size_t len) {
  (void)n;
  snprintf(buf, len, "conv3x3_f32_wino42_chain_kernel<%d>", srk_conv_wino42_nmt(args[0]));
  return SRK_OK;
}

// 1: launched as one chain kernel, 0: not eligible (nothing launched), < 0: error
int srk_launch_conv_w42_chain(const srk_conv_args* args, int n, hipStream_t st) {
  if (!w42_chain_eligible(args, n)) return 0;
  if (srk_chain_resting(true)) return 0;
  srk_chain_args A;
  const srk_conv_args& f = args[0];
  const int nmt = srk_conv_wino42_nmt(f);
  const dim3 grid((unsigned)(f.N * srk_div_up(f.H, 16 * nmt) * srk_div_up(f.W, SRK_TW)));
  const int rc = srk_chain_begin(st, n, (int)grid.x, 1, &A);
  if (rc != 1) return rc;
  for (int c = 0; c < n; ++c) A.c[c] = args[c];
  if (nmt == 2) hipLaunchKernelGGL(conv3x3_f32_wino42_chain_kernel<2>, grid, dim3(256), 0, st, A);
  else hipLaunchKernelGGL(conv3x3_f32_wino42_chain_kernel<1>, grid, dim3(256), 0, st, A);
  const bool ok = hipGetLastError() == hipSuccess;
  const int rc2 = srk_chain_end(st, ok);
  return ok ? (rc2 ? rc2 : 1) : SRK_ERR_LAUNCH;
}

#ifdef SRK_STAMP
extern "C" int srk_debug_set_w42_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w42_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -5;
}
#endif
