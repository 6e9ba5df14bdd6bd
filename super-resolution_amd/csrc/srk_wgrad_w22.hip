// srk_wgrad_w22.hip -- the 2-D Winograd weight-gradient kernel ("wino22") for gfx950.  Its own translation unit because it is
// built without SLP vectorisation: the compiler must not pair scalar fp32 operations into v_pk_*_f32 on its own (the pairs it
// finds need register moves to line up: 657 -> 638 us on the dense-block batch with them off in the tile-owner form; the 1-D
// kernels of srk_wgrad.hip measured slightly better WITH them).  The row-owner form writes its packed instructions by hand.
#include "srk_wgrad_internal.h"

using namespace srkw;
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

// ---------------------------------------------------------------------------------------------------------
// 2-D Winograd weight gradient ("wino22"): the transposed F(2,3) algorithm in BOTH image directions.  A 2 x 2 patch of dy and
// the 4 x 4 patch of x around it give the nine taps from 16 products instead of 36 (the 1-D kernel above: 24):
//     U = A g A^T (4 x 4 from the 2 x 2 dy patch),  V = B^T d B (4 x 4 from the x patch),  M_pq += U_pq V_pq  over all patches,
//     dW = G^T M G  once, at the end          (A, B, G: the 1-D matrices of the kernel above, applied along columns then rows)
// i.e. 4/9 of the direct kernel's MFMAs (1-D: 2/3).  The MFMA K index runs over PATCHES.
// Shape: like the wino42 conv kernel, ONE wave per SIMD: workgroup = 4 waves = one 64 x 64 (cout, cin) chunk, wave (wa, wb) =
// one 32 x 32 tile x 16 positions = 16 accumulator tiles (all 256 AGPRs); the operands of a k-step (two patches) are formed in
// registers from 4 + 16 raw LDS dwords per lane (44 VALU per 16 MFMAs), one step ahead, in the shadow of the MFMAs.  Pixel tiles
// of 8 rows x 16 columns (+ halo) go global -> LDS by DMA, two buffers; the tile barrier sits in front of the LAST k-step, whose
// shadow already prepares the first step of the next tile from the other buffer.
constexpr int W22_TP = W22_TH * WTW;                       // 128 dy pixels
constexpr int W22_IH = W22_TH + 2, W22_IW = WTW + 2;       // 10 x 18 halo
constexpr int W22_NHP = W22_IH * W22_IW;                   // 180
constexpr int W22_TILE_FLOATS = 80 * 256;                  // (128 + 180) pixels x 64 channels, padded to 80 DMA slots: 81,920 B per buffer
constexpr int W22_THREADS = 256;
#ifndef W22_XS0
#define W22_XS0 7        // the two MFMA gaps of a k-step that hold its vector-ALU work, and the transform group the second starts with
#define W22_XS1 11
#define W22_XG 5
#endif
#ifndef W22_NPS
#define W22_NPS 1        // row-owner form: DMA piece pairs per k-step (1: a tile's 20 pieces per wave ride on ten k-steps; 2: on five)
#endif
#ifndef W24_PIECE_PLAN
#define W24_PIECE_PLAN 0
#endif
#ifndef W22_DEFAULT_FORM
#define W22_DEFAULT_FORM 2
#endif
#ifndef W22_RG
#define W22_RG 7         // row-owner form: the MFMA gap of a k-step that holds ALL its vector-ALU work (raw reads: gaps 0-5)
#endif

#ifdef SRK_STAMP      // diagnostic build (make stamp; tools/stamp_w22.py): per-workgroup phase stamps.  Only OUTSIDE the tile loop: a chained node
                      // (s_memtime, volatile asm) inside it makes the instruction selector abandon the source order of the builtin MFMAs
__device__ unsigned long long* g_w22_stamps = nullptr;
#define W22_STAMP(k) do { if (wv == 0 && g_w22_stamps) {   /* (wave-uniform test: a lane-divergent branch here made the compiler treat the tile contexts as divergent) */ g_w22_stamps[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); g_w22_stamps[blockIdx.x * 16 + 8 + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
#else
#define W22_STAMP(k) do { } while (0)
#endif

template <int DYMODE, bool ROWS>
__global__ __launch_bounds__(W22_THREADS) void wgrad_f32_wino22_kernel(const WBatch B, float* part, float* pbias) {
  __shared__ __attribute__((aligned(16))) float smem[2 * W22_TILE_FLOATS];     // 163,840 B: all of the LDS
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  W22_STAMP(0);
  const int hl = lane >> 5, l32 = lane & 31;
  const int wa = wv & 1, wb = wv >> 1;
  int p, chunk;
  {
    const int id = blockIdx.x, nc = B.n_chunks, pm = B.P & ~7;      // XCD-aware placement, as in wgrad_f32_wino_kernel
    if (id < pm * nc) { const int s_ = id >> 3; p = (id & 7) + 8 * (s_ / nc); chunk = s_ - (s_ / nc) * nc; }
    else { const int r_ = id - pm * nc; p = pm + r_ / nc; chunk = r_ - (r_ / nc) * nc; }
  }
  const int pi = __builtin_amdgcn_readfirstlane(B.c_prob[chunk]);
  const int cy = __builtin_amdgcn_readfirstlane(B.c_cy[chunk]), cz = __builtin_amdgcn_readfirstlane(B.c_cz[chunk]);
  const WProb& a = B.prob[pi];
  const int cin0 = cy * 64, cout0 = cz * 64;
  const bool active = (cout0 + 32 * wa < a.Cout) && (cin0 + 32 * wb < a.Cin);
  const bool do_bias = (a.db != nullptr) && cy == 0 && wb == 0 && (cout0 + 32 * wa < a.Cout);
  const int Cps = a.Cout >> 2;

  f32x16 acc[16];
#pragma unroll
  for (int t = 0; t < 16; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int t_begin = p * B.tpb;
  int t_end = t_begin + B.tpb;
  if (t_end > B.total_tiles) t_end = B.total_tiles;

  const long x_img = (long)B.H * B.W * a.x_ldc;
  const long dy_img = (long)B.OH * B.OW * a.dy_ldc * (DYMODE == SRK_IN_UNSHUFFLE ? 4 : 1);
  const long xb_l = ((long)(B.H * B.W - 1) * a.x_ldc + a.Cin) * 4, db_l = dy_img * 4;
  const unsigned xbytes = (unsigned)(xb_l > 0x7fffffffL ? 0x7fffffffL : xb_l);
  const unsigned dbytes = (unsigned)(db_l > 0x7fffffffL ? 0x7fffffffL : db_l);
  // ---- DMA plan.  A wave-wide instruction moves 4 consecutive pixels x 16 channel quads (1 KB, 8 cache lines): 32 of dy,
  // 45 of x per tile, 80 slots (the tile buffer is padded to 80 KB so that all four waves own exactly 20: instruction 4 j + wv).
  // Everything that depends on the lane is computed ONCE: the byte offset of the piece relative to the tile origin and its
  // column relative to the tile origin.  Issuing a piece is then 6 instructions, none of them a branch (the k-steps stay ONE
  // basic block, so the hand-placed order below survives): rows outside the image fall out of the per-image buffer range by
  // themselves (negative offsets wrap), only the column needs a test.
  constexpr int NDY = W22_TP * 16 / 64;                            // 32 dy instructions
  constexpr int NINST = NDY + W22_NHP * 16 / 64;                   // 77 live instructions
  constexpr int NPW = 20;
  static_assert(NDY % 4 == 0 && NINST <= 4 * NPW && 4 * NPW * 256 <= W22_TILE_FLOATS, "DMA slots");
  const int c4 = lane & 15, lp = lane >> 4;
  const int co = cout0 + 4 * c4, ci = cin0 + 4 * c4;
  int dyc, dyij = 0;
  if (DYMODE == SRK_IN_UNSHUFFLE) { dyij = co / Cps; dyc = co - dyij * Cps; } else { dyc = co; }
  const bool co_ok = co < a.Cout, ci_ok = ci < a.Cin;
  unsigned rel[NPW];
  int colx[NPW];
#pragma unroll
  for (int j = 0; j < NPW; ++j) {
    const int i = 4 * j + wv;
    if (j < NDY / 4) {
      const int r = i >> 2, c = 4 * (i & 3) + lp;                  // pixel 4i + lp of the 8 x 16 tile
      if (DYMODE == SRK_IN_UNSHUFFLE) rel[j] = (unsigned)((((2 * r + (dyij >> 1)) * (2 * B.OW) + 2 * c + (dyij & 1)) * a.dy_ldc + dyc) * 4);
      else rel[j] = (unsigned)(((r * B.OW + c) * a.dy_ldc + dyc) * 4);
      colx[j] = co_ok ? c : -(1 << 20);
    } else {
      const int hp = 4 * (i - NDY) + lp;                           // halo pixel 0..179 (beyond: the padding slots)
      const int hy = hp / W22_IW, hx = hp - hy * W22_IW;
      rel[j] = (unsigned)((((hy - 1) * B.W + (hx - 1)) * a.x_ldc + ci) * 4);
      colx[j] = (ci_ok && i < NINST) ? hx - 1 : -(1 << 20);
    }
  }
  struct TileCtx { int ow0; unsigned org_dy, org_x; __amdgpu_buffer_rsrc_t xr, dr; };
  struct TilePos { int tx, ty, n; };
  // Tile bookkeeping is wave-uniform scalar work that the instruction selector lets float to the head of the one-block loop
  // body (sched_barriers only bind the machine scheduler), i.e. in front of idle matrix pipes: as tile -> (n, ty, tx) divisions,
  // formed for tile + 1 and again for tile + 2, it was ~240 instructions per tile.  So the position is stepped incrementally (no
  // division in the loop) and the context of tile + 2 is formed ONCE and carried into the next iteration.  (Pinning the
  // arithmetic into empty MFMA slots with an empty volatile asm was tried: the asm nodes make the selector abandon source
  // order for the builtin MFMAs -- every slot emptied, 60 spills.)
  auto tile_pos = [&](int tile) {
    TilePos q;
    int tt = tile;
    q.tx = tt % B.tilesW; tt /= B.tilesW;
    q.ty = tt % B.tilesH; q.n = tt / B.tilesH;
    return q;
  };
  auto pos_step = [&](TilePos& q, bool go) {                      // q <- position of the next tile (unchanged when !go)
    int tx = q.tx + 1, ty = q.ty, n = q.n;
    const bool wx = tx == B.tilesW;
    tx = wx ? 0 : tx; ty += wx ? 1 : 0;
    const bool wy = ty == B.tilesH;
    ty = wy ? 0 : ty; n += wy ? 1 : 0;
    q.tx = go ? tx : q.tx; q.ty = go ? ty : q.ty; q.n = go ? n : q.n;
  };
  auto ctx_offsets = [&](const TilePos& q, TileCtx& c) {
    const int oh0 = q.ty * W22_TH;
    c.ow0 = q.tx * WTW;
    c.org_dy = (unsigned)((DYMODE == SRK_IN_UNSHUFFLE ? (2 * oh0 * 2 * B.OW + 2 * c.ow0) : (oh0 * B.OW + c.ow0)) * a.dy_ldc * 4);
    c.org_x = (unsigned)((oh0 * B.W + c.ow0) * a.x_ldc * 4);
  };
  auto ctx_rsrcs = [&](const TilePos& q, TileCtx& c) {
    c.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + q.n * x_img + a.x_coff), 0, xbytes, 0x00020000);
    c.dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + q.n * dy_img + a.dy_coff), 0, dbytes, 0x00020000);
  };
  // piece j of this wave (j: compile-time after unrolling); `live` false (wave-uniform): issued out of range -- nothing is read,
  // zeros land in a buffer nobody reads any more
  auto piece = [&](const TileCtx& c, int b, int j, bool live) {
    float* dst = smem + b * W22_TILE_FLOATS + (4 * j + wv) * 256;
    const bool isdy = j < NDY / 4;
    const unsigned off = rel[j] + (isdy ? c.org_dy : c.org_x);
    const int ow = live ? c.ow0 : (1 << 28);                       // (dead pieces fail the column test: one scalar select, no mask AND per piece)
    const bool ok = (unsigned)(colx[j] + ow) < (unsigned)(isdy ? B.OW : B.W);
    wdma16(isdy ? c.dr : c.xr, dst, ok ? off : W_OOB);
  };

  if constexpr (!ROWS) {
  // per-lane LDS offsets (floats) of patch (row 0, columns 2 * 0 + hl)
  const int aoff = (2 * hl) * 64 + 32 * wa + l32;                              // dy pixel (0, 2 hl)
  const int boff = W22_TP * 64 + (2 * hl) * 64 + 32 * wb + l32;                // halo pixel (0, 2 hl)

  // K-step order: DOWN the patch columns, kk = 4 * (column pair) + (patch row).  Vertically adjacent patches share two of their
  // four x rows, so for patch rows 1-3 only the two new rows are read and column-transformed (12 instead of 20 LDS dwords, 36
  // instead of 44 VALU per step); the transformed rows 2, 3 of the previous step are carried in registers as rows 0, 1.
  float U0[16], V0[16], U1[16], V1[16], d[16];
  auto raw_ptrs = [&](int bb, int kk, const float*& ap, const float*& bp) {
    const int pr = kk & 3, cp = kk >> 2;
    ap = smem + bb * W22_TILE_FLOATS + aoff + ((2 * pr) * 16 + 4 * cp) * 64;
    bp = smem + bb * W22_TILE_FLOATS + boff + ((2 * pr) * W22_IW + 4 * cp) * 64;
  };
  // the VALU operations of the operand transform in groups of 4 (one group per MFMA slot).  The four dy values are loaded
  // straight into U[0], U[3], U[12], U[15] (the corner positions ARE the raw values); the column-transformed x rows live in two
  // register sets that swap roles from patch row to patch row (rows 2, 3 of one patch are rows 0, 1 of the next: no copies).
  float a01, a02, a11, a12, bx[8], by[8];
  auto coltf = [&](float (&o)[8], int h, int r) {                 // o[4h ..] <- column transform of raw row r
    o[4 * h + 0] = d[4 * r + 0] - d[4 * r + 2]; o[4 * h + 1] = d[4 * r + 1] + d[4 * r + 2];
    o[4 * h + 2] = d[4 * r + 2] - d[4 * r + 1]; o[4 * h + 3] = d[4 * r + 1] - d[4 * r + 3];
  };
  auto xform = [&](int grp, int pr, float (&U)[16], float (&V)[16], bool count_bias) {
    float (&up)[8] = (pr & 1) ? by : bx;                           // transformed rows 0, 1 of this patch
    float (&lo)[8] = (pr & 1) ? bx : by;                           // rows 2, 3 (new)
    if (grp == 0) { a01 = U[0] + U[3]; a02 = U[0] - U[3]; a11 = U[12] + U[15]; a12 = U[12] - U[15]; }
    if (grp == 1) { U[4] = U[0] + U[12]; U[5] = a01 + a11; U[6] = a02 + a12; U[7] = U[3] + U[15]; }
    if (grp == 2) {
      U[8] = U[0] - U[12]; U[9] = a01 - a11; U[10] = a02 - a12; U[11] = U[3] - U[15];
      U[1] = a01; U[2] = a02; U[13] = a11; U[14] = a12;
      // bias: the patch sum IS U[5]; every wave adds (one instruction, no per-wave select in the loop), the bias waves store.
      // count_bias is a literal `true` except in the last k-step of a tile (the operands formed behind the last tile are not real)
      bsum += count_bias ? a01 + a11 : 0.f;
    }
    if (grp == 3 && pr == 0) coltf(up, 0, 0);
    if (grp == 4 && pr == 0) coltf(up, 1, 1);
    if (grp == 5) coltf(lo, 0, 2);
    if (grp == 6) coltf(lo, 1, 3);
    if (grp == 7) { for (int q = 0; q < 4; ++q) V[q] = up[q] - lo[q]; }
    if (grp == 8) { for (int q = 0; q < 4; ++q) V[4 + q] = up[4 + q] + lo[q]; }
    if (grp == 9) { for (int q = 0; q < 4; ++q) V[8 + q] = lo[q] - up[4 + q]; }
    if (grp == 10) { for (int q = 0; q < 4; ++q) V[12 + q] = up[4 + q] - lo[4 + q]; }
  };
  // One k-step = 16 MFMAs on (U, V).  Between them, by hand (one sched_barrier per MFMA): gaps 0-4 the raw reads of the NEXT step
  // (buffer nb, step nk; rows 0, 1 of x only at the top of a column); gaps W22_XS0 and W22_XS1 its transform into (UN, VN) and
  // one DMA piece each (2 dj, 2 dj + 1) of the tile described by dc into buffer db
  TilePos pos2;                                                    // position / context of the newest tile known (tile + 2 at most)
  TileCtx c2;
  auto kstep = [&](const float (&U)[16], const float (&V)[16], int nb, int nk, float (&UN)[16], float (&VN)[16],
                   const TileCtx& dc, int db, int dj, bool dlive, bool next_real, int mk = 0, bool mk_go = false) {
    const float *ap, *bp;
    raw_ptrs(nb, nk, ap, bp);
    const bool top = (nk & 3) == 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(U[i], V[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);                           // (the gap's work stays BEHIND its MFMA: VALU in front of it would open a second VALU gap)
#ifndef W22_NO_LDS     // (-DW22_NO_LDS / NO_XFORM / NO_DMA: timing-only ablation builds of tools/debug/run_wgvar.sh -- wrong results)
      if (i == 0) { UN[0] = ap[0]; UN[3] = ap[64]; UN[12] = ap[16 * 64]; UN[15] = ap[17 * 64]; }
      if (i == 1 || i == 2 || (top && (i == 3 || i == 4))) {
        const int r = i <= 2 ? i + 1 : i - 3;                      // rows 2, 3 first, then (top) rows 0, 1
#pragma unroll
        for (int j = 0; j < 4; ++j) d[4 * r + j] = bp[(r * W22_IW + j) * 64];
      }
#endif
      if (mk == 1 && i == 3) { pos_step(pos2, mk_go); ctx_offsets(pos2, c2); ctx_rsrcs(pos2, c2); }
      // ALL vector-ALU work of the step sits in W22_XSLOTS gaps (tools/ubench/lds_beside_mfma.hip: the fp32 MFMA and the vector
      // ALU do not overlap within a wave -- a gap costs 64 + 14 + 4 n cycles for n VALU instructions, so the 14 are paid per gap
      // that holds any; LDS reads and DMA issue are free beside the MFMAs)
#ifndef W22_NO_XFORM
      if (i == W22_XS0) {
#pragma unroll
        for (int g = 0; g < W22_XG; ++g) xform(g, nk & 3, UN, VN, next_real);
      }
      if (i == W22_XS1) {
#pragma unroll
        for (int g = W22_XG; g < 11; ++g) xform(g, nk & 3, UN, VN, next_real);
      }
#endif
#ifndef W22_NO_DMA
      if (dj >= 0 && i == W22_XS0) piece(dc, db, 2 * dj, dlive);
      if (dj >= 0 && i == W22_XS1) piece(dc, db, 2 * dj + 1, dlive);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  TileCtx cn;                                                      // context of tile + 1 at the head of every iteration
  if (t_begin < t_end) {
    pos2 = tile_pos(t_begin);
    TileCtx c0;
    ctx_offsets(pos2, c0); ctx_rsrcs(pos2, c0);
#pragma unroll
    for (int j = 0; j < NPW; ++j) piece(c0, 0, j, true);
    pos_step(pos2, t_begin + 1 < t_end);
    ctx_offsets(pos2, cn); ctx_rsrcs(pos2, cn);
    c2 = cn;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0)
  __syncthreads();
  if (t_begin < t_end) {
    if (t_begin + 1 < t_end) { piece(cn, 1, 0, true); piece(cn, 1, 1, true); }   // (the state every tile starts in: pieces 0, 1 of the next one issued)
    {
      const float *ap, *bp;
      raw_ptrs(0, 0, ap, bp);
      U0[0] = ap[0]; U0[3] = ap[64]; U0[12] = ap[16 * 64]; U0[15] = ap[17 * 64];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) d[4 * r + j] = bp[(r * W22_IW + j) * 64];
    }
#pragma unroll
    for (int grp = 0; grp < 11; ++grp) xform(grp, 0, U0, V0, true);
  }
  int b = 0;
  W22_STAMP(1);
  for (int tile = t_begin; tile < t_end; ++tile) {
    // tile sits in buffer b, tile + 1 is in flight into b ^ 1 (issued during the previous tile); tile + 2 goes into b once the
    // barrier in front of the last k-step has released it: its pieces ride on that step and on steps 0-8 of the next tile.
    // cn: the tile whose pieces 2-19 are issued during THIS tile's steps 0-8 (past the end: any valid context, pieces dead).
    const bool more1 = tile + 1 < t_end, more2 = tile + 2 < t_end;
    const bool livep = more1;
    const int bnp = b ^ 1;
#pragma unroll
    for (int kk = 0; kk < 14; kk += 2) {
      kstep(U0, V0, b, kk + 1, U1, V1, cn, bnp, kk < 9 ? kk + 1 : -1, livep, true, kk == 10 ? 1 : 0, more2);
      kstep(U1, V1, b, kk + 2, U0, V0, cn, bnp, kk + 1 < 9 ? kk + 2 : -1, livep, true, 0, false);
    }
    kstep(U0, V0, b, 15, U1, V1, cn, bnp, -1, false, true);        // k-step 14
    __builtin_amdgcn_s_waitcnt(0x0070);                            // every piece of tile + 1 has landed (no other VMEM in flight)
    __builtin_amdgcn_s_barrier();
    kstep(U1, V1, bnp, 0, U0, V0, c2, b, 0, more2, more1);         // k-step 15: pieces 0, 1 of tile + 2; first operands of tile + 1
    cn = c2;
    b ^= 1;
  }
  W22_STAMP(2);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();

  // G^T M G: taps from the 4 x 4 positions (columns q first, then rows p)
  f32x16 tap[9];
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    float T[4][3];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const float m0 = acc[4 * pp][e], m1 = acc[4 * pp + 1][e], m2 = acc[4 * pp + 2][e], m3 = acc[4 * pp + 3][e];
      const float hs = 0.5f * (m1 + m2);
      T[pp][0] = m0 + hs; T[pp][1] = 0.5f * (m1 - m2); T[pp][2] = hs - m3;
    }
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      const float hs = 0.5f * (T[1][sx] + T[2][sx]);
      tap[sx][e] = T[0][sx] + hs;
      tap[3 + sx][e] = 0.5f * (T[1][sx] - T[2][sx]);
      tap[6 + sx][e] = hs - T[3][sx];
    }
  }
  if (active) {
    float* dst = part + ((size_t)p * B.n_chunks + chunk) * CHUNK_FLOATS;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * hl;
        dst[(t * 64 + 32 * wa + i) * 64 + 32 * wb + l32] = tap[t][reg];
      }
  }
  if (do_bias) {
    const float tot = bsum + __shfl_xor(bsum, 32);
    if (hl == 0) pbias[((size_t)p * B.n_chunks + chunk) * 64 + 32 * wa + l32] = tot;
  }
  } else {
  // =====================================================================================================================
  // ROW-OWNER form.  Wave wv owns ROW wv of the 4 x 4 positions for the whole 64 x 64 chunk (2 cout tiles x 2 cin tiles x 4
  // positions = the same 16 accumulator tiles), instead of all 16 positions of one 32 x 32 tile.  A row of U = A g A^T and of
  // V = B^T d B needs only ONE combination of two raw rows (row pass first: t_j = d[ra][j] + s d[rb][j], with (ra, rb, s) wave-
  // uniform -- two LDS base offsets and one scalar coefficient, so the four waves run the SAME code), then the four column
  // combinations: 4 + 4 operations per cin tile, 2 + 2 per cout tile = 24 per 16 MFMAs instead of 36-44 -- as 12 PACKED instructions
  // (below) -- for 24 instead of 12-20 LDS dwords (free beside fp32 MFMAs, which vector-ALU work is not).  The rows meet in
  // G^T M G at the end, through the LDS.
  //   B^T rows: (d0 - d2, d1 + d2, d2 - d1, d1 - d3);  A rows: (g0, g0 + g1, g0 - g1, g1)
  const int xra = wv == 0 ? 0 : (wv == 2 ? 2 : 1);
  const int xrb = wv == 0 ? 2 : (wv == 1 ? 2 : (wv == 2 ? 1 : 3));
  const float xs = wv == 1 ? 1.f : -1.f;
  const int dra = wv == 3 ? 1 : 0;
  const int drb = (wv == 1 || wv == 2) ? 1 : dra;
  const float dsc = wv == 1 ? 1.f : (wv == 2 ? -1.f : 0.f);
  const int aoffA = (2 * hl) * 64 + l32 + dra * 16 * 64, aoffB = (2 * hl) * 64 + l32 + drb * 16 * 64;
  // (the second cout tile's dy base is opaque: seen as aoffA + 32, the two tiles' reads are paired into ds_read2_b32, whose 8-bit
  // offsets need a v_add per k-step for the base -- a third vector-ALU gap; unrelated bases pair along j, as ds_read2st64_b32
  // with the whole step offset in the instruction)
  int aoffA1 = aoffA + 32, aoffB1 = aoffB + 32;
  asm volatile("" : "+v"(aoffA1), "+v"(aoffB1));
  const int boffA = W22_TP * 64 + (2 * hl) * 64 + l32 + xra * W22_IW * 64, boffB = W22_TP * 64 + (2 * hl) * 64 + l32 + xrb * W22_IW * 64;
  float bs0 = 0.f, bs1 = 0.f;
  // The transform runs on PACKED fp32 instructions (tools/ubench/pk_beside_mfma.hip: beside fp32 MFMAs a v_pk_add_f32 / v_pk_fma_f32
  // costs what a v_add_f32 does, so two results per instruction halve the bill): the pairs are the two pixels j, j + 1 that one
  // ds_read2st64_b32 delivers into an aligned register pair; op_sel / neg modifiers do the broadcasts and signs of the column pass.
  // Plain (non-volatile) asm: no chain, so the selector keeps the source order of the builtin MFMAs.
  //   operands: U[4 m + q] (cout tile m), V[4 n + q] (cin tile n); raw pairs gA/gB[m] = rows ra / rb, pixels (0, 1); dA/dB[2 n + h] =
  //   rows ra / rb, pixels (2 h, 2 h + 1)
  float U0[8], V0[8], U1[8], V1[8];
  f32x2 gA[2], gB[2], dA[4], dB[4];
  const f32x2 xs2 = {xs, xs}, dsc2 = {dsc, dsc};
  auto rd_g = [&](int bb, int kk) {
    const int pr = kk & 3, cp = kk >> 2;
    const int o = bb * W22_TILE_FLOATS + ((2 * pr) * 16 + 4 * cp) * 64;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        gA[m][j] = smem[(m ? aoffA1 : aoffA) + o + 64 * j];
        gB[m][j] = smem[(m ? aoffB1 : aoffB) + o + 64 * j];
      }
  };
  auto rd_d = [&](int bb, int kk, int n, bool second) {
    const int pr = kk & 3, cp = kk >> 2;
    const float* q = smem + bb * W22_TILE_FLOATS + (second ? boffB : boffA) + ((2 * pr) * W22_IW + 4 * cp) * 64 + 32 * n;
#pragma unroll
    for (int j = 0; j < 4; ++j) (second ? dB : dA)[2 * n + (j >> 1)][j & 1] = q[64 * j];
  };
  auto xf_u = [&](float (&U)[8], bool count_bias) {                // 2 packed + 1 (bias: row 1, position 1 IS the patch sum) per cout tile
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      f32x2 t, u12;
      asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(t) : "v"(dsc2), "v"(gB[m]), "v"(gA[m]));                    // (t0, t1) = gA + s gB
      asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_hi:[0,1]" : "=v"(u12) : "v"(t));      // (t0 + t1, t0 - t1)
      U[4 * m + 0] = t[0]; U[4 * m + 1] = u12[0]; U[4 * m + 2] = u12[1]; U[4 * m + 3] = t[1];
    }
    bs0 += count_bias ? U[1] : 0.f;
    bs1 += count_bias ? U[5] : 0.f;
  };
  f32x2 tt[4];
  auto xf_vrow = [&](int n) {                                      // 2 packed: t_j = dA_j + s dB_j
#pragma unroll
    for (int h = 0; h < 2; ++h) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(tt[2 * n + h]) : "v"(xs2), "v"(dB[2 * n + h]), "v"(dA[2 * n + h]));
  };
  auto xf_vcol = [&](float (&V)[8], int n) {                       // 2 packed: (t0 - t2, t1 + t2), (t2 - t1, t1 - t3)
    f32x2 v01, v23;
    asm("v_pk_add_f32 %0, %1, %2 op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(v01) : "v"(tt[2 * n]), "v"(tt[2 * n + 1]));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(v23) : "v"(tt[2 * n + 1]), "v"(tt[2 * n]));
    V[4 * n + 0] = v01[0]; V[4 * n + 1] = v01[1]; V[4 * n + 2] = v23[0]; V[4 * n + 3] = v23[1];
  };
  TilePos pos2;
  TileCtx c2;
  auto kstep = [&](const float (&U)[8], const float (&V)[8], int nb, int nk, float (&UN)[8], float (&VN)[8],
                   const TileCtx& dc, int db, int dj, bool dlive, bool next_real, int mk = 0, bool mk_go = false) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {                                 // accumulator tile i = 4 q + 2 m + n
      acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(U[4 * ((i >> 1) & 1) + (i >> 2)], V[4 * (i & 1) + (i >> 2)], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#ifndef W22_NO_LDS
      if (i == 0) rd_g(nb, nk);
      if (i == 1) rd_d(nb, nk, 0, false);
      if (i == 2) rd_d(nb, nk, 0, true);
      if (i == 4) rd_d(nb, nk, 1, false);
      if (i == 5) rd_d(nb, nk, 1, true);
#endif
      if (mk == 1 && i == 3) { pos_step(pos2, mk_go); ctx_offsets(pos2, c2); ctx_rsrcs(pos2, c2); }
      // ONE vector-ALU gap per k-step: the whole transform (12 packed + 2 scalar instructions) and both DMA pieces' offset selects
#ifndef W22_NO_XFORM
      if (i == W22_RG) { xf_u(UN, next_real); xf_vrow(0); xf_vcol(VN, 0); xf_vrow(1); xf_vcol(VN, 1); }
#endif
#ifndef W22_NO_DMA
      if (dj >= 0 && dj < NPW / (2 * W22_NPS) && i == W22_RG) {
#pragma unroll
        for (int u = 0; u < 2 * W22_NPS; ++u) piece(dc, db, 2 * W22_NPS * dj + u, dlive);
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  TileCtx cn;
  if (t_begin < t_end) {
    pos2 = tile_pos(t_begin);
    TileCtx c0;
    ctx_offsets(pos2, c0); ctx_rsrcs(pos2, c0);
#pragma unroll
    for (int j = 0; j < NPW; ++j) piece(c0, 0, j, true);
    pos_step(pos2, t_begin + 1 < t_end);
    ctx_offsets(pos2, cn); ctx_rsrcs(pos2, cn);
    c2 = cn;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();
  if (t_begin < t_end) {
    if (t_begin + 1 < t_end) {
#pragma unroll
      for (int u = 0; u < 2 * W22_NPS; ++u) piece(cn, 1, u, true);
    }
    rd_g(0, 0);
    rd_d(0, 0, 0, false); rd_d(0, 0, 0, true); rd_d(0, 0, 1, false); rd_d(0, 0, 1, true);
    xf_u(U0, true);
    xf_vrow(0); xf_vcol(V0, 0); xf_vrow(1); xf_vcol(V0, 1);
  }
  int b = 0;
  W22_STAMP(1);
  for (int tile = t_begin; tile < t_end; ++tile) {                 // (the tile pipeline of the form above)
    const bool more1 = tile + 1 < t_end, more2 = tile + 2 < t_end;
    const bool livep = more1;
    const int bnp = b ^ 1;
#pragma unroll
    for (int kk = 0; kk < 14; kk += 2) {
      kstep(U0, V0, b, kk + 1, U1, V1, cn, bnp, kk < 9 ? kk + 1 : -1, livep, true, kk == 10 ? 1 : 0, more2);
      kstep(U1, V1, b, kk + 2, U0, V0, cn, bnp, kk + 1 < 9 ? kk + 2 : -1, livep, true, 0, false);
    }
    kstep(U0, V0, b, 15, U1, V1, cn, bnp, -1, false, true);
    __builtin_amdgcn_s_waitcnt(0x0070);
    __builtin_amdgcn_s_barrier();
    kstep(U1, V1, bnp, 0, U0, V0, c2, b, 0, more2, more1);
    cn = c2;
    b ^= 1;
  }
  W22_STAMP(2);
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();

  // G^T M G: the column pass (over q) inside the wave, the row pass (over the four waves) through the LDS, one cout tile per
  // round (4 waves x 2 cin tiles x 16 registers x 3 values x 64 lanes = 96 KB).  Round m, wave w then combines cin tile w & 1,
  // registers 8 (w >> 1) .. + 7 and stores the nine taps.
  float* dst = part + ((size_t)p * B.n_chunks + chunk) * CHUNK_FLOATS;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (m) __syncthreads();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = acc[2 * m + n][e], m1 = acc[4 + 2 * m + n][e], m2 = acc[8 + 2 * m + n][e], m3 = acc[12 + 2 * m + n][e];
        const float hs = 0.5f * (m1 + m2);
        float* o = smem + ((wv * 2 + n) * 16 + e) * 192 + lane;
        o[0] = m0 + hs; o[64] = 0.5f * (m1 - m2); o[128] = hs - m3;
      }
    __syncthreads();
    const int n = wv & 1;
    const bool tile_ok = (cout0 + 32 * m < a.Cout) && (cin0 + 32 * n < a.Cin);
    if (tile_ok) {
#pragma unroll
      for (int e8 = 0; e8 < 8; ++e8) {
        const int e = 8 * (wv >> 1) + e8;
        const int i = (e & 3) + 8 * (e >> 2) + 4 * hl;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
          const float* q = smem + (n * 16 + e) * 192 + sx * 64 + lane;
          const float T0 = q[0], T1 = q[2 * 16 * 192], T2 = q[4 * 16 * 192], T3 = q[6 * 16 * 192];
          const float hs = 0.5f * (T1 + T2);
          dst[((sx) * 64 + 32 * m + i) * 64 + 32 * n + l32] = T0 + hs;
          dst[((3 + sx) * 64 + 32 * m + i) * 64 + 32 * n + l32] = 0.5f * (T1 - T2);
          dst[((6 + sx) * 64 + 32 * m + i) * 64 + 32 * n + l32] = hs - T3;
        }
      }
    }
  }
  if (a.db != nullptr && cy == 0 && wv == 1) {
    const float t0 = bs0 + __shfl_xor(bs0, 32), t1 = bs1 + __shfl_xor(bs1, 32);
    float* pb = pbias + ((size_t)p * B.n_chunks + chunk) * 64 + l32;
    if (hl == 0 && cout0 < a.Cout) pb[0] = t0;
    if (hl == 0 && cout0 + 32 < a.Cout) pb[32] = t1;
  }
  }
  W22_STAMP(3);
}


// =========================================================================================================================
// "wino24" (round 4): the transposed F(2,3) along the image COLUMN (as above) times the transposed F(4,3) along the image ROW: a 2 x 4
// patch of dy and the 4 x 6 patch of x around it give the nine taps from 24 products instead of 72 -- a THIRD of the direct kernel's
// MFMAs (wino22: 4/9), the algorithm the F(2x4,3x3) conv kernels run forwards:
//     U = A_h g A_w^T (4 x 6 from the 2 x 4 dy patch),  V = B_h^T d B_w (4 x 6 from the x patch),  M_pq += U_pq V_pq,  dW = G_h^T M G_w
//     A_w (6 x 4) = transpose of F(4,3)'s A^T:  u0 = g0, u1 = g0+g1+g2+g3, u2 = g0-g1+g2-g3, u3 = g0+2g1+4g2+8g3, u4 = g0-2g1+4g2-8g3, u5 = g3
//     B_w^T = F(4,3)'s (4,0,-5,0,1,0 | 0,-4,-4,1,1,0 | 0,4,-4,-1,1,0 | 0,-2,-1,2,1,0 | 0,2,-1,-2,1,0 | 0,4,0,-5,0,1);  G_w = its G (6 x 3)
// Row-owner form only: wave w owns ROW w of the 4 x 6 positions for the whole 64 x 64 chunk = 6 x 2 x 2 = 24 accumulator tiles (384 registers:
// 16 tiles in the accumulation registers, 8 in vector registers; the MFMAs are inline assembly with the class spelled out, as in
// srk_conv_w42.hip; tools/check_w42_hazards.py lints them).  Same tiles (8 x 16 dy pixels + halo), same LDS image, same DMA pieces, same
// partial-block format as wino22: a tile is 8 k-steps (pairs of horizontally adjacent patches) of 24 MFMAs instead of 16 of 16.
// Operand transform per k-step: 30 packed instructions + 1 (bias), the pairs being the two cout tiles (U) resp. the two cin tiles (V) --
// each raw value is read by its own ds_read_b32 into one half of an aligned register pair (LDS reads are free beside fp32 MFMAs).
// The DMA plan is not held per piece (2 x 20 registers in wino22; here there are none to spare): a dy piece's lane offset is the same for
// every piece, an x piece's is one of two (pieces that cross the end of an 18-pixel halo row), the piece's part is wave-uniform.
template <int DYMODE>
__global__ __launch_bounds__(W22_THREADS) void wgrad_f32_wino24_kernel(const WBatch B, float* part, float* pbias) {
  __shared__ __attribute__((aligned(16))) float smem[2 * W22_TILE_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hl = lane >> 5, l32 = lane & 31;
  int p, chunk;
  {
    const int id = blockIdx.x, nc = B.n_chunks, pm = B.P & ~7;
    if (id < pm * nc) { const int s_ = id >> 3; p = (id & 7) + 8 * (s_ / nc); chunk = s_ - (s_ / nc) * nc; }
    else { const int r_ = id - pm * nc; p = pm + r_ / nc; chunk = r_ - (r_ / nc) * nc; }
  }
  const int pi = __builtin_amdgcn_readfirstlane(B.c_prob[chunk]);
  const int cy = __builtin_amdgcn_readfirstlane(B.c_cy[chunk]), cz = __builtin_amdgcn_readfirstlane(B.c_cz[chunk]);
  const WProb& a = B.prob[pi];
  const int cin0 = cy * 64, cout0 = cz * 64;
  const int Cps = a.Cout >> 2;

  f32x16 acc[24];
#pragma unroll
  for (int t = 0; t < 24; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  auto mfma = [&](int t, float va, float vb) {       // t is a constant after unrolling
    if (t < 16) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(acc[t]) : "v"(va), "v"(vb));
    else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc[t]) : "v"(va), "v"(vb));
  };

  const int t_begin = p * B.tpb;
  int t_end = t_begin + B.tpb;
  if (t_end > B.total_tiles) t_end = B.total_tiles;

  const long x_img = (long)B.H * B.W * a.x_ldc;
  const long dy_img = (long)B.OH * B.OW * a.dy_ldc * (DYMODE == SRK_IN_UNSHUFFLE ? 4 : 1);
  const long xb_l = ((long)(B.H * B.W - 1) * a.x_ldc + a.Cin) * 4, db_l = dy_img * 4;
  const unsigned xbytes = (unsigned)(xb_l > 0x7fffffffL ? 0x7fffffffL : xb_l);
  const unsigned dbytes = (unsigned)(db_l > 0x7fffffffL ? 0x7fffffffL : db_l);
  // ---- DMA plan: piece 4 j + wv of the tile buffer's 80 one-KB slots (4 consecutive pixels x 16 channel quads), j < 8: dy row j,
  // pixels 4 wv .. 4 wv + 3; j >= 8: halo pixels 4 k .. 4 k + 3, k = 4 (j - 8) + wv (k >= 45: padding).
  constexpr int NPW = 20;
  const int c4 = lane & 15, lp = lane >> 4;
  const int co = cout0 + 4 * c4, ci = cin0 + 4 * c4;
  int dyc, dyij = 0;
  if (DYMODE == SRK_IN_UNSHUFFLE) { dyij = co / Cps; dyc = co - dyij * Cps; } else { dyc = co; }
  const bool co_ok = co < a.Cout, ci_ok = ci < a.Cin;
  const int cdy = 4 * wv + lp;                                     // this lane's dy column inside the tile, the same in every dy piece
  unsigned laneDy = DYMODE == SRK_IN_UNSHUFFLE ? (unsigned)((((dyij >> 1) * (2 * B.OW) + 2 * cdy + (dyij & 1)) * a.dy_ldc + dyc) * 4)
                                                     : (unsigned)((cdy * a.dy_ldc + dyc) * 4);
  const unsigned dyRow = (unsigned)((DYMODE == SRK_IN_UNSHUFFLE ? 4 * B.OW : B.OW) * a.dy_ldc * 4);
  int lpd = co_ok ? cdy : -(1 << 20);
  // halo pixel hp = 4 k + lp = 18 hy + hx.  With 4 k = 18 a + b (b even, <= 16) only b = 16 lets lanes lp >= 2 run into the next row:
  // their offset is the plain one + (W - 18) pixels, their column the plain one - 18
  unsigned laneX = (unsigned)((lp * a.x_ldc + ci) * 4);
  unsigned laneXw = laneX + (lp >= 2 ? (unsigned)((B.W - W22_IW) * a.x_ldc * 4) : 0u);
  int lpx = ci_ok ? lp : -(1 << 20);
  int lpxw = ci_ok ? (lp >= 2 ? lp - W22_IW : lp) : -(1 << 20);
  struct TileCtx { int ow0; unsigned org_dy, org_x; __amdgpu_buffer_rsrc_t xr, dr; };
  struct TilePos { int tx, ty, n; };
  auto tile_pos = [&](int tile) {
    TilePos q;
    int tt = tile;
    q.tx = tt % B.tilesW; tt /= B.tilesW;
    q.ty = tt % B.tilesH; q.n = tt / B.tilesH;
    return q;
  };
  auto pos_step = [&](TilePos& q, bool go) {
    int tx = q.tx + 1, ty = q.ty, n = q.n;
    const bool wx = tx == B.tilesW;
    tx = wx ? 0 : tx; ty += wx ? 1 : 0;
    const bool wy = ty == B.tilesH;
    ty = wy ? 0 : ty; n += wy ? 1 : 0;
    q.tx = go ? tx : q.tx; q.ty = go ? ty : q.ty; q.n = go ? n : q.n;
  };
  auto ctx_offsets = [&](const TilePos& q, TileCtx& c) {
    const int oh0 = q.ty * W22_TH;
    c.ow0 = q.tx * WTW;
    c.org_dy = (unsigned)((DYMODE == SRK_IN_UNSHUFFLE ? (2 * oh0 * 2 * B.OW + 2 * c.ow0) : (oh0 * B.OW + c.ow0)) * a.dy_ldc * 4);
    c.org_x = (unsigned)((oh0 * B.W + c.ow0) * a.x_ldc * 4);
  };
  auto ctx_rsrcs = [&](const TilePos& q, TileCtx& c) {
    c.xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x + q.n * x_img + a.x_coff), 0, xbytes, 0x00020000);
    c.dr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy + q.n * dy_img + a.dy_coff), 0, dbytes, 0x00020000);
  };
  // piece j of this wave (j: compile-time after unrolling).  Everything about the piece itself is wave-uniform scalar arithmetic; the offset
  // goes out through the VECTOR offset (the hardware's range check, which drops the rows above / below the image, does not cover the scalar one)
  // piece_off: the vector-ALU half of a piece (its per-lane offset, out of range where the lane has nothing to fetch); piece_go: the DMA
  // instruction itself.  Apart in the k-step: the offsets of a step's pieces are formed in its ONE vector-ALU gap, the instructions go out one
  // per MFMA gap behind it -- back to back they queue in the address unit (measured on the conv kernel: five in a row ~180 cycles each; here:
  // 45 us of a 480 us launch with all of a step's pieces in one gap).
  auto piece_off = [&](const TileCtx& c, int j, bool live) -> unsigned {
#ifdef W24_CHEAP_PIECES      // (timing-only ablation: one add per piece, no validity tests, no row-wrap select.  NOT a measure of the offsets' cost:
                             //  the pieces then fetch the same few lines -- 467 -> 425 us; a correct form with half the vector-ALU work: 470 -> 466)
    if (j < 8) return laneDy + c.org_dy + (unsigned)j * dyRow;
    return laneX + c.org_x + (unsigned)((4 * (j - 8) + wv) * 16);
#endif
    if (j < 8) {
      const unsigned so = c.org_dy + (unsigned)j * dyRow;
#ifdef W24_DEAD_DY          // (timing-only ablations: only the dy / only the x pieces fetch)
      live = false;
#endif
      const int ow = live ? c.ow0 : (1 << 28);
      const bool ok = (unsigned)(lpd + ow) < (unsigned)B.OW;
      return ok ? laneDy + so : W_OOB;
    }
#ifdef W24_DEAD_X
    live = false;
#endif
    const int k = 4 * (j - 8) + wv, hp0 = 4 * k, ar = hp0 / W22_IW, bq = hp0 - W22_IW * ar;
    const bool isw = bq == 16;
    const unsigned so = c.org_x + (unsigned)(((ar - 1) * B.W + bq - 1) * a.x_ldc * 4);
    const int ow = (live && k < W22_NHP / 4) ? c.ow0 : (1 << 28);
    const int lc = isw ? lpxw : lpx;
    const unsigned lo = isw ? laneXw : laneX;
    const bool ok = (unsigned)(lc + bq - 1 + ow) < (unsigned)B.W;
    return ok ? lo + so : W_OOB;
  };
  auto piece_go = [&](const TileCtx& c, int b, int j, unsigned voff) {
    wdma16(j < 8 ? c.dr : c.xr, smem + b * W22_TILE_FLOATS + (4 * j + wv) * 256, voff);
  };
  auto piece = [&](const TileCtx& c, int b, int j, bool live) { piece_go(c, b, j, piece_off(c, j, live)); };

  //   B_h^T rows: (d0 - d2, d1 + d2, d2 - d1, d1 - d3);  A_h rows: (g0, g0 + g1, g0 - g1, g1)   -- one combination of two raw rows per wave
  const int xra = wv == 0 ? 0 : (wv == 2 ? 2 : 1);
  const int xrb = wv == 0 ? 2 : (wv == 1 ? 2 : (wv == 2 ? 1 : 3));
  const float xs = wv == 1 ? 1.f : -1.f;
  const int dra = wv == 3 ? 1 : 0;
  const int drb = (wv == 1 || wv == 2) ? 1 : dra;
  const float dsc = wv == 1 ? 1.f : (wv == 2 ? -1.f : 0.f);
  // LDS BYTE addresses of (row ra / rb of the patch, column 4 hl [+ 3 for the second x pointer], channel l32) of the k-step whose raw values are
  // read next.  They RUN with the k-steps (advanced in the transform gap, by constants) and are opaque to the compiler: every read is
  // pointer + a small immediate -- derived from one base per buffer, the step offsets exceed what a ds_read2_b32 can address and every
  // group of reads got a v_add of its own, i.e. a vector-ALU gap (14 cycles beside fp32 MFMAs) per group.
  unsigned pgA = (unsigned)(((4 * hl) * 64 + l32 + dra * 16 * 64) * 4), pgB = (unsigned)(((4 * hl) * 64 + l32 + drb * 16 * 64) * 4);
  unsigned pdA = (unsigned)((W22_TP * 64 + (4 * hl) * 64 + l32 + xra * W22_IW * 64) * 4), pdB = (unsigned)((W22_TP * 64 + (4 * hl) * 64 + l32 + xrb * W22_IW * 64) * 4);
  unsigned pdA3 = pdA + 3 * 64 * 4, pdB3 = pdB + 3 * 64 * 4;
  const f32x2 xs2 = {xs, xs}, dsc2 = {dsc, dsc};
  f32x2 bs2 = {0.f, 0.f};
  f32x2 U0[6], V0[6], U1[6], V1[6];
  f32x2 gA[4], gB[4], dA[6], dB[6];          // raw pairs: [pixel j][cout tile m] resp. [pixel j][cin tile n]
  const char* sm = reinterpret_cast<const char*>(smem);
  auto ldsf = [&](unsigned addr, int off_floats) { return *reinterpret_cast<const float*>(sm + addr + 4 * off_floats); };
  auto rd_g = [&](bool second) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int m = 0; m < 2; ++m) (second ? gB : gA)[j][m] = ldsf(second ? pgB : pgA, 64 * j + 32 * m);
  };
  auto rd_d = [&](int j0, bool second) {
#pragma unroll
    for (int j = j0; j < j0 + 3; ++j)
#pragma unroll
      for (int n = 0; n < 2; ++n) (second ? dB : dA)[j][n] = ldsf(j0 ? (second ? pdB3 : pdA3) : (second ? pdB : pdA), 64 * (j - j0) + 32 * n);
  };
  // the pointers move on to the k-step behind the one just read: dg / dx BYTES (compile-time except across the tile boundary)
  auto advance = [&](int dg, int dx) {
    pgA += dg; pgB += dg; pdA += dx; pdB += dx; pdA3 += dx; pdB3 += dx;
    asm volatile("" : "+v"(pgA), "+v"(pgB), "+v"(pdA), "+v"(pdB), "+v"(pdA3), "+v"(pdB3));
  };
#define W24_PKFMA(r, k, x, y) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "s"(k), "v"(x), "v"(y))
#define W24_PKADD(r, x, y) asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y))
#define W24_PKSUB(r, x, y) asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(x), "v"(y))
  // U (both cout tiles): row pass t_j = gA_j + s gB_j, then A_w: 12 packed.  The patch sum (row g0 + g1, position 1) counts towards the bias --
  // unless the operands are formed behind the last tile (realmask = 0: stale LDS contents, possibly NaN bit patterns: masked, not multiplied)
  auto xf_u = [&](f32x2 (&U)[6], bool always, unsigned realmask) {
    const f32x2 k4 = {4.f, 4.f}, k2 = {2.f, 2.f}, kM2 = {-2.f, -2.f};
    f32x2 t[4], s02, s13, aa, b4;
#pragma unroll
    for (int j = 0; j < 4; ++j) W24_PKFMA(t[j], dsc2, gB[j], gA[j]);
    W24_PKADD(s02, t[0], t[2]); W24_PKADD(s13, t[1], t[3]);
    W24_PKFMA(aa, k4, t[2], t[0]); W24_PKFMA(b4, k4, t[3], t[1]);
    W24_PKADD(U[1], s02, s13); W24_PKSUB(U[2], s02, s13);
    W24_PKFMA(U[3], k2, b4, aa); W24_PKFMA(U[4], kM2, b4, aa);
    U[0] = t[0]; U[5] = t[3];
    if (always) {
      W24_PKADD(bs2, bs2, U[1]);
    } else {
      // (as assembly: written as two scalar ANDs in C++, the compiler formed the first and COPIED it into the second half)
      f32x2 u1m;
      asm("v_and_b32 %0, %2, %3\n\tv_and_b32 %1, %2, %4" : "=&v"(u1m[0]), "=&v"(u1m[1]) : "s"(realmask), "v"(U[1][0]), "v"(U[1][1]));
      W24_PKADD(bs2, bs2, u1m);
    }
  };
  // V (both cin tiles): row pass d_j = dA_j + s dB_j, then B_w^T in 12: 18 packed
  auto xf_v = [&](f32x2 (&V)[6]) {
    const f32x2 kM4 = {-4.f, -4.f}, k4 = {4.f, 4.f}, kM5 = {-5.f, -5.f}, k2 = {2.f, 2.f}, kM2 = {-2.f, -2.f};
    f32x2 d[6], t1, t2, t3, t4, u0, u5;
#pragma unroll
    for (int j = 0; j < 6; ++j) W24_PKFMA(d[j], xs2, dB[j], dA[j]);
    W24_PKFMA(t1, kM4, d[2], d[4]); W24_PKFMA(t2, kM4, d[1], d[3]);
    W24_PKSUB(t3, d[4], d[2]); W24_PKSUB(t4, d[3], d[1]);
    W24_PKFMA(u0, kM5, d[2], d[4]); W24_PKFMA(V[0], k4, d[0], u0);
    W24_PKADD(V[1], t1, t2); W24_PKSUB(V[2], t1, t2);
    W24_PKFMA(V[3], k2, t4, t3); W24_PKFMA(V[4], kM2, t4, t3);
    W24_PKFMA(u5, kM5, d[3], d[5]); W24_PKFMA(V[5], k4, d[1], u5);
  };
  // One k-step = 24 MFMAs on (U, V): accumulator tile i = 4 q + 2 m + n.  In their shadow, by hand (one sched_barrier per MFMA): gaps 0-5 the
  // raw reads of the NEXT step (buffer nb, step nk), gap W24_RG its whole transform into (UN, VN) and the DMA pieces dj0 .. dj0 + djn - 1 of
  // the tile dc into buffer db -- ONE vector-ALU gap per k-step (a gap that holds any costs 14 cycles on top of 4 per instruction).
  constexpr int W24_RG = 8;
  constexpr int W24_DG = 2 * 16 * 64 * 4, W24_DX = 2 * W22_IW * 64 * 4;          // one patch row down (bytes)
  constexpr int W24_DG3 = (8 * 64 - 3 * 2 * 16 * 64) * 4, W24_DX3 = (8 * 64 - 3 * 2 * W22_IW * 64) * 4;      // patch row 3 -> row 0 of the next column pair
  constexpr int W24_G7 = (6 * 16 + 8) * 64 * 4, W24_X7 = (6 * W22_IW + 8) * 64 * 4;            // k-step 7 -> k-step 0 (of the other buffer: +- the buffer size)
  TilePos pos2;
  TileCtx c2;
  // dg / dx: how far the read pointers move behind this step's reads (to the k-step the NEXT call reads)
  auto kstep = [&](const f32x2 (&U)[6], const f32x2 (&V)[6], f32x2 (&UN)[6], f32x2 (&VN)[6], int dg, int dx,
                   const TileCtx& dc, int db, int dj0, int djn, bool dlive, bool always, unsigned realmask, int mk = 0, bool mk_go = false) {
    unsigned pvo[9];
#pragma unroll
    for (int i = 0; i < 24; ++i) {
      mfma(i, U[i >> 2][(i >> 1) & 1], V[i >> 2][i & 1]);
      __builtin_amdgcn_sched_barrier(0);
#ifndef W24_NO_LDS        // (-DW24_NO_LDS / NO_XFORM / NO_DMA: timing-only ablation builds, tools/debug/build_w22_var.sh -- wrong results)
      if (i == 0) rd_g(false);
      if (i == 1) rd_g(true);
      if (i == 2) rd_d(0, false);
      if (i == 3) rd_d(3, false);
      if (i == 4) rd_d(0, true);
      if (i == 5) rd_d(3, true);
#endif
      if (mk == 1 && i == 6) { pos_step(pos2, mk_go); ctx_offsets(pos2, c2); ctx_rsrcs(pos2, c2); }
      if (i == W24_RG) {
#ifndef W24_NO_XFORM
        xf_u(UN, always, realmask);
        xf_v(VN);
#endif
        advance(dg, dx);
#ifndef W24_NO_DMA
        if (djn > 0) {
          // (the lane parts of the piece offsets pass through here so that the pieces' vector-ALU work stays in THIS gap)
          asm volatile("" : "+v"(laneDy), "+v"(lpd), "+v"(laneX), "+v"(laneXw), "+v"(lpx), "+v"(lpxw));
#pragma unroll
          for (int u = 0; u < djn; ++u) { pvo[u] = piece_off(dc, dj0 + u, dlive); asm volatile("" : "+v"(pvo[u])); }
        }
#endif
      }
#ifndef W24_NO_DMA
      if (i > W24_RG && i - W24_RG - 1 < djn) piece_go(dc, db, dj0 + i - W24_RG - 1, pvo[i - W24_RG - 1]);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  TileCtx cn;
  if (t_begin < t_end) {
    pos2 = tile_pos(t_begin);
    TileCtx c0;
    ctx_offsets(pos2, c0); ctx_rsrcs(pos2, c0);
#pragma unroll
    for (int j = 0; j < NPW; ++j) piece(c0, 0, j, true);
    pos_step(pos2, t_begin + 1 < t_end);
    ctx_offsets(pos2, cn); ctx_rsrcs(pos2, cn);
    c2 = cn;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);                              // vmcnt(0) lgkmcnt(0)
  __syncthreads();
  if (t_begin < t_end) {
    if (t_begin + 1 < t_end) { piece(cn, 1, 0, true); piece(cn, 1, 1, true); }   // (the state every tile starts in: pieces 0, 1 of the next one issued)
    rd_g(false); rd_g(true);
    rd_d(0, false); rd_d(3, false); rd_d(0, true); rd_d(3, true);
    xf_u(U0, true, 0u);
    xf_v(V0);
    advance(W24_DG, W24_DX);                                       // -> k-step 1 of buffer 0
  }
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_nop 4");                                        // (VALU results two wait states ahead of the first MFMA that reads them)
  int b = 0;
  for (int tile = t_begin; tile < t_end; ++tile) {
    // tile sits in buffer b; tile + 1 is in flight into b ^ 1: pieces 0, 1 went out on the previous tile's last step, 2 .. 19 go out on this
    // tile's steps 0 .. 3 (five, five, four, four), which leaves steps 4 .. 6 for them to land; tile + 2's pieces 0, 1 ride on the last step,
    // behind the barrier that releases buffer b
    const bool more1 = tile + 1 < t_end, more2 = tile + 2 < t_end;
    const int bnp = b ^ 1;
    const int flip = (b ? -1 : 1) * W22_TILE_FLOATS * 4;            // to the other buffer (bytes)
#ifdef W24_DEAD_DMA         // (timing-only ablation: the pieces are issued, but out of range -- issue cost without memory traffic)
    const bool plive = false;
#else
    const bool plive = more1;
#endif
#if W24_PIECE_PLAN == 1     // 6 + 6 + 6 on steps 0 - 2
    kstep(U0, V0, U1, V1, W24_DG, W24_DX, cn, bnp, 2, 6, plive, true, 0u);            // runs step 0, reads step 1
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 8, 6, plive, true, 0u);            // reads step 2
    kstep(U0, V0, U1, V1, W24_DG3, W24_DX3, cn, bnp, 14, 6, plive, true, 0u);         // reads step 3
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 0, 0, false, true, 0u);            // reads step 4
#elif W24_PIECE_PLAN == 2   // 9 + 9 on steps 0, 1
    kstep(U0, V0, U1, V1, W24_DG, W24_DX, cn, bnp, 2, 9, plive, true, 0u);
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 11, 9, plive, true, 0u);
    kstep(U0, V0, U1, V1, W24_DG3, W24_DX3, cn, bnp, 0, 0, false, true, 0u);
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 0, 0, false, true, 0u);
#else                       // 5 + 5 + 4 + 4 on steps 0 - 3
    kstep(U0, V0, U1, V1, W24_DG, W24_DX, cn, bnp, 2, 5, plive, true, 0u);            // runs step 0, reads step 1
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 7, 5, plive, true, 0u);            // reads step 2
    kstep(U0, V0, U1, V1, W24_DG3, W24_DX3, cn, bnp, 12, 4, plive, true, 0u);         // reads step 3
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 16, 4, plive, true, 0u);           // reads step 4
#endif
    kstep(U0, V0, U1, V1, W24_DG, W24_DX, cn, bnp, 0, 0, false, true, 0u, 1, more2);  // reads step 5
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, cn, bnp, 0, 0, false, true, 0u);            // reads step 6
    kstep(U0, V0, U1, V1, flip - W24_G7, flip - W24_X7, cn, bnp, 0, 0, false, true, 0u);      // runs step 6, reads step 7; then on to step 0 of tile + 1
    __builtin_amdgcn_s_waitcnt(0x0070);                            // every piece of tile + 1 has landed (no other VMEM in flight)
    __builtin_amdgcn_s_barrier();
    kstep(U1, V1, U0, V0, W24_DG, W24_DX, c2, b, 0, 2, more2, false, more1 ? ~0u : 0u);             // runs step 7: pieces 0, 1 of tile + 2; first operands of tile + 1
    cn = c2;
    b ^= 1;
  }
  __builtin_amdgcn_s_waitcnt(0x0070);
  __syncthreads();

  // G_h^T M G_w: the column pass (over the six q) inside the wave, the row pass (over the four waves) through the LDS, one cout tile per
  // round, exactly as in the wino22 row-owner form
  float* dst = part + ((size_t)p * B.n_chunks + chunk) * CHUNK_FLOATS;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    if (m) __syncthreads();
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const float m0 = acc[2 * m + n][e], m1 = acc[4 + 2 * m + n][e], m2 = acc[8 + 2 * m + n][e], m3 = acc[12 + 2 * m + n][e],
                    m4 = acc[16 + 2 * m + n][e], m5 = acc[20 + 2 * m + n][e];
        const float s12 = m1 + m2, s34 = m3 + m4;
        float* o = smem + ((wv * 2 + n) * 16 + e) * 192 + lane;
        o[0] = 0.25f * m0 - (1.f / 6.f) * s12 + (1.f / 24.f) * s34;
        o[64] = (1.f / 6.f) * (m2 - m1) + (1.f / 12.f) * (m3 - m4);
        o[128] = (1.f / 6.f) * (s34 - s12) + m5;
      }
    __syncthreads();
    const int n = wv & 1;
    const bool tile_ok = (cout0 + 32 * m < a.Cout) && (cin0 + 32 * n < a.Cin);
    if (tile_ok) {
#pragma unroll
      for (int e8 = 0; e8 < 8; ++e8) {
        const int e = 8 * (wv >> 1) + e8;
        const int i = (e & 3) + 8 * (e >> 2) + 4 * hl;
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
          const float* q = smem + (n * 16 + e) * 192 + sx * 64 + lane;
          const float T0 = q[0], T1 = q[2 * 16 * 192], T2 = q[4 * 16 * 192], T3 = q[6 * 16 * 192];
          const float hs = 0.5f * (T1 + T2);
          dst[((sx) * 64 + 32 * m + i) * 64 + 32 * n + l32] = T0 + hs;
          dst[((3 + sx) * 64 + 32 * m + i) * 64 + 32 * n + l32] = 0.5f * (T1 - T2);
          dst[((6 + sx) * 64 + 32 * m + i) * 64 + 32 * n + l32] = hs - T3;
        }
      }
    }
  }
  if (a.db != nullptr && cy == 0 && wv == 1) {
    const float t0 = bs2[0] + __shfl_xor(bs2[0], 32), t1 = bs2[1] + __shfl_xor(bs2[1], 32);
    float* pb = pbias + ((size_t)p * B.n_chunks + chunk) * 64 + l32;
    if (hl == 0 && cout0 < a.Cout) pb[0] = t0;
    if (hl == 0 && cout0 + 32 < a.Cout) pb[32] = t1;
  }
}

}  // namespace

// Which form of the kernel is launched: 2 = wino24, the F(2,3) x F(4,3) kernel above (a third of the direct kernel's MFMAs); 1 the row-owner
// form of wino22 (4/9); 0 its tile-owner form (every wave forms all 16 positions of its own 32 x 32 tile).  SRK_WGRAD_W22_FORM /
// srk_debug_set_wgrad_w22_form (A/B measurements, tests).
static int g_w22_rows = -1;
int srk_wgrad_wino22_rows() {
  if (g_w22_rows < 0) { const char* e = getenv("SRK_WGRAD_W22_FORM"); g_w22_rows = e ? atoi(e) : W22_DEFAULT_FORM; if (g_w22_rows < 0 || g_w22_rows > 2) g_w22_rows = W22_DEFAULT_FORM; }
  return g_w22_rows;
}
extern "C" int srk_debug_set_wgrad_w22_form(int rows) { g_w22_rows = rows < 0 ? -1 : (rows > 2 ? 2 : rows); return SRK_OK; }

int srk_launch_wgrad_wino22(const WBatch& B, float* part, float* pbias, hipStream_t st) {
  const int rows = srk_wgrad_wino22_rows();
  const dim3 grid(B.P * B.n_chunks), blk(W22_THREADS);
  if (B.dy_mode == SRK_IN_UNSHUFFLE) {
    if (rows == 2) hipLaunchKernelGGL((wgrad_f32_wino24_kernel<SRK_IN_UNSHUFFLE>), grid, blk, 0, st, B, part, pbias);
    else if (rows) hipLaunchKernelGGL((wgrad_f32_wino22_kernel<SRK_IN_UNSHUFFLE, true>), grid, blk, 0, st, B, part, pbias);
    else hipLaunchKernelGGL((wgrad_f32_wino22_kernel<SRK_IN_UNSHUFFLE, false>), grid, blk, 0, st, B, part, pbias);
  } else {
    if (rows == 2) hipLaunchKernelGGL((wgrad_f32_wino24_kernel<SRK_IN_PLAIN>), grid, blk, 0, st, B, part, pbias);
    else if (rows) hipLaunchKernelGGL((wgrad_f32_wino22_kernel<SRK_IN_PLAIN, true>), grid, blk, 0, st, B, part, pbias);
    else hipLaunchKernelGGL((wgrad_f32_wino22_kernel<SRK_IN_PLAIN, false>), grid, blk, 0, st, B, part, pbias);
  }
  SRK_CHECK_LAUNCH();
  return SRK_OK;
}

#ifdef SRK_STAMP
extern "C" int srk_debug_set_w22_stamps(void* p) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_w22_stamps), &p, sizeof(p)) == hipSuccess ? 0 : -5;
}
#endif
