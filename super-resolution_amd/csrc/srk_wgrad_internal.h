// Definitions shared by the weight-gradient translation units (srk_wgrad.hip, srk_wgrad_w22.hip); not part of the C ABI.
#pragma once
#include "srk_internal.h"

namespace srkw {

constexpr int WTW = 16;
constexpr int MAX_PROB = 8;
constexpr int MAX_CHUNK = 64;

template <int S> struct WGeo {
  static constexpr int TH = (S == 1) ? 4 : 2;
  static constexpr int TP = TH * WTW;              // pixels per tile
  static constexpr int IH = (TH - 1) * S + 3;
  static constexpr int IW = (WTW - 1) * S + 3;
  static constexpr int NHP = IH * IW;
};

struct WProb {
  const float* x; const float* dy; float* dw; float* db;
  int x_ldc, x_coff, dy_ldc, dy_coff, Cin, Cout, accumulate;
  float in_slope, scale;
};

struct WBatch {
  int N, H, W, OH, OW;
  int P, tpb, tilesH, tilesW, total_tiles, n_chunks, n_prob, dy_mode, wino;
  int h16;                  // x / dy are 16-bit tensors (srk_wgrad_args.precision 3 / 4): srk_wgrad_h16.hip only
  int bias_lo;              // the bias partials are (hi, lo) float pairs of a double sum; the lo words sit lo_off floats behind `part`
  unsigned long long lo_off;//   (the direct kernel: the discriminator's bias gradients are differences of nearly equal sums, srk_wgrad.hip)
  WProb prob[MAX_PROB];
  unsigned char c_prob[MAX_CHUNK], c_cy[MAX_CHUNK], c_cz[MAX_CHUNK];
};

constexpr size_t CHUNK_FLOATS = 9 * 64 * 64;

constexpr unsigned W_OOB = 0x80000000u;

__device__ __forceinline__ void wdma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, unsigned voffset) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, voffset, 0, 0, 0);
}

constexpr int W22_TH = 8;                                  // pixel rows per tile of the 2-D Winograd kernel (srk_wgrad_w22.hip)
constexpr int W16_TH = 8;                                  // pixel rows per tile of the 16-bit-storage kernel (srk_wgrad_h16.hip)

}  // namespace srkw

int srk_launch_wgrad_wino22(const srkw::WBatch& B, float* part, float* pbias, hipStream_t st);
int srk_wgrad_wino22_rows();            // 1: the row-owner form of the kernel is the one launched
int srk_launch_wgrad_h16(const srkw::WBatch& B, int precision, float* part, float* pbias, hipStream_t st);   // precision 3 (fp16) / 4 (bf16)
