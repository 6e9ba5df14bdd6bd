/* srk.h -- C ABI of the MI355X (gfx950) ESRGAN hot-path kernels.
 *
 * The reference (lukas-blecher/super-resolution) has no FFI: its hot path is the
 * chain of ATen ops dispatched by /root/reference/models.py.  Each entry point
 * below names the reference op(s) it replaces (file:line in the reference).
 *
 * Conventions
 *  - all data pointers are DEVICE pointers to fp32; activations are NHWC
 *    "views": element (n,h,w,c) of a view lives at
 *        base[((n*H + h)*W + w) * ldc + coff + c]
 *    so a conv can read a channel prefix / write a channel slice of a shared
 *    dense buffer (the concat-free DenseResidualBlock, models.py:34-41).
 *  - weights are passed PRE-PACKED (srk_pack_weights) from the canonical OIHW
 *    fp32 nn.Conv2d parameters; the OIHW tensors stay the source of truth.
 *  - `stream` is a hipStream_t passed as void*; every call is asynchronous on it,
 *    allocates nothing and never synchronises (hipGraph-capturable).
 *  - return value: 0 = ok, negative = srk_status; nothing throws across the ABI.
 *  - the library owns no memory: buffers and workspaces are caller-provided.
 */
#ifndef SRK_H
#define SRK_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRK_VERSION 100

typedef enum srk_status {
  SRK_OK = 0,
  SRK_ERR_BAD_ARG = -1,        /* null pointer / non-positive dimension             */
  SRK_ERR_UNSUPPORTED = -2,    /* combination of modes not implemented              */
  SRK_ERR_ALIGNMENT = -3,      /* view not 16-byte aligned where a vector path needs */
  SRK_ERR_WORKSPACE = -4,      /* workspace too small                                */
  SRK_ERR_LAUNCH = -5,         /* hipLaunchKernel reported an error                 */
  SRK_ERR_CHAIN_TIMEOUT = -6   /* a chain launch (srk_conv3x3_seq) gave up: a neighbouring tile did not publish within the bound.
                                  Nothing was launched by THIS call; call srk_chain_recover(), then repeat the iteration */
} srk_status;

/* how the logical conv input is read from memory */
typedef enum srk_in_mode {
  SRK_IN_PLAIN = 0,
  SRK_IN_UNSHUFFLE = 1, /* logical (h,w,k*Cps+c), k=2i+j  <-  mem (2h+i, 2w+j, c): the inverse of
                           nn.PixelShuffle(2) (models.py:89) folded into the load (backward of the
                           upsampling conv) */
  SRK_IN_ZERO_UPSAMPLE = 2 /* logical (uh,uw) = mem (uh/2,uw/2) if both even else 0: data-gradient of
                              a stride-2 conv (models.py:144) as a stride-1 conv */
} srk_in_mode;

/* A fused 3x3 / pad 1 convolution (forward or data-gradient):
 *   t = sum_{r,s,c} Wp[o][r][s][c] * lrelu_in(X[n, S*oh + r - 1, S*ow + s - 1, c])  + bias[o]
 *   t = alpha * t + beta1 * R1[...] + beta2 * R2[...]
 *   t = t > 0 ? t : slope * t                     (slope == 1 -> none)
 *   t = t * (M[...] > 0 ? 1 : mask_slope)          (M == NULL -> none; LeakyReLU backward)
 *   Y[n, oh, ow, o] = t        or, with ps_out, Y[n, 2oh+i, 2ow+j, c] for packed o = (2i+j)*Cout/4 + c
 * R1, R2, M are read at the same physical (pixel, channel) as the Y store, through their own ldc/coff.
 *
 * Replaces: nn.Conv2d(.,.,3,stride,1) + nn.LeakyReLU + torch.cat + .mul(res_scale)+x + torch.add +
 * nn.PixelShuffle of models.py:19-21,36-41,53,63,67,86-90,97-99,126,142-145,168 and their autograd
 * data-gradients. */
typedef struct srk_conv_args {
  int32_t N, H, W;          /* logical input extent (for SRK_IN_ZERO_UPSAMPLE: extent of the stored tensor) */
  int32_t OH, OW;           /* logical output extent (before ps_out) */
  int32_t Cin, Cout;
  int32_t stride;           /* 1 or 2 */
  int32_t in_mode;          /* srk_in_mode */
  int32_t ps_out;           /* 0 | 1: PixelShuffle(2) folded into the store */
  const float* x;  int32_t x_ldc, x_coff;
  float in_slope;           /* LeakyReLU applied to X while it is staged (1 = none): pre-activation chaining of
                               the discriminator, conv(lrelu(z)) (models.py:142-145) */
  const float* wp;          /* packed weights, srk_pack_weights layout for (Cin, Cout) */
  const float* bias;        /* [Cout] in packed-o order, or NULL */
  float* y;        int32_t y_ldc, y_coff;
  float alpha;
  const float* r1; int32_t r1_ldc, r1_coff; float beta1;
  const float* r2; int32_t r2_ldc, r2_coff; float beta2;
  float slope;
  const float* mask; int32_t m_ldc, m_coff; float mask_slope;
  int32_t wp_format;        /* 0: fp32 fragments, exact-fp32 MFMA (default).  1: split-bf16 ("bf16x3") fragments from
                               srk_pack_weights_bf16x3 -> 3 bf16 MFMAs per product, fp32 accumulate, ~2^-16 relative
                               operand precision; opt-in, needs srk_conv3x3_bf16x3_supported().
                               2: same packed weights, plain bf16 operands (hi parts only; mixed precision)
                               3: Winograd F(2,3)-along-W fp32 fragments (srk_pack_entry.fmt = 3): exact-fp32 MFMA on 2/3
                                  of the products; stride 1, Cin % 8 == 0, Cout % 64 == 0, plain / unshuffle input
                               5: Winograd F(4,3)-along-W fp32 fragments (fmt = 5): half of the products, 32x16 workgroup
                                  tiles; as 3, plus in_slope == 1
                               6: 2-D Winograd F(2x4, 3x3) fp32 fragments (fmt = 6): a third of the products (F(4,3) along W
                                  times F(2,3) along H), 32x16 workgroup tiles, one wave per SIMD; same contract as 5
                               7 / 8: 16-BIT ACTIVATION STORAGE, fp16 (7) / bf16 (8) MFMA operands, fp32 accumulate (BASELINE configs[4]):
                                  x, y, r1, r2, mask point to 16-bit elements (ldc / coff in elements, multiples of 8), bias stays fp32,
                                  wp = fragments of srk_pack_entry.fmt 7 / 8; stride 1, Cin % 32 == 0, Cout % 8 == 0, plain / unshuffle
                                  input, in_slope == 1 */
  int32_t flags;            /* SRK_CONV_OUT_F32: (wp_format 7 / 8 only) y is fp32 [.., y_ldc] (y_ldc / y_coff in fp32 elements, any Cout);
                               excludes r1 / r2 / mask / ps_out -- the generator's last conv writes the fp32 image
                               SRK_CONV_WRITE_SIGNS / SRK_CONV_MASK_SIGNS: see `signs` */
  void* signs;              /* SIGN BITS of an output tensor (wp_format 7 / 8, 16-bit output, no ps_out; srk_conv3x3_signs_bytes() bytes, 16-byte
                               aligned, layout private to the kernels: 128 bits per lane of a workgroup tile).  SRK_CONV_WRITE_SIGNS: besides
                               y, the conv writes (y > 0) per stored element here.  SRK_CONV_MASK_SIGNS: the LeakyReLU' mask comes from such
                               bits, written by an earlier conv of the SAME geometry (N, OH, OW, Cout) and format, instead of from `mask`
                               (which must be NULL): the data-gradient conv of a dense block then reads 1 MB instead of the 16.8 MB of the
                               forward activation.  Same results as with the mask tensor.  (The fp32 F(2x4,3x3) kernel, wp_format 6, has no
                               register to spare for them: srk_conv3x3_seq_signs_bytes says 0 there.) */
} srk_conv_args;
#define SRK_CONV_OUT_F32 1
#define SRK_CONV_WRITE_SIGNS 2
#define SRK_CONV_MASK_SIGNS 4
/* bytes of the `signs` buffer for this conv (by its geometry, format and the kernel form its launch takes); 0: the launch does not support sign bits */
size_t srk_conv3x3_signs_bytes(const srk_conv_args* a);
/* the same for a whole srk_conv3x3_seq call (bytes of ONE conv's buffer; every conv of the sequence needs the same); 0: this sequence's
 * launches offer no sign bits (the caller keeps passing the mask tensors) */
size_t srk_conv3x3_seq_signs_bytes(const srk_conv_args* args, int n);
/* layout tag of those bits (tile rows | wp_format << 8; 0: none).  The sequence that writes sign bits and the one that reads them may be
 * dispatched to different kernel forms: use the bits only if both report the same tag, else pass the mask tensors. */
int srk_conv3x3_seq_signs_tag(const srk_conv_args* args, int n);

int srk_conv3x3(const srk_conv_args* a, void* stream);
/* n of them launched back to back on `stream`, in array order, from ONE call: the five forward (or five data-gradient) convolutions of
 * a DenseResidualBlock (models.py:34-41), whose buffers all exist before the first launch.  Stops at the first failing launch and
 * returns its status. */
int srk_conv3x3_seq(const srk_conv_args* args, int n, void* stream);
/* CHAIN FORMS (16-bit storage, wp_format 7 / 8; the fp32 F(2x4,3x3) kernel, wp_format 6): a sequence in which every convolution takes at
 * most its LAST 64 input channels from its predecessor's output (a dense block: conv k reads slices 0..k-1 of one buffer and writes
 * slice k), all of one geometry, 64-channel slices on 128-byte lines, 64 outputs, at most one workgroup tile per CU, goes out as ONE
 * persistent launch: every workgroup keeps its tile for the whole sequence, neighbouring tiles hand each other the new slice through
 * device-scope flags instead of a kernel boundary, and the next conv's first (old-slice) stage streams in beside the previous conv's
 * end.  Results: wp_format 6 bit-identical to the separate launches; 7 / 8 identical up to the order of the fp32 sums (one unit in the
 * last place of the 16-bit outputs).  At most one chain kernel is in flight per device (launches on different streams are ordered by
 * an event).
 * SINGLE TENANT: a chain kernel's tiles wait for their neighbours, which must therefore be resident: the process has the GPU to itself
 * (one process per GPU).  Every wait is bounded: SRK_CHAIN_WAIT_MS, default 50 ms (nothing of this process holds a CU that long).  A tile
 * whose neighbour has not published in time (another process on the GPU, a kernel of another stream holding CUs for that long) raises a
 * host-visible fault word, poisons a device word and drains -- no further waiting, neither in this tile nor, once they see the poison,
 * in any other: the launch ends in its usual time; what it writes is garbage in activation buffers.  From then on
 *   - srk_adam_count_step tells the optimizer step that follows to skip itself, ON THE DEVICE (no weight ever sees a gradient computed
 *     from a launch that gave up, however far the host has run ahead),
 *   - the next srk_conv3x3_seq / srk_adam_* call returns SRK_ERR_CHAIN_TIMEOUT and launches nothing,
 * until srk_chain_recover(): it waits for the device, clears the fault and lets the chain forms rest (64, 512, 4096, ... sequence calls go
 * conv by conv; then they are tried again).  The caller repeats the iteration (train.Stepper does).
 * SRK_H16_CHAIN=0 / SRK_W42_CHAIN=0 (srk_debug_set_h16_chain / _w42_chain(0)) disable the forms; for wp_format 7 / 8 the default (1)
 * uses the form where the 16-row kernel would run, 2 wherever the sequence is eligible.
 * srk_conv3x3_seq_kernel_name: name of the one kernel the sequence goes to, "" if it is launched conv by conv. */
int srk_conv3x3_seq_kernel_name(const srk_conv_args* args, int n, char* buf, size_t len);
int srk_debug_set_h16_chain(int mode);
int srk_debug_set_w42_chain(int mode);
/* MFMA shape of the 16-bit chain kernel: 1 (default; SRK_H16_CHAIN_M16) = v_mfma_f32_16x16x32 with the weights as the row operand and an
 * epilogue that stores straight from the accumulators; 0 = the 32x32x16 form (A/B measurements, tests) */
int srk_debug_set_h16_chain_m16(int on);
/* pending fault code (0: none) after waiting for the device, cleared; < 0: error.  See above. */
int srk_chain_recover(void);
/* chain launches so far / wrap resets of the flag epoch / recovered time-outs / sequence calls left in the back-off (any may be NULL) */
int srk_chain_stats(unsigned long long* launches, unsigned long long* resets, int* strikes, long* off_calls);
/* The flag epoch is 32-bit and flags are compared as differences; before it would pass 2^30 the library zeroes it and the flag array on
 * the launching stream.  This is that decision as pure arithmetic (CPU-testable): given the current epoch and the next launch (n convs)
 * it returns the epoch the launch uses and whether the reset precedes it. */
int srk_chain_epoch_plan(unsigned epoch, int n, unsigned* epoch_out, int* reset);
/* bound of every wait of a chain launch in microseconds (0: back to SRK_CHAIN_WAIT_MS / 50 ms).  A data-parallel job sets it to seconds:
 * a collective's kernel holds CUs while a peer rank is late, and a tile behind it has to wait that out. */
int srk_chain_set_wait_us(unsigned us);
/* test aids: set the epoch (to cross the wrap in a test) and the back-off; pretend a launch timed out; occupy `workgroups` CUs for `usec`
 * microseconds on `stream` (one 4-wave workgroup with 64 KB of LDS each: no chain workgroup fits beside it) -- the stand-in for a
 * collective's kernel in tools/debug/holder_bench.py */
int srk_debug_chain_set(unsigned epoch, long off_calls);
int srk_debug_chain_inject_fault(unsigned code);
int srk_debug_chain_inject_fault_async(void* stream);     /* the same from the device side of `stream`, in stream order */
/* start skew of a chain kernel kind (0: 16-bit, 1: fp32): workgroup b begins (b >> 3) % groups * ns late (experiment: spreading the
 * epilogues' store bursts, measured without effect; test aid: tiles whose neighbours are late, deterministically) */
int srk_debug_chain_skew(int kind, unsigned ns, unsigned groups);
int srk_debug_hold_cus(int workgroups, int usec, void* stream);
/* Test aid: fills the whole LDS of every CU with NaN bit patterns, so that a kernel that lets LDS bytes it never wrote reach a result
   fails deterministically in the launch that follows. */
int srk_debug_poison_lds(void* stream);

/* Weight-gradient of the same convolution:
 *   dW[o][c][r][s] = scale * sum_{n,oh,ow} DY[n,oh,ow,o] * X[n, S*oh+r-1, S*ow+s-1, c]
 *   db[o]          = scale * sum DY[n,oh,ow,o]                       (db may be NULL)
 * written in canonical OIHW order (accumulate == 0: overwrite, 1: add to dW/db).
 * DY may be read through SRK_IN_UNSHUFFLE (dy_mode); then o is the packed order and the result is
 * un-permuted on store.  Two launches: partial sums per pixel-split into `workspace`, then a
 * deterministic (fixed-order) reduction -- no float atomics.
 * Replaces the autograd weight/bias gradient of nn.Conv2d (models.py:19,63,67,87,97,99,142,144,168). */
typedef struct srk_wgrad_args {
  int32_t N, H, W;          /* x extent */
  int32_t OH, OW;           /* dy logical extent */
  int32_t Cin, Cout;
  int32_t stride;
  int32_t dy_mode;          /* SRK_IN_PLAIN | SRK_IN_UNSHUFFLE */
  const float* x;  int32_t x_ldc, x_coff;
  float in_slope;           /* LeakyReLU applied to X while staged (1 = none) */
  const float* dy; int32_t dy_ldc, dy_coff;
  float* dw;                /* [Cout][Cin][3][3] */
  float* db;                /* [Cout] or NULL */
  float scale;
  int32_t accumulate;
  void* workspace; size_t workspace_bytes;
  int32_t precision;        /* 0: exact fp32 MFMA (default).  1: split-bf16 operands, 3 bf16 MFMAs per product, fp32
                               accumulate (opt-in; stride 1, Cin % 8 == 0, Cout % 8 == 0, 16-byte addressable views).
                               2: plain bf16 operands (hi parts only), fp32 accumulate
                               3 / 4: x and dy are 16-bit tensors, fp16 (3) / bf16 (4), ldc / coff in elements (multiples of 8);
                                  dw / db stay fp32; stride 1, in_slope == 1 */
} srk_wgrad_args;

int srk_conv3x3_wgrad(const srk_wgrad_args* a, void* stream);
int srk_conv3x3_wgrad_workspace(const srk_wgrad_args* a, size_t* bytes);

/* n (<= 8) weight-gradient problems that share N,H,W,OH,OW,stride,dy_mode in ONE pair of launches (the five
 * convs of a DenseResidualBlock, models.py:24-28: their 15 (64 cout x 64 cin) chunks fill the chip with few
 * pixel-splits, so the partial-sum traffic stays small).  The workspace of args[0] is used for all. */
int srk_conv3x3_wgrad_batched(const srk_wgrad_args* args, int n, void* stream);
int srk_conv3x3_wgrad_batched_workspace(const srk_wgrad_args* args, int n, size_t* bytes);
/* n INDEPENDENT weight-gradient problems (any geometries: the layers of a discriminator, models.py:149-174) launched back to back on
 * `stream` from ONE call; every args[i] carries its own workspace fields (they may all name the same buffer: the launches are
 * ordered on the stream).  Stops at the first failing launch and returns its status. */
int srk_conv3x3_wgrad_seq(const srk_wgrad_args* args, int n, void* stream);
/* Test aid: routing of convolutions with <= 4 channels on one side to the HBM-bound kernels of srk_conv_small.hip:
 * 0 = never, 1 = when their 16x16 tiles fill the chip (default), 2 = whenever the shape allows (small parity cases). */
int srk_debug_set_conv_small(int mode);
/* Test aid: M tiles per workgroup of the F(2x4, 3x3) conv kernel (wp_format 6): 0 = by launch size (default), 1 = 16-row tiles,
 * 2 = 32-row tiles. */
int srk_debug_set_wino42_nmt(int nmt);
/* Test aid: form of the 2-D Winograd weight-gradient kernel (fp32, stride 1, > 32 channels each way): 2 = wino24 (default since round 4:
 * transposed F(2,3) along H x F(4,3) along W, a third of the direct kernel's MFMAs; each wave owns one row of the 4 x 6 transform
 * positions for the whole 64 x 64 chunk), 1 = wino22, row-owner form (F(2,3) both ways, 4/9 of the MFMAs; one row of the 4 x 4 positions
 * per wave), 0 = wino22, tile-owner form (each wave owns all 16 positions of a 32 x 32 tile), < 0 = back to the default /
 * SRK_WGRAD_W22_FORM. */
int srk_debug_set_wgrad_w22_form(int rows);
/* Measurement aid: writes the name (as rocprofv3 prints it) of the kernel srk_conv3x3 dispatches to for these arguments into
 * buf (NUL-terminated, truncated to len).  Launches nothing. */
int srk_conv3x3_kernel_name(const srk_conv_args* args, char* buf, size_t len);
/* Measurement aid: writes the name (as rocprofv3 prints it) of the main kernel srk_conv3x3_wgrad_batched dispatches to
 * for these arguments into buf (NUL-terminated, truncated to len).  Launches nothing. */
int srk_conv3x3_wgrad_kernel_name(const srk_wgrad_args* args, int n, char* buf, size_t len);

/* Weight packing (OIHW fp32 -> MFMA-fragment order).  One launch packs a whole table.
 * Destination layout for a conv with K input channels and M outputs, Mp = round_up(M, 32):
 *   dst[q][tap][h][m][e]  (q = K/8 chunks, tap = 3r+s, h in {0,1}, e in 0..3)
 *       = scale * SRC(m, k = 8q + 4h + e, tap)      (0 where k >= K or m >= M)
 * forward entry  (transpose == 0): SRC(m,k,tap) = W[m'][c_begin + k - k_off][tap]
 * backward entry (transpose == 1): SRC(m,k,tap) = W[k' ][c_begin + m][8 - tap], k' from k - k_off
 * stride-2 data gradient (transpose == 2; M = 4 * Cin): the data gradient of a STRIDE-2 conv written as a stride-1 conv on dy whose
 *   4 x Cin outputs are PixelShuffled into dx (srk_conv3x3 with ps_out; even H, W): output m = (2a + b) * Cin + c is dx channel c at
 *   pixel parity (a, b), SRC(m,k,3u+v) = W[k][c_begin + c][r(a,u)][s(b,v)] with r(0,1) = 1, r(1,1) = 2, r(1,2) = 0, else no tap (0):
 *   a quarter of the multiply-adds of the zero-upsample form, and no 16 -> 32 channel padding
 * where x' = ps ? 4*(x % (Cout_src/4)) + x / (Cout_src/4) : x maps packed PixelShuffle order to OIHW rows.
 * Several entries may fill disjoint k ranges [k_off, k_off + k_len) of one dst (the backward of a
 * DenseResidualBlock reads the concatenation of dy5..dy(j+1)). */
typedef struct srk_pack_entry {
  const float* src;         /* OIHW [src_cout][src_cin][3][3] */
  float* dst;
  int32_t src_cout, src_cin;
  int32_t transpose;
  int32_t c_begin;          /* first src input channel used */
  int32_t M;                /* logical outputs of dst (fwd: src_cout; bwd: slice width) */
  int32_t k_off, k_len;     /* dst k range filled by this entry (k_off % 8 == 0) */
  int32_t K_total;          /* total K of dst (for zero fill of the tail chunk) */
  int32_t ps;
  float scale;
  int32_t fmt;              /* 0: fp32 fragments [K/8][tap][h][Mp][4];  1: split bf16 [K/16][tap][hi|lo][h][Mp][8]
                               (same byte size; k_off % 16 == 0);  3: Winograd fp32 fragments [K/8][3 rows x 4 pos][h][Mp][4]
                               with u = G w folded in (srk_pack_weights; 4/3 the size);  5: F(4,3) fragments [K/8][3 rows x 6 pos][h][Mp][4] (twice the
                               size);  6: F(2x4, 3x3) fragments [K/8][2 channel pairs][4 row x 6 column positions][h][Mp][2] (8/3
                               the size);  7 / 8: fp16 / bf16 fragments [K/16][tap][h][Mp64][8] of 16-bit elements, Mp64 = M rounded up
                               to 64 (k_off % 16 == 0; transpose 0 / 1; 16-channel chunks of dst that no entry covers -- K_total padded to the
                               conv kernel's 32-channel stages -- are not written: the caller zeroes dst once).  One table = one format. */
  int64_t elem_begin;       /* prefix sum of work items, filled by srk_pack_plan */
} srk_pack_entry;

/* fills elem_begin for each entry, returns total work items in *total */
int srk_pack_plan(srk_pack_entry* host_entries, int n, int64_t* total);
/* device_entries: the same table copied to device memory by the caller */
int srk_pack_weights(const srk_pack_entry* device_entries, int n, int64_t total, void* stream);
/* the same for a table whose entries have fmt == 1 */
int srk_pack_weights_bf16x3(const srk_pack_entry* device_entries, int n, int64_t total, void* stream);
/* the same for a table whose entries have fmt == 7 (fp16) or 8 (bf16): 16-bit fragments of the 16-bit-storage kernels */
int srk_pack_weights_h16(const srk_pack_entry* device_entries, int n, int64_t total, int fmt, void* stream);
size_t srk_packed_floats(int K, int M);   /* rounds K up to 16: valid for formats 0 and 1 */
size_t srk_packed_floats_wino(int K, int M);   /* fmt 3: 12 transformed taps instead of 9 */
size_t srk_packed_floats_wino4(int K, int M);  /* fmt 5: 18 transformed taps */
size_t srk_packed_floats_wino42(int K, int M); /* fmt 6: 24 transformed taps */
size_t srk_packed_floats_h16(int K, int M);    /* fmt 7 / 8: size of the 16-bit fragment buffer in FLOAT units (4 bytes) */
/* Rows per wave of the 16-bit-storage conv kernel (wp_format 7 / 8): 0 = by launch size (default), 2 = 8-row tiles, 4 = 16-row tiles,
 * 1 = the shared-CU form (8-row tiles, 16-channel stages, two workgroups per CU: the engine's two-chain mode selects it) */
int srk_debug_set_h16_mt(int mt);
/* 1 if srk_conv3x3 accepts wp_format == 1 for this geometry (stride 1, Cin % 16 == 0, 16-byte addressable input) */
int srk_conv3x3_bf16x3_supported(const srk_conv_args* a);

/* Standalone nn.PixelShuffle(2) forward / backward on NHWC (models.py:89); C = channels of the
 * shuffled tensor, x is [N,H,W,4C] in OIHW channel order (c*4 + 2i + j), y is [N,2H,2W,C]. */
int srk_pixel_shuffle_fwd(const float* x, float* y, int N, int H, int W, int C, void* stream);
int srk_pixel_shuffle_bwd(const float* dy, float* dx, int N, int H, int W, int C, void* stream);

/* NCHW <-> NHWC view copies (boundary tensors are NCHW, esrgan.py:404-405) */
int srk_nchw_to_nhwc(const float* x, float* y, int y_ldc, int y_coff, int N, int C, int H, int W, void* stream);
int srk_nhwc_to_nchw(const float* x, int x_ldc, int x_coff, float* y, int N, int C, int H, int W, void* stream);

/* k*k sum pooling NHWC/NCHW with C folded into N (SumPool2d, models.py:297-305) and its gradient */
int srk_sum_pool_fwd(const float* x, float* y, int NC, int H, int W, int k, void* stream);
int srk_sum_pool_bwd(const float* dy, float* dx, int NC, int H, int W, int k, void* stream);

/* ---- Optional physics loss heads of the generator phase (esrgan.py:522-547), each ONE fused read pass forward and one
 * read+write pass backward over dense fp32 tensors; reductions are two-stage, fixed-order (deterministic).  `workspace`
 * must hold srk_loss_workspace_bytes() bytes.
 *   srk_sigmoid_*     y = sigmoid(scale*x + shift): softgreater(x,val,sigma,delta) = (sigma, sigma*(delta-val)) and
 *                     nnz_mask(x,sigma) = (sigma, 0) as stand-alone functions            utils.py:259-261,271-272
 *   srk_soft_count_*  out[b] = sum_i sigmoid(sigma*(x[b,i]-val))  (hard != 0: count(x > val), the target)
 *                     = softgreater(x,val,sigma).sum(1).sum(1).sum(1)                    esrgan.py:523-524
 *   srk_mask_l1_*     out[0] = mean_i |sigmoid(sigma*a_i) - sigmoid(sigma*b_i)| = L1Loss(nnz_mask(a), nnz_mask(b)),
 *                     gradient w.r.t. a                                                 esrgan.py:527-529
 *   srk_hitogram_*    t [BC][H][W] -> out[i*f+j] = mean_{bc,p,q} sigmoid(sig*(t[bc,f*p+i,f*q+j]-thr)) (sig <= 0: t)
 *                     = get_hitogram(t, f, thr, sig); f in {1,2,4,8}                      utils.py:264-268
 *   srk_soft_hist_*   (positive_only = 1; 0: sum over all i)  out[k] = sum_{x_i>0} sigmoid(sigma(x_i-c_k+d_k/2)) - sigmoid(sigma(x_i-c_k-d_k/2)), K <= 64
 *                     = DiffableHistogram(binedges, sigma)(x[x > 0])                     models.py:308-342, esrgan.py:533-536 */
int srk_loss_workspace_bytes(size_t* out);
int srk_sigmoid_fwd(const float* x, float* y, long n, float scale, float shift, void* stream);
int srk_sigmoid_bwd(const float* y, const float* gy, float* dx, long n, float scale, void* stream);
/* out[i] = g[i] * (x[i] > 0 ? 1 : slope): LeakyReLU' applied to a gradient, one pass.  Replaces the torch.where / mul / gt chain autograd
 * builds for the double backward of `LeakyReLU(0.2)` in the gradient penalty (models.py:149,151; esrgan.py:598-606). */
int srk_lrelu_grad_mul(const float* x, const float* g, float* out, long n, float slope, void* stream);
int srk_soft_count_fwd(const float* x, float* out, int B, long per_image, float sigma, float val, int hard, void* workspace,
                       size_t ws_bytes, void* stream);
int srk_soft_count_bwd(const float* x, const float* gout, float* dx, int B, long per_image, float sigma, float val, void* stream);
int srk_mask_l1_fwd(const float* a, const float* b, float* out, long n, float sigma, void* workspace, size_t ws_bytes, void* stream);
int srk_mask_l1_bwd(const float* a, const float* b, const float* gout, float* da, long n, float sigma, void* stream);
int srk_hitogram_fwd(const float* t, float* out, int BC, int H, int W, int factor, float thr, float sig, void* workspace,
                     size_t ws_bytes, void* stream);
int srk_hitogram_bwd(const float* t, const float* gout, float* dt, int BC, int H, int W, int factor, float thr, float sig, void* stream);
int srk_soft_hist_fwd(const float* x, long n, const float* centers, const float* delta, int K, float sigma, int positive_only,
                      float* out, void* workspace, size_t ws_bytes, void* stream);
int srk_soft_hist_bwd(const float* x, long n, const float* centers, const float* delta, int K, float sigma, int positive_only,
                      const float* gout, float* dx, void* stream);

/* Sparse jet decode: the batched form of datasets.py:136-145 `extract` (+ ThresholdImageCutter, datasets.py:170-175).
 * rows [B][row_stride] fp32 = interleaved (pos_i, E_i) pairs, L per event, zero-padded; out [B][etaBins][phiBins]
 * (= NCHW with one channel): out[b][pos % etaBins][pos // etaBins] += E in list order up to the first E == 0; pixels
 * <= threshold are zeroed (threshold < 0: no cut).  One workgroup per event, bit-identical to the sequential loop. */
int srk_jet_extract(const float* rows, int B, int L, int row_stride, int etaBins, int phiBins, float threshold, float* out,
                    void* stream);

/* Flat, self-contained entry points on CANONICAL OIHW fp32 weights (SURVEY.md section 8(b)'s signatures): the weights are
 * packed into the caller's workspace on `stream`, then the fused kernel above runs.  For callers that do not keep
 * packed weights alive (one conv at a time from another framework); the training engine uses srk_conv3x3 directly.
 * dtype: 0 = fp32 (the only one accepted here).  H, W = extent of the conv INPUT x (dx for dgrad); the output extent is
 * ceil(H/stride).  `residual` (optional) is dense [N,OH,OW,Cout]: y = res_scale * (conv + bias) + residual, the
 * DenseResidualBlock / RRDB tail (models.py:40,53); it excludes lrelu_slope != 1.  pixel_shuffle_r = 2 folds
 * nn.PixelShuffle(2) into the store (fwd; y is [N,2H,2W,Cout/4]) or its inverse into the load (dgrad; dy likewise).
 * Replaces: F.conv2d + LeakyReLU (+ residual) forward and its autograd input-gradient, models.py:19-41,63,67,86-99,142-168. */
enum { SRK_OP_CONV_FWD = 0, SRK_OP_CONV_DGRAD = 1, SRK_OP_CONV_WGRAD = 2 };
int srk_workspace_bytes(int op, int N, int H, int W, int Cin, int Cout, int dtype, size_t* out);
int srk_conv3x3_fwd(const void* x, int ldc_in, int c_in_off, int Cin, const void* w, const void* bias, void* y,
                    int ldc_out, int c_out_off, int Cout, int N, int H, int W, int stride, float lrelu_slope,
                    const void* residual, float res_scale, int pixel_shuffle_r, int dtype, void* workspace,
                    size_t ws_bytes, void* stream);
int srk_conv3x3_dgrad(const void* dy, int ldc_dy, int c_dy_off, int Cout, const void* w, void* dx, int ldc_dx,
                      int c_dx_off, int Cin, int N, int H, int W, int stride, int pixel_shuffle_r, int dtype,
                      void* workspace, size_t ws_bytes, void* stream);
/* Weight / bias gradient of the same convolution in flat form: dw [Cout,Cin,3,3] (canonical OIHW) and dbias [Cout] (may be
 * NULL) from the conv input x [N,H,W,ldc_in] and the output gradient dy [N,ceil(H/stride),ceil(W/stride),ldc_dy]
 * (pixel_shuffle_r = 2: dy is the gradient of the SHUFFLED output, [N,2H,2W,ldc_dy] with Cout/4 channels per pixel).
 * dw = scale * sum (+ dw if accumulate).  Workspace: srk_workspace_bytes(SRK_OP_CONV_WGRAD, ...).  Deterministic (fixed-order
 * split reduction, no float atomics).  Replaces the autograd weight gradient of nn.Conv2d, models.py:19,63,67,87,97,99,142-168. */
int srk_conv3x3_wgrad_flat(const void* x, int ldc_in, int c_in_off, int Cin, const void* dy, int ldc_dy, int c_dy_off, int Cout,
                           void* dw, void* dbias, int N, int H, int W, int stride, float scale, int accumulate,
                           int pixel_shuffle_r, int dtype, void* workspace, size_t ws_bytes, void* stream);
/* (srk_conv3x3_wgrad / srk_conv3x3_wgrad_batched above are the struct forms the engine uses: packed dense-block batches) */

/* torch.optim.Adam.step() of a whole parameter list in ONE launch (esrgan.py:299,305,487,623: the generator's and the discriminators'
 * optimizers): fp32 parameters, gradients, exp_avg, exp_avg_sq behind a device table of pointers.  Arithmetic of ATen's fused Adam
 * (amsgrad off, maximize off): g /= *grad_scale (if given); g += weight_decay * p; exp_avg = lerp(exp_avg, g, 1 - beta1);
 * exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) g^2; p -= lr / (1 - beta1^t) * exp_avg / (sqrt(exp_avg_sq) / sqrt(1 - beta2^t) + eps) with
 * t = *step (device, already incremented by the caller: srk_adam_count_step); nothing is touched when *found_inf != 0 (torch.amp.GradScaler's
 * contract; srk_adam_count_step folds a pending chain fault into that word).
 * srk_adam_plan fills chunk_begin of a host table and returns the launch size; the caller copies the table to the device. */
typedef struct srk_adam_entry {
  float* p; const float* g; float* m; float* v;
  int64_t n;
  int64_t chunk_begin;
} srk_adam_entry;
int srk_adam_plan(srk_adam_entry* host_entries, int n, int64_t* total_chunks);
int srk_adam_step(const srk_adam_entry* device_entries, int n, int64_t total_chunks, float lr, float beta1, float beta2, float eps,
                  float weight_decay, const float* step, const float* grad_scale, const float* found_inf, void* stream);
/* n <= 64 tensors from the HOST table itself (it travels in the kernel arguments: nothing to copy when the gradients are new tensors
 * every step, as a discriminator's are) */
int srk_adam_step_small(const srk_adam_entry* host_entries, int n, int64_t total_chunks, float lr, float beta1, float beta2, float eps,
                        float weight_decay, const float* step, const float* grad_scale, const float* found_inf, void* stream);

/* The counting half of an Adam step, and the ONE place where a step decides whether it happens: skip = (*found_inf != 0, found_inf may
 * be NULL) or a chain launch's fault is pending (srk_conv3x3_seq above; read on the device).  *skip_out = skip ? 1 : 0 (may be NULL),
 * *step += 1 unless skip (ATen: _foreach_add_(steps, 1) ... _foreach_sub_(steps, found_inf): a skipped update is not counted).  The
 * caller passes skip_out as the found_inf of srk_adam_step: all of its workgroups then act on the same decision. */
int srk_adam_count_step(float* step, const float* found_inf, float* skip_out, void* stream);

const char* srk_strerror(int status);
int srk_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SRK_H */
