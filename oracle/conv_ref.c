/* conv_ref.c -- plain-C restatement of the hot path's primitive ops.  TEST INFRASTRUCTURE ONLY (see
 * oracle/esrgan_oracle.py): used by tests/ to pin the PyTorch-CPU oracle's primitives to first principles.
 * Semantics follow the ops the reference dispatches:
 *   nn.Conv2d(cin, cout, 3, stride, 1)   /root/reference/models.py:19,63,67,87,97,99,142,144,168
 *   nn.LeakyReLU(slope)                  /root/reference/models.py:21,143
 *   nn.PixelShuffle(2)                   /root/reference/models.py:89
 *   SumPool2d                            /root/reference/models.py:297-305
 * NCHW fp32, double accumulation (so it bounds both the oracle's and the kernel's rounding).
 * Build: gcc -O2 -shared -fPIC oracle/conv_ref.c -o oracle/_build/libconvref.so
 */
#include <stddef.h>

void ref_conv3x3(const float* x, const float* w, const float* b, float* y, int N, int Cin, int H, int W, int Cout,
                 int stride, float slope) {
  const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
  for (int n = 0; n < N; ++n)
    for (int o = 0; o < Cout; ++o)
      for (int oh = 0; oh < OH; ++oh)
        for (int ow = 0; ow < OW; ++ow) {
          double acc = b ? (double)b[o] : 0.0;
          for (int c = 0; c < Cin; ++c)
            for (int r = 0; r < 3; ++r) {
              const int ih = oh * stride + r - 1;
              if (ih < 0 || ih >= H) continue;
              for (int s = 0; s < 3; ++s) {
                const int iw = ow * stride + s - 1;
                if (iw < 0 || iw >= W) continue;
                acc += (double)w[((size_t)(o * Cin + c) * 3 + r) * 3 + s] * (double)x[((size_t)(n * Cin + c) * H + ih) * W + iw];
              }
            }
          float v = (float)acc;
          y[((size_t)(n * Cout + o) * OH + oh) * OW + ow] = v > 0.f ? v : v * slope;
        }
}

/* out[n,c,2h+i,2w+j] = in[n,4c+2i+j,h,w] */
void ref_pixel_shuffle2(const float* x, float* y, int N, int C4, int H, int W) {
  const int C = C4 / 4;
  for (int n = 0; n < N; ++n)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w)
          for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
              y[((size_t)(n * C + c) * 2 * H + 2 * h + i) * 2 * W + 2 * w + j] =
                  x[((size_t)(n * C4 + 4 * c + 2 * i + j) * H + h) * W + w];
}

void ref_sum_pool(const float* x, float* y, int NC, int H, int W, int k) {
  const int OH = H / k, OW = W / k;
  for (int p = 0; p < NC; ++p)
    for (int oh = 0; oh < OH; ++oh)
      for (int ow = 0; ow < OW; ++ow) {
        double s = 0;
        for (int i = 0; i < k; ++i)
          for (int j = 0; j < k; ++j) s += x[((size_t)p * H + oh * k + i) * W + ow * k + j];
        y[((size_t)p * OH + oh) * OW + ow] = (float)s;
      }
}
