"""GPU parity of the raw C-ABI kernels (libsrk.so) against the CPU oracle.  Tolerance: 1e-4 relative to
the output's max-abs for a single conv (fp32 MFMA is an fp32 fma chain; only summation order differs),
well inside BASELINE.json's 1e-3; PixelShuffle indexing is bit-exact on integer data."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.fixture(scope="module")
def U():
    import srk_testutil
    return srk_testutil


@pytest.mark.parametrize("ci,co,h,w,n", [
    (1, 64, 8, 8, 2), (3, 16, 9, 7, 1), (8, 32, 16, 16, 1), (16, 16, 20, 33, 2), (64, 64, 64, 64, 2),
    (320, 64, 16, 16, 1), (64, 256, 8, 24, 1), (64, 1, 33, 17, 1), (12, 40, 10, 10, 1), (64, 96, 8, 16, 1)])
def test_conv_fwd_plain(U, ci, co, h, w, n):
    L = U.L
    x = _rand((n, ci, h, w), 1)
    wt = _rand((co, ci, 3, 3), 2, 1.0 / np.sqrt(9 * ci))
    b = _rand((co,), 3, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b), 0.01)
    xd = U.nhwc(x)
    wp, _ = U.pack_fwd(wt)
    y = torch.full((n, h, w, co), float("nan"), device="cuda")
    L.conv3x3(L.View(xd), wp, b.cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, slope=0.01)
    assert U.rel_err(U.nchw(y), ref) < TOL


def test_conv_slices_residuals_mask(U):
    """concat-free dense-block addressing: read a channel prefix, write a channel slice, two residuals,
    alpha, and the LeakyReLU-backward mask (models.py:36-41,53)."""
    L = U.L
    n, h, w, F_ = 2, 12, 20, 16
    xfull = _rand((n, 5 * F_, h, w), 4)
    wt = _rand((F_, 3 * F_, 3, 3), 5, 0.05)
    b = _rand((F_,), 6, 0.1)
    r1 = _rand((n, F_, h, w), 7)
    r2 = _rand((n, F_, h, w), 8)
    m = _rand((n, F_, h, w), 9)
    ref = 0.2 * O.conv3x3(xfull[:, F_:4 * F_], wt, b) + 0.5 * r1 + 1.0 * r2
    ref = ref * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.01))
    buf = U.nhwc(xfull)                       # [n,h,w,5F]; read channels F..4F
    out = torch.zeros(n, h, w, 2 * F_, device="cuda")
    wp, _ = U.pack_fwd(wt)
    L.conv3x3(L.View(buf, F_, 3 * F_), wp, b.cuda(), L.View(out, F_, F_), N=n, H=h, W=w, OH=h, OW=w, Cin=3 * F_, Cout=F_,
              alpha=0.2, r1=L.View(U.nhwc(r1)), beta1=0.5, r2=L.View(U.nhwc(r2, ldc=24, coff=8), 8, F_), beta2=1.0,
              mask=L.View(U.nhwc(m)), mask_slope=0.01)
    assert U.rel_err(U.nchw(out, F_, F_), ref) < TOL
    assert out[..., :F_].abs().max().item() == 0.0      # neighbours of the slice untouched


@pytest.mark.parametrize("F_", [8, 16, 64])
def test_conv_pixel_shuffle_store(U, F_):
    """conv F->4F + LeakyReLU + PixelShuffle(2) fused (models.py:86-90)."""
    L = U.L
    n, h, w = 2, 9, 13
    x = _rand((n, F_, h, w), 10)
    wt = _rand((4 * F_, F_, 3, 3), 11, 0.08)
    b = _rand((4 * F_,), 12, 0.1)
    ref = O.pixel_shuffle(O.lrelu(O.conv3x3(x, wt, b), 0.01), 2)
    wp, _ = U.pack_fwd(wt, ps=True)
    # bias in packed order: packed o' = ij*F + c  <->  OIHW o = 4c + ij
    bp = b.view(F_, 4).t().contiguous().view(-1).cuda()
    y = torch.full((n, 2 * h, 2 * w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(x)), wp, bp, L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, ps_out=True, slope=0.01)
    assert U.rel_err(U.nchw(y), ref) < TOL


def test_pixel_shuffle_standalone_bit_exact(U, golden_dir):
    L = U.L
    d = np.load(os.path.join(golden_dir, "G1_pixel_shuffle.npz"))
    x = torch.from_numpy(d["x"]); yref = torch.from_numpy(d["y"])
    n, c4, h, w = x.shape
    xd = U.nhwc(x)
    y = torch.empty(n, 2 * h, 2 * w, c4 // 4, device="cuda")
    L.pixel_shuffle_fwd(xd, y, n, h, w, c4 // 4)
    assert torch.equal(U.nchw(y), yref)
    back = torch.empty_like(xd)
    L.pixel_shuffle_bwd(y, back, n, h, w, c4 // 4)
    assert torch.equal(back.cpu(), xd.cpu())


@pytest.mark.parametrize("ci,co,h,w", [(1, 16, 32, 32), (16, 16, 32, 32), (32, 32, 17, 23), (64, 64, 16, 16), (16, 32, 75, 75)])
def test_conv_stride2(U, ci, co, h, w):
    """discriminator_block's stride-2 conv + LeakyReLU(0.2) (models.py:144-145)."""
    L = U.L
    n = 2
    x = _rand((n, ci, h, w), 13)
    wt = _rand((co, ci, 3, 3), 14, 1.0 / np.sqrt(9 * ci))
    b = _rand((co,), 15, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b, stride=2), 0.2)
    oh, ow = ref.shape[2:]
    wp, _ = U.pack_fwd(wt)
    y = torch.full((n, oh, ow, co), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(x)), wp, b.cuda(), L.View(y), N=n, H=h, W=w, OH=oh, OW=ow, Cin=ci, Cout=co, stride=2, slope=0.2)
    assert U.rel_err(U.nchw(y), ref) < TOL


@pytest.mark.parametrize("ci,co,h,w,stride", [(64, 64, 16, 16, 1), (192, 64, 10, 18, 1), (1, 16, 16, 16, 1), (16, 16, 32, 32, 2),
                                                (32, 32, 17, 23, 2), (64, 1, 9, 9, 1), (3, 64, 8, 8, 1)])
def test_conv_dgrad(U, ci, co, h, w, stride):
    """data-gradient = the same kernel on transposed+flipped packed weights; stride 2 through zero-upsampling."""
    L = U.L
    n = 2
    x = _rand((n, ci, h, w), 16).requires_grad_(True)
    wt = _rand((co, ci, 3, 3), 17, 1.0 / np.sqrt(9 * ci))
    y = O.conv3x3(x, wt, None, stride=stride)
    dy = _rand(y.shape, 18)
    y.backward(dy)
    oh, ow = y.shape[2:]
    wpb, _ = U.pack_bwd(wt)
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    if stride == 1:
        L.conv3x3(L.View(U.nhwc(dy)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=co, Cout=ci)
    else:
        L.conv3x3(L.View(U.nhwc(dy)), wpb, None, L.View(dx), N=n, H=oh, W=ow, OH=h, OW=w, Cin=co, Cout=ci,
                  in_mode=L.IN_ZERO_UPSAMPLE)
    assert U.rel_err(U.nchw(dx), x.grad) < TOL


def test_conv_dgrad_unshuffle(U):
    """backward of conv->LeakyReLU->PixelShuffle: dy arrives in shuffled layout and is read through
    SRK_IN_UNSHUFFLE with ps-permuted transposed weights."""
    L = U.L
    n, F_, h, w = 2, 16, 8, 12
    x = _rand((n, F_, h, w), 19).requires_grad_(True)
    wt = _rand((4 * F_, F_, 3, 3), 20, 0.08)
    y = O.pixel_shuffle(O.conv3x3(x, wt, None), 2)
    g = _rand(y.shape, 21)
    y.backward(g)
    wpb, _ = U.pack_bwd(wt, ps=True)
    dx = torch.full((n, h, w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(g)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=4 * F_, Cout=F_, in_mode=L.IN_UNSHUFFLE)
    assert U.rel_err(U.nchw(dx), x.grad) < TOL


@pytest.mark.parametrize("ci,co,h,w,stride,n", [(64, 64, 16, 16, 1, 2), (128, 64, 12, 20, 1, 3), (1, 16, 16, 16, 1, 2),
                                                  (16, 16, 32, 32, 2, 2), (32, 32, 17, 23, 2, 1), (64, 1, 9, 9, 1, 2),
                                                  (320, 64, 64, 64, 1, 2), (3, 64, 8, 8, 1, 1), (64, 64, 75, 75, 2, 1),
                                                  # pixel-split ("KS") variants: 1 or 2 live wave tiles per chunk
                                                  (16, 32, 33, 40, 1, 2), (32, 64, 20, 24, 2, 2), (64, 32, 20, 24, 1, 1),
                                                  (48, 24, 11, 13, 2, 2), (128, 32, 16, 16, 1, 1), (16, 96, 16, 16, 2, 1),
                                                  # Winograd weight-gradient kernel (stride 1, > 32 channels each way): odd / ragged extents
                                                  (64, 64, 15, 17, 1, 2), (128, 128, 7, 9, 1, 1), (64, 192, 1, 1, 1, 3), (96, 64, 33, 3, 1, 1)])
def test_conv_wgrad(U, ci, co, h, w, stride, n):
    L = U.L
    x = _rand((n, ci, h, w), 22)
    wt = _rand((co, ci, 3, 3), 23, 0.05).requires_grad_(True)
    b = torch.zeros(co, requires_grad=True)
    y = O.conv3x3(x, wt, b, stride=stride)
    dy = _rand(y.shape, 24)
    y.backward(dy)
    oh, ow = y.shape[2:]
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    db = torch.full((co,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(dy)), dw, db, N=n, H=h, W=w, OH=oh, OW=ow, Cin=ci, Cout=co, stride=stride)
    assert U.rel_err(dw.cpu(), wt.grad) < TOL
    assert U.rel_err(db.cpu(), b.grad) < TOL
    # accumulate + scale
    L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(dy)), dw, db, N=n, H=h, W=w, OH=oh, OW=ow, Cin=ci, Cout=co, stride=stride,
                    scale=0.5, accumulate=True)
    assert U.rel_err(dw.cpu(), 1.5 * wt.grad) < TOL


@pytest.mark.parametrize("ci,co,h,w,n", [(64, 64, 15, 17, 2), (320, 64, 64, 64, 2), (96, 64, 33, 3, 1), (64, 192, 1, 1, 3), (128, 128, 7, 9, 1)])
def test_conv_wgrad_wino22_forms(U, ci, co, h, w, n):
    """The forms of the 2-D Winograd weight-gradient kernel (1: wino22 row-owner, 0: wino22 tile-owner, 2: wino24 = F(2,3) x F(4,3)) against
    the oracle; the two wino22 forms against each other to rounding (they add the same products, the transforms in a different order)."""
    L = U.L
    x = _rand((n, ci, h, w), 31)
    wt = _rand((co, ci, 3, 3), 32, 0.05).requires_grad_(True)
    b = torch.zeros(co, requires_grad=True)
    y = O.conv3x3(x, wt, b)
    dy = _rand(y.shape, 33)
    y.backward(dy)
    got = []
    try:
        for form in (1, 0, 2):
            L.lib().srk_debug_set_wgrad_w22_form(form)
            L.poison_lds()               # (what a kernel reads of LDS bytes it never wrote must not reach a result)
            dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
            db = torch.full((co,), float("nan"), device="cuda")
            L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(dy)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co)
            assert U.rel_err(dw.cpu(), wt.grad) < TOL and U.rel_err(db.cpu(), b.grad) < TOL, form
            got.append((dw.cpu(), db.cpu()))
    finally:
        L.lib().srk_debug_set_wgrad_w22_form(-1)
    assert U.rel_err(got[0][0], got[1][0]) < 1e-5 and U.rel_err(got[0][1], got[1][1]) < 1e-5


def test_conv_wgrad_unshuffle_wino(U):
    """upsampling conv at F=64 (Cout = 256): dy read through SRK_IN_UNSHUFFLE by the Winograd weight-gradient kernel."""
    L = U.L
    n, F_, h, w = 2, 64, 8, 12
    x = _rand((n, F_, h, w), 91)
    wt = _rand((4 * F_, F_, 3, 3), 92, 0.05).requires_grad_(True)
    b = torch.zeros(4 * F_, requires_grad=True)
    y = O.pixel_shuffle(O.conv3x3(x, wt, b), 2)
    g = _rand(y.shape, 93)
    y.backward(g)
    dw = torch.full((4 * F_, F_, 3, 3), float("nan"), device="cuda")
    db = torch.full((4 * F_,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(g)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, dy_mode=L.IN_UNSHUFFLE)
    assert U.rel_err(dw.cpu(), wt.grad) < TOL and U.rel_err(db.cpu(), b.grad) < TOL


def test_conv_wgrad_unshuffle(U):
    L = U.L
    n, F_, h, w = 2, 16, 8, 12
    x = _rand((n, F_, h, w), 25)
    wt = _rand((4 * F_, F_, 3, 3), 26, 0.08).requires_grad_(True)
    b = torch.zeros(4 * F_, requires_grad=True)
    y = O.pixel_shuffle(O.conv3x3(x, wt, b), 2)
    g = _rand(y.shape, 27)
    y.backward(g)
    dw = torch.full((4 * F_, F_, 3, 3), float("nan"), device="cuda")
    db = torch.full((4 * F_,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(g)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, dy_mode=L.IN_UNSHUFFLE)
    assert U.rel_err(dw.cpu(), wt.grad) < TOL
    assert U.rel_err(db.cpu(), b.grad) < TOL


def test_golden_G2(U, golden_dir):
    """single conv + bias + LeakyReLU(0.01) vs the reference-generated fixture (models.py:19-21)."""
    L = U.L
    d = np.load(os.path.join(golden_dir, "G2_conv_lrelu.npz"))
    for (ci, co) in [(1, 64), (64, 64), (320, 64), (64, 256), (64, 1)]:
        x = torch.from_numpy(d[f"x_{ci}_{co}"]); yref = torch.from_numpy(d[f"y_{ci}_{co}"])
        sd = O.closed_form_fill({"weight": torch.empty(co, ci, 3, 3), "bias": torch.empty(co)})
        wp, _ = U.pack_fwd(sd["weight"])
        n, _, h, w = x.shape
        y = torch.empty(n, h, w, co, device="cuda")
        L.conv3x3(L.View(U.nhwc(x)), wp, sd["bias"].cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, slope=0.01)
        assert U.rel_err(U.nchw(y), yref) < TOL, (ci, co)


def test_layout_and_sumpool(U, golden_dir):
    L = U.L
    x = _rand((2, 5, 6, 7), 30)
    buf = torch.zeros(2, 6, 7, 8, device="cuda")
    L.nchw_to_nhwc(x.cuda(), L.View(buf, 2, 5), 2, 5, 6, 7)
    assert torch.equal(U.nchw(buf, 2, 5), x)
    back = torch.empty(2, 5, 6, 7, device="cuda")
    L.nhwc_to_nchw(L.View(buf, 2, 5), back, 2, 5, 6, 7)
    assert torch.equal(back.cpu(), x)
    d = np.load(os.path.join(golden_dir, "G9_sumpool.npz"))
    xs = torch.from_numpy(d["x"]).cuda()
    for k, key in ((4, "y4"), (2, "y2")):
        y = torch.empty(2, 1, 16 // k, 16 // k, device="cuda")
        L.sum_pool_fwd(xs, y, 2, 16, 16, k)
        assert torch.allclose(y.cpu(), torch.from_numpy(d[key]), rtol=1e-6, atol=1e-6)
        dx = torch.empty_like(xs)
        L.sum_pool_bwd(y, dx, 2, 16, 16, k)
        assert torch.allclose(dx.cpu(), F.interpolate(y.cpu(), scale_factor=k, mode="nearest"))


def test_bad_args_report_status(U):
    L = U.L
    a = L.ConvArgs()
    assert L.lib().srk_conv3x3(None, None) == -1
    import ctypes
    assert L.lib().srk_conv3x3(ctypes.byref(a), None) == -1
    with pytest.raises(RuntimeError):
        L.check(-2, "x")


# ---------------------------------------------------------------------------------------------- split-bf16 mode
def _pack_fmt1(U, w_oihw, transpose=False, ps=False):
    L = U.L
    co, ci = w_oihw.shape[:2]
    K, M = (co, ci) if transpose else (ci, co)
    src = w_oihw.contiguous().cuda()
    dst = torch.empty(L.packed_floats(K, M), dtype=torch.float32, device="cuda")
    t = L.PackTable(src.device, fmt=1)
    t.add(src, dst, M=M, k_off=0, k_len=K, K_total=K, transpose=transpose, ps=ps)
    t.run()
    torch.cuda.synchronize()
    return dst, src


BF_TOL = 1e-4     # bf16x3: ~2^-16 relative operand precision -> a few 1e-5 of the output scale


@pytest.mark.parametrize("ci,co,h,w,n", [(16, 64, 8, 32, 1), (64, 64, 64, 64, 2), (320, 64, 16, 40, 1), (64, 256, 9, 33, 1),
                                         (128, 16, 20, 13, 2), (32, 96, 8, 8, 1)])
def test_conv_bf16x3_fwd(U, ci, co, h, w, n):
    L = U.L
    x = _rand((n, ci, h, w), 41)
    wt = _rand((co, ci, 3, 3), 42, 1.0 / np.sqrt(9 * ci))
    b = _rand((co,), 43, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b), 0.01)
    wp, _ = _pack_fmt1(U, wt)
    y = torch.full((n, h, w, co), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(x)), wp, b.cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, slope=0.01, wp_format=1)
    err = U.rel_err(U.nchw(y), ref)
    assert err < BF_TOL, err


def test_conv_bf16x3_slices_residual_mask_and_dgrad(U):
    L = U.L
    n, h, w, F_ = 2, 12, 40, 16
    xfull = _rand((n, 5 * F_, h, w), 44)
    wt = _rand((F_, 3 * F_, 3, 3), 45, 0.05)
    b = _rand((F_,), 46, 0.1)
    r1 = _rand((n, F_, h, w), 47)
    m = _rand((n, F_, h, w), 48)
    ref = 0.2 * O.conv3x3(xfull[:, F_:4 * F_], wt, b) + 0.5 * r1
    ref = ref * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.01))
    buf = U.nhwc(xfull)
    out = torch.zeros(n, h, w, 2 * F_, device="cuda")
    wp, _ = _pack_fmt1(U, wt)
    L.conv3x3(L.View(buf, F_, 3 * F_), wp, b.cuda(), L.View(out, F_, F_), N=n, H=h, W=w, OH=h, OW=w, Cin=3 * F_, Cout=F_,
              alpha=0.2, r1=L.View(U.nhwc(r1)), beta1=0.5, mask=L.View(U.nhwc(m)), mask_slope=0.01, wp_format=1)
    assert U.rel_err(U.nchw(out, F_, F_), ref) < BF_TOL
    # data gradient through the transposed packing
    x = _rand((n, 32, h, w), 49).requires_grad_(True)
    w2 = _rand((48, 32, 3, 3), 50, 0.06)
    y = O.conv3x3(x, w2, None)
    dy = _rand(y.shape, 51)
    y.backward(dy)
    wpb, _ = _pack_fmt1(U, w2, transpose=True)
    dx = torch.full((n, h, w, 32), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(dy)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=48, Cout=32, wp_format=1)
    assert U.rel_err(U.nchw(dx), x.grad) < BF_TOL


def test_conv_bf16x3_pixel_shuffle_and_unshuffle(U):
    L = U.L
    n, F_, h, w = 1, 64, 8, 12
    x = _rand((n, F_, h, w), 52).requires_grad_(True)
    wt = _rand((4 * F_, F_, 3, 3), 53, 0.04)
    b = _rand((4 * F_,), 54, 0.1)
    ref = O.pixel_shuffle(O.lrelu(O.conv3x3(x, wt, b), 0.01), 2)
    wp, _ = _pack_fmt1(U, wt, ps=True)
    bp = b.view(F_, 4).t().contiguous().view(-1).cuda()
    y = torch.full((n, 2 * h, 2 * w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(x.detach())), wp, bp, L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, ps_out=True, slope=0.01, wp_format=1)
    assert U.rel_err(U.nchw(y), ref.detach()) < BF_TOL
    lin = O.pixel_shuffle(O.conv3x3(x, wt, None), 2)
    g = _rand(lin.shape, 55)
    lin.backward(g)
    wpb, _ = _pack_fmt1(U, wt, transpose=True, ps=True)
    dx = torch.full((n, h, w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(g)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=4 * F_, Cout=F_, in_mode=L.IN_UNSHUFFLE, wp_format=1)
    assert U.rel_err(U.nchw(dx), x.grad) < BF_TOL


def test_conv_bf16x3_rejects_unsupported(U):
    L = U.L
    x = torch.zeros(1, 8, 8, 8, device="cuda"); y = torch.zeros(1, 8, 8, 64, device="cuda")
    wp = torch.zeros(L.packed_floats(8, 64), device="cuda")
    with pytest.raises(RuntimeError):
        L.conv3x3(L.View(x), wp, None, L.View(y), N=1, H=8, W=8, OH=8, OW=8, Cin=8, Cout=64, wp_format=1)   # Cin % 16 != 0


@pytest.mark.parametrize("ci,co,h,w,n", [(64, 64, 16, 16, 2), (128, 64, 12, 20, 3), (320, 64, 64, 64, 1), (16, 16, 9, 33, 2), (64, 8, 8, 8, 1)])
def test_conv_wgrad_bf16x3(U, ci, co, h, w, n):
    """split-bf16 weight gradient (transposed LDS reads, 3 MFMAs per product)."""
    L = U.L
    x = _rand((n, ci, h, w), 61)
    wt = _rand((co, ci, 3, 3), 62, 0.05).requires_grad_(True)
    b = torch.zeros(co, requires_grad=True)
    y = O.conv3x3(x, wt, b)
    dy = _rand(y.shape, 63)
    y.backward(dy)
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda")
    db = torch.full((co,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(dy)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, precision=1)
    assert U.rel_err(dw.cpu(), wt.grad) < BF_TOL
    assert U.rel_err(db.cpu(), b.grad) < BF_TOL


def test_conv_wgrad_bf16x3_batched_and_unshuffle(U):
    L = U.L
    n, F_, h, w = 2, 16, 8, 24
    buf = _rand((n, 5 * F_, h, w), 64)
    E = _rand((n, 5 * F_, h, w), 65)
    bufd, Ed = U.nhwc(buf), U.nhwc(E)
    probs, refs = [], []
    for k in (1, 3, 5):
        wt = torch.zeros(F_, k * F_, 3, 3, requires_grad=True); bb = torch.zeros(F_, requires_grad=True)
        y = O.conv3x3(buf[:, :k * F_], wt, bb)
        y.backward(E[:, (5 - k) * F_:(6 - k) * F_])
        dw = torch.full((F_, k * F_, 3, 3), float("nan"), device="cuda"); db = torch.full((F_,), float("nan"), device="cuda")
        probs.append(dict(x=L.View(bufd, 0, k * F_), dy=L.View(Ed, (5 - k) * F_, F_), dw=dw, db=db, Cin=k * F_, Cout=F_, scale=0.5))
        refs.append((wt.grad, bb.grad))
    L.conv3x3_wgrad_batched(probs, N=n, H=h, W=w, OH=h, OW=w, precision=1)
    for pr, (gw, gb) in zip(probs, refs):
        assert U.rel_err(pr["dw"].cpu(), 0.5 * gw) < BF_TOL and U.rel_err(pr["db"].cpu(), 0.5 * gb) < BF_TOL
    # PixelShuffle-side dy
    F2 = 32
    x = _rand((1, F2, 8, 12), 66)
    wt = _rand((4 * F2, F2, 3, 3), 67, 0.05).requires_grad_(True)
    b = torch.zeros(4 * F2, requires_grad=True)
    y = O.pixel_shuffle(O.conv3x3(x, wt, b), 2)
    g = _rand(y.shape, 68)
    y.backward(g)
    dw = torch.full((4 * F2, F2, 3, 3), float("nan"), device="cuda"); db = torch.full((4 * F2,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(U.nhwc(x)), L.View(U.nhwc(g)), dw, db, N=1, H=8, W=12, OH=8, OW=12, Cin=F2, Cout=4 * F2,
                    dy_mode=L.IN_UNSHUFFLE, precision=1)
    assert U.rel_err(dw.cpu(), wt.grad) < BF_TOL and U.rel_err(db.cpu(), b.grad) < BF_TOL


# ---------------------------------------------------------------------------------------------------------------
# flat entry points on canonical OIHW weights (SURVEY.md section 8(b) signatures): srk_conv3x3_fwd / srk_conv3x3_dgrad
@pytest.mark.parametrize("ci,co,h,w,n,stride,slope,res,ps", [
    (64, 64, 16, 16, 2, 1, 0.01, False, 0),      # DenseResidualBlock b1 (models.py:24)
    (320, 64, 16, 16, 1, 1, 1.0, True, 0),       # b5 + 0.2 * out + x (models.py:40)
    (16, 16, 33, 20, 2, 2, 0.2, False, 0),       # discriminator stride-2 conv (models.py:144)
    (16, 64, 8, 12, 2, 1, 0.01, False, 2),       # upsampling conv + LeakyReLU + PixelShuffle (models.py:86-89)
    (1, 16, 16, 16, 2, 1, 1.0, False, 0), (64, 1, 9, 9, 1, 1, 1.0, False, 0)])
def test_flat_conv_fwd(U, ci, co, h, w, n, stride, slope, res, ps):
    L = U.L
    x = _rand((n, ci, h, w), 50)
    wt = _rand((co, ci, 3, 3), 51, 1.0 / np.sqrt(9 * ci))
    b = _rand((co,), 52, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b, stride=stride), slope)
    oh, ow = ref.shape[2:]
    r = _rand(ref.shape, 53) if res else None
    if res:
        ref = 0.2 * ref + r
    if ps:
        ref = O.pixel_shuffle(ref, 2)
    # read a channel slice of a wider buffer, write into a channel slice
    xb = U.nhwc(x, ldc=ci + 8, coff=4)
    cy = ref.shape[1]
    y = torch.full((n, ref.shape[2], ref.shape[3], cy + 4), float("nan"), device="cuda")
    L.conv3x3_fwd_flat(L.View(xb, 4, ci), wt.cuda(), b.cuda(), L.View(y, 4, cy), N=n, H=h, W=w, Cin=ci, Cout=co, stride=stride,
                       slope=slope, residual=U.nhwc(r) if res else None, res_scale=0.2, ps=ps)
    assert U.rel_err(U.nchw(y, 4, cy), ref) < TOL
    assert torch.isnan(y[..., :4]).all()          # neighbouring channels untouched


@pytest.mark.parametrize("ci,co,h,w,n,stride,ps", [(64, 64, 16, 16, 2, 1, 0), (128, 64, 12, 20, 1, 1, 0), (16, 32, 33, 20, 2, 2, 0),
                                                   (32, 32, 16, 16, 1, 2, 0), (16, 64, 8, 12, 2, 1, 2), (1, 16, 16, 16, 1, 1, 0)])
def test_flat_conv_dgrad(U, ci, co, h, w, n, stride, ps):
    L = U.L
    x = _rand((n, ci, h, w), 60).requires_grad_(True)
    wt = _rand((co, ci, 3, 3), 61, 1.0 / np.sqrt(9 * ci))
    y = O.conv3x3(x, wt, None, stride=stride)
    if ps:
        y = O.pixel_shuffle(y, 2)
    g = _rand(y.shape, 62)
    y.backward(g)
    dx = torch.full((n, h, w, ci), float("nan"), device="cuda")
    L.conv3x3_dgrad_flat(L.View(U.nhwc(g)), wt.cuda(), L.View(dx), N=n, H=h, W=w, Cin=ci, Cout=co, stride=stride, ps=ps)
    assert U.rel_err(U.nchw(dx), x.grad) < TOL


@pytest.mark.parametrize("ci,co,h,w,n,stride,ps", [(64, 64, 16, 16, 2, 1, 0), (320, 64, 12, 20, 1, 1, 0), (16, 32, 33, 20, 2, 2, 0),
                                                   (1, 16, 32, 32, 2, 1, 0), (16, 64, 8, 12, 2, 1, 2), (64, 3, 9, 9, 1, 1, 0)])
def test_flat_conv_wgrad(U, ci, co, h, w, n, stride, ps):
    """srk_conv3x3_wgrad_flat: dw / dbias in canonical OIHW from strided channel slices, scale + accumulate."""
    L = U.L
    x = _rand((n, ci, h, w), 70)
    wt = _rand((co, ci, 3, 3), 71, 1.0 / np.sqrt(9 * ci)).requires_grad_(True)
    b = torch.zeros(co, requires_grad=True)
    y = O.conv3x3(x, wt, b, stride=stride)
    if ps:
        y = O.pixel_shuffle(y, 2)
    g = _rand(y.shape, 72)
    y.backward(g)
    xb = U.nhwc(x, ldc=ci + 8, coff=4)
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda"); db = torch.full((co,), float("nan"), device="cuda")
    L.conv3x3_wgrad_flat(L.View(xb, 4, ci), L.View(U.nhwc(g)), dw, db, N=n, H=h, W=w, Cin=ci, Cout=co, stride=stride, ps=ps)
    assert U.rel_err(dw.cpu(), wt.grad) < TOL and U.rel_err(db.cpu(), b.grad) < TOL
    L.conv3x3_wgrad_flat(L.View(xb, 4, ci), L.View(U.nhwc(g)), dw, None, N=n, H=h, W=w, Cin=ci, Cout=co, stride=stride, ps=ps,
                         scale=0.5, accumulate=True)
    assert U.rel_err(dw.cpu(), 1.5 * wt.grad) < TOL


def test_flat_workspace_and_errors(U):
    L = U.L
    import ctypes as C
    assert L.workspace_bytes(L.OP_CONV_FWD, 2, 16, 16, 64, 64) >= L.packed_floats(64, 64) * 4
    assert L.workspace_bytes(L.OP_CONV_WGRAD, 2, 16, 16, 64, 64) > 0
    n = C.c_size_t(0)
    assert L.lib().srk_workspace_bytes(L.OP_CONV_FWD, 2, 16, 16, 64, 64, 1, C.byref(n)) == -2      # dtype != fp32: unsupported
    assert L.lib().srk_workspace_bytes(7, 2, 16, 16, 64, 64, 0, C.byref(n)) == -1
    x = torch.zeros(1, 8, 8, 8, device="cuda"); y = torch.zeros(1, 8, 8, 8, device="cuda"); w = torch.zeros(8, 8, 3, 3, device="cuda")
    ws = torch.zeros(16, device="cuda")
    rc = L.lib().srk_conv3x3_fwd(x.data_ptr(), 8, 0, 8, w.data_ptr(), None, y.data_ptr(), 8, 0, 8, 1, 8, 8, 1, 1.0, None, 1.0, 0, 0,
                                 ws.data_ptr(), ws.numel() * 4, None)
    assert rc == -4                                                                                 # workspace too small


# ---------------------------------------------------------------------------------------------------------------
# Winograd kernels: wp_format 3 = F(2,3) along W (2/3 of the MFMAs), 5 = F(4,3) along W (1/2), 6 = 2-D F(2x4,3x3) (1/3; "62" below =
# the same format with 32-row workgroup tiles forced, 6 = 16-row tiles forced: both forms of the kernel on every shape); fp32 throughout
@pytest.fixture(autouse=True)
def _reset_w42_form(U):
    yield
    U.L.lib().srk_debug_set_wino42_nmt(0)


def _w42_form(U, fmt):
    """test parameter -> wp_format, selecting the workgroup-tile form of the wino42 kernel"""
    U.L.lib().srk_debug_set_wino42_nmt(2 if fmt == 62 else (1 if fmt == 6 else 0))
    return 6 if fmt == 62 else fmt


@pytest.mark.parametrize("ci,co,h,w,n", [(64, 64, 64, 64, 2), (320, 64, 16, 16, 1), (8, 64, 9, 7, 2), (64, 128, 33, 17, 1),
                                         (128, 64, 20, 40, 1), (16, 64, 5, 3, 1), (64, 64, 1, 1, 1), (24, 192, 16, 31, 2)])
@pytest.mark.parametrize("fmt", [3, 5, 6, 62])
def test_wino_conv_fwd(U, ci, co, h, w, n, fmt):
    L = U.L
    fmt = _w42_form(U, fmt)
    x = _rand((n, ci, h, w), 71)
    wt = _rand((co, ci, 3, 3), 72, 1.0 / np.sqrt(9 * ci))
    b = _rand((co,), 73, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b), 0.01)
    wp, _ = U.pack_fwd(wt, fmt=fmt)
    y = torch.full((n, h, w, co), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(x)), wp, b.cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, slope=0.01, wp_format=fmt)
    assert U.rel_err(U.nchw(y), ref) < TOL


@pytest.mark.parametrize("fmt", [3, 5, 6, 62])
def test_wino_slices_residuals_mask_and_dgrad(U, fmt):
    fmt = _w42_form(U, fmt)
    """the dense-block addressing (channel prefix in, channel slice out, two residuals, alpha, LeakyReLU' mask) and the
    data gradient (transposed, tap-flipped weights through the same transform) on the Winograd kernel."""
    L = U.L
    n, h, w, F_ = 2, 12, 20, 64
    xfull = _rand((n, 5 * F_, h, w), 74)
    wt = _rand((F_, 3 * F_, 3, 3), 75, 0.03)
    b = _rand((F_,), 76, 0.1)
    r1, r2, m = _rand((n, F_, h, w), 77), _rand((n, F_, h, w), 78), _rand((n, F_, h, w), 79)
    ref = 0.2 * O.conv3x3(xfull[:, F_:4 * F_], wt, b) + 0.5 * r1 + 1.0 * r2
    ref = ref * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.01))
    buf = U.nhwc(xfull)
    out = torch.zeros(n, h, w, 2 * F_, device="cuda")
    wp, _ = U.pack_fwd(wt, fmt=fmt)
    L.conv3x3(L.View(buf, F_, 3 * F_), wp, b.cuda(), L.View(out, F_, F_), N=n, H=h, W=w, OH=h, OW=w, Cin=3 * F_, Cout=F_, alpha=0.2,
              r1=L.View(U.nhwc(r1)), beta1=0.5, r2=L.View(U.nhwc(r2)), beta2=1.0, mask=L.View(U.nhwc(m)), mask_slope=0.01, wp_format=fmt)
    assert U.rel_err(U.nchw(out, F_, F_), ref) < TOL
    assert out[..., :F_].abs().max().item() == 0
    # data gradient: dx = conv(dy, flip(W)^T)
    x = _rand((n, F_, h, w), 80).requires_grad_(True)
    w2 = _rand((2 * F_, F_, 3, 3), 81, 0.03)
    y = O.conv3x3(x, w2, None)
    g = _rand(y.shape, 82)
    y.backward(g)
    wpb, _ = U.pack_bwd(w2, fmt=fmt)
    dx = torch.full((n, h, w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(g)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=2 * F_, Cout=F_, wp_format=fmt)
    assert U.rel_err(U.nchw(dx), x.grad) < TOL


@pytest.mark.parametrize("fmt", [3, 5, 6, 62])
def test_wino_pixel_shuffle_fold_and_unshuffle(U, fmt):
    fmt = _w42_form(U, fmt)
    L = U.L
    n, F_, h, w = 2, 64, 8, 12
    x = _rand((n, F_, h, w), 83).requires_grad_(True)
    wt = _rand((4 * F_, F_, 3, 3), 84, 0.04)
    b = _rand((4 * F_,), 85, 0.1)
    y = O.pixel_shuffle(O.lrelu(O.conv3x3(x, wt, b), 0.01), 2)
    wp, _ = U.pack_fwd(wt, ps=True, fmt=fmt)
    bp = b.view(-1, 4).t().contiguous().view(-1).cuda()
    out = torch.full((n, 2 * h, 2 * w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(x.detach())), wp, bp, L.View(out), N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, slope=0.01, ps_out=True, wp_format=fmt)
    assert U.rel_err(U.nchw(out), y.detach()) < TOL
    g = _rand(y.shape, 86)
    y2 = O.pixel_shuffle(O.conv3x3(x, wt, None), 2)
    y2.backward(g)
    wpb, _ = U.pack_bwd(wt, ps=True, fmt=fmt)
    dx = torch.full((n, h, w, F_), float("nan"), device="cuda")
    L.conv3x3(L.View(U.nhwc(g)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=4 * F_, Cout=F_, in_mode=L.IN_UNSHUFFLE, wp_format=fmt)
    assert U.rel_err(U.nchw(dx), x.grad) < TOL


def test_wino_rejects_unsupported(U):
    L = U.L
    x = torch.zeros(1, 8, 8, 8, device="cuda"); y = torch.zeros(1, 8, 8, 32, device="cuda"); wp = torch.zeros(L.packed_floats(8, 32, 3), device="cuda")
    with pytest.raises(RuntimeError):
        L.conv3x3(L.View(x), wp, None, L.View(y), N=1, H=8, W=8, OH=8, OW=8, Cin=8, Cout=32, wp_format=3)      # Cout % 64
    y2 = torch.zeros(1, 4, 4, 64, device="cuda")
    with pytest.raises(RuntimeError):
        L.conv3x3(L.View(x), wp, None, L.View(y2), N=1, H=8, W=8, OH=4, OW=4, Cin=8, Cout=64, stride=2, wp_format=3)


@pytest.fixture
def force_small(U):
    """Route every eligible conv to the HBM-bound small-channel kernels (srk_conv_small.hip), whatever the tile count."""
    U.L.lib().srk_debug_set_conv_small(2)
    yield
    U.L.lib().srk_debug_set_conv_small(1)


@pytest.mark.parametrize("ci,co,h,w,n", [(1, 16, 32, 32, 2), (1, 64, 19, 37, 1), (3, 16, 16, 16, 2), (3, 64, 33, 18, 1), (2, 8, 9, 7, 1), (4, 32, 20, 20, 1),
                                         (16, 1, 32, 32, 2), (64, 1, 21, 35, 1), (64, 3, 16, 48, 1), (8, 2, 9, 9, 1), (40, 4, 17, 16, 1), (16, 1, 256, 256, 1)])
def test_small_channel_kernels_fwd(U, force_small, ci, co, h, w, n):
    """conv3x3_cin_small_kernel (Cin <= 4) / conv3x3_cout_small_kernel (Cout <= 4): bias + LeakyReLU epilogue, ragged edges,
    against the CPU oracle (models.py:63,99,142,168)."""
    L = U.L
    x = _rand((n, ci, h, w), 41)
    wt = _rand((co, ci, 3, 3), 42, 1.0 / np.sqrt(9 * ci))
    b = _rand((co,), 43, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b), 0.2)
    wp, _ = U.pack_fwd(wt)
    y = torch.full((n, h, w, co), float("nan"), device="cuda")
    a = L.ConvArgs(); a.wp_format = 0; a.stride = 1; a.in_mode = 0; a.N, a.H, a.W, a.OH, a.OW, a.Cin, a.Cout = n, h, w, h, w, ci, co
    xd = U.nhwc(x); a.x, a.x_ldc, a.x_coff = xd.data_ptr(), ci, 0; a.wp = wp.data_ptr(); a.y, a.y_ldc, a.y_coff = y.data_ptr(), co, 0
    assert ("cin_small" if ci <= 4 else "cout_small") in L._conv_kernel_name(a)
    L.conv3x3(L.View(xd), wp, b.cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, slope=0.2)
    assert U.rel_err(U.nchw(y), ref) < TOL


@pytest.mark.parametrize("ci,co", [(1, 16), (3, 8), (16, 1), (24, 3)])
def test_small_channel_kernels_views_and_fused_epilogue(U, force_small, ci, co):
    """channel-slice views on both sides, input LeakyReLU (the discriminator's pre-activation chain), alpha, two residuals and
    the LeakyReLU' output mask -- the same fused epilogue as the MFMA kernels."""
    L = U.L
    n, h, w = 2, 18, 21
    xfull = _rand((n, ci + 8, h, w), 44)
    wt = _rand((co, ci, 3, 3), 45, 0.2)
    b = _rand((co,), 46, 0.1)
    r1, r2, m = _rand((n, co, h, w), 47), _rand((n, co, h, w), 48), _rand((n, co, h, w), 49)
    ref = 0.3 * O.conv3x3(O.lrelu(xfull[:, 4:4 + ci], 0.2), wt, b) + 0.5 * r1 - 1.5 * r2
    ref = O.lrelu(ref, 0.1) * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.01))
    buf = U.nhwc(xfull)
    out = torch.zeros(n, h, w, co + 8, device="cuda")
    wp, _ = U.pack_fwd(wt)
    L.conv3x3(L.View(buf, 4, ci), wp, b.cuda(), L.View(out, 4, co), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, in_slope=0.2, alpha=0.3, slope=0.1,
              r1=L.View(U.nhwc(r1)), beta1=0.5, r2=L.View(U.nhwc(r2, ldc=co + 4, coff=4), 4, co), beta2=-1.5, mask=L.View(U.nhwc(m)), mask_slope=0.01)
    assert U.rel_err(U.nchw(out, 4, co), ref) < TOL
    assert out[..., :4].abs().max().item() == 0.0 and out[..., 4 + co:].abs().max().item() == 0.0


@pytest.mark.parametrize("ci,co,h,w,n", [(16, 16, 32, 32, 2), (32, 32, 64, 48, 1), (64, 64, 16, 16, 2), (16, 32, 8, 24, 1), (16, 16, 33, 20, 1), (16, 16, 256, 256, 1)])
def test_stride2_dgrad_as_pixel_shuffle_conv(U, ci, co, h, w, n):
    """Data gradient of the discriminator's stride-2 layers (models.py:144) through the transpose == 2 packing: a stride-1 conv on dy
    with 4 * ci outputs stored through the PixelShuffle epilogue, LeakyReLU' mask of the layer below fused in; odd extents take the
    zero-upsample form.  Against autograd on the CPU oracle."""
    import importlib
    ops = importlib.import_module("super-resolution_amd.ops")
    models = importlib.import_module("super-resolution_amd.models")
    conv = models.Conv3x3(ci, co, stride=2).cuda()
    wt = _rand((co, ci, 3, 3), 92, 1.0 / np.sqrt(9 * ci))
    conv.weight.data.copy_(wt)
    x = _rand((n, ci, h, w), 91).requires_grad_(True)
    y = O.conv3x3(O.lrelu(x, 0.2), wt, None, stride=2)
    dy = _rand(y.shape, 93)
    y.backward(dy)
    pc = ops.PackedConvs([conv])
    pc.refresh(need_bwd=True)
    assert getattr(pc.bwd[0], "s2pack", None) is not None
    dx = ops.conv_dgrad_raw(U.nhwc(dy), conv.weight, U.nhwc(x.detach()), 2, 0.2, h, w, wpt=pc.bwd[0])
    assert U.rel_err(U.nchw(dx), x.grad) < TOL
