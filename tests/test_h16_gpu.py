"""GPU parity of the 16-bit-storage kernels (wp_format 7 = fp16, 8 = bf16; BASELINE configs[4]'s reduced-precision path) against
the CPU oracle.  The oracle runs in fp32 on the SAME 16-bit-rounded inputs and weights, so what is left is accumulation order and the
one rounding of the 16-bit output: tolerance 2 ulp of the storage type at the output's max-abs (fp16: 2^-10, bf16: 2^-7); the fp32
outputs (weight gradients, the last conv's fp32 image) are held to 2e-4 / 2e-3."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
DT = {7: torch.float16, 8: torch.bfloat16}
TOL16 = {7: 2.0 ** -10, 8: 2.0 ** -7}


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


@pytest.fixture(scope="module")
def U():
    import srk_testutil
    return srk_testutil


@pytest.fixture(autouse=True)
def _reset_form(U):
    yield
    U.L.lib().srk_debug_set_h16_mt(0)
    U.L.lib().srk_debug_set_h16_chain(1)
    U.L.lib().srk_debug_set_h16_chain_m16(1)


def _q(t, fmt):
    """round to the storage type and back: what the kernel sees"""
    return t.to(DT[fmt]).float()


def _nhwc16(U, x_nchw, fmt, ldc=None, coff=0):
    return U.nhwc(x_nchw, ldc, coff).to(DT[fmt]).contiguous()


def _pack(U, w, fmt, transpose=False, ps=False, k_pad=None, scale=1.0, c_begin=0, c_len=None):
    L = U.L
    co, ci = w.shape[:2]
    src = w.contiguous().cuda()
    if transpose:
        K, M = co, (c_len or ci)
    else:
        K, M = ci, co
    Kt = k_pad or K
    dst = torch.zeros(L.packed_floats(Kt, M, fmt), dtype=torch.float32, device="cuda")
    t = L.PackTable(src.device, fmt)
    t.add(src, dst, M=M, k_off=0, k_len=K, K_total=Kt, transpose=transpose, ps=ps, scale=scale, c_begin=c_begin)
    t.run()
    torch.cuda.synchronize()
    return L_PW(dst, fmt)


class L_PW:
    def __init__(self, t, fmt):
        self.t, self.fmt = t, fmt

    def data_ptr(self):
        return self.t.data_ptr()


@pytest.mark.parametrize("fmt", [7, 8])
@pytest.mark.parametrize("mt", [1, 2, 4])          # 1 = the shared-CU form (conv3x3_h16s_kernel), 2 / 4 = 8- / 16-row tiles
@pytest.mark.parametrize("ci,co,h,w,n", [(64, 64, 32, 32, 2), (320, 64, 16, 32, 1), (32, 64, 9, 7, 2), (64, 128, 33, 17, 1),
                                         (128, 64, 20, 40, 1), (32, 64, 5, 3, 1), (64, 64, 1, 1, 1), (96, 192, 16, 35, 2)])
def test_h16_conv_fwd(U, fmt, mt, ci, co, h, w, n):
    L = U.L
    L.lib().srk_debug_set_h16_mt(mt)
    x = _q(_rand((n, ci, h, w), 71), fmt)
    wt = _q(_rand((co, ci, 3, 3), 72, 1.0 / np.sqrt(9 * ci)), fmt)
    b = _rand((co,), 73, 0.1)
    ref = O.lrelu(O.conv3x3(x, wt, b), 0.01)
    wp = _pack(U, wt, fmt)
    y = torch.full((n, h, w, co), float("nan"), device="cuda", dtype=DT[fmt])
    L.conv3x3(L.View(_nhwc16(U, x, fmt)), wp, b.cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co, slope=0.01)
    assert U.rel_err(U.nchw(y.float()), ref) < TOL16[fmt]


@pytest.mark.parametrize("fmt", [7, 8])
@pytest.mark.parametrize("mt", [1, 2, 4])
def test_h16_slices_residuals_mask_and_dgrad(U, fmt, mt):
    """the dense-block addressing (channel prefix in, channel slice out, two residuals, alpha, LeakyReLU' mask) and the data
    gradient (transposed, tap-flipped weights) on the 16-bit kernels"""
    L = U.L
    L.lib().srk_debug_set_h16_mt(mt)
    n, h, w, F_ = 2, 12, 37, 64
    xfull = _q(_rand((n, 5 * F_, h, w), 74), fmt)
    wt = _q(_rand((F_, 3 * F_, 3, 3), 75, 0.03), fmt)
    b = _rand((F_,), 76, 0.1)
    r1, r2, m = _q(_rand((n, F_, h, w), 77), fmt), _q(_rand((n, F_, h, w), 78), fmt), _q(_rand((n, F_, h, w), 79), fmt)
    ref = 0.2 * O.conv3x3(xfull[:, F_:4 * F_], wt, b) + 0.5 * r1 + 1.0 * r2
    ref = ref * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.01))
    buf = _nhwc16(U, xfull, fmt)
    out = torch.zeros(n, h, w, 2 * F_, device="cuda", dtype=DT[fmt])
    wp = _pack(U, wt, fmt)
    for aux in ("r1r2m", "r1", "m", "r1r2"):
        kw = {}
        refa = 0.2 * O.conv3x3(xfull[:, F_:4 * F_], wt, b)
        if "r1" in aux:
            kw.update(r1=L.View(_nhwc16(U, r1, fmt)), beta1=0.5); refa = refa + 0.5 * r1
        if "r2" in aux:
            kw.update(r2=L.View(_nhwc16(U, r2, fmt)), beta2=1.0); refa = refa + r2
        if "m" in aux:
            kw.update(mask=L.View(_nhwc16(U, m, fmt)), mask_slope=0.01)
            refa = refa * torch.where(m > 0, torch.ones_like(m), torch.full_like(m, 0.01))
        out.zero_()
        L.conv3x3(L.View(buf, F_, 3 * F_), wp, b.cuda(), L.View(out, F_, F_), N=n, H=h, W=w, OH=h, OW=w, Cin=3 * F_, Cout=F_, alpha=0.2, **kw)
        assert U.rel_err(U.nchw(out.float(), F_, F_), refa) < TOL16[fmt], aux
        assert out[..., :F_].abs().max().item() == 0
    # data gradient: dx = conv(dy, flip(W)^T)
    x = _rand((n, F_, h, w), 80).requires_grad_(True)
    w2 = _q(_rand((2 * F_, F_, 3, 3), 81, 0.03), fmt)
    y = O.conv3x3(x, w2, None)
    g = _q(_rand(y.shape, 82), fmt)
    y.backward(g)
    wpb = _pack(U, w2, fmt, transpose=True)
    dx = torch.full((n, h, w, F_), float("nan"), device="cuda", dtype=DT[fmt])
    L.conv3x3(L.View(_nhwc16(U, g, fmt)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=2 * F_, Cout=F_)
    assert U.rel_err(U.nchw(dx.float()), x.grad) < TOL16[fmt]


@pytest.mark.parametrize("fmt", [7, 8])
@pytest.mark.parametrize("mt", [1, 2, 4])
def test_h16_pixel_shuffle_fold_and_unshuffle(U, fmt, mt):
    L = U.L
    L.lib().srk_debug_set_h16_mt(mt)
    n, F_, h, w = 2, 64, 8, 12
    x = _q(_rand((n, F_, h, w), 83), fmt).requires_grad_(True)
    wt = _q(_rand((4 * F_, F_, 3, 3), 84, 0.04), fmt)
    b = _rand((4 * F_,), 85, 0.1)
    y = O.pixel_shuffle(O.lrelu(O.conv3x3(x, wt, b), 0.01), 2)
    wp = _pack(U, wt, fmt, ps=True)
    bp = b.view(-1, 4).t().contiguous().view(-1).cuda()
    out = torch.full((n, 2 * h, 2 * w, F_), float("nan"), device="cuda", dtype=DT[fmt])
    L.conv3x3(L.View(_nhwc16(U, x.detach(), fmt)), wp, bp, L.View(out), N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, slope=0.01, ps_out=True)
    assert U.rel_err(U.nchw(out.float()), y.detach()) < TOL16[fmt]
    g = _q(_rand(y.shape, 86), fmt)
    y2 = O.pixel_shuffle(O.conv3x3(x, wt, None), 2)
    y2.backward(g)
    wpb = _pack(U, wt, fmt, transpose=True, ps=True)
    dx = torch.full((n, h, w, F_), float("nan"), device="cuda", dtype=DT[fmt])
    L.conv3x3(L.View(_nhwc16(U, g, fmt)), wpb, None, L.View(dx), N=n, H=h, W=w, OH=h, OW=w, Cin=4 * F_, Cout=F_, in_mode=L.IN_UNSHUFFLE)
    assert U.rel_err(U.nchw(dx.float()), x.grad) < TOL16[fmt]


@pytest.mark.parametrize("fmt", [7, 8])
def test_h16_boundary_convs_padded_channels_and_fp32_output(U, fmt):
    """the image-side convs of the generator in the 16-bit mode: conv1 reads the image zero-padded to 32 channels (K_total = 32),
    conv3.2 writes the fp32 image (SRK_CONV_OUT_F32, Cout = 3) -- models.py:63,99"""
    L = U.L
    n, h, w, F_, C_ = 2, 19, 41, 64, 3
    img = _q(_rand((n, C_, h, w), 91), fmt)
    w1 = _q(_rand((F_, C_, 3, 3), 92, 0.2), fmt)
    b1 = _rand((F_,), 93, 0.1)
    ref = O.conv3x3(img, w1, b1)
    x16 = _nhwc16(U, img, fmt, ldc=32)
    y = torch.full((n, h, w, F_), float("nan"), device="cuda", dtype=DT[fmt])
    L.conv3x3(L.View(x16), _pack(U, w1, fmt, k_pad=32), b1.cuda(), L.View(y), N=n, H=h, W=w, OH=h, OW=w, Cin=32, Cout=F_)
    assert U.rel_err(U.nchw(y.float()), ref) < TOL16[fmt]
    feat = _q(_rand((n, F_, h, w), 94), fmt)
    w3 = _q(_rand((C_, F_, 3, 3), 95, 0.05), fmt)
    b3 = _rand((C_,), 96, 0.1)
    ref3 = O.conv3x3(feat, w3, b3)
    out = torch.full((n, h, w, C_), float("nan"), device="cuda", dtype=torch.float32)
    L.conv3x3(L.View(_nhwc16(U, feat, fmt)), _pack(U, w3, fmt), b3.cuda(), L.View(out), N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=C_,
              flags=L.CONV_OUT_F32)
    assert U.rel_err(U.nchw(out), ref3) < 2e-5 * (1 if fmt == 7 else 8) + 1e-5


def test_h16_rejects_unsupported(U):
    L = U.L
    x = torch.zeros(1, 8, 8, 48, device="cuda", dtype=torch.float16); y = torch.zeros(1, 8, 8, 64, device="cuda", dtype=torch.float16)
    wp = L_PW(torch.zeros(L.packed_floats(64, 64, 7), device="cuda"), 7)
    with pytest.raises(RuntimeError):
        L.conv3x3(L.View(x), wp, None, L.View(y), N=1, H=8, W=8, OH=8, OW=8, Cin=48, Cout=64)      # Cin % 32
    with pytest.raises(ValueError):
        L.conv3x3(L.View(x.float()), wp, None, L.View(y), N=1, H=8, W=8, OH=8, OW=8, Cin=32, Cout=64)   # fp32 view with a 16-bit format


# ---------------------------------------------------------------------------------------------------------------- weight gradient
WPREC = {7: 3, 8: 4}


@pytest.mark.parametrize("fmt", [7, 8])
@pytest.mark.parametrize("ci,co,h,w,n", [(64, 64, 16, 16, 2), (128, 64, 24, 40, 1), (64, 128, 9, 7, 2), (320, 64, 8, 16, 1), (16, 64, 33, 17, 1),
                                         (64, 16, 5, 50, 1), (40, 72, 20, 20, 1)])
def test_h16_wgrad(U, fmt, ci, co, h, w, n):
    """dW / db in fp32 from 16-bit x and dy (precision 3 / 4) vs autograd on the same rounded tensors: products of 16-bit values are
    exact in fp32, so only the summation order differs"""
    L = U.L
    x = _q(_rand((n, ci, h, w), 101), fmt).requires_grad_(False)
    wt = torch.zeros(co, ci, 3, 3, requires_grad=True)
    b = torch.zeros(co, requires_grad=True)
    dy = _q(_rand((n, co, h, w), 102), fmt)
    O.conv3x3(x, wt, b).backward(dy)
    dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda"); db = torch.full((co,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(_nhwc16(U, x, fmt)), L.View(_nhwc16(U, dy, fmt)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co,
                    precision=WPREC[fmt], scale=0.5)
    assert U.rel_err(dw.cpu(), 0.5 * wt.grad) < 2e-4
    assert U.rel_err(db.cpu(), 0.5 * b.grad) < 2e-4


@pytest.mark.parametrize("fmt", [7, 8])
def test_h16_wgrad_dense_block_batch_unshuffle_and_padded_channels(U, fmt):
    L = U.L
    n, h, w, F_ = 2, 24, 32, 64
    # the five convs of a dense block in ONE batched launch: conv k reads the channel prefix [0, kF), its dy is a slice of E
    D = _q(_rand((n, 5 * F_, h, w), 111), fmt)
    E = _q(_rand((n, 5 * F_, h, w), 112), fmt)
    Dd, Ed = _nhwc16(U, D, fmt), _nhwc16(U, E, fmt)
    probs, refs = [], []
    for k in range(1, 6):
        wt = torch.zeros(F_, k * F_, 3, 3, requires_grad=True); b = torch.zeros(F_, requires_grad=True)
        O.conv3x3(D[:, :k * F_], wt, b).backward(E[:, (5 - k) * F_:(6 - k) * F_])
        dw = torch.full((F_, k * F_, 3, 3), float("nan"), device="cuda"); db = torch.full((F_,), float("nan"), device="cuda")
        probs.append(dict(x=L.View(Dd, 0, k * F_), dy=L.View(Ed, (5 - k) * F_, F_), dw=dw, db=db, Cin=k * F_, Cout=F_, scale=1.0))
        refs.append((wt.grad, b.grad))
    L.conv3x3_wgrad_batched(probs, N=n, H=h, W=w, OH=h, OW=w, precision=WPREC[fmt])
    for p, (gw, gb) in zip(probs, refs):
        assert U.rel_err(p["dw"].cpu(), gw) < 2e-4 and U.rel_err(p["db"].cpu(), gb) < 2e-4
    # upsampling conv: dy is the gradient of the PixelShuffled output, read through SRK_IN_UNSHUFFLE
    x = _q(_rand((n, F_, h, w), 113), fmt)
    wt = torch.zeros(4 * F_, F_, 3, 3, requires_grad=True); b = torch.zeros(4 * F_, requires_grad=True)
    g = _q(_rand((n, F_, 2 * h, 2 * w), 114), fmt)
    O.pixel_shuffle(O.conv3x3(x, wt, b), 2).backward(g)
    dw = torch.full((4 * F_, F_, 3, 3), float("nan"), device="cuda"); db = torch.full((4 * F_,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(_nhwc16(U, x, fmt)), L.View(_nhwc16(U, g, fmt)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_,
                    dy_mode=L.IN_UNSHUFFLE, precision=WPREC[fmt])
    assert U.rel_err(dw.cpu(), wt.grad) < 2e-4 and U.rel_err(db.cpu(), b.grad) < 2e-4
    # image-side convs: 3 channels zero-padded to 32 on the x side (conv1) resp. the dy side (conv3.2)
    img = _q(_rand((n, 3, h, w), 115), fmt)
    w1 = torch.zeros(F_, 3, 3, 3, requires_grad=True); b1 = torch.zeros(F_, requires_grad=True)
    g1 = _q(_rand((n, F_, h, w), 116), fmt)
    O.conv3x3(img, w1, b1).backward(g1)
    dw = torch.full((F_, 3, 3, 3), float("nan"), device="cuda"); db = torch.full((F_,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(_nhwc16(U, img, fmt, ldc=32), 0, 32), L.View(_nhwc16(U, g1, fmt)), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=3, Cout=F_,
                    precision=WPREC[fmt])
    assert U.rel_err(dw.cpu(), w1.grad) < 2e-4 and U.rel_err(db.cpu(), b1.grad) < 2e-4
    w3 = torch.zeros(3, F_, 3, 3, requires_grad=True); b3 = torch.zeros(3, requires_grad=True)
    g3 = _q(_rand((n, 3, h, w), 117), fmt)
    O.conv3x3(x, w3, b3).backward(g3)
    dw = torch.full((3, F_, 3, 3), float("nan"), device="cuda"); db = torch.full((3,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(_nhwc16(U, x, fmt)), L.View(_nhwc16(U, g3, fmt, ldc=32), 0, 32), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=3,
                    precision=WPREC[fmt])
    assert U.rel_err(dw.cpu(), w3.grad) < 2e-4 and U.rel_err(db.cpu(), b3.grad) < 2e-4


# ------------------------------------------------------------------------------------------------------------ whole generator
@pytest.mark.parametrize("fmt", [7, 8])
@pytest.mark.parametrize("mt", [1, 2, 4])
def test_h16_sign_bits_equal_the_mask_tensor(U, fmt, mt):
    """srk_conv_args.signs: a forward conv writes the sign bits of what it stores (1 MB per dense-block conv instead of the 16.8 MB
    slice); a data-gradient conv that takes its LeakyReLU' mask from them gives the SAME bits as with the mask tensor -- on ragged
    multi-tile images, every tile form, values that round to zero included."""
    L = U.L
    L.lib().srk_debug_set_h16_mt(mt)
    n, h, w, F_ = 2, 37, 70, 64
    x = _rand((n, 2 * F_, h, w), 90)
    x[:, :, ::5, ::7] = 0                                     # exact zeros in the input: outputs that are exactly / nearly zero
    wt = _rand((F_, 2 * F_, 3, 3), 91, 0.03)
    b = _rand((F_,), 92, 0.05)
    xb = _nhwc16(U, x, fmt)
    y = torch.zeros(n, h, w, F_, device="cuda", dtype=DT[fmt])
    kw = dict(N=n, H=h, W=w, OH=h, OW=w, Cin=2 * F_, Cout=F_, slope=0.2)
    nb = L.conv_signs_bytes(L.View(xb), _pack(U, wt, fmt), b.cuda(), L.View(y), **kw)
    assert nb > 0 and nb % 16 == 0
    signs = torch.zeros(nb, dtype=torch.uint8, device="cuda")
    wp = _pack(U, wt, fmt)
    L.conv3x3(L.View(xb), wp, b.cuda(), L.View(y), signs_out=signs, **kw)
    y_plain = torch.zeros_like(y)
    L.conv3x3(L.View(xb), wp, b.cuda(), L.View(y_plain), **kw)
    assert torch.equal(y, y_plain)                            # writing the bits changes nothing else
    # total number of set bits = number of positive stored values (the layout is private; the count is not)
    bits = torch.from_numpy(__import__("numpy").unpackbits(signs.cpu().numpy())).sum().item()
    assert bits == int((y.float() > 0).sum().item())
    # the data gradient of a following conv, masked by y: from the tensor and from the bits
    g = _q(_rand((n, F_, h, w), 93), fmt)
    w2 = _rand((F_, F_, 3, 3), 94, 0.05)
    wpb = _pack(U, w2, fmt, transpose=True)
    gb = _nhwc16(U, g, fmt)
    kw2 = dict(N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=F_, mask_slope=0.2)
    d_ref = torch.zeros(n, h, w, F_, device="cuda", dtype=DT[fmt])
    L.conv3x3(L.View(gb), wpb, None, L.View(d_ref), mask=L.View(y), **kw2)
    d_bits = torch.zeros_like(d_ref)
    L.conv3x3(L.View(gb), wpb, None, L.View(d_bits), mask_signs=signs, **kw2)
    assert torch.equal(d_bits, d_ref)
    assert (d_ref.float().abs() > 0).float().mean().item() > 0.5


def test_h16_sign_bits_contract(U):
    L = U.L
    n, h, w, F_ = 1, 16, 32, 64
    xb = _nhwc16(U, _rand((n, F_, h, w), 95), 7)
    wp = _pack(U, _rand((F_, F_, 3, 3), 96, 0.05), 7)
    y = torch.zeros(n, h, w, F_, device="cuda", dtype=torch.float16)
    kw = dict(N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=F_)
    signs = torch.zeros(L.conv_signs_bytes(L.View(xb), wp, None, L.View(y), **kw), dtype=torch.uint8, device="cuda")
    with pytest.raises(RuntimeError):          # both a mask tensor and mask bits
        L.conv3x3(L.View(xb), wp, None, L.View(y), mask=L.View(y), mask_signs=signs, **kw)
    with pytest.raises(RuntimeError):          # the fp32-output form has no sign bits
        L.conv3x3(L.View(xb), wp, None, L.View(torch.zeros(n, h, w, F_, device="cuda")), signs_out=signs, flags=L.CONV_OUT_F32, **kw)
    # the fp32 kernels do not offer them
    x32 = U.nhwc(_rand((n, F_, h, w), 97)).cuda()
    w32 = U.pack_fwd(_rand((F_, F_, 3, 3), 98, 0.05), fmt=6)[0]

    class PW6:
        fmt = 6

        def data_ptr(self):
            return w32.data_ptr()
    y32 = torch.zeros(n, h, w, F_, device="cuda")
    assert L.conv_signs_bytes(L.View(x32), PW6(), None, L.View(y32), **kw) == 0
    with pytest.raises(RuntimeError):
        L.conv3x3(L.View(x32), PW6(), None, L.View(y32), signs_out=signs, **kw)


# ---------------------------------------------------------------------------------------------- the chain form (one launch per dense block)
def _dense_block_calls(U, fmt, n, h, w, backward, seed):
    """the engine's two sequences (engine.py _drb_forward / _drb_backward) on random data: conv k reads slices 0..k-1 of D and writes
    slice k (bias + LeakyReLU, or the LeakyReLU' mask), the fifth writes another buffer with alpha and two residuals"""
    L, F_ = U.L, 64
    D = torch.zeros(n, h, w, 5 * F_, device="cuda", dtype=DT[fmt])
    D[..., :F_] = U.nhwc(_rand((n, F_, h, w), seed)).to(DT[fmt])
    out = torch.zeros(n, h, w, 2 * F_, device="cuda", dtype=DT[fmt])
    M = U.nhwc(_rand((n, 5 * F_, h, w), seed + 1)).to(DT[fmt]).contiguous()
    R2 = U.nhwc(_rand((n, F_, h, w), seed + 2)).to(DT[fmt]).contiguous()
    keep, calls = [M, R2], []
    geo = dict(N=n, H=h, W=w, OH=h, OW=w, Cout=F_)
    for k in range(1, 6):
        wt = _rand((F_, k * F_, 3, 3), seed + 10 + k, 1.0 / (3.0 * (k * F_) ** 0.5))
        b = None if backward else _rand((F_,), seed + 20 + k, 0.1).cuda()
        wp = _pack(U, wt, fmt)
        keep += [wp, b]
        if k < 5 and backward:
            calls.append((L.View(D, 0, k * F_), wp, None, L.View(D, k * F_, F_), dict(Cin=k * F_, mask=L.View(M, k * F_, F_), mask_slope=0.2, **geo)))
        elif k < 5:
            calls.append((L.View(D, 0, k * F_), wp, b, L.View(D, k * F_, F_), dict(Cin=k * F_, slope=0.2, **geo)))
        else:
            calls.append((L.View(D, 0, 5 * F_), wp, b, L.View(out, F_, F_), dict(Cin=5 * F_, alpha=0.2, r1=L.View(D, 0, F_), beta1=1.0, r2=L.View(R2), beta2=0.5, **geo)))
    return D, out, calls, keep


def _seq_kernel(L, calls):
    import ctypes as C
    arr = (L.ConvArgs * len(calls))()
    for a, (x, wp, bias, y, kw) in zip(arr, calls):
        L._fill_conv_args(a, x, wp, bias, y, **kw)
    buf = C.create_string_buffer(96)
    L.check(L.lib().srk_conv3x3_seq_kernel_name(arr, len(calls), buf, 96), "srk_conv3x3_seq_kernel_name")
    return buf.value.decode()


@pytest.mark.parametrize("fmt", [7, 8])
@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("n,h,w", [(1, 16, 32), (1, 40, 70), (2, 48, 96), (3, 33, 31)])
def test_h16_chain_matches_separate_launches(U, fmt, backward, n, h, w):
    """One persistent launch per dense block (srk_conv3x3_seq, chain form; tiles exchange halos through flags) against the same five
    convolutions launched one by one (each checked against the fp32 reference above): equal up to the order of the fp32 sums,
    i.e. one unit in the last place of the 16-bit outputs per conv; and bit-identical from run to run (a race would not be)."""
    L = U.L
    D, out, calls, keep = _dense_block_calls(U, fmt, n, h, w, backward, 300 + n + h)
    L.lib().srk_debug_set_h16_chain(0)
    assert _seq_kernel(L, calls) == ""
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.float().clone(), out.float().clone()
    assert refD[..., 64:].abs().max().item() > 0.1 and refO[..., 64:].abs().max().item() > 0.1
    L.lib().srk_debug_set_h16_chain(2)
    assert _seq_kernel(L, calls).startswith("conv3x3_h16_chain_kernel<")
    first = None
    for rep in range(4):
        D[..., 64:] = 0
        out.zero_()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        scale = refD.abs().max().item()
        tol = 6 * TOL16[fmt] * scale          # five convs, each within an ulp of its (differently ordered) fp32 sum
        assert (D.float() - refD).abs().max().item() <= tol
        assert (out.float() - refO).abs().max().item() <= tol
        assert out[..., :64].abs().max().item() == 0
        if first is None:
            first = (D.clone(), out.clone())
        else:
            assert torch.equal(D, first[0]) and torch.equal(out, first[1])


@pytest.mark.parametrize("m16", [1, 0])
def test_h16_chain_writes_and_reads_the_same_sign_bits_as_separate_launches(U, m16):
    """Forward sequence with sign bits through the chain form and conv by conv, then the data-gradient sequence masked by them.
    m16 = 0 (the 32x32x16 form of the chain kernel): the SAME bit layout as the one-conv kernels -- equal buffers, and either path may read
    the other's bits.  m16 = 1 (the default 16x16x32 form): a layout of its own, told apart by srk_conv3x3_seq_signs_tag -- each path reads
    the bits IT wrote, and the two data-gradient results agree within the usual last place."""
    L = U.L
    L.lib().srk_debug_set_h16_chain_m16(m16)
    try:
        n, h, w = 8, 128, 128
        D, out, calls, keep = _dense_block_calls(U, 7, n, h, w, False, 600)
        nb = L.conv_signs_bytes(calls[0][0], calls[0][1], calls[0][2], calls[0][3], **calls[0][4])
        sg = [torch.zeros(4, nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
        tags = []
        for mode, s_ in ((0, sg[0]), (1, sg[1])):
            L.lib().srk_debug_set_h16_chain(mode)
            D[..., 64:] = 0
            cs = [(x, wp, b, y, dict(kw, signs_out=s_[k]) if k < 4 else kw) for k, (x, wp, b, y, kw) in enumerate(calls)]
            assert (_seq_kernel(L, cs) != "") == (mode == 1)
            nbs, tag = L.conv_seq_signs(cs)
            assert nbs == nb and tag != 0
            tags.append(tag)
            L.conv3x3_seq(cs)
            torch.cuda.synchronize()
        assert (tags[0] == tags[1]) == (m16 == 0)
        if m16 == 0:
            # the two forward paths differ in the last place of a few outputs (order of the fp32 sums): so may a few sign bits next to zero
            assert (sg[0] != sg[1]).float().mean().item() < 1e-3
        else:
            # same number of positive outputs (up to those few), different places
            np = __import__("numpy")
            pop = [int(np.unpackbits(t.cpu().numpy()).sum()) for t in sg]
            assert abs(pop[0] - pop[1]) < 1e-3 * pop[0] and pop[1] == int((D[..., 64:].float() > 0).sum().item())
        Db, outb, callsb, keepb = _dense_block_calls(U, 7, n, h, w, True, 601)
        res = []
        for mode in (0, 1):
            L.lib().srk_debug_set_h16_chain(mode)
            Db[..., 64:] = 0
            outb.zero_()
            cs = []
            for k, (x, wp, b, y, kw) in enumerate(callsb):
                kw = dict(kw)
                if k < 4:
                    kw.pop("mask")
                    kw["mask_signs"] = sg[1 if m16 == 0 else mode][k]       # (m16: every path reads the bits its own form wrote)
                cs.append((x, wp, b, y, kw))
            L.conv3x3_seq(cs)
            torch.cuda.synchronize()
            res.append((Db.float().clone(), outb.float().clone()))
        # (m16: the two mask sets come from two forward runs that differ in the last place of a few outputs: a few masks next to zero differ)
        tol = 6 * TOL16[7] * res[0][0].abs().max().item()
        if m16 == 0:
            assert (res[0][0] - res[1][0]).abs().max().item() <= tol and (res[0][1] - res[1][1]).abs().max().item() <= tol
        else:
            bad = ((res[0][0] - res[1][0]).abs() > tol).float().mean().item()
            assert bad < 1e-3, bad
    finally:
        L.lib().srk_debug_set_h16_chain_m16(1)


def test_h16_chain_eligibility(U):
    """what srk_conv3x3_seq sends out as one kernel: only sequences in which a conv takes at most its LAST 64 input channels from its
    predecessor's output, one geometry, <= 64 outputs, whole 64-channel slices -- everything else goes conv by conv"""
    L = U.L
    L.lib().srk_debug_set_h16_chain(2)
    D, out, calls, keep = _dense_block_calls(U, 7, 1, 16, 32, False, 400)
    assert _seq_kernel(L, calls).startswith("conv3x3_h16_chain_kernel<_Float16, ")
    assert _seq_kernel(L, calls[:1]) == ""                                     # a single conv
    assert _seq_kernel(L, calls[1:3]).startswith("conv3x3_h16_chain_kernel")   # any sub-sequence of the pattern
    # conv 3 reading, as an OLD slice, what conv 2 has just written: not the pattern (swap the order of two convs)
    assert _seq_kernel(L, [calls[0], calls[2], calls[1]]) == ""
    # a second conv with 64 inputs would need its predecessor's output as its FIRST stage
    x, wp, b, y, kw = calls[0]
    assert _seq_kernel(L, [calls[0], (L.View(D, 64, 64), wp, b, L.View(D, 128, 64), kw)]) == ""
    # in place
    assert _seq_kernel(L, [calls[0], (calls[1][0], calls[1][1], calls[1][2], L.View(D, 64, 64), calls[1][4])]) == ""
    # mixed geometry
    kw2 = dict(calls[1][4]); kw2.update(H=8, OH=8)
    assert _seq_kernel(L, [calls[0], (calls[1][0], calls[1][1], calls[1][2], calls[1][3], kw2)]) == ""
    L.lib().srk_debug_set_h16_chain(1)      # default: only where the 16-row form would run (>= 200 tiles)
    assert _seq_kernel(L, calls) == ""
    L.lib().srk_debug_set_h16_chain(0)
    assert _seq_kernel(L, calls) == ""


@pytest.mark.parametrize("backward", [False, True])
def test_h16_chain_full_size_repeatable(U, backward):
    """BASELINE configs[4]'s trunk geometry (8 x 128 x 128: 256 tiles, one per CU, on all eight XCDs), default dispatch: the chain result
    agrees with the separate launches and is bit-identical over 12 runs"""
    L = U.L
    D, out, calls, keep = _dense_block_calls(U, 7, 8, 128, 128, backward, 500)
    L.lib().srk_debug_set_h16_chain(0)
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.float().clone(), out.float().clone()
    L.lib().srk_debug_set_h16_chain(1)
    assert _seq_kernel(L, calls).startswith("conv3x3_h16_chain_kernel<")
    first = None
    for rep in range(12):
        D[..., 64:] = 0
        out.zero_()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        tol = 6 * TOL16[7] * refD.abs().max().item()
        assert (D.float() - refD).abs().max().item() <= tol and (out.float() - refO).abs().max().item() <= tol
        if first is None:
            first = (D.clone(), out.clone())
        else:
            assert torch.equal(D, first[0]) and torch.equal(out, first[1])


def _mrel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


# (output, gradient) bounds of the two storage types on the small generator / on configs[4]'s full architecture: max-abs error
# relative to the tensor's max-abs against the CPU fp32 oracle.  A 16-bit rounding per stored activation (fp16: 2^-11, bf16: 2^-8
# relative) accumulates along 30 / 351 convolutions; measured values are in DESIGN.md section 4d.
SMALL_TOL = {"fp16": (2e-3, 1e-2), "bf16s": (1.5e-2, 8e-2)}
FULL_TOL = {"fp16": (4e-3, 5e-3), "bf16s": (3e-2, 3e-2)}       # measured: fp16 1.8e-3 / 1.6e-3, bf16 1.6e-2 / 1.0e-2


@pytest.mark.parametrize("mode", ["fp16", "bf16s"])
def test_h16_generator_small_all_gradients_vs_oracle(srk, mode):
    """configs[4] in small (C = 3, F = 64, 4x, 2 RRDBs, ragged 24 x 20 input, batch 2) with 16-bit activation storage: forward and
    EVERY weight / bias gradient vs the fp32 oracle; jets (C = 1) and a final-layer RRDB as well."""
    for ch, nfin in ((3, 0), (1, 1)):
        gen = srk.GeneratorRRDB(ch, filters=64, num_res_blocks=2, num_upsample=2, num_final_layer_res=nfin).cuda()
        sd = O.default_init_generator(11, channels=ch, filters=64, num_res_blocks=2, num_upsample=2, num_final_layer_res=nfin)
        gen.load_state_dict(sd)
        g = torch.Generator().manual_seed(5)
        x = torch.rand(2, ch, 24, 20, generator=g)
        sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
        yo, _ = O.generator_forward(sdo, x, 2, 2, 0.2, training=True, num_final_layer_res=nfin)
        tgt = torch.rand(yo.shape, generator=g)
        (yo - tgt).abs().mean().backward()
        gen._engine.precision = mode
        y = gen(x.cuda())
        assert y.dtype == torch.float32
        otol, gtol = SMALL_TOL[mode]
        assert _mrel(y.detach().cpu(), yo.detach()) < otol, (mode, ch)
        # (loss scale 1024: what train.Stepper's GradScaler does for fp16; exact power of two, divided out below)
        ((y - tgt.cuda()).abs().mean() * 1024.0).backward()
        worst = max((_mrel(p.grad.cpu() / 1024.0, sdo[k].grad), k) for k, p in gen.named_parameters() if p.grad is not None)
        assert worst[0] < gtol, (mode, ch, worst)


@pytest.mark.parametrize("mode", ["fp16", "bf16s"])
def test_h16_configs4_full_architecture_forward_and_gradients_vs_oracle(srk, mode):
    """BASELINE configs[4]'s architecture and image size: GeneratorRRDB(3, 64, 23, num_upsample=2) on 1 x 3 x 128 x 128 -> 512 x 512,
    16-bit activation storage: forward and weight gradients at the head, in the first, a middle and the last RRDB and in the tail
    vs the CPU fp32 oracle (models.py:9-135 + autograd; ~15 s of CPU)."""
    gen = srk.GeneratorRRDB(3, filters=64, num_res_blocks=23, num_upsample=2).cuda()
    sd = O.default_init_generator(0, channels=3, filters=64, num_res_blocks=23, num_upsample=2)
    gen.load_state_dict(sd)
    g = torch.Generator().manual_seed(4321)
    hr = torch.rand(1, 3, 512, 512, generator=g)
    lr = torch.nn.functional.avg_pool2d(hr, 4)
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, lr, 23, 2, 0.2, training=True)
    O.warmup_loss(yo, hr).backward()
    gen._engine.precision = mode
    y = gen(lr.cuda())
    otol, gtol = FULL_TOL[mode]
    err = _mrel(y.detach().cpu(), yo.detach())
    assert err < otol, (mode, err)
    scale = 65536.0 if mode == "fp16" else 1.0
    ((y - hr.cuda()).abs().mean() * scale).backward()
    named = dict(gen.named_parameters())
    keys = ["conv1.weight", "conv1.bias", "res_blocks.0.dense_blocks.0.b1.0.weight", "res_blocks.11.dense_blocks.1.b5.0.weight",
            "res_blocks.11.dense_blocks.1.b3.0.bias", "res_blocks.22.dense_blocks.2.b5.0.weight", "res_blocks.22.dense_blocks.0.b2.0.weight",
            "conv2.weight", "upsampling.0.weight", "upsampling.3.weight", "conv3.0.weight", "conv3.2.weight", "conv3.2.bias"]
    errs = {k: _mrel(named[k].grad.cpu() / scale, sdo[k].grad) for k in keys}
    print(mode, "forward", err, "gradients", errs)
    assert all(torch.isfinite(named[k].grad).all() for k in keys)
    worst = max(errs.items(), key=lambda kv: kv[1])
    assert worst[1] < gtol, (mode, worst)


@pytest.mark.parametrize("mode", ["fp16", "bf16s"])
def test_h16_configs4_batch8_launch_set_vs_oracle(srk, mode):
    """BASELINE configs[4]'s ACTUAL launch set: batch 8 of 3 x 128 x 128 (256 sixteen-row tiles = one per CU) on GeneratorRRDB(3, 64, 2,
    num_upsample=2), default dispatch.  At this size every dense block's forward and data-gradient sequence goes out as ONE
    conv3x3_h16_chain_kernel launch and its five weight gradients as the LOADER form of wgrad_h16_kernel (both only selected at >= 200
    tiles / >= 8 tiles per workgroup, which the batch-1 test above never reaches): asserted from the names the C side reports for the
    launches of this very forward / backward (srk_conv3x3_seq_kernel_name, srk_conv3x3_wgrad_kernel_name).  Forward and EVERY weight / bias
    gradient vs the CPU fp32 oracle (models.py:34-41,58,63,99, esrgan.py:416-427; ~15 s of CPU) at the full-architecture bounds."""
    L = srk._lib
    R = 2
    gen = srk.GeneratorRRDB(3, filters=64, num_res_blocks=R, num_upsample=2).cuda()
    sd = O.default_init_generator(3, channels=3, filters=64, num_res_blocks=R, num_upsample=2)
    gen.load_state_dict(sd)
    g = torch.Generator().manual_seed(99)
    hr = torch.rand(8, 3, 512, 512, generator=g)
    lr = torch.nn.functional.avg_pool2d(hr, 4)
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, lr, R, 2, 0.2, training=True)
    O.warmup_loss(yo, hr).backward()
    gen._engine.precision = mode
    tname = "_Float16" if mode == "fp16" else "__bf16"
    L.KernelTimer.start()
    try:
        y = gen(lr.cuda())
        otol, gtol = FULL_TOL[mode]
        err = _mrel(y.detach().cpu(), yo.detach())
        scale = 65536.0 if mode == "fp16" else 1.0
        ((y - hr.cuda()).abs().mean() * scale).backward()
    finally:
        ran = L.KernelTimer.stop()
    # ---- what ran: 3 R forward + 3 R data-gradient chain launches, 3 R loader-form batched weight gradients
    chain = f"conv3x3_h16_chain_kernel<{tname}, true>"
    loader = f"wgrad_h16_kernel<{tname}, 0, true, 8>+reduce"
    assert chain in ran and ran[chain]["n"] == 6 * R, sorted(ran)
    assert loader in ran and ran[loader]["n"] >= 3 * R, sorted(ran)
    # (nothing of the trunk fell back to one-conv launches: what is left outside the chains are conv1, conv2, two upsampling convs and
    # conv3.0 / conv3.2, forward and data gradient)
    single = sum(v["n"] for k, v in ran.items() if k.startswith("conv3x3_h16") and not k.startswith("conv3x3_h16_chain"))
    assert single <= 12, sorted(ran)
    assert err < otol, (mode, err)
    named = dict(gen.named_parameters())
    errs = {k: _mrel(p.grad.cpu() / scale, sdo[k].grad) for k, p in named.items() if p.grad is not None}
    assert len(errs) == sum(1 for k in sd if k not in ("power", "multiplier"))          # every weight and bias of the model
    assert all(torch.isfinite(p.grad).all() for p in named.values() if p.grad is not None)
    worst = max(errs.items(), key=lambda kv: kv[1])
    print(mode, "batch 8 forward", err, "worst gradient", worst)
    assert worst[1] < gtol, (mode, worst)


def test_h16_warmup_step_with_loss_scaling_matches_oracle_update(srk):
    """train.Stepper in the fp16 mode: dynamic loss scaling (GradScaler) around the generator's backward; two warm-up iterations
    (esrgan.py:416-427) move the weights like the oracle's Adam does, and the scale stays at its initial 2^16 (no overflow)."""
    import importlib
    train = importlib.import_module("super-resolution_amd.train")
    st = train.Stepper(workload="g_only", res_blocks=1, filters=64, device=torch.device("cuda"), hr=64, factor=2, res_scale=0.2, channels=3)
    st.generator._engine.precision = "fp16"
    gsd = {k: v.detach().cpu().clone() for k, v in st.generator.state_dict().items()}
    g = torch.Generator().manual_seed(3)
    hr = torch.rand(2, 3, 64, 64, generator=g); lr = torch.nn.functional.avg_pool2d(hr, 2)
    params = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in gsd.items()}
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=2e-4, betas=(0.9, 0.999))
    for _ in range(2):
        opt.zero_grad()
        yo, _ = O.generator_forward(params, lr, 1, 1, 0.2, training=True)
        lo = O.warmup_loss(yo, hr)
        lo.backward()
        opt.step()
        out = st.step(lr.cuda(), hr.cuda())
        assert abs(out["g_loss"].item() - lo.item()) < 2e-3 * max(1.0, abs(lo.item()))
    assert st._grad_scaler is not None and st._grad_scaler.get_scale() == 65536.0
    new = st.generator.state_dict()
    for k in ("conv1.weight", "res_blocks.0.dense_blocks.1.b3.0.weight", "conv3.2.bias", "upsampling.0.weight"):
        upd_ref = params[k].detach() - gsd[k]
        upd = new[k].cpu() - gsd[k]
        # Adam's first steps are ~lr * sign(g): compare where the oracle's gradient is not tiny
        assert ((upd - upd_ref).abs().mean() / upd_ref.abs().mean()).item() < 0.1, k


def test_h16_gan_iteration_with_fp16_generator(srk):
    """The full G+D iteration (esrgan.py:457-626) with the GENERATOR in fp16 activation storage (the discriminators stay fp32): the
    G-phase loss and generator gradients against the exact-fp32 path on the same weights and batch, then whole iterations with the
    loss scaler in place (finite losses, weights move, scale stays at 2^16)."""
    import importlib
    train = importlib.import_module("super-resolution_amd.train")
    lr, hr = O.jet_images(4, 1, 64, 64, 31, 2)
    res = {}
    for mode in ("f32", "fp16"):
        torch.manual_seed(0)
        st = train.Stepper(workload="gan", res_blocks=1, filters=64, device=torch.device("cuda"), hr=64, factor=2, res_scale=0.2)
        st.generator._engine.precision = mode
        loss_G, generated, gt, parts = st.g_phase_loss(lr.cuda(), hr.cuda())
        (loss_G * (1024.0 if mode == "fp16" else 1.0)).backward()
        named = dict(st.generator.named_parameters())
        res[mode] = (loss_G.item(), {k: named[k].grad.detach().cpu() / (1024.0 if mode == "fp16" else 1.0)
                                     for k in ("conv1.weight", "res_blocks.0.dense_blocks.1.b3.0.weight", "upsampling.0.weight", "conv3.2.weight")})
    assert abs(res["fp16"][0] - res["f32"][0]) < 2e-3 * abs(res["f32"][0])
    for k, g in res["f32"][1].items():
        assert _mrel(res["fp16"][1][k], g) < 2e-2, k
    st.generator.zero_grad(set_to_none=True)
    w0 = st.generator.conv2.weight.detach().clone()
    for _ in range(3):
        out = st.step(lr.cuda(), hr.cuda())
        assert torch.isfinite(out["g_loss"]).all() and all(torch.isfinite(v).all() for v in out["d_loss"].values())
    assert st._grad_scaler is not None and st._grad_scaler.get_scale() == 65536.0
    assert not torch.equal(st.generator.conv2.weight.detach(), w0)
