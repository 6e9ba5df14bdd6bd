"""The chain form of the fp32 F(2x4,3x3) conv kernel (srk_conv3x3_seq: a dense block's five convolutions as ONE persistent launch,
csrc/srk_chain.h) against the same convolutions launched one by one -- which the conv tests check against the oracle.  Same
arithmetic in the same order: the results must agree BIT FOR BIT, run after run (tiles hand each other their halos through flags:
an ordering error would show as a difference).  Reference: models.py:34-41 (DenseResidualBlock.forward) and its autograd."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
F_ = 64


@pytest.fixture(scope="module")
def U():
    import srk_testutil
    return srk_testutil


@pytest.fixture(autouse=True)
def _reset(U):
    yield
    U.L.lib().srk_debug_set_w42_chain(1)
    U.L.lib().srk_debug_set_wino42_nmt(0)


class PW:
    def __init__(self, t, fmt):
        self.t, self.fmt = t, fmt

    def data_ptr(self):
        return self.t.data_ptr()


def _rand(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


def _block(U, n, h, w, backward, seed):
    """the engine's two sequences (engine.py _drb_forward / _drb_backward) on random data"""
    L = U.L
    D = torch.zeros(n, h, w, 5 * F_, device="cuda")
    D[..., :F_] = U.nhwc(_rand((n, F_, h, w), seed)).cuda()
    out = torch.zeros(n, h, w, 2 * F_, device="cuda")
    M = U.nhwc(_rand((n, 5 * F_, h, w), seed + 1)).cuda().contiguous()
    R2 = U.nhwc(_rand((n, F_, h, w), seed + 2)).cuda().contiguous()
    keep, calls = [M, R2], []
    geo = dict(N=n, H=h, W=w, OH=h, OW=w, Cout=F_)
    for k in range(1, 6):
        wt = _rand((F_, k * F_, 3, 3), seed + 10 + k, 1.0 / (3.0 * (k * F_) ** 0.5))
        b = None if backward else _rand((F_,), seed + 20 + k, 0.1).cuda()
        wp = PW(U.pack_fwd(wt, fmt=6)[0], 6)
        keep += [wp, b]
        if k < 5 and backward:
            calls.append((L.View(D, 0, k * F_), wp, None, L.View(D, k * F_, F_), dict(Cin=k * F_, mask=L.View(M, k * F_, F_), mask_slope=0.2, **geo)))
        elif k < 5:
            calls.append((L.View(D, 0, k * F_), wp, b, L.View(D, k * F_, F_), dict(Cin=k * F_, slope=0.2, **geo)))
        else:
            calls.append((L.View(D, 0, 5 * F_), wp, b, L.View(out, F_, F_), dict(Cin=5 * F_, alpha=0.2, r1=L.View(D, 0, F_), beta1=1.0, r2=L.View(R2), beta2=0.5, **geo)))
    return D, out, calls, keep


def _seq_kernel(L, calls):
    arr = (L.ConvArgs * len(calls))()
    for a, (x, wp, bias, y, kw) in zip(arr, calls):
        L._fill_conv_args(a, x, wp, bias, y, **kw)
    buf = C.create_string_buffer(96)
    L.check(L.lib().srk_conv3x3_seq_kernel_name(arr, len(calls), buf, 96), "srk_conv3x3_seq_kernel_name")
    return buf.value.decode()


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("nmt", [1, 2])
@pytest.mark.parametrize("n,h,w", [(1, 32, 16), (1, 40, 70), (2, 64, 48), (3, 33, 31)])
def test_wino42_chain_bit_identical_to_separate_launches(U, backward, nmt, n, h, w):
    L = U.L
    L.lib().srk_debug_set_wino42_nmt(nmt)
    D, out, calls, keep = _block(U, n, h, w, backward, 700 + n + h)
    L.lib().srk_debug_set_w42_chain(0)
    assert _seq_kernel(L, calls) == ""
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.clone(), out.clone()
    assert refD[..., F_:].abs().max().item() > 0.1 and refO[..., F_:].abs().max().item() > 0.1
    L.lib().srk_debug_set_w42_chain(1)
    assert _seq_kernel(L, calls) == f"conv3x3_f32_wino42_chain_kernel<{nmt}>"
    for rep in range(4):
        D[..., F_:] = 0
        out.zero_()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        assert torch.equal(D, refD) and torch.equal(out, refO)


@pytest.mark.parametrize("backward", [False, True])
@pytest.mark.parametrize("n,hw", [(32, 64), (16, 64)])
def test_wino42_chain_full_size_repeatable(U, backward, n, hw):
    """the headline's trunk geometry (32 x 64 x 64: 256 32-row tiles, one per CU, all eight XCDs) and configs[1]'s (batch 16: 256 16-row
    tiles), default dispatch, 12 runs each"""
    L = U.L
    D, out, calls, keep = _block(U, n, hw, hw, backward, 800 + n)
    L.lib().srk_debug_set_w42_chain(0)
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.clone(), out.clone()
    L.lib().srk_debug_set_w42_chain(1)
    assert _seq_kernel(L, calls).startswith("conv3x3_f32_wino42_chain_kernel<")
    for rep in range(12):
        D[..., F_:] = 0
        out.zero_()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        assert torch.equal(D, refD) and torch.equal(out, refO)


def test_wino42_chain_eligibility(U):
    L = U.L
    D, out, calls, keep = _block(U, 1, 32, 16, False, 900)
    assert _seq_kernel(L, calls).startswith("conv3x3_f32_wino42_chain_kernel<")
    assert _seq_kernel(L, calls[:1]) == ""
    assert _seq_kernel(L, calls[1:3]).startswith("conv3x3_f32_wino42_chain_kernel<")
    assert _seq_kernel(L, [calls[0], calls[2], calls[1]]) == ""            # conv 3 would read conv 2's slice as an OLD one
    x, wp, b, y, kw = calls[1]
    assert _seq_kernel(L, [calls[0], (x, wp, b, L.View(D, 64, 64), kw)]) == ""          # in place
    kw2 = dict(kw); kw2.update(H=16, OH=16)
    assert _seq_kernel(L, [calls[0], (x, wp, b, y, kw2)]) == ""                       # two geometries
    # more tiles than CUs: 1024 16 x 16 tiles
    D2, out2, calls2, keep2 = _block(U, 4, 256, 256, False, 901)
    assert _seq_kernel(L, calls2) == ""
    L.lib().srk_debug_set_w42_chain(0)
    assert _seq_kernel(L, calls) == ""


def test_chain_launches_from_two_streams_are_ordered_by_the_library(U):
    """At most one chain kernel may be in flight per device (two of them could each hold part of the CUs and wait for tiles that cannot
    become resident).  Sequences issued alternately on two streams -- the library orders them with an event -- give the single-stream
    results, at the full trunk geometry where a launch needs every CU."""
    L = U.L
    blocks = [_block(U, 32, 64, 64, bw, 950 + i) for i, bw in enumerate((False, True))]
    refs = []
    for D, out, calls, keep in blocks:
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        refs.append((D.clone(), out.clone()))
    s = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(6):
        for (D, out, calls, keep) in blocks:
            D[..., F_:] = 0
            out.zero_()
        torch.cuda.synchronize()
        for i, (D, out, calls, keep) in enumerate(blocks):
            with torch.cuda.stream(s[(i + rep) & 1]):
                L.conv3x3_seq(calls)
                L.conv3x3_seq(calls)          # (idempotent: same inputs, same outputs)
        torch.cuda.synchronize()
        for (D, out, calls, keep), (rD, rO) in zip(blocks, refs):
            assert torch.equal(D, rD) and torch.equal(out, rO)


def test_chain_form_is_skipped_under_stream_capture(U):
    """the launch epoch is a kernel argument: a captured graph must hold the conv-by-conv launches (which replay correctly)"""
    L = U.L
    D, out, calls, keep = _block(U, 2, 64, 48, False, 960)
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.clone(), out.clone()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        D[..., F_:] = 0
        out.zero_()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            L.conv3x3_seq(calls)
    for rep in range(3):
        D[..., F_:] = 0
        out.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(D, refD) and torch.equal(out, refO)



def test_wino42_offers_no_sign_bits(U):
    """srk_conv_args.signs is a feature of the 16-bit kernels (tests/test_h16_gpu.py); the fp32 F(2x4,3x3) kernels have no register to
    spare for it (built into their chain form it was correct and 8 % slower): the probe says 0 and the engine keeps the mask tensors"""
    L = U.L
    D, out, calls, keep = _block(U, 2, 64, 48, False, 970)
    assert _seq_kernel(L, calls).startswith("conv3x3_f32_wino42_chain_kernel<")
    assert L.conv_seq_signs_bytes(calls) == 0
    signs = torch.zeros(4, 4096, dtype=torch.uint8, device="cuda")
    cs = [(x, wp, b, y, dict(kw, signs_out=signs[k]) if k < 4 else kw) for k, (x, wp, b, y, kw) in enumerate(calls)]
    with pytest.raises(RuntimeError):
        L.conv3x3_seq(cs)
