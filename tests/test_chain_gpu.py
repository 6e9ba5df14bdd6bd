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


# ----------------------------------------------------------------------------------------------- protocol (round 4): bounded waits, wrap, recovery
@pytest.fixture
def _protocol_reset(U):
    """every protocol test leaves the device as it found it: no fault, no back-off, default wait bound, epoch 0"""
    yield
    L = U.L
    torch.cuda.synchronize()
    L.lib().srk_debug_chain_inject_fault(0)
    L.chain_recover()
    L.lib().srk_debug_chain_set(0, 0)
    L.lib().srk_chain_set_wait_us(0)


def test_chain_epoch_wrap_reset_is_invisible(U, _protocol_reset):
    """flags hold `epoch + k + 1` in 32 bits and are compared as differences; before the epoch passes 2^30 the library zeroes the flags
    on the launching stream (srk_chain_epoch_plan).  Start 7 convs short of the wrap and chain 12 blocks across it: bit-identical
    results, exactly one reset."""
    L = U.L
    D, out, calls, keep = _block(U, 2, 64, 48, False, 1234)
    L.lib().srk_debug_set_w42_chain(0)
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.clone(), out.clone()
    L.lib().srk_debug_set_w42_chain(1)
    L.lib().srk_debug_chain_set((1 << 30) - 7, 0)
    r0 = L.chain_stats()["resets"]
    for rep in range(12):
        D[..., F_:] = 0
        out.zero_()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        assert torch.equal(D, refD) and torch.equal(out, refO), rep
    assert L.chain_stats()["resets"] == r0 + 1


@pytest.mark.parametrize("kind", ["w42", "h16"])
def test_chain_gives_up_when_neighbours_are_late_then_recovers(U, kind, _protocol_reset):
    """A tile whose neighbour does not publish within the bound must not sit out a long wait: it raises the fault word, poisons the device
    word and drains; every other tile sees the poison and drains too, and the launch ends at once.  Made deterministic with the start
    skew (tile phases begin 200 us apart: what a foreign kernel on some CUs does to the tiles that wait for a CU) and a wait bound of
    20 us.  Then: the next sequence call is refused with ChainTimeout, optimizer steps are refused, chain_recover reports the fault, the
    sequence runs conv by conv while the forms rest and gives the reference result, and the forms come back."""
    import time
    L = U.L
    if kind == "w42":
        D, out, calls, keep = _block(U, 32, 64, 64, False, 4321)
        setter = L.lib().srk_debug_set_w42_chain
    else:
        import test_h16_gpu as H
        D, out, calls, keep = H._dense_block_calls(U, 7, 8, 128, 128, False, 77)
        setter = L.lib().srk_debug_set_h16_chain
    ki = 1 if kind == "w42" else 0
    setter(0)
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.clone(), out.clone()
    setter(1)
    assert "chain_kernel" in _seq_kernel(L, calls)
    try:
        # late neighbours WITHIN the bound are no fault: phases 50 us apart, bound 50 ms -> the reference result
        L.check(L.lib().srk_debug_chain_skew(ki, 50000, 8), "srk_debug_chain_skew")
        D[..., F_:] = 0
        out.zero_()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        if kind == "w42":
            assert torch.equal(D, refD) and torch.equal(out, refO)
        # beyond it: phases 200 us apart, bound 20 us
        L.lib().srk_chain_set_wait_us(20)
        L.check(L.lib().srk_debug_chain_skew(ki, 200000, 8), "srk_debug_chain_skew")
        t0 = time.perf_counter()
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        assert time.perf_counter() - t0 < 2.0
    finally:
        L.lib().srk_debug_chain_skew(ki, 0, 1)
        L.lib().srk_chain_set_wait_us(0)
    with pytest.raises(L.ChainTimeout):
        L.conv3x3_seq(calls)
    # an optimizer step while the fault is pending: refused on the host ...
    step = torch.zeros((), device="cuda"); skip = torch.zeros((), device="cuda")
    with pytest.raises(L.ChainTimeout):
        L.adam_count_step(step, None, skip)
    assert L.chain_recover() == 1
    st = L.chain_stats()
    assert st["strikes"] >= 1 and st["off_calls"] > 0
    assert _seq_kernel(L, calls) == ""        # resting: conv by conv
    D[..., F_:] = 0
    out.zero_()
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    if kind == "w42":
        assert torch.equal(D, refD) and torch.equal(out, refO)
    else:
        assert torch.equal(D.float(), refD.float()) and torch.equal(out.float(), refO.float())
    # ... and the forms come back once the rest is over
    L.lib().srk_debug_chain_set(0, 0)
    assert "chain_kernel" in _seq_kernel(L, calls)
    D[..., F_:] = 0
    out.zero_()
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    if kind == "w42":
        assert torch.equal(D, refD) and torch.equal(out, refO)


def test_chain_launch_beside_a_kernel_that_holds_cus_is_late_not_wrong(U, _protocol_reset):
    """Partial residency is not a fault as long as every tile's neighbours turn up within the bound: a full-chip chain launch behind a
    kernel that holds 36 CUs for 5 ms (bound: the default 50 ms) starts on the CUs that are free, its tiles next to the missing ones
    wait for them, and the result is the reference -- 5 ms late.  (What the generator's backward does all the time: a chain launch beside
    the previous block's weight gradient.)"""
    L = U.L
    D, out, calls, keep = _block(U, 32, 64, 64, False, 777)
    L.lib().srk_debug_set_w42_chain(0)
    L.conv3x3_seq(calls)
    torch.cuda.synchronize()
    refD, refO = D.clone(), out.clone()
    L.lib().srk_debug_set_w42_chain(1)
    side = torch.cuda.Stream()
    for rep in range(3):
        D[..., F_:] = 0
        out.zero_()
        torch.cuda.synchronize()
        L.check(L.lib().srk_debug_hold_cus(36, 5000, side.cuda_stream), "srk_debug_hold_cus")
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        assert torch.equal(D, refD) and torch.equal(out, refO)
    assert L.chain_stats()["strikes"] == 0


def test_adam_skips_itself_on_the_device_while_a_chain_fault_is_pending(srk, U, _protocol_reset):
    """The host may be iterations ahead of the GPU when a chain launch gives up: the optimizer steps it has ALREADY queued must not touch
    the weights.  The fault is raised here from the device side of the stream (in stream order behind a 20 ms kernel), after the host has
    queued the step: parameters, moments and step counter stay as they were; after recovery the same step goes through."""
    L = U.L
    optim = __import__("importlib").import_module("super-resolution_amd.optim")
    # (make sure the fault word exists: one chain launch)
    D, out, calls, keep = _block(U, 1, 32, 16, False, 5)
    L.conv3x3_seq(calls)
    ps = [torch.nn.Parameter(torch.randn(n, device="cuda")) for n in (5000, 300, 70000)]
    opt = optim.Adam(ps, lr=1e-2)
    for p in ps:
        p.grad = torch.randn_like(p)
    opt.step()                                  # a normal first step (allocates the state)
    torch.cuda.synchronize()
    before = [p.detach().clone() for p in ps]
    m_before = [opt.state[p]["exp_avg"].clone() for p in ps]
    s = torch.cuda.current_stream()
    L.check(L.lib().srk_debug_hold_cus(4, 20000, s.cuda_stream), "hold")
    L.check(L.lib().srk_debug_chain_inject_fault_async(s.cuda_stream), "inject")
    opt.step()                                  # host-side checks pass (the word is still clear); on the device the fault precedes it
    torch.cuda.synchronize()
    assert all(torch.equal(a, p.detach()) for a, p in zip(before, ps))
    assert all(torch.equal(a, opt.state[p]["exp_avg"]) for a, p in zip(m_before, ps))
    assert float(opt.param_groups[0]["_srk_step"]) == 1.0
    with pytest.raises(L.ChainTimeout):
        opt.step()
    assert L.chain_recover() != 0
    opt.step()
    torch.cuda.synchronize()
    assert not any(torch.equal(a, p.detach()) for a, p in zip(before, ps))
    assert float(opt.param_groups[0]["_srk_step"]) == 2.0


def test_stepper_repeats_an_iteration_whose_chain_launch_gave_up(srk, U, _protocol_reset):
    """train.Stepper.step: a ChainTimeout in the middle of an iteration (here: injected behind the forward, in front of the first weight
    gradient) -> recover, run the iteration again conv by conv.  fp32 chain and conv-by-conv results are bit-identical, so the weights
    after the repaired iteration must EQUAL those of an undisturbed Stepper bit for bit (esrgan.py:416-427)."""
    import importlib
    from oracle import esrgan_oracle as O
    train = importlib.import_module("super-resolution_amd.train")
    L = U.L
    lr, hr = O.jet_images(16, 1, 256, 256, 3, 4)

    def mk():
        torch.manual_seed(0)
        return train.Stepper(workload="g_only", res_blocks=1, filters=64, device=torch.device("cuda"), hr=256, factor=4, res_scale=0.2)
    ref = mk()
    for _ in range(2):
        out_ref = ref.step(lr.cuda(), hr.cuda())
    torch.cuda.synchronize()
    st = mk()
    st.step(lr.cuda(), hr.cuda())
    assert L.chain_stats()["launches"] > 0, "this geometry was meant to run the chain form"
    orig = L.conv3x3_wgrad_batched
    fired = []

    def faulty(*a, **kw):
        if not fired:
            fired.append(1)
            L.lib().srk_debug_chain_inject_fault(1)
        return orig(*a, **kw)
    L.conv3x3_wgrad_batched = faulty
    try:
        out = st.step(lr.cuda(), hr.cuda())
    finally:
        L.conv3x3_wgrad_batched = orig
    torch.cuda.synchronize()
    assert fired and getattr(st, "chain_recoveries", 0) == 1
    assert torch.equal(out["g_loss"], out_ref["g_loss"])
    for (k, a), (_, b) in zip(st.generator.state_dict().items(), ref.generator.state_dict().items()):
        assert torch.equal(a, b), k


def test_two_full_size_chain_launch_streams_from_two_host_threads(U, _protocol_reset):
    """At most ONE chain kernel may be in flight per device (two full-chip persistent kernels would each hold part of the CUs and fail
    their census).  Two host threads, each with its own stream and its own dense block at the full trunk geometry (32 x 64 x 64 = 256
    tiles), launch 10 chains each at the same time: the library orders the launches by events; every result is the reference, and no
    launch gives up."""
    import threading
    L = U.L
    blocks = [_block(U, 32, 64, 64, bool(i), 2000 + i) for i in range(2)]
    refs = []
    L.lib().srk_debug_set_w42_chain(0)
    for D, out, calls, keep in blocks:
        L.conv3x3_seq(calls)
        torch.cuda.synchronize()
        refs.append((D.clone(), out.clone()))
    L.lib().srk_debug_set_w42_chain(1)
    n0 = L.chain_stats()["launches"]
    errs = []

    def worker(i):
        try:
            D, out, calls, keep = blocks[i]
            s = torch.cuda.Stream()
            with torch.cuda.stream(s):
                for rep in range(10):
                    D[..., F_:] = 0
                    out.zero_()
                    L.conv3x3_seq(calls)
                    s.synchronize()
                    if not (torch.equal(D, refs[i][0]) and torch.equal(out, refs[i][1])):
                        errs.append((i, rep, "mismatch"))
        except Exception as e:      # noqa: BLE001
            errs.append((i, repr(e)))
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    assert not errs, errs
    st = L.chain_stats()
    assert st["launches"] == n0 + 20 and st["strikes"] == 0
