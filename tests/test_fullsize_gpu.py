"""GPU: size-independent properties at BASELINE.json's FULL sizes (F=64, 64x64 LR, batch 16, Cin up to 320, R=23), where
the CPU oracle is too slow to be the checker for every case:
  * adjointness  <conv(x), y> == <x, conv^T(y)>  and  <dW, V> == <dy, conv_V(x)>  (data-/weight-gradient kernels vs the
    forward kernel, no oracle needed),
  * linearity of the un-activated convolution,
  * batch independence of the generator (image i does not depend on the other images or its slot; bit-exact at equal
    batch size, 1e-5 across batch sizes where another kernel variant may be selected),
  * run-to-run determinism of forward + backward (all 702 gradient tensors bit-identical),
  * plus one direct oracle comparison of the full-depth generator on 2 images (a few seconds of CPU).
Both the exact-fp32 kernels and the opt-in split-bf16 kernels are exercised.
"""
import os
import sys

import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
N, H, W, F = 16, 64, 64, 64


@pytest.fixture(scope="module")
def U():
    import srk_testutil
    return srk_testutil


def _dot(a, b):
    return (a.double() * b.double()).sum().item()


def _pack(L, w, fmt, transpose=False):
    co, ci = w.shape[:2]
    K, M = (co, ci) if transpose else (ci, co)
    dst = torch.empty(L.packed_floats(K, M, fmt), dtype=torch.float32, device="cuda")
    t = L.PackTable(w.device, fmt=fmt)
    t.add(w, dst, M=M, k_off=0, k_len=K, K_total=K, transpose=transpose)
    t.run()
    return dst


@pytest.fixture(autouse=True)
def _reset_w42_form(U):
    yield
    U.L.lib().srk_debug_set_wino42_nmt(0)


# direct fp32, split-bf16, Winograd F(2,3), Winograd F(4,3), 2-D Winograd F(2x4,3x3) in its 16-row form (61: what configs[1]'s
# launch set selects at N = 16, 64 x 64) and in its 32-row form (62: the batch-32 launch set of configs[2])
@pytest.mark.parametrize("fmt", [0, 1, 3, 5, 61, 62])
@pytest.mark.parametrize("ci", [64, 320])
def test_adjoint_and_linearity_at_full_dense_block_size(U, fmt, ci):
    L = U.L
    if fmt in (61, 62):
        L.lib().srk_debug_set_wino42_nmt(fmt - 60)
        fmt = 6
    g = torch.Generator(device="cuda").manual_seed(ci + fmt)
    x = torch.randn(N, H, W, 5 * F, device="cuda", generator=g)
    x2 = torch.randn(N, H, W, 5 * F, device="cuda", generator=g)
    y = torch.randn(N, H, W, F, device="cuda", generator=g)
    w = torch.randn(F, ci, 3, 3, device="cuda", generator=g) * 0.02
    wp, wpt = _pack(L, w, fmt), _pack(L, w, fmt, transpose=True)
    geo = dict(N=N, H=H, W=W, OH=H, OW=W)

    def conv(inp):
        out = torch.empty(N, H, W, F, device="cuda")
        L.conv3x3(L.View(inp, 0, ci), wp, None, L.View(out), Cin=ci, Cout=F, wp_format=fmt, **geo)
        return out
    cx = conv(x)
    # <conv(x), y> == <x, dgrad(y)>
    dx = torch.empty(N, H, W, ci, device="cuda")
    L.conv3x3(L.View(y), wpt, None, L.View(dx), Cin=F, Cout=ci, wp_format=fmt, **geo)
    lhs, rhs = _dot(cx, y), _dot(x[..., :ci], dx)
    tol = 2e-4 if fmt == 1 else 2e-5
    assert abs(lhs - rhs) <= tol * max(abs(lhs), abs(rhs), (cx.double().norm() * y.double().norm()).item() * 1e-2), (lhs, rhs)
    # linearity: conv(2x - 3x2) == 2conv(x) - 3conv(x2)
    lin = conv(2 * x - 3 * x2)
    ref = 2 * cx - 3 * conv(x2)
    assert ((lin - ref).abs().max() / ref.abs().max()).item() < (1e-4 if fmt == 1 else 2e-5)
    # <dW, V> == <y, conv_V(x)>  (weight gradient vs forward with weights V)
    dw = torch.empty(F, ci, 3, 3, device="cuda"); db = torch.empty(F, device="cuda")
    L.conv3x3_wgrad(L.View(x, 0, ci), L.View(y), dw, db, Cin=ci, Cout=F, precision=(fmt if fmt == 1 else 0), **geo)
    V = torch.randn(F, ci, 3, 3, device="cuda", generator=g) * 0.02
    cv = torch.empty(N, H, W, F, device="cuda")
    L.conv3x3(L.View(x, 0, ci), _pack(L, V, fmt), None, L.View(cv), Cin=ci, Cout=F, wp_format=fmt, **geo)
    lhs, rhs = _dot(dw, V), _dot(y, cv)
    assert abs(lhs - rhs) <= (3e-4 if fmt == 1 else 5e-5) * max(abs(lhs), abs(rhs), (dw.double().norm() * V.double().norm()).item() * 1e-2), (lhs, rhs)
    assert abs(_dot(db, torch.ones_like(db)) - y.double().sum().item()) <= 1e-4 * y.double().abs().sum().item()


@pytest.fixture(scope="module")
def full_generator(srk):
    torch.manual_seed(0)
    gen = srk.GeneratorRRDB(1, filters=64, num_res_blocks=23, num_upsample=2).cuda()
    sd = O.default_init_generator(0, channels=1, filters=64, num_res_blocks=23, num_upsample=2)
    gen.load_state_dict(sd)
    return gen, sd


@pytest.mark.parametrize("mode", ["f32", "bf16x3"])
def test_full_generator_properties(full_generator, mode):
    """Full depth (351 convs): batch independence and run-to-run determinism, both bit-exact.  The kernels use no
    atomics and fixed reduction orders, so any difference would be a race in the LDS pipelines / loader waves."""
    gen, sd = full_generator
    gen._engine.precision = mode
    lr, hr = O.jet_images(4, 1, 256, 256, 77, 4)
    lr, hr = lr.cuda(), hr.cuda()
    with torch.no_grad():
        y = gen(lr)
        assert torch.isfinite(y).all()
        # image 1 in another slot, among other neighbours (same batch size: the kernel selection -- F(4,3) / F(2,3) /
        # direct -- depends on how many tiles the batch offers, and the variants differ at the 1e-6 level)
        y2 = gen(lr[[1, 0, 3, 3]])
        assert torch.equal(y2[0], y[1]) and torch.equal(y2[1], y[0]) and torch.equal(y2[2], y[3])
        y1 = gen(lr[1:2])                       # alone: same image, possibly another kernel variant
        assert ((y1 - y[1:2]).abs().max() / y.abs().max()).item() < 1e-5
    grads = []
    for _ in range(2):
        gen.zero_grad(set_to_none=True)
        out = gen(lr)
        (out - hr).abs().mean().backward()
        grads.append([p.grad.clone() for p in gen.parameters() if p.grad is not None])
    assert len(grads[0]) == 702
    for a, b in zip(*grads):
        assert torch.equal(a, b)
    gen._engine.precision = "f32"


def test_full_generator_forward_vs_oracle(full_generator):
    """F=64, R=23, 4x: HIP forward vs the CPU oracle on identical weights/inputs (BASELINE's 1e-3 bar) in both modes."""
    gen, sd = full_generator
    lr, _ = O.jet_images(2, 1, 256, 256, 4321, 4)
    with torch.no_grad():
        ref, _ = O.generator_forward(sd, lr, 23, 2, 0.2, training=True)
        for mode, tol in (("f32", 2e-5), ("bf16x3", 1e-4)):
            gen._engine.precision = mode
            y = gen(lr.cuda()).cpu()
            err = ((y - ref).abs().max() / ref.abs().max()).item()
            assert err < tol, (mode, err)
    gen._engine.precision = "f32"


def test_generator_with_f43_kernels_on_ragged_width_vs_oracle(srk):
    """A launch big enough for the engine to pick the Winograd F(4,3) kernels (>= 200 workgroups of 32 x 16) with a width that
    is not a multiple of the tile (72 = 4.5 tiles): forward and all weight gradients of a one-RRDB generator vs the oracle."""
    gen = srk.GeneratorRRDB(1, filters=64, num_res_blocks=1, num_upsample=1).cuda()
    sd = O.default_init_generator(3, channels=1, filters=64, num_res_blocks=1, num_upsample=1)
    gen.load_state_dict(sd)
    N, H, W = 20, 64, 72
    assert gen._engine._wino4_levels((N, H, W)) == (True, True)
    g = torch.Generator().manual_seed(9)
    x = torch.rand(N, 1, H, W, generator=g) * (torch.rand(N, 1, H, W, generator=g) < 0.2) * 5
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, x, 1, 1, 0.2, training=True)
    tgt = torch.rand(yo.shape, generator=g)
    (yo - tgt).abs().mean().backward()
    y = gen(x.cuda())
    assert ((y.detach().cpu() - yo.detach()).abs().max() / yo.detach().abs().max()).item() < 2e-5
    (y - tgt.cuda()).abs().mean().backward()
    worst = max(((p.grad.cpu() - sdo[k].grad).abs().max() / sdo[k].grad.abs().max().clamp_min(1e-4)).item()
                for k, p in gen.named_parameters() if p.grad is not None)
    assert worst < 2e-3, worst


def test_full_depth_generator_gradients_vs_oracle(full_generator):
    """F=64, R=23, 4x (BASELINE configs[1]/[2] architecture), one 64x64 jet image: forward + L1 backward through all 351
    convolutions; weight gradients at the head, in the first, a middle and the last RRDB, and in the tail vs the CPU oracle
    (models.py:9-135 + autograd).  ~10 s of CPU."""
    gen, sd = full_generator
    gen._engine.precision = "f32"
    lr, hr = O.jet_images(1, 1, 256, 256, 77, 4)
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, lr, 23, 2, 0.2, training=True)
    O.warmup_loss(yo, hr).backward()
    gen.zero_grad()
    y = gen(lr.cuda())
    assert ((y.detach().cpu() - yo.detach()).abs().max() / yo.detach().abs().max()).item() < 2e-5
    (y - hr.cuda()).abs().mean().backward()
    named = dict(gen.named_parameters())
    keys = ["conv1.weight", "conv1.bias", "res_blocks.0.dense_blocks.0.b1.0.weight", "res_blocks.11.dense_blocks.1.b5.0.weight",
            "res_blocks.11.dense_blocks.1.b3.0.bias", "res_blocks.22.dense_blocks.2.b5.0.weight", "res_blocks.22.dense_blocks.0.b2.0.weight",
            "conv2.weight", "upsampling.0.weight", "upsampling.3.weight", "conv3.0.weight", "conv3.2.weight", "conv3.2.bias"]
    for k in keys:
        ref = sdo[k].grad
        err = ((named[k].grad.cpu() - ref).abs().max() / ref.abs().max().clamp_min(1e-12)).item()
        assert err < 2e-3, (k, err)


def test_batch32_shaped_launch_set_gradients_vs_oracle(srk):
    """N = 26 >= 25 images of 64x64 (the batch-32-shaped launch set: the engine picks the Winograd F(4,3) conv kernels at the
    LR level, the batched Winograd weight-gradient kernel runs 13 pixel splits) on a 2-RRDB generator: forward and EVERY weight
    and bias gradient through engine.backward vs the CPU oracle."""
    gen = srk.GeneratorRRDB(1, filters=64, num_res_blocks=2, num_upsample=1, res_scale=0.1).cuda()
    sd = O.default_init_generator(5, channels=1, filters=64, num_res_blocks=2, num_upsample=1)
    gen.load_state_dict(sd)
    Nb, Hb, Wb = 26, 64, 64
    assert gen._engine._wino4_levels((Nb, Hb, Wb)) == (True, True)
    lr, hr = O.jet_images(Nb, 1, 2 * Hb, 2 * Wb, 78, 2)
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, lr, 2, 1, 0.1, training=True)
    O.warmup_loss(yo, hr).backward()
    y = gen(lr.cuda())
    assert ((y.detach().cpu() - yo.detach()).abs().max() / yo.detach().abs().max()).item() < 2e-5
    (y - hr.cuda()).abs().mean().backward()
    worst = max((((p.grad.cpu() - sdo[k].grad).abs().max() / sdo[k].grad.abs().max().clamp_min(1e-12)).item(), k)
                for k, p in gen.named_parameters() if p.grad is not None)
    assert worst[0] < 2e-3, worst
