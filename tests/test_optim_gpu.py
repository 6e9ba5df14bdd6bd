"""super-resolution_amd.optim.Adam (ONE launch per step, csrc/srk_optim.hip) against torch.optim.Adam, the optimizer the reference steps
(esrgan.py:299,305,427,487,623): same updates over several steps, weight decay, the GradScaler contract (unscale inside the step, skip on
non-finite gradients without counting the step), state_dict interchange."""
import importlib

import pytest
import torch

pytestmark = pytest.mark.gpu
srk = importlib.import_module("super-resolution_amd")
optim = importlib.import_module("super-resolution_amd.optim")

SHAPES = [(64, 64, 3, 3), (64,), (3, 64, 3, 3), (5,), (64, 320, 3, 3), (1,), (4097,), (8192,), (123, 7)]


def _params(seed, device):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.rand(s, generator=g) - 0.5).to(device)) for s in SHAPES]


def _grads(step, device, scale=1.0):
    g = torch.Generator().manual_seed(1000 + step)
    return [((torch.rand(s, generator=g) - 0.5) * scale).to(device) for s in SHAPES]


@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_adam_matches_torch_adam_over_steps(wd):
    ours = _params(1, "cuda")
    ref = _params(1, "cpu")
    o = optim.Adam(ours, lr=2e-3, betas=(0.9, 0.999), weight_decay=wd)
    rp = [torch.nn.Parameter(p.detach().double()) for p in ref]
    r = torch.optim.Adam(rp, lr=2e-3, betas=(0.9, 0.999), weight_decay=wd)
    for step in range(6):
        gs = _grads(step, "cpu")
        for p, q, g in zip(ours, rp, gs):
            p.grad = g.cuda()
            q.grad = g.double()
        o.step()
        r.step()
    for p, q in zip(ours, rp):
        assert (p.detach().cpu().double() - q.detach()).abs().max().item() < 2e-6        # fp32 vs float64 arithmetic over 6 steps of 2e-3
    assert float(o.state[ours[0]]["step"]) == 6.0


def test_adam_is_bit_compatible_with_atens_fused_adam():
    a, b = _params(2, "cuda"), _params(2, "cuda")
    o = optim.Adam(a, lr=1e-3, betas=(0.9, 0.999))
    t = torch.optim.Adam(b, lr=1e-3, betas=(0.9, 0.999), fused=True)
    for step in range(4):
        for p, q, g in zip(a, b, _grads(step, "cuda")):
            p.grad, q.grad = g.clone(), g.clone()
        o.step()
        t.step()
    worst = max((p - q).abs().max().item() for p, q in zip(a, b))
    assert worst < 2e-7          # same formula in fp32; a last-place difference per step at most


def test_adam_grad_scaler_contract_and_skipped_step():
    a, b = _params(3, "cuda"), _params(3, "cuda")
    o = optim.Adam(a, lr=1e-3)
    t = torch.optim.Adam(b, lr=1e-3, fused=True)
    sa, sb = torch.amp.GradScaler("cuda", init_scale=1024.0), torch.amp.GradScaler("cuda", init_scale=1024.0)
    sa.scale(torch.zeros((), device="cuda")); sb.scale(torch.zeros((), device="cuda"))       # (the scalers create their scale tensors lazily)
    for step in range(5):
        gs = _grads(step, "cuda", scale=1024.0 if step <= 2 else 512.0)          # (as if scaled by the scaler's current scale)
        if step == 2:
            gs[4][0, 0, 0, 0] = float("inf")                       # this step must be skipped by both, and the scale halved
        for p, q, g in zip(a, b, gs):
            p.grad, q.grad = g.clone(), g.clone()
        before = [p.detach().clone() for p in a]
        sa.step(o); sa.update()
        sb.step(t); sb.update()
        if step == 2:
            assert all(torch.equal(p, q) for p, q in zip(a, before))
    assert sa.get_scale() == sb.get_scale() == 512.0
    assert float(o.state[a[0]]["step"]) == 4.0 and float(t.state[b[0]]["step"]) == 4.0
    assert max((p - q).abs().max().item() for p, q in zip(a, b)) < 2e-7


def _through_a_file(sd):
    """as a checkpoint would carry it (load_state_dict alone keeps referring to the other optimizer's tensors)"""
    import io
    f = io.BytesIO()
    torch.save(sd, f)
    f.seek(0)
    return torch.load(f, map_location="cpu")


def test_adam_state_dict_interchange_with_torch_adam():
    a, b = _params(4, "cuda"), _params(4, "cuda")
    o = optim.Adam(a, lr=1e-3)
    for step in range(2):
        for p, g in zip(a, _grads(step, "cuda")):
            p.grad = g
        o.step()
    t = torch.optim.Adam(b, lr=1e-3, fused=True)
    with torch.no_grad():
        for p, q in zip(a, b):
            q.copy_(p)
    t.load_state_dict(_through_a_file(o.state_dict()))
    o2 = optim.Adam(_params(4, "cuda"), lr=1e-3)
    with torch.no_grad():
        for p, q in zip(a, o2.param_groups[0]["params"]):
            q.copy_(p)
    o2.load_state_dict(_through_a_file(t.state_dict()))
    c = o2.param_groups[0]["params"]
    for step in range(2, 4):
        for p, q, r, g in zip(a, b, c, _grads(step, "cuda")):
            p.grad, q.grad, r.grad = g.clone(), g.clone(), g.clone()
        o.step(); t.step(); o2.step()
    assert max((p - q).abs().max().item() for p, q in zip(a, b)) < 2e-7
    assert all(torch.equal(p, r) for p, r in zip(a, c))


def test_adam_refuses_cpu_parameters():
    p = [torch.nn.Parameter(torch.zeros(4))]
    o = optim.Adam(p, lr=1e-3)
    p[0].grad = torch.ones(4)
    with pytest.raises(RuntimeError):
        o.step()
