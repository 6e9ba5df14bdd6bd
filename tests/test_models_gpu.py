"""GPU parity of the drop-in modules against the reference-generated golden fixtures and the CPU oracle.
Tolerance: BASELINE.json's 1e-3 relative (fp32) on outputs; gradients 2e-3 of the tensor's max-abs."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import esrgan_oracle as O  # noqa: E402  (checker only)

pytestmark = pytest.mark.gpu
OUT_TOL = 1e-3
GRAD_TOL = 2e-3


def rel(a, b):
    # references that are exactly 0 (the final D bias: the relativistic loss is invariant to it) get an absolute floor
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-4)).item()


def _load_closed_form(mod, gain=1.0):
    sd = O.closed_form_fill({k: v.cpu() for k, v in mod.state_dict().items()}, gain=gain)
    mod.load_state_dict(sd)
    return sd


@pytest.mark.parametrize("name", ["G4_gen_f16_r1_u2", "G6_gen_resscale01", "G4b_gen_c3_f16_r1_u1", "G5_config0"])
def test_generator_golden(srk, golden_dir, name):
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    c, f, r, u = [int(v) for v in d["cfg"]]
    gen = srk.GeneratorRRDB(c, filters=f, num_res_blocks=r, num_upsample=u, res_scale=float(d["res_scale"])).cuda()
    _load_closed_form(gen, float(d["gain"]))
    assert int(d["n_keys"]) == len(gen.state_dict())
    x = torch.from_numpy(d["x"]).cuda()
    gen.train()
    y = gen(x)
    assert rel(y.cpu(), torch.from_numpy(d["y_train"])) < OUT_TOL
    assert rel(gen.srs.cpu(), torch.from_numpy(d["srs"])) < OUT_TOL
    loss = (y - torch.from_numpy(d["target"]).cuda()).abs().mean()
    assert abs(loss.item() - float(d["loss"])) < 1e-4 * max(1.0, float(d["loss"]))
    loss.backward()
    checked = 0
    for k, p in gen.named_parameters():
        key = "grad." + k
        if key in d.files:
            ref = torch.from_numpy(d[key])
            assert rel(p.grad.cpu(), ref) < GRAD_TOL, k
            checked += 1
    if name != "G5_config0":
        assert checked == len(list(gen.parameters())) - 2
    gen.eval()
    with torch.no_grad():
        ye = gen(x)
    ye_ref = torch.from_numpy(d["y_eval"])
    assert (ye.cpu() - ye_ref).abs().max().item() < OUT_TOL * max(ye_ref.abs().max().item(), torch.from_numpy(d["y_train"]).abs().max().item())
    assert torch.equal(ye, torch.relu(ye))


def test_generator_vs_oracle_ragged_and_final_blocks(srk):
    """80x80-style ragged tiles (not multiples of the 8x16 tile), batch 3, final-layer RRDBs, input gradient."""
    torch.manual_seed(0)
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=2, num_upsample=1, res_scale=0.1, num_final_layer_res=1).cuda()
    sd = _load_closed_form(gen)
    x = (torch.rand(3, 1, 20, 13) * 2).requires_grad_(True)
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, x, 2, 1, 0.1, training=True, num_final_layer_res=1)
    tgt = torch.rand_like(yo)
    (yo - tgt).abs().mean().backward()
    xg = x.detach().cuda().requires_grad_(True)
    y = gen(xg)
    assert rel(y.detach().cpu(), yo.detach()) < OUT_TOL
    (y - tgt.cuda()).abs().mean().backward()
    assert rel(xg.grad.cpu(), x.grad) < GRAD_TOL
    for k, p in gen.named_parameters():
        if p.grad is not None:
            assert rel(p.grad.cpu(), sdo[k].grad) < GRAD_TOL, k


def test_generator_three_channel_photographic_config(srk):
    """BASELINE.json configs[4] in small: C=3, F=64, 4x (128->512 style), forward + all gradients vs the oracle in exact fp32,
    and the reduced-precision MFMA modes (bf16x3 / bf16: this build's counterpart of the config's fp16 path) within their
    stated tolerances.  C=3 exercises the non-vector staging of conv1 and the Cout=3 tail conv."""
    gen = srk.GeneratorRRDB(3, filters=64, num_res_blocks=2, num_upsample=2).cuda()
    sd = O.default_init_generator(11, channels=3, filters=64, num_res_blocks=2, num_upsample=2)
    gen.load_state_dict(sd)
    g = torch.Generator().manual_seed(5)
    x = torch.rand(2, 3, 24, 20, generator=g)
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    yo, _ = O.generator_forward(sdo, x, 2, 2, 0.2, training=True)
    tgt = torch.rand(yo.shape, generator=g)
    (yo - tgt).abs().mean().backward()
    assert yo.shape == (2, 3, 96, 80)
    for mode, otol, gtol in (("f32", OUT_TOL, GRAD_TOL), ("bf16x3", 1e-4, 1e-3), ("bf16", 3e-2, 2e-1)):
        gen._engine.precision = mode
        gen.zero_grad(set_to_none=True)
        y = gen(x.cuda())
        assert rel(y.detach().cpu(), yo.detach()) < otol, mode
        (y - tgt.cuda()).abs().mean().backward()
        worst = max(rel(p.grad.cpu(), sdo[k].grad) for k, p in gen.named_parameters() if p.grad is not None)
        assert worst < gtol, (mode, worst)
    gen._engine.precision = "f32"


def test_generator_with_dropout_branch(srk):
    """drop_rate > 0 (models.py:29-31,38-39): Dropout2d on each dense block's last conv, run module-wise on the HIP convolutions.
    Eval mode equals the no-dropout generator; in train mode the channel masks drawn by torch are captured with forward hooks
    and replayed in the oracle (forward and all weight gradients)."""
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=1, num_upsample=1, drop_rate=0.3).cuda()
    sd = _load_closed_form(gen)
    assert list(gen.state_dict().keys()) == list(srk.GeneratorRRDB(1, filters=16, num_res_blocks=1, num_upsample=1).state_dict().keys())
    x = torch.rand(3, 1, 12, 10) * 2
    gen.eval()
    with torch.no_grad():
        ye = gen(x.cuda()).cpu()
    yo_eval, _ = O.generator_forward(sd, x, 1, 1, 0.2, training=False)
    assert rel(ye, yo_eval) < OUT_TOL
    gen.train()
    masks, hooks = {}, []
    for name, m in gen.named_modules():
        if isinstance(m, torch.nn.Dropout2d):
            def hook(mod, inp, out, name=name):
                i = inp[0].detach()
                ratio = torch.where(i != 0, out.detach() / torch.where(i != 0, i, torch.ones_like(i)), torch.zeros_like(i))
                masks[name.rsplit(".", 1)[0]] = ratio.amax(dim=(2, 3), keepdim=True).cpu()     # [N, F, 1, 1]: 0 or 1/(1-p)
            hooks.append(m.register_forward_hook(hook))
    y = gen(x.cuda())
    tgt = torch.rand(y.shape, generator=torch.Generator().manual_seed(1))
    (y - tgt.cuda()).abs().mean().backward()
    for h in hooks:
        h.remove()
    assert len(masks) == 3
    for m in masks.values():
        assert all(v == 0.0 or abs(v - 1 / 0.7) < 1e-5 for v in torch.unique(m).tolist())
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    O.DROP_MASKS = masks
    try:
        yo, _ = O.generator_forward(sdo, x, 1, 1, 0.2, training=True)
    finally:
        O.DROP_MASKS = None
    assert rel(y.detach().cpu(), yo.detach()) < OUT_TOL
    (yo - tgt).abs().mean().backward()
    for k, p in gen.named_parameters():
        if p.grad is not None:
            assert rel(p.grad.cpu(), sdo[k].grad) < GRAD_TOL, k


def test_generator_power_multiplier(srk):
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=1, num_upsample=1, power=0.5, multiplier=2.0).cuda()
    sd = _load_closed_form(gen)
    sd["power"] = torch.tensor([0.5]); sd["multiplier"] = torch.tensor([2.0])
    gen.load_state_dict(sd)
    x = torch.rand(2, 1, 8, 8) + 0.1
    yo, so = O.generator_forward(sd, x, 1, 1, 0.2, training=True)
    y = gen(x.cuda())
    assert rel(y.cpu(), yo) < OUT_TOL and rel(gen.srs.cpu(), so) < OUT_TOL


def test_drb_golden_generic_path(srk, golden_dir):
    """DenseResidualBlock used stand-alone (generic Conv3x3 path) against G3 incl. all gradients."""
    d = np.load(os.path.join(golden_dir, "G3_drb16.npz"))
    drb = srk.DenseResidualBlock(16).cuda()
    _load_closed_form(drb)
    x = torch.from_numpy(d["x"]).cuda().requires_grad_(True)
    y = drb(x)
    assert rel(y.detach().cpu(), torch.from_numpy(d["y"])) < OUT_TOL
    y.backward(torch.from_numpy(d["gout"]).cuda())
    assert rel(x.grad.cpu(), torch.from_numpy(d["dx"])) < GRAD_TOL
    for k, p in drb.named_parameters():
        assert rel(p.grad.cpu(), torch.from_numpy(d["grad." + k])) < GRAD_TOL, k


def test_discriminator_golden_with_gradient_penalty(srk, golden_dir):
    """Markovian_Discriminator forward, relativistic D loss and the gradient penalty's double backward
    (esrgan.py:569-616) against G7."""
    d = np.load(os.path.join(golden_dir, "G7_discriminator.npz"))
    D = srk.Markovian_Discriminator((1, 32, 32), [16, 32, 32, 64]).cuda()
    _load_closed_form(D, gain=2.0)
    assert tuple(D.output_shape) == (1, 2, 2)
    gt, gen, eps = [torch.from_numpy(d[k]).cuda() for k in ("gt", "gen", "eps")]
    pred_real = D(gt, None)
    pred_fake = D(gen, None)
    assert rel(pred_real.detach().cpu(), torch.from_numpy(d["pred_real"])) < OUT_TOL
    assert rel(pred_fake.detach().cpu(), torch.from_numpy(d["pred_fake"])) < OUT_TOL
    crit = torch.nn.BCEWithLogitsLoss()
    valid = torch.ones(3, *D.output_shape, device="cuda"); fake = torch.zeros_like(valid)
    loss_D = (crit(1e-7 + pred_real - pred_fake.mean(0, keepdim=True), valid) +
              crit(1e-7 + pred_fake - pred_real.mean(0, keepdim=True), fake)) / 2
    interp = (eps * gt + (1 - eps) * gen)
    interp.requires_grad = True
    pi = D(interp, None)
    grads = torch.autograd.grad(outputs=pi, inputs=interp, grad_outputs=valid, create_graph=True, retain_graph=True, only_inputs=True)[0]
    assert rel(grads.detach().cpu(), torch.from_numpy(d["input_grad_gp"])) < GRAD_TOL
    gp = ((grads.view(3, -1).norm(2, dim=1) - 1) ** 2).mean() * 0.01 / 2
    tot = loss_D + gp
    assert abs(gp.item() - float(d["gp"])) < 1e-3 * float(d["gp"])
    assert abs(tot.item() - float(d["loss"])) < 1e-4
    tot.backward()
    for k, p in D.named_parameters():
        assert rel(p.grad.cpu(), torch.from_numpy(d["grad." + k])) < GRAD_TOL, k


def test_conditional_discriminator_golden(srk, golden_dir):
    """Conditional_Discriminator (models.py:189-223) forward on (HR, LR condition), relativistic D loss + gradient penalty
    double backward with the LR ground truth as condition (esrgan.py:569-606), against the reference-generated G13; and the
    state_dict key contract."""
    d = np.load(os.path.join(golden_dir, "G13_conditional_discriminator.npz"))
    cfg = [int(v) for v in d["cfg"]]
    ch, nu, hr = cfg[:4], cfg[4], cfg[5]
    D = srk.Conditional_Discriminator((1, hr, hr), ch, num_upsample=nu).cuda()
    assert list(D.state_dict().keys()) == open(os.path.join(golden_dir, "G13_state_keys.txt")).read().split()
    _load_closed_form(D, gain=2.0)
    assert tuple(D.output_shape) == tuple(int(v) for v in d["out_shape"])
    lr, gt, gen, eps = [torch.from_numpy(d[k]).cuda() for k in ("lr", "gt", "gen", "eps")]
    pred_real, pred_fake = D(gt, lr), D(gen, lr)
    assert rel(pred_real.detach().cpu(), torch.from_numpy(d["pred_real"])) < OUT_TOL
    assert rel(pred_fake.detach().cpu(), torch.from_numpy(d["pred_fake"])) < OUT_TOL
    crit = torch.nn.BCEWithLogitsLoss()
    valid = torch.ones(3, *D.output_shape, device="cuda"); fake = torch.zeros_like(valid)
    loss_D = (crit(1e-7 + pred_real - pred_fake.mean(0, keepdim=True), valid) + crit(1e-7 + pred_fake - pred_real.mean(0, keepdim=True), fake)) / 2
    interp = (eps * gt + (1 - eps) * gen)
    interp.requires_grad = True
    pi = D(interp, lr)
    grads = torch.autograd.grad(outputs=pi, inputs=interp, grad_outputs=valid, create_graph=True, retain_graph=True, only_inputs=True)[0]
    assert rel(grads.detach().cpu(), torch.from_numpy(d["input_grad_gp"])) < GRAD_TOL
    gp = ((grads.view(3, -1).norm(2, dim=1) - 1) ** 2).mean() * 0.01 / 2
    tot = loss_D + gp
    assert abs(gp.item() - float(d["gp"])) < 1e-3 * float(d["gp"]) and abs(tot.item() - float(d["loss"])) < 1e-4
    tot.backward()
    for k, p in D.named_parameters():
        assert rel(p.grad.cpu(), torch.from_numpy(d["grad." + k])) < GRAD_TOL, k
    # the condition branch is differentiable too (the G phase feeds pool(G(lr)) as condition, esrgan.py:494)
    c = lr.clone().requires_grad_(True)
    D(gen, c).sum().backward()
    assert c.grad is not None and torch.isfinite(c.grad).all() and c.grad.abs().max() > 0


def test_three_channel_discriminator_gradient_penalty(srk):
    """C = 3 (real NCHW <-> NHWC copies at the boundary): the input gradient is contiguous like nn.Conv2d's, so the reference's
    ``gradients.view(batch_size, -1)`` (esrgan.py:604) works, and the penalty's double backward matches the oracle."""
    ch = [8, 16]
    D = srk.Markovian_Discriminator((3, 24, 20), ch).cuda()
    sd = O.closed_form_fill({k: v.cpu() for k, v in D.state_dict().items()}, gain=2.0)
    D.load_state_dict(sd)
    g = torch.Generator().manual_seed(3)
    gt, gen = torch.rand(2, 3, 24, 20, generator=g), torch.rand(2, 3, 24, 20, generator=g)
    eps = torch.rand(2, 1, 1, 1, generator=g)
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo, gpo = O.d_phase_loss(sdo, gt, gen, eps, 0.01, d_channels=ch)
    lo.backward()
    interp = (eps.cuda() * gt.cuda() + (1 - eps.cuda()) * gen.cuda()).requires_grad_(True)
    pi = D(interp, None)
    grads = torch.autograd.grad(outputs=pi, inputs=interp, grad_outputs=torch.ones_like(pi), create_graph=True, retain_graph=True, only_inputs=True)[0]
    assert grads.is_contiguous()
    gp = ((grads.view(2, -1).norm(2, dim=1) - 1) ** 2).mean() * 0.01 / 2
    assert abs(gp.item() - gpo.item()) < 2e-3 * abs(gpo.item())
    crit = torch.nn.BCEWithLogitsLoss()
    pr, pf = D(gt.cuda(), None), D(gen.cuda(), None)
    loss = (crit(1e-7 + pr - pf.mean(0, keepdim=True), torch.ones_like(pr)) + crit(1e-7 + pf - pr.mean(0, keepdim=True), torch.zeros_like(pr))) / 2 + gp
    loss.backward()
    for k, p in D.named_parameters():
        assert rel(p.grad.cpu(), sdo[k].grad) < GRAD_TOL, k


def test_standard_discriminator_golden(srk, golden_dir):
    """Standard_Discriminator (models.py:177-186: patch trunk without its last conv + Linear-ReLU-Linear head) forward,
    relativistic loss and gradients against the reference-generated G15."""
    d = np.load(os.path.join(golden_dir, "G15_standard_discriminator.npz"))
    S = srk.Standard_Discriminator((1, 32, 32), [8, 16]).cuda()
    assert list(S.state_dict().keys()) == open(os.path.join(golden_dir, "G15_state_keys.txt")).read().split()
    _load_closed_form(S, gain=2.0)
    assert tuple(S.output_shape) == tuple(int(v) for v in d["out_shape"])
    gt, gen = torch.from_numpy(d["gt"]).cuda(), torch.from_numpy(d["gen"]).cuda()
    pr, pf = S(gt, None), S(gen, None)
    assert rel(pr.detach().cpu(), torch.from_numpy(d["pred_real"])) < OUT_TOL and rel(pf.detach().cpu(), torch.from_numpy(d["pred_fake"])) < OUT_TOL
    crit = torch.nn.BCEWithLogitsLoss()
    loss = (crit(1e-7 + pr - pf.mean(0, keepdim=True), torch.ones_like(pr)) + crit(1e-7 + pf - pr.mean(0, keepdim=True), torch.zeros_like(pr))) / 2
    assert abs(loss.item() - float(d["loss"])) < 1e-4 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    for k, p in S.named_parameters():
        g = p.grad.cpu()
        if k == "fc.0.weight":
            assert abs(g.double().abs().sum().item() - float(d["gradsum." + k])) < 2e-3 * float(d["gradsum." + k])
            g = g[:16]
        assert rel(g, torch.from_numpy(d["grad." + k])) < GRAD_TOL, k


def test_discriminator_ragged_shapes(srk):
    for shp in [(1, 80, 80), (1, 75, 75), (3, 40, 24)]:
        D = srk.Markovian_Discriminator(shp, [16, 32, 32, 64]).cuda()
        sd = _load_closed_form(D, gain=2.0)
        x = torch.rand(2, *shp)
        ref = O.discriminator_forward(sd, x)
        y = D(x.cuda())
        assert tuple(y.shape[1:]) == tuple(D.output_shape) == O.discriminator_output_shape(shp)
        assert rel(y.detach().cpu(), ref) < OUT_TOL


def test_discriminator_full_size_256_vs_oracle(srk):
    """BASELINE configs[2] size: Markovian_Discriminator((1,256,256),[16,32,32,64]) (models.py:149-174) on 2 jet images --
    forward, relativistic D loss, the gradient penalty's double backward (esrgan.py:596-606) and every weight gradient against
    the CPU oracle.  This is the size at which the small-channel kernels, the pixel-split weight-gradient variants and the
    Cin = 1 streaming weight gradient are selected in the training step."""
    D = srk.Markovian_Discriminator((1, 256, 256), [16, 32, 32, 64]).cuda()
    sd = _load_closed_form(D, gain=2.0)
    assert tuple(D.output_shape) == (1, 16, 16) == O.discriminator_output_shape((1, 256, 256))
    _, gt = O.jet_images(2, 1, 256, 256, 31, 4)
    _, gen = O.jet_images(2, 1, 256, 256, 32, 4)
    gen = gen * 0.7 + 0.05
    eps = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(9))
    sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lo, gpo = O.d_phase_loss(sdo, gt, gen, eps, 0.01)
    lo.backward()
    pr, pf = D(gt.cuda(), None), D(gen.cuda(), None)
    assert rel(pr.detach().cpu(), O.discriminator_forward(sd, gt)) < OUT_TOL
    assert rel(pf.detach().cpu(), O.discriminator_forward(sd, gen)) < OUT_TOL
    interp = (eps.cuda() * gt.cuda() + (1 - eps.cuda()) * gen.cuda()).requires_grad_(True)
    pi = D(interp, None)
    grads = torch.autograd.grad(outputs=pi, inputs=interp, grad_outputs=torch.ones_like(pi), create_graph=True, retain_graph=True, only_inputs=True)[0]
    # The input gradient itself.  LeakyReLU' is discontinuous at 0: among the ~3 M pre-activations of this case a few lie within
    # fp32 rounding of 0, where the CPU's and the GPU's summation orders disagree on the sign; such a unit changes the gradient
    # inside its receptive field only (measured: one 5x5 and one 23x42 pixel patch = 0.56 % of the pixels, off by up to 3 % of
    # the tensor's max; forward unaffected).  So: all but 1 % of the pixels within the gradient tolerance and none off by more
    # than 10 % of the max; an indexing or masking error in a kernel would miss both by orders of magnitude.
    io = (eps * gt + (1 - eps) * gen).requires_grad_(True)
    go = torch.autograd.grad(O.discriminator_forward(sd, io).sum(), io)[0]
    gd = grads.detach().cpu()
    err = (gd - go).abs().flatten()
    assert (err > GRAD_TOL * go.abs().max()).float().mean().item() < 0.01
    assert err.max().item() < 0.1 * go.abs().max().item()
    gp = ((grads.view(2, -1).norm(2, dim=1) - 1) ** 2).mean() * 0.01 / 2
    no = go.view(2, -1).norm(2, dim=1)
    assert abs(gp.item() - gpo.item()) < 1e-2 * 0.01 * float(((no - 1).abs() * no).mean()) + 1e-9
    crit = torch.nn.BCEWithLogitsLoss()
    loss = (crit(1e-7 + pr - pf.mean(0, keepdim=True), torch.ones_like(pr)) + crit(1e-7 + pf - pr.mean(0, keepdim=True), torch.zeros_like(pr))) / 2 + gp
    assert abs(loss.item() - lo.item()) < 1e-4 * max(1.0, abs(lo.item()))
    loss.backward()
    for k, p in D.named_parameters():
        assert rel(p.grad.cpu(), sdo[k].grad) < GRAD_TOL, k


def test_sumpool_module(srk):
    x = torch.rand(2, 1, 16, 16).requires_grad_(True)
    ref = O.sum_pool(x, 4)
    ref.sum().backward()
    xg = x.detach().cuda().requires_grad_(True)
    y = srk.SumPool2d(4)(xg)
    assert torch.allclose(y.detach().cpu(), ref.detach(), rtol=1e-6, atol=1e-6)
    y.sum().backward()
    assert torch.allclose(xg.grad.cpu(), x.grad)


@pytest.mark.parametrize("n", [4096 * 33, 1023, 7])
def test_lrelu_grad_mul_is_the_where_chain(srk, n):
    """srk_lrelu_grad_mul (one pass) == torch.where(x > 0, g, g * s), bit for bit, incl. x == 0 and unaligned tails: the op the
    gradient penalty's double backward applies between two conv nodes (models.py:149,151; esrgan.py:598-606)."""
    import importlib
    L = importlib.import_module("super-resolution_amd")._lib
    g = torch.Generator().manual_seed(n)
    x = torch.randn(n, generator=g)
    x[::5] = 0.0
    gr = torch.randn(n, generator=g)
    x, gr = x.cuda(), gr.cuda()
    out = torch.empty_like(gr)
    L.lrelu_grad_mul(x, gr, out, 0.2)
    assert torch.equal(out, torch.where(x > 0, gr, gr * 0.2))
    with pytest.raises(ValueError):
        L.lrelu_grad_mul(x[:-1], gr, out, 0.2)


def test_cpu_tensor_fails_loudly(srk):
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=1)
    with pytest.raises(RuntimeError):
        gen(torch.rand(1, 1, 8, 8))


def test_weights_updated_by_fused_adam_are_never_stale(srk):
    """Fused Adam updates parameters WITHOUT bumping Parameter._version, so nothing may cache packed weights by
    version: after an optimizer step the very next forward must use the new values (G and D)."""
    D = srk.Markovian_Discriminator((1, 32, 32), [16, 32, 32, 64]).cuda()
    _load_closed_form(D, gain=2.0)
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=1, num_upsample=1).cuda()
    _load_closed_form(gen)
    x = torch.rand(2, 1, 32, 32, device="cuda")
    xl = torch.rand(2, 1, 16, 16, device="cuda")
    for mod, inp in ((D, x), (gen, xl)):
        opt = torch.optim.Adam([p for p in mod.parameters() if p.requires_grad], lr=1e-2, fused=True)
        y0 = mod(inp)
        y0.square().mean().backward()
        opt.step()
        with torch.no_grad():
            y1 = mod(inp)
        sd = {k: v.cpu() for k, v in mod.state_dict().items()}
        ref = O.discriminator_forward(sd, inp.cpu()) if mod is D else O.generator_forward(sd, inp.cpu(), 1, 1, 0.2, training=True)[0]
        assert rel(y1.cpu(), ref) < OUT_TOL
        assert (y1 - y0.detach()).abs().max().item() > 1e-3


def test_eval_mode_calculate_metrics(srk):
    """the eval-mode inference caller (evaluation/eval.py:455-494 minus EMD)."""
    es = __import__("importlib").import_module("super-resolution_amd.esrgan")
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=2, num_upsample=1, res_scale=0.1).cuda()
    sd = _load_closed_form(gen)
    ds = es.SyntheticJets(10, 1, 32, 32, 2, seed=3)
    got = srk.evaluation.calculate_metrics(gen, ds, torch.device("cuda"), batch_size=4, factor=2)
    batches = [(ds.lr[i:i + 4], ds.hr[i:i + 4]) for i in range(0, 10, 4)]
    ref = O.calculate_metrics(sd, batches, 2, 1, 0.1, 2)
    for k in ("hr_l1", "lr_l1"):
        for m in ("mean", "std"):
            assert abs(got[k][m] - ref[k][m]) < 1e-4 * max(1.0, abs(ref[k][m])), (k, m, got, ref)
    assert not gen.training


def test_reference_written_checkpoint_loads_and_matches(srk, golden_dir):
    """a .pth written by the reference's torch.save(generator.state_dict()) (esrgan.py:385) loads
    into the drop-in (eval.py:441 path) and reproduces the reference's eval-mode output."""
    sd = torch.load(os.path.join(golden_dir, "G10_ref_generator.pth"), map_location="cuda")
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=2, num_upsample=2, res_scale=0.1).cuda()
    gen.load_state_dict(sd)
    d = np.load(os.path.join(golden_dir, "G10_ref_generator_io.npz"))
    gen.eval()
    with torch.no_grad():
        y = gen(torch.from_numpy(d["x"]).cuda())
    ref = torch.from_numpy(d["y_eval"])
    assert (y.cpu() - ref).abs().max().item() < OUT_TOL * max(ref.abs().max().item(), 1e-3)


@pytest.mark.parametrize("name", ["G4_gen_f16_r1_u2", "G5_config0"])
def test_generator_golden_bf16x3_mode(srk, golden_dir, name):
    """Opt-in split-bf16 precision (engine.precision = "bf16x3"): forward / data-gradient convs run as three bf16 MFMAs
    per product.  Still inside BASELINE's 1e-3 on outputs; gradients within 5e-3 of each tensor's max-abs."""
    d = np.load(os.path.join(golden_dir, name + ".npz"))
    c, f, r, u = [int(v) for v in d["cfg"]]
    gen = srk.GeneratorRRDB(c, filters=f, num_res_blocks=r, num_upsample=u, res_scale=float(d["res_scale"])).cuda()
    gen._engine.precision = "bf16x3"
    _load_closed_form(gen, float(d["gain"]))
    x = torch.from_numpy(d["x"]).cuda()
    gen.train()
    y = gen(x)
    assert any(v == 1 for v in gen._engine.fmt_f.values()) and any(v == 1 for v in gen._engine.fmt_b.values())
    err = rel(y.cpu(), torch.from_numpy(d["y_train"]))
    assert err < OUT_TOL, err
    loss = (y - torch.from_numpy(d["target"]).cuda()).abs().mean()
    assert abs(loss.item() - float(d["loss"])) < 1e-4 * max(1.0, float(d["loss"]))
    loss.backward()
    for k, p in gen.named_parameters():
        key = "grad." + k
        if key in d.files:
            assert rel(p.grad.cpu(), torch.from_numpy(d[key])) < 5e-3, k


def test_hipgraph_replay_matches_eager(srk):
    """engine.use_graphs: forward/backward captured into hipGraphs and replayed; results must equal the eager launches
    bit for bit (same kernels, same order), across optimizer updates and new inputs."""
    torch.manual_seed(0)
    gen = srk.GeneratorRRDB(1, filters=16, num_res_blocks=2, num_upsample=2).cuda()
    _load_closed_form(gen)
    opt = torch.optim.Adam(gen.parameters(), lr=1e-3, fused=True)
    xs = [torch.rand(2, 1, 16, 16, device="cuda") for _ in range(3)]
    tgt = torch.rand(2, 1, 64, 64, device="cuda")

    def run(use_graphs):
        _load_closed_form(gen)
        opt.state.clear()
        gen._engine.use_graphs = use_graphs
        outs = []
        for x in xs:
            opt.zero_grad(set_to_none=True)
            y = gen(x)
            loss = (y - tgt).abs().mean()
            loss.backward()
            outs.append((y.detach().clone(), gen.conv1.weight.grad.clone(), gen.res_blocks[1].dense_blocks[2].b5[0].weight.grad.clone()))
            opt.step()
        with torch.no_grad():
            gen.eval(); outs.append((gen(xs[0]).clone(),)); gen.train()
        return outs
    eager = run(False)
    graphed = run(True)
    assert len(gen._engine._graphs) >= 1
    for a, b in zip(eager, graphed):
        for ta, tb in zip(a, b):
            assert torch.equal(ta, tb)
    gen._engine.use_graphs = False


def test_plain_bf16_mixed_precision_mode(srk, golden_dir):
    """engine.precision = "bf16" (BASELINE config 4 style mixed precision: bf16 MFMA operands, fp32 accumulate, fp32
    master weights and activations in HBM).  8-bit mantissas: loose bounds; the point is that the path runs, is
    deterministic and stays close to fp32."""
    d = np.load(os.path.join(golden_dir, "G4b_gen_c3_f16_r1_u1.npz"))       # 3-channel (photographic-style) generator
    c, f, r, u = [int(v) for v in d["cfg"]]
    gen = srk.GeneratorRRDB(c, filters=f, num_res_blocks=r, num_upsample=u, res_scale=float(d["res_scale"])).cuda()
    gen._engine.precision = "bf16"
    _load_closed_form(gen, float(d["gain"]))
    x = torch.from_numpy(d["x"]).cuda()
    y = gen(x)
    ref = torch.from_numpy(d["y_train"])
    assert rel(y.detach().cpu(), ref) < 3e-2
    (y - torch.from_numpy(d["target"]).cuda()).abs().mean().backward()
    g1 = gen.conv2.weight.grad.clone()
    assert rel(g1.cpu(), torch.from_numpy(d["grad.conv2.weight"])) < 0.1
    gen.zero_grad()
    y2 = gen(x)
    assert torch.equal(y2, y)


@pytest.mark.parametrize("tag,kw", [("tc", dict(use_transposed_conv=True)), ("full", dict(fully_tconv_upsample=True))])
def test_transposed_conv_generators_golden(srk, golden_dir, tag, kw):
    """G16 (from the imported reference): the ConvTranspose2d upsampling variants (models.py:69-83) -- 3x3 convs on the HIP
    kernels, the transposed convs on PyTorch -- forward, input gradient and weight gradients."""
    d = np.load(os.path.join(golden_dir, "G16_tconv_generators.npz"))
    gen = srk.GeneratorRRDB(1, 16, 1, num_upsample=2, res_scale=0.1, **kw).cuda()
    _load_closed_form(gen)
    x = torch.from_numpy(d[f"{tag}.lr"]).cuda().requires_grad_(True)
    y = gen(x)
    assert rel(y.detach().cpu(), torch.from_numpy(d[f"{tag}.y"])) < OUT_TOL
    (y - torch.from_numpy(d[f"{tag}.tgt"]).cuda()).abs().mean().backward()
    assert rel(x.grad.cpu(), torch.from_numpy(d[f"{tag}.dx"])) < GRAD_TOL
    named = dict(gen.named_parameters())
    for k in [k[len(tag) + 6:] for k in d.files if k.startswith(tag + ".grad.")]:
        assert rel(named[k].grad.cpu(), torch.from_numpy(d[f"{tag}.grad.{k}"])) < GRAD_TOL, k
