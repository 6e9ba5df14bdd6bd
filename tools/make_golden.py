#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the *imported* reference (build container only).

Runs /root/reference/models.py unmodified on CPU, checks that oracle/esrgan_oracle.py
reproduces it (<=1e-6), and stores inputs + expected outputs as small fixtures.
Only data (inputs/outputs) is written; no reference source travels.

    python tools/make_golden.py            # writes tests/golden/G*.npz

Weights are closed-form (oracle.closed_form_fill) so they need no storage.
"""
import os
import sys
sys.dont_write_bytecode = True   # importing the reference must not leave __pycache__ inside /root/reference
import math
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present; goldens can only be regenerated in the build container")
sys.path.insert(0, REF)
import models as ref  # noqa: E402  (the reference, imported read-only)
from oracle import esrgan_oracle as O  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(8)


def close(a, b, tol=1e-6, what=""):
    d = (a - b).abs().max().item()
    s = max(b.abs().max().item(), 1e-30)
    assert d <= tol * max(1.0, s), f"oracle != reference for {what}: {d} (scale {s})"


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KB")


def seeded(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(shape, generator=g) * 2 - 1) * scale


# ---- G1: PixelShuffle index map, bit-exact (models.py:89)
x = torch.arange(2 * 8 * 3 * 3, dtype=torch.float32).reshape(2, 8, 3, 3)
y = torch.nn.PixelShuffle(2)(x)
assert torch.equal(O.pixel_shuffle(x, 2), y)
save("G1_pixel_shuffle", x=x, y=y)

# ---- G2: single conv3x3 + bias + LeakyReLU(0.01) (models.py:19-21)
g2 = {}
for (ci, co) in [(1, 64), (64, 64), (320, 64), (64, 256), (64, 1)]:
    conv = torch.nn.Conv2d(ci, co, 3, 1, 1, bias=True)
    sd = O.closed_form_fill({"weight": conv.weight.data, "bias": conv.bias.data})
    conv.load_state_dict(sd)
    xin = seeded((2, ci, 8, 8), 100 + ci + co)
    yref = torch.nn.LeakyReLU()(conv(xin))
    yo = O.lrelu(O.conv3x3(xin, sd["weight"], sd["bias"]), O.G_SLOPE)
    close(yo, yref, what=f"G2 {ci}->{co}")
    g2[f"x_{ci}_{co}"] = xin
    g2[f"y_{ci}_{co}"] = yref
save("G2_conv_lrelu", **g2)

# ---- G3: one DenseResidualBlock(16): fwd, input grad, weight grads (models.py:9-41)
drb = ref.DenseResidualBlock(16)
sd = O.closed_form_fill(drb.state_dict())
drb.load_state_dict(sd)
xin = seeded((2, 16, 8, 8), 3).requires_grad_(True)
yref = drb(xin)
gout = seeded(yref.shape, 33)
yref.backward(gout)
# oracle
sdo = {("p." + k): v.clone().requires_grad_(True) for k, v in sd.items()}
xo = xin.detach().clone().requires_grad_(True)
yo = O.dense_residual_block(sdo, "p", xo)
yo.backward(gout)
close(yo, yref, what="G3 fwd")
close(xo.grad, xin.grad, what="G3 dx")
arrs = dict(x=xin.detach(), gout=gout, y=yref.detach(), dx=xin.grad)
for k, p in drb.named_parameters():
    close(sdo["p." + k].grad, p.grad, what="G3 " + k)
    arrs["grad." + k] = p.grad
save("G3_drb16", **arrs)


# ---- G4/G6: small generators, train+eval fwd, srs, L1 grads (models.py:56-135)
def gen_case(name, channels, filters, nrb, nup, res_scale, xshape, seed, store_grads=True, gain=1.0):
    gen = ref.GeneratorRRDB(channels, filters=filters, num_res_blocks=nrb, num_upsample=nup, res_scale=res_scale)
    sd = O.closed_form_fill(gen.state_dict(), gain=gain)
    gen.load_state_dict(sd)
    xin = seeded(xshape, seed).abs()  # non-negative like jet images
    gen.train()
    ytr = gen(xin)
    srs = gen.srs
    tgt = seeded(ytr.shape, seed + 1).abs()
    loss = torch.nn.L1Loss()(ytr, tgt)
    gen.zero_grad()
    loss.backward()
    gen.eval()
    with torch.no_grad():
        yev = gen(xin)
    assert torch.equal(yev, torch.relu(ytr.detach()))
    # oracle
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v.clone()) for k, v in sd.items()}
    yo, so = O.generator_forward(sdo, xin, nrb, nup, res_scale, training=True)
    lo = O.warmup_loss(yo, tgt)
    lo.backward()
    with torch.no_grad():
        yeo, _ = O.generator_forward(sdo, xin, nrb, nup, res_scale, training=False)
    close(yo, ytr, what=name + " train fwd")
    close(so, srs, what=name + " srs")
    close(yeo, yev, what=name + " eval fwd")
    close(lo, loss, what=name + " loss")
    arrs = dict(x=xin, y_train=ytr.detach(), y_eval=yev, srs=srs.detach(), target=tgt, loss=loss.detach(),
                cfg=np.array([channels, filters, nrb, nup], dtype=np.int64), res_scale=np.float64(res_scale),
                gain=np.float64(gain))
    for k, p in gen.named_parameters():
        if p.grad is None:
            continue
        close(sdo[k].grad, p.grad, tol=2e-6, what=name + " grad " + k)
        if store_grads:
            arrs["grad." + k] = p.grad
    keys = list(gen.state_dict().keys())
    arrs["n_keys"] = np.int64(len(keys))
    save(name, **arrs)
    return keys


gen_case("G4_gen_f16_r1_u2", 1, 16, 1, 2, 0.2, (2, 1, 8, 8), 4)
gen_case("G6_gen_resscale01", 1, 16, 1, 1, 0.1, (2, 1, 8, 8), 6)
gen_case("G4b_gen_c3_f16_r1_u1", 3, 16, 1, 1, 0.2, (1, 3, 8, 8), 44)
# ---- G5: config 0 exactly: GeneratorRRDB(1, 32, 2) on 4x1x32x32 (BASELINE.json configs[0])
keys = gen_case("G5_config0", 1, 32, 2, 1, 0.2, (4, 1, 32, 32), 5, store_grads=False)
assert len(keys) == 72, len(keys)
# key-name contract for the state_dict
with open(os.path.join(OUT, "G5_state_keys.txt"), "w") as f:
    f.write("\n".join(keys) + "\n")

# ---- G7: Markovian_Discriminator fwd, grads, gradient penalty double backward
# (models.py:140-174, esrgan.py:596-606)
D = ref.Markovian_Discriminator((1, 32, 32), [16, 32, 32, 64])
sd = O.closed_form_fill(D.state_dict(), gain=2.0)
D.load_state_dict(sd)
assert tuple(D.output_shape) == O.discriminator_output_shape((1, 32, 32))
for shp in [(1, 80, 80), (1, 75, 75), (1, 256, 256)]:
    assert tuple(ref.Markovian_Discriminator(shp, [16, 32, 32, 64]).output_shape) == O.discriminator_output_shape(shp)
gt = seeded((3, 1, 32, 32), 7).abs()
gen = seeded((3, 1, 32, 32), 8).abs()
eps_ = torch.rand(3, 1, 1, 1, generator=torch.Generator().manual_seed(9))
# reference-side D phase (esrgan.py:569-616 restated against the reference module)
crit = torch.nn.BCEWithLogitsLoss()
pred_real = D(gt, None)
pred_fake = D(gen, None)
valid = torch.ones(3, *D.output_shape)
fake = torch.zeros(3, *D.output_shape)
loss_real = crit(1e-7 + pred_real - pred_fake.mean(0, keepdim=True), valid)
loss_fake = crit(1e-7 + pred_fake - pred_real.mean(0, keepdim=True), fake)
loss_D = (loss_real + loss_fake) / 2
interp = (eps_ * gt + (1 - eps_) * gen)
interp.requires_grad = True
pi = D(interp, None)
grads = torch.autograd.grad(outputs=pi, inputs=interp, grad_outputs=valid, create_graph=True, retain_graph=True, only_inputs=True)[0]
gp = ((grads.view(3, -1).norm(2, dim=1) - 1) ** 2).mean() * 0.01 / 2
tot = loss_D + gp
D.zero_grad()
tot.backward()
sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
lo, gpo = O.d_phase_loss(sdo, gt, gen, eps_, 0.01)
lo.backward()
close(O.discriminator_forward(sdo, gt), pred_real, what="G7 fwd")
close(gpo, gp, what="G7 gp")
close(lo, tot, what="G7 loss")
arrs = dict(gt=gt, gen=gen, eps=eps_, pred_real=pred_real.detach(), pred_fake=pred_fake.detach(),
            input_grad_gp=grads.detach(), gp=gp.detach(), loss=tot.detach())
for k, p in D.named_parameters():
    close(sdo[k].grad, p.grad, tol=2e-6, what="G7 grad " + k)
    arrs["grad." + k] = p.grad
save("G7_discriminator", **arrs)
with open(os.path.join(OUT, "G7_state_keys.txt"), "w") as f:
    f.write("\n".join(D.state_dict().keys()) + "\n")

# ---- G9: SumPool2d (models.py:297-305)
x = seeded((2, 1, 16, 16), 10).abs()
save("G9_sumpool", x=x, y4=ref.SumPool2d(4)(x), y2=ref.SumPool2d(2)(x))
assert torch.equal(O.sum_pool(x, 4), ref.SumPool2d(4)(x))
print("all goldens written; oracle == reference on every case")
