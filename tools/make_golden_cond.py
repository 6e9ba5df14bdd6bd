#!/usr/bin/env python3
"""Generate tests/golden/G13_conditional_discriminator.npz from the imported reference (build container only):
``models.Conditional_Discriminator`` (models.py:189-223) forward on (HR image, LR condition), the relativistic D loss with
gradient penalty as esrgan.py:569-606 applies it (condition = LR ground truth for all three calls), input gradient of the
penalty and all weight gradients.  Closed-form weights; only inputs/outputs are stored."""
import os
import sys
sys.dont_write_bytecode = True
import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present")
sys.path.insert(0, REF)
import models as ref  # noqa: E402
from oracle import esrgan_oracle as O  # noqa: E402

torch.set_num_threads(8)
CH, NU, HR, F = [8, 16, 16, 32], 2, 32, 4
D = ref.Conditional_Discriminator((1, HR, HR), CH, num_upsample=NU)
sd = O.closed_form_fill(D.state_dict(), gain=2.0)
D.load_state_dict(sd)
lr, gt = O.jet_images(3, 1, HR, HR, 41, F)
_, gen = O.jet_images(3, 1, HR, HR, 42, F)
eps_ = torch.rand(3, 1, 1, 1, generator=torch.Generator().manual_seed(43))
crit = torch.nn.BCEWithLogitsLoss()
pred_real = D(gt, lr)
pred_fake = D(gen, lr)
valid = torch.ones(3, *D.output_shape); fake = torch.zeros(3, *D.output_shape)
loss_D = (crit(1e-7 + pred_real - pred_fake.mean(0, keepdim=True), valid) + crit(1e-7 + pred_fake - pred_real.mean(0, keepdim=True), fake)) / 2
interp = (eps_ * gt + (1 - eps_) * gen)
interp.requires_grad = True
pi = D(interp, lr)
grads = torch.autograd.grad(outputs=pi, inputs=interp, grad_outputs=valid, create_graph=True, retain_graph=True, only_inputs=True)[0]
gp = ((grads.view(3, -1).norm(2, dim=1) - 1) ** 2).mean() * 0.01 / 2
tot = loss_D + gp
D.zero_grad()
tot.backward()
# oracle cross-check
sdo = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
lo, gpo = O.d_phase_loss(sdo, gt, gen, eps_, 0.01, d_channels=CH, cond=lr, num_upsample=NU)
lo.backward()


def close(a, b, tol=1e-6, what=""):
    d = (a - b).abs().max().item(); s = max(b.abs().max().item(), 1e-30)
    assert d <= tol * max(1.0, s), (what, d, s)


close(O.conditional_discriminator_forward(sdo, gt, lr, CH, NU), pred_real, what="fwd")
close(gpo, gp, what="gp"); close(lo, tot, what="loss")
arrs = dict(cfg=np.array(CH + [NU, HR, F], dtype=np.int64), lr=lr, gt=gt, gen=gen, eps=eps_, pred_real=pred_real.detach(),
            pred_fake=pred_fake.detach(), input_grad_gp=grads.detach(), gp=gp.detach(), loss=tot.detach(),
            out_shape=np.array(D.output_shape, dtype=np.int64))
for k, p in D.named_parameters():
    close(sdo[k].grad, p.grad, tol=2e-6, what="grad " + k)
    arrs["grad." + k] = p.grad
path = os.path.join(ROOT, "tests", "golden", "G13_conditional_discriminator.npz")
np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs.items()})
with open(os.path.join(ROOT, "tests", "golden", "G13_state_keys.txt"), "w") as f:
    f.write("\n".join(D.state_dict().keys()) + "\n")
print("wrote", path, os.path.getsize(path), "output_shape", D.output_shape, "keys", len(D.state_dict()))

# ---- G15: Standard_Discriminator (models.py:177-186): forward, relativistic loss (mean over the batch of a [B, 1] output), grads
S = ref.Standard_Discriminator((1, HR, HR), [8, 16])
ssd = O.closed_form_fill(S.state_dict(), gain=2.0)
S.load_state_dict(ssd)
pr, pf = S(gt, lr), S(gen, lr)
v1, f1 = torch.ones(3, *S.output_shape), torch.zeros(3, *S.output_shape)
lS = (crit(1e-7 + pr - pf.mean(0, keepdim=True), v1) + crit(1e-7 + pf - pr.mean(0, keepdim=True), f1)) / 2
S.zero_grad(); lS.backward()
so = {k: v.clone().requires_grad_(True) for k, v in ssd.items()}
pro, pfo = O.standard_discriminator_forward(so, gt, [8, 16]), O.standard_discriminator_forward(so, gen, [8, 16])
lo2 = (O.bce_logits(1e-7 + pro - pfo.mean(0, keepdim=True), v1) + O.bce_logits(1e-7 + pfo - pro.mean(0, keepdim=True), f1)) / 2
lo2.backward()
close(pro, pr, what="std fwd"); close(lo2, lS, what="std loss")
arrs2 = dict(gt=gt, gen=gen, pred_real=pr.detach(), pred_fake=pf.detach(), loss=lS.detach(), out_shape=np.array(S.output_shape, dtype=np.int64))
for k, p in S.named_parameters():
    close(so[k].grad, p.grad, tol=2e-6, what="std grad " + k)
    # fc.0.weight's gradient is 1024 x 1024: keep a 16-row slice and its total instead of 4 MB
    arrs2["grad." + k] = p.grad[:16].clone() if k == "fc.0.weight" else p.grad
    if k == "fc.0.weight":
        arrs2["gradsum." + k] = p.grad.double().abs().sum()
path2 = os.path.join(ROOT, "tests", "golden", "G15_standard_discriminator.npz")
np.savez_compressed(path2, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrs2.items()})
with open(os.path.join(ROOT, "tests", "golden", "G15_state_keys.txt"), "w") as f:
    f.write("\n".join(S.state_dict().keys()) + "\n")
print("wrote", path2, os.path.getsize(path2), "output_shape", S.output_shape)
