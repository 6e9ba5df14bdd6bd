"""is the distance of a discriminator BIAS gradient from the float64 oracle (1.0e-2 on D1 model.6.bias at batch 32,
CPU float32: 2.2e-3) made by the summation of dy inside the weight-gradient path, or is it already in dy?
For every weight-gradient launch of the D phase (tests/test_train_gpu.py's batch-32 setup) this compares the bias gradient the library
returned with a float64 sum of the SAME dy tensor it was given:  |db - sum64(dy)| / max|sum64(dy)|  is what the summation adds; the
rest of the distance to the oracle was in dy before the kernel saw it."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import esrgan_oracle as O  # noqa: E402  (checker only)
import test_train_gpu as T  # noqa: E402

L = importlib.import_module("super-resolution_amd._lib")
Nb = int(os.environ.get("NB", "32"))
lr, hr = O.jet_images(Nb, 1, 256, 256, 21, 4)
st, gsd, dsds = T._mk("gan", res_blocks=2, filters=64, hr=256, factor=4)
with torch.no_grad():
    gen = [st.generator(lr.cuda()), None]
    gen[1] = st.generator.srs
eps = torch.rand(Nb, 1, 1, 1, generator=torch.Generator().manual_seed(5))
rec = []
orig_seq, orig_one = L.conv3x3_wgrad_seq, L.conv3x3_wgrad


def seq(calls):
    for x, dy, dw, db, kw in calls:
        rec.append((dy.t, db, kw))
    return orig_seq(calls)


def one(x, dy, dw, db, **kw):
    rec.append((dy.t, db, kw))
    return orig_one(x, dy, dw, db, **kw)


L.conv3x3_wgrad_seq, L.conv3x3_wgrad = seq, one
gt = [hr.cuda(), hr.cuda() ** st.scaling_power]
for k in range(2):
    rec.clear()
    st.discriminators[k].zero_grad()
    loss_D, gp = st.d_phase_loss(k, gt[k], gen[k].detach(), eps.cuda())
    loss_D.backward()
    torch.cuda.synchronize()
    # oracle in float64
    dk = {n: v.clone().double().requires_grad_(True) for n, v in dsds[k].items()}
    lD, _ = O.d_phase_loss(dk, hr.double() if k == 0 else (hr ** st.scaling_power).double(), gen[k].detach().cpu().double(), eps.double(), 0.01)
    lD.backward()
    print(f"D{k}: {len(rec)} weight-gradient launches with a bias; per launch: Cout, pixels, |db - sum64(dy)| / max|sum64(dy)|")
    for dy, db, kw in rec:
        if db is None:
            continue
        s64 = dy.double().sum((0, 1, 2))
        print(f"   Cout {kw['Cout']:3d} stride {kw.get('stride', 1)} px {dy.shape[0] * dy.shape[1] * dy.shape[2]:8d}:"
              f" {((db.double() - s64).abs().max() / s64.abs().max()).item():.2e}   max|sum| {s64.abs().max().item():.3e}  sum|dy| {dy.double().abs().sum((0,1,2)).max().item():.3e}")
    named = dict(st.discriminators[k].named_parameters())
    for n in ("model.0.bias", "model.6.bias", "model.8.bias"):
        ref = dk[n].grad
        print(f"   {n}: |HIP - float64 oracle| / max = {((named[n].grad.cpu().double() - ref).abs().max() / ref.abs().max()).item():.2e}")
