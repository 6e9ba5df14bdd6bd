#!/usr/bin/env python3
"""Generate tests/golden/G11_loss_heads.npz from the reference's own helper functions (build container only):
``utils.softgreater / nnz_mask / get_hitogram / KLD_hist`` and ``models.DiffableHistogram`` evaluated, with gradients,
on seeded jet-like inputs, composed exactly as esrgan.py:522-547 composes them.  ``utils.py`` imports packages that
are absent here and only serve plotting / dataset I/O (h5py, energyflow, torchvision): empty stand-ins are placed in
sys.modules first.  Only inputs and outputs are stored."""
import os
import sys
sys.dont_write_bytecode = True
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present")
sys.path.insert(0, ROOT)
from oracle import esrgan_oracle as O  # noqa: E402

for name in ["torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets", "h5py", "energyflow",
             "energyflow.emd", "pyjet"]:
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision"].utils = sys.modules["torchvision.utils"]
sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
sys.modules["torchvision.datasets"].STL10 = object
sys.modules["energyflow"].emd = sys.modules["energyflow.emd"]
os.chdir(REF)
sys.path.insert(0, REF)
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import utils as rutils  # noqa: E402  (the reference, read-only)
import models as rmodels  # noqa: E402

torch.manual_seed(0)
B, H, W, FACT = 3, 16, 24, 4
_, gt = O.jet_images(B, 1, H, W, 31, FACT)
g = torch.Generator().manual_seed(32)
# a "generated" image: ground truth perturbed, some exact zeros, some small negatives clipped like GeneratorRRDB.out does
gen = (gt + 0.3 * torch.randn(gt.shape, generator=g) * (torch.rand(gt.shape, generator=g) < 0.3)).clamp_min(0)
gen = gen.clone().requires_grad_(True)
out = dict(gen=gen.detach().numpy(), gt=gt.numpy(), factor=np.int64(FACT))

# --- elementwise functions
for name, fn in (("softgreater", lambda t: rutils.softgreater(t, 0.2, 30, 0.05)), ("nnz_mask", lambda t: rutils.nnz_mask(t, 7.0))):
    y = fn(gen)
    w = torch.rand(y.shape, generator=g)
    gy, = torch.autograd.grad((y * w).sum(), gen)
    out[name + ".y"], out[name + ".w"], out[name + ".dx"] = y.detach().numpy(), w.numpy(), gy.numpy()

# --- nnz head (esrgan.py:522-525); sigma 50000 as the reference uses, plus a gentler one whose gradient is not all-zero
for tag, sig in (("nnz", 50000.0), ("nnz_soft", 3.0)):
    gen_nnz = rutils.softgreater(gen, 0, sig).sum(1).sum(1).sum(1)
    target = (gt > 0).sum(1).sum(1).sum(1).float()
    loss = torch.nn.MSELoss()(gen_nnz, target)
    d, = torch.autograd.grad(loss, gen)
    out[tag + ".count"], out[tag + ".target"], out[tag + ".loss"], out[tag + ".dx"] = gen_nnz.detach().numpy(), target.numpy(), loss.detach().numpy(), d.numpy()

# --- mask head (esrgan.py:526-529)
for tag, sig in (("mask", 5e4), ("mask_soft", 2.0)):
    loss = torch.nn.L1Loss()(rutils.nnz_mask(gen, sig), rutils.nnz_mask(gt, sig))
    d, = torch.autograd.grad(loss, gen)
    out[tag + ".loss"], out[tag + ".dx"] = loss.detach().numpy(), d.numpy()

# --- hit head (esrgan.py:543-547), reference defaults hit_threshold 0.5, sigma 500; and a gentle one; and sig <= 0
for tag, thr, sig in (("hit", 0.5, 500.0), ("hit_soft", 0.5, 2.0), ("hit_mean", 0.5, -1.0)):
    gh = rutils.get_hitogram(gen, FACT, thr, sig)
    th = rutils.get_hitogram(gt, FACT, thr, sig)
    loss = torch.nn.MSELoss()(gh, th)
    d, = torch.autograd.grad(loss, gen)
    out[tag + ".gen"], out[tag + ".target"], out[tag + ".loss"], out[tag + ".dx"] = gh.detach().numpy(), th.numpy(), loss.detach().numpy(), d.numpy()

# --- hist head (esrgan.py:441-456, 530-538)
nnz = gt.reshape(-1).numpy()
nnz = nnz[nnz > 0]
edges = O.hist_binedges(nnz, 6, 1.0)
# the reference's inline recipe, restated with its own calls for the cross-check of O.hist_binedges
from sklearn.cluster import KMeans  # noqa: E402
c, b = np.histogram(nnz, 100)
e_max = b[(np.cumsum(c) > len(nnz) * .9).argmax()]
sn = np.sort(nnz); sn = sn[sn <= e_max]
km = np.sort(KMeans(n_clusters=6, random_state=0).fit(sn.reshape(-1, 1)).cluster_centers_.flatten())
edges_ref = np.array([0, *(np.diff(km) / 2 + km[:-1]), e_max])
assert np.array_equal(edges, edges_ref)
out["hist.nnz"], out["hist.edges"] = nnz, edges
for tag, sig in (("hist", 500.0), ("hist_soft", 4.0)):
    hist = rmodels.DiffableHistogram(edges, sigma=sig, batchwise=False)
    crit = rutils.KLD_hist(torch.from_numpy(edges))
    gen_hist = hist(gen[gen > 0])
    real_hist = hist(gt[gt > 0])
    out[tag + ".gen"], out[tag + ".real"] = gen_hist.detach().numpy().copy(), real_hist.numpy().copy()
    loss = crit(gen_hist, real_hist)
    d, = torch.autograd.grad(loss, gen)
    out[tag + ".loss"], out[tag + ".dx"] = loss.detach().numpy(), d.numpy()
    full = hist(gen.detach())
    out[tag + ".all"] = full.numpy()

path = os.path.join(ROOT, "tests", "golden", "G11_loss_heads.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
for k in sorted(out):
    v = np.asarray(out[k])
    print(k, v.shape, float(np.abs(v).max()) if v.size else 0)
