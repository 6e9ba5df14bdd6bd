#!/usr/bin/env python3
"""Generate tests/golden/G14_sparse_jets.npz from the reference's ``datasets.py`` (build container only): synthetic sparse
event rows (position/energy pair lists with duplicates, early zero terminators, a trailing label column) decoded by the
reference's own ``extract`` + ``Cutter`` + ``SumPool2d`` exactly as ``SparseJetDataset.__getitem__`` does
(datasets.py:236-249; the pandas/HDF5 file access itself needs PyTables, absent here, so the row table is handed to the
same code path directly).  Stand-ins for h5py / torchvision (absent, not used by this path) as in make_golden_train.py."""
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
for name in ["torchvision", "torchvision.transforms", "torchvision.utils", "torchvision.datasets", "h5py"]:
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
sys.path.insert(0, REF)
import datasets as rds  # noqa: E402  (the reference, read-only)

rng = np.random.RandomState(3)
ETA, PHI, F, NEV, L = 20, 24, 2, 12, 30
rows = np.zeros((NEV, 2 * L + 1), dtype=np.float32)
for e in range(NEV):
    n = rng.randint(3, L)                       # constituents; the rest of the row stays zero (terminator)
    pos = rng.randint(0, ETA * PHI, size=n)
    if e % 3 == 0:
        pos[n // 2] = pos[0]; pos[-1] = pos[0]  # duplicates: energies must accumulate in list order
    en = (rng.rand(n) * 5 + 0.01).astype(np.float32)
    if e % 4 == 1:
        en[n // 2] = 0.0                        # an early zero: everything after it is ignored
    rows[e, 0:2 * n:2] = pos
    rows[e, 1:2 * n:2] = en
    rows[e, -1] = e                             # label column, dropped by [:-1]
out = dict(rows=rows, cfg=np.array([ETA, PHI, F, L], dtype=np.int64))
pool = rds.SumPool2d(F)
for tag, thr, nh, pre in (("plain", None, None, 1), ("thres", 1.5, None, 1), ("nhard", None, 4, 1), ("pre2", None, None, 2)):
    cut = rds.Cutter(thr, nh)
    hrs, lrs = [], []
    for e in range(NEV):
        eta, phi = ETA * pre, PHI * pre
        r = rows[e].copy()
        if pre > 1:                             # positions re-drawn for the finer grid, same energies
            k = (r[1:-1:2] != 0).sum()
            r[0:2 * k:2] = np.random.RandomState(100 + e).randint(0, eta * phi, size=k)
        out.setdefault("rows_" + tag, []).append(r)
        img = cut(rds.extract(torch.Tensor(r[:-1]).view(-1, 2).t(), eta, phi))[None, ...]      # datasets.py:237
        if pre > 1:
            img = rds.SumPool2d(pre)(img)
        lr, hr = pool(img)[0], img[0].clone()
        lo, ho = O.sparse_jet_item(torch.from_numpy(r), ETA, PHI, F, pre, thr, nh)
        assert torch.equal(lo, lr) and torch.equal(ho, hr), tag
        hrs.append(hr.numpy()); lrs.append(lr.numpy())
    out["rows_" + tag] = np.stack(out["rows_" + tag])
    out["hr_" + tag], out["lr_" + tag] = np.stack(hrs), np.stack(lrs)
path = os.path.join(ROOT, "tests", "golden", "G14_sparse_jets.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), {k: np.asarray(v).shape for k, v in out.items()})
