#!/usr/bin/env python3
"""Generate tests/golden/G8_train_trajectory.npz (and, with ``--heads``, G12_train_heads_trajectory.npz) by DRIVING the reference's own ``esrgan.train()`` (build container
only).  The reference loop runs unmodified on CPU; three inert accommodations, all off the hot path:
  1. empty stand-in modules for packages that are absent here and only needed by plotting / dataset I/O
     (torchvision, h5py, energyflow), placed in sys.modules before the import;
  2. ``ReduceLROnPlateau`` accepts and ignores the ``verbose`` kwarg removed in torch 2.10 (esrgan.py:308,310);
  3. ``esrgan.get_dataset`` returns a synthetic jet dataset; model constructors are wrapped so the freshly built
     modules get closed-form weights; ``torch.rand`` is wrapped to record the gradient-penalty epsilons.
Stored: the exact batches in the order the reference consumed them, the epsilons, and the per-iteration loss series
from the reference's info.json.  No reference source is stored.
"""
import json
import os
import sys
sys.dont_write_bytecode = True   # importing the reference must not leave __pycache__ inside /root/reference
import tempfile
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present")
sys.path.insert(0, ROOT)
from oracle import esrgan_oracle as O  # noqa: E402

# ---- (1) stand-ins for absent, off-path packages
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

# ---- (2) scheduler kwarg
_RLP = torch.optim.lr_scheduler.ReduceLROnPlateau


class _RLPCompat(_RLP):
    def __init__(self, *a, verbose=None, **k):
        super().__init__(*a, **k)


torch.optim.lr_scheduler.ReduceLROnPlateau = _RLPCompat

os.chdir(REF)
sys.path.insert(0, REF)
import matplotlib  # noqa: E402
matplotlib.use("Agg")
import esrgan as ref  # noqa: E402  (the reference, imported read-only)

HEADS = "--heads" in sys.argv      # second fixture: the optional physics loss heads switched on (esrgan.py:522-547)
CFG = dict(hr=16, factor=2, R=1, batch=4, n_batches=5, warmup=2, seed=7, n_items=16)
HEAD_FLAGS = ["--lambda_nnz", "1e-5", "--lambda_mask", "0.5", "--lambda_hit", "20", "--lambda_hist", "0.05", "--bins", "4",
              "--sigma", "5", "--hit_threshold", "0.5"]

# ---- (3) dataset, closed-form weights, epsilon recorder
lr_all, hr_all = O.jet_images(CFG["n_items"], 1, CFG["hr"], CFG["hr"], 99, CFG["factor"])
consumed = []


class DS(torch.utils.data.Dataset):
    def __len__(self):
        return CFG["n_items"]

    def __getitem__(self, i):
        consumed.append(i)
        return {"lr": lr_all[i], "hr": hr_all[i]}


ref.get_dataset = lambda *a, **k: DS()
_G, _D = ref.GeneratorRRDB, ref.Markovian_Discriminator
_dcount = [0]


def G_wrapped(*a, **k):
    g = _G(*a, **k)
    g.load_state_dict(O.closed_form_fill(g.state_dict()))
    return g


def D_wrapped(*a, **k):
    d = _D(*a, **k)
    d.load_state_dict(O.closed_form_fill(d.state_dict(), gain=2.0 + _dcount[0]))
    _dcount[0] += 1
    return d


ref.GeneratorRRDB, ref.Markovian_Discriminator = G_wrapped, D_wrapped
eps_log = []
_rand = torch.rand


def rand_logged(*size, **k):
    out = _rand(*size, **k)
    if len(out.shape) == 4 and out.shape[1:] == (1, 1, 1):
        eps_log.append(out.clone())
    return out


torch.rand = rand_logged

tmp = tempfile.mkdtemp()
# options through the reference's own parser (it defines attributes that options/default.json does not)
sys.argv = ["esrgan.py", "--n_epochs", "10", "--batch_size", str(CFG["batch"]), "--factor", str(CFG["factor"]),
            "--hr_height", str(CFG["hr"]), "--hr_width", str(CFG["hr"]), "--residual_blocks", str(CFG["R"]),
            "--warmup_batches", str(CFG["warmup"]), "--n_batches", str(CFG["n_batches"]), "--report_freq", "1",
            "--set_seed", str(CFG["seed"]), "--root", tmp, "--model_path", "m", "--sample_interval", "-1",
            "--name", "g8", "--res_scale", "0.1"] + (HEAD_FLAGS if HEADS else [])
opt = ref.get_parser()
opt.save = False
opt.save_info = True
os.chdir(tmp)            # the reference writes info.json relative to the cwd (esrgan.py:163): never inside /root/reference
ret = ref.train(opt)
torch.rand = _rand
info = json.load(open(os.path.join(tmp, "m", "g8_info.json")))
loss = info["loss"]
nb = len(loss["g_loss"])
print("iterations recorded:", nb, "consumed items:", consumed, "eps draws:", len(eps_log))
order = np.array(consumed, dtype=np.int64).reshape(-1, CFG["batch"])
arrs = dict(cfg=np.array([CFG["hr"], CFG["factor"], CFG["R"], CFG["batch"], CFG["warmup"]], dtype=np.int64),
            lr=lr_all[order.reshape(-1)].reshape(order.shape[0], CFG["batch"], 1, CFG["hr"] // CFG["factor"], CFG["hr"] // CFG["factor"]),
            hr=hr_all[order.reshape(-1)].reshape(order.shape[0], CFG["batch"], 1, CFG["hr"], CFG["hr"]),
            eps=torch.stack(eps_log) if eps_log else torch.zeros(0))
if HEADS:
    arrs["binedges0"] = np.array(info["binedges0"], dtype=np.float64)
    arrs["binedges1"] = np.array(info["binedges1"], dtype=np.float64)
    arrs["head_flags"] = np.array([float(x) for x in HEAD_FLAGS[1::2]], dtype=np.float64)   # nnz, mask, hit, hist, bins, sigma, hit_threshold
for k in ["g_loss", "d_loss_def", "d_loss_pow", "def_loss", "pow_loss", "adv_loss", "adv_loss_pow", "pixel_loss", "pixel_loss_pow",
          "lr_loss", "lr_loss_pow"] + (["hist_loss", "hist_loss_pow", "nnz_loss", "nnz_loss_pow", "mask_loss", "mask_loss_pow",
                                       "hit_loss", "hit_loss_pow"] if HEADS else []):
    arrs["loss." + k] = np.array(loss[k], dtype=np.float64)
    print(k, loss[k])
out = os.path.join(ROOT, "tests", "golden", "G12_train_heads_trajectory.npz" if HEADS else "G8_train_trajectory.npz")
np.savez_compressed(out, **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in arrs.items()})
print("wrote", out, os.path.getsize(out))
