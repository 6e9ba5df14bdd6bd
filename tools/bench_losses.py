#!/usr/bin/env python3
"""Fused loss-head kernels (csrc/srk_loss.hip) at the headline size (32 x 1 x 256 x 256 fp32 = 8.4 MB per tensor):
time, algorithmic GB/s against the 8 TB/s HBM roofline, and the same head composed from ATen ops the way
esrgan.py:522-547 does (forward + backward), for reference."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sr = importlib.import_module("super-resolution_amd")
LS = sr.losses
from bench_conv import timeit
B, H = int(os.environ.get("N", 32)), 256
g = torch.Generator().manual_seed(0)
gt = (10 * torch.rand(B, 1, H, H, generator=g) * (torch.rand(B, 1, H, H, generator=g) < 0.1)).cuda()
gen = (gt + 0.1 * torch.randn(B, 1, H, H, generator=g).cuda()).clamp_min(0).requires_grad_(True)
nbytes = gen.numel() * 4
edges = np.linspace(0, 10, 11)
hist = LS.DiffableHistogram(edges, sigma=500.0).to("cuda")
crit = LS.KLD_hist(torch.from_numpy(edges)).to("cuda")
mse = torch.nn.MSELoss()


def fused_nnz():
    return mse(LS.soft_count(gen, 0.0, 50000.0), LS.hard_count(gt))
def aten_nnz():
    return mse(torch.sigmoid(50000 * gen).sum(1).sum(1).sum(1), (gt > 0).sum(1).sum(1).sum(1).float())
def fused_mask():
    return LS.mask_l1(gen, gt)
def aten_mask():
    return (torch.sigmoid(5e4 * gen) - torch.sigmoid(5e4 * gt)).abs().mean()
def hit_aten(t):
    return torch.sigmoid(500 * (torch.cat(torch.split(torch.cat(torch.split(t, 4, -2)), 4, -1)) - 0.5)).mean((0, 1))
def fused_hit():
    return mse(LS.get_hitogram(gen, 4, 0.5, 500.0), LS.get_hitogram(gt, 4, 0.5, 500.0))
def aten_hit():
    return mse(hit_aten(gen), hit_aten(gt))
def fused_hist():
    return crit(hist.forward_positive(gen), hist.forward_positive(gt))
c_t = hist.centers.reshape(-1); d_t = hist.delta.reshape(-1)
def hist_aten(x):
    x = x.view(1, -1)
    x = x[:, None, :] - c_t[None, :, None]
    x = torch.sigmoid(500.0 * (x + d_t[None, :, None] / 2)) - torch.sigmoid(500.0 * (x - d_t[None, :, None] / 2))
    return x.sum(2)
def aten_hist():
    return crit(hist_aten(gen[gen > 0]), hist_aten(gt[gt > 0]))


# algorithmic HBM bytes, forward + backward: fwd reads gen (+gt); bwd reads gen (+gt) and writes d gen
alg = {"nnz": (1 + 1 + 1 + 1) * nbytes, "mask": (2 + 2 + 1) * nbytes, "hit": (1 + 1 + 1 + 1) * nbytes, "hist": (1 + 1 + 1 + 1) * nbytes}
for name, fused, aten in (("nnz", fused_nnz, aten_nnz), ("mask", fused_mask, aten_mask), ("hit", fused_hit, aten_hit), ("hist", fused_hist, aten_hist)):
    def run(fn):
        def f():
            gen.grad = None
            fn().backward()
        return f
    tf = timeit(run(fused), iters=20)
    ta = timeit(run(aten), iters=10)
    lf, la = fused().item(), aten().item()
    print(f"{name:5s} fused fwd+bwd {tf*1e6:8.1f} us  {alg[name]/tf/1e9:7.0f} GB/s ({alg[name]/tf/8e12*100:4.1f} % of 8 TB/s)   "
          f"ATen composition {ta*1e6:8.1f} us  ({ta/tf:4.1f}x)   loss {lf:.6g} vs {la:.6g}")
