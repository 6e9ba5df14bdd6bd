#!/usr/bin/env python3
"""Sparse-jet decode throughput: one srk_jet_extract launch per batch (+ SumPool for LR) vs the reference-style per-event
Python loop of datasets.py:136-145 (timed with the oracle's restatement on the host, a bounded sample)."""
import importlib, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sr = importlib.import_module("super-resolution_amd")
from oracle import esrgan_oracle as O   # CPU baseline only
from bench_conv import timeit
B, ETA, PHI, L = int(os.environ.get("N", 256)), 256, 256, 200
rng = np.random.RandomState(0)
rows = np.zeros((B, 2 * L + 1), dtype=np.float32)
for b in range(B):
    n = rng.randint(30, L)
    rows[b, 0:2 * n:2] = rng.randint(0, ETA * PHI, size=n)
    rows[b, 1:2 * n:2] = rng.rand(n) * 10 + 0.1
ds = sr.datasets.SparseJetDataset(rows, etaBins=ETA, phiBins=PHI, factor=4)
dev_rows = torch.from_numpy(rows).cuda()
dt = timeit(lambda: ds.decode_batch(dev_rows), iters=50)
out_bytes = B * ETA * PHI * 4 * (1 + 1 + 1 / 16)          # hr write, hr read by the pool, lr write
print(f"GPU decode of {B} events ({ETA}x{PHI}, <= {L} constituents) + LR pool: {dt*1e6:8.1f} us  = {B/dt/1e6:6.2f} M events/s, "
      f"{out_bytes/dt/1e9:6.0f} GB/s of image traffic")
h2d = timeit(lambda: torch.from_numpy(rows).cuda(), iters=20)
print(f"host->device copy of the raw rows ({rows.nbytes/1e3:.0f} KB): {h2d*1e6:.1f} us")
t0 = time.perf_counter()
nb = 16
for b in range(nb):
    O.sparse_jet_item(torch.from_numpy(rows[b]), ETA, PHI, 4)
cpu = (time.perf_counter() - t0) / nb
print(f"reference-style per-event Python loop on the host: {cpu*1e6:.0f} us/event = {1/cpu:.0f} events/s  -> GPU batch decode is {cpu*B/dt:.0f}x")
