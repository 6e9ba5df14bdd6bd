#!/usr/bin/env python3
"""Stability soak: N full-size GAN iterations on changing synthetic batches; reports throughput per 100 iterations, HBM use
(allocated / peak: must not grow) and that every loss stays finite."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
train = importlib.import_module("super-resolution_amd.train")
iters = int(os.environ.get("ITERS", 600))
torch.manual_seed(0)                      # (same initial weights every run: trajectories of two runs are comparable)
st = train.Stepper(workload="gan", res_blocks=23, device=torch.device("cuda"), hr=256, factor=4)
g = torch.Generator().manual_seed(0)
pool = [(10 * torch.rand(32, 1, 256, 256, generator=g) * (torch.rand(32, 1, 256, 256, generator=g) < 0.1)).cuda() for _ in range(8)]
t0 = time.perf_counter()
for it in range(iters):
    hr = pool[it % len(pool)]
    lr = torch.nn.functional.avg_pool2d(hr, 4) * 16
    out = st.step(lr, hr)
    if (it + 1) % 100 == 0:
        vals = st.loss_scalars(out)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0; t0 = time.perf_counter()
        assert all(v == v and abs(v) < 1e30 for v in vals.values()), vals
        print(f"iter {it+1}: {dt*10:.1f} ms/iter  g_loss {vals['g_loss']:.4f} d_loss {vals['d_loss_def']:.4f}/{vals['d_loss_pow']:.4f}  "
              f"HBM allocated {torch.cuda.memory_allocated()/2**30:.2f} GiB peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
print("soak ok")
