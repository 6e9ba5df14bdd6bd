#!/usr/bin/env python3
"""Weight-gradient kernel on the generator's shapes: the batched 5-conv DenseResidualBlock launch and the single
convolutions, through the C ABI.  Env knobs are read once per process (SRK_WGRAD_T, SRK_WGRAD_KSPLIT, ...)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
from bench_conv import timeit
N, H, W, F = int(os.environ.get("N", 16)), 64, 64, 64
prec = int(os.environ.get("PREC", 0))
buf = torch.randn(N, H, W, 5 * F, device="cuda")
E = torch.randn(N, H, W, 5 * F, device="cuda")
probs, fl = [], 0.0
for k in range(1, 6):
    ci = k * F
    probs.append(dict(x=L.View(buf, 0, ci), dy=L.View(E, (5 - k) * F, F), dw=torch.empty(F, ci, 3, 3, device="cuda"),
                      db=torch.empty(F, device="cuda"), Cin=ci, Cout=F))
    fl += 2.0 * N * H * W * F * ci * 9
IT = int(os.environ.get("ITERS", 30))
dt = timeit(lambda: L.conv3x3_wgrad_batched(probs, N=N, H=H, W=W, OH=H, OW=W, precision=prec), iters=IT, warm=max(3, IT // 4))
print(f"batched DRB wgrad (5 convs, N={N}): {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF/s")
for k in (1, 5):
    p = probs[k - 1]
    f1 = 2.0 * N * H * W * F * p['Cin'] * 9
    dt = timeit(lambda: L.conv3x3_wgrad(p['x'], p['dy'], p['dw'], p['db'], N=N, H=H, W=W, OH=H, OW=W, Cin=p['Cin'], Cout=F, precision=prec), iters=30)
    print(f"single wgrad Cin={p['Cin']:3d}: {dt*1e6:8.1f} us  {f1/dt/1e12:6.1f} TF/s")
