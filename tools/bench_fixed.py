#!/usr/bin/env python3
"""Fixed per-launch cost of the conv kernel: small Cin sweep at 1 block/CU."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
from bench_conv import timeit
H = W = 64; F = 64
N = int(os.environ.get("N", 8))
for ci in (8, 16, 32, 64, 128):
    buf = torch.randn(N, H, W, 320, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    wp = torch.empty(L.packed_floats(ci, F), device="cuda")
    t = L.PackTable(buf.device); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
    b = torch.zeros(F, device="cuda")
    dt = timeit(lambda: L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, slope=0.01), iters=50)
    print(f"Cin={ci:3d} N={N} chunks={ci//8:2d} {dt*1e6:8.1f} us")
# empty-ish kernel for launch floor
x = torch.zeros(1024, device="cuda")
dt = timeit(lambda: x.add_(1.0), iters=100)
print(f"torch add_ launch floor {dt*1e6:.1f} us")
