#!/usr/bin/env python3
"""Conv kernel throughput vs grid size (batch) -- diagnoses workgroup placement / tail effects."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
from bench_conv import timeit
H = W = 64; F = 64
for ci in (64, 320):
    for N in (4, 8, 12, 16, 24, 32, 48, 64):
        buf = torch.randn(N, H, W, 5 * F, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
        w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
        wp = torch.empty(L.packed_floats(ci, F), device="cuda")
        t = L.PackTable(buf.device); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
        b = torch.zeros(F, device="cuda")
        dt = timeit(lambda: L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, slope=0.01), iters=30)
        fl = 2.0 * N * H * W * F * ci * 9
        print(f"Cin={ci:3d} N={N:2d} blocks={N*32:5d}  {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF/s")
