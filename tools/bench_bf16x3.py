#!/usr/bin/env python3
"""Micro-benchmark: split-bf16 (bf16x3) conv kernel vs the exact-fp32 kernel on DenseResidualBlock shapes."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
from bench_conv import timeit
N, H, W, F = int(os.environ.get("N", 16)), 64, 64, 64
buf = torch.randn(N, H, W, 5 * F, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
for ci in (64, 128, 192, 256, 320):
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    b = torch.zeros(F, device="cuda")
    res = []
    for fmt in (0, 1):
        wp = torch.empty(L.packed_floats(ci, F), device="cuda")
        t = L.PackTable(buf.device, fmt=fmt); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
        dt = timeit(lambda: L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, slope=0.01, wp_format=fmt), iters=30)
        res.append(dt)
    fl = 2.0 * N * H * W * F * ci * 9
    print(f"Cin={ci:3d}: fp32 {res[0]*1e6:7.1f} us ({fl/res[0]/1e12:6.1f} TF/s)   bf16x3 {res[1]*1e6:7.1f} us ({fl/res[1]/1e12:6.1f} TF/s fp32-equivalent)  x{res[0]/res[1]:.2f}")
Hh = 256
x = torch.randn(N, Hh, Hh, F, device="cuda"); y = torch.empty(N, Hh, Hh, F, device="cuda")
w = torch.randn(F, F, 3, 3, device="cuda") * 0.02
for fmt in (0, 1):
    wp = torch.empty(L.packed_floats(F, F), device="cuda")
    t = L.PackTable(x.device, fmt=fmt); t.add(w, wp, M=F, k_off=0, k_len=F, K_total=F); t.run()
    dt = timeit(lambda: L.conv3x3(L.View(x), wp, None, L.View(y), N=N, H=Hh, W=Hh, OH=Hh, OW=Hh, Cin=F, Cout=F, slope=0.01, wp_format=fmt))
    print(f"HR 64->64 256^2 fmt {fmt}: {dt*1e6:8.1f} us  {2.0*N*Hh*Hh*F*F*9/dt/1e12:6.1f} TF/s")
