#!/usr/bin/env python3
"""Micro-benchmark of the raw conv / wgrad kernels on DenseResidualBlock shapes (run on the GPU box)."""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def main():
    N, H, W, F = int(os.environ.get("N", 16)), 64, 64, 64
    FMT = int(os.environ.get("FMT", 0))          # 0: direct fp32 kernel, 3 / 5: Winograd F(2,3) / F(4,3) along W
    dev = "cuda"
    buf = torch.randn(N, H, W, 5 * F, device=dev)
    out = torch.empty(N, H, W, F, device=dev)
    for k in range(1, 6):
        ci = k * F
        w = torch.randn(F, ci, 3, 3, device=dev) * 0.02
        wp = torch.empty(L.packed_floats(ci, F, FMT), device=dev)
        t = L.PackTable(buf.device, FMT); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
        b = torch.zeros(F, device=dev)
        fl = 2.0 * N * H * W * F * ci * 9
        dt = timeit(lambda: L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, slope=0.01, wp_format=FMT))
        print(f"fwd  Cin={ci:3d}->64  {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF/s")
        dw = torch.empty(F, ci, 3, 3, device=dev); db = torch.empty(F, device=dev)
        dt = timeit(lambda: L.conv3x3_wgrad(L.View(buf, 0, ci), L.View(out), dw, db, N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F))
        print(f"wgrad Cin={ci:3d}->64 {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF/s")
    # HR convs
    Hh = 256
    x = torch.randn(N, Hh, Hh, F, device=dev); y = torch.empty(N, Hh, Hh, F, device=dev)
    w = torch.randn(F, F, 3, 3, device=dev) * 0.02
    wp = torch.empty(L.packed_floats(F, F, FMT), device=dev)
    t = L.PackTable(x.device, FMT); t.add(w, wp, M=F, k_off=0, k_len=F, K_total=F); t.run()
    fl = 2.0 * N * Hh * Hh * F * F * 9
    dt = timeit(lambda: L.conv3x3(L.View(x), wp, None, L.View(y), N=N, H=Hh, W=Hh, OH=Hh, OW=Hh, Cin=F, Cout=F, slope=0.01, wp_format=FMT))
    print(f"fwd HR 64->64 256^2 {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF/s")
    w1 = torch.randn(1, F, 3, 3, device=dev) * 0.02
    wp1 = torch.empty(L.packed_floats(F, 1), device=dev)
    t = L.PackTable(x.device); t.add(w1, wp1, M=1, k_off=0, k_len=F, K_total=F); t.run()
    y1 = torch.empty(N, Hh, Hh, 1, device=dev)
    dt = timeit(lambda: L.conv3x3(L.View(x), wp1, None, L.View(y1), N=N, H=Hh, W=Hh, OH=Hh, OW=Hh, Cin=F, Cout=1))
    print(f"fwd HR 64->1 256^2 {dt*1e6:8.1f} us  ({2.0*N*Hh*Hh*F*9/dt/1e12:.2f} real TF/s, {N*Hh*Hh*F*4/dt/1e9:.0f} GB/s in)")

if __name__ == "__main__":
    main()
