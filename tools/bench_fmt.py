#!/usr/bin/env python3
"""Forward-conv micro-benchmark of the Winograd formats side by side on dense-block shapes (run on the GPU box).
N=32 by default (the GAN step's launch shapes); FMTS=5,6 selects the formats."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib

def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def main():
    N, F = int(os.environ.get("N", 32)), 64
    fmts = [int(f) for f in os.environ.get("FMTS", "5,6").split(",")]
    dev = "cuda"
    for H in (64, 256):
        n = N if H == 64 else max(N // 8, 1)
        buf = torch.randn(n, H, H, 5 * F, device=dev)
        out = torch.empty(n, H, H, F, device=dev)
        mask = torch.randn(n, H, H, F, device=dev)
        for k in (1, 2, 3, 4, 5):
            ci = k * F
            w = torch.randn(F, ci, 3, 3, device=dev) * 0.02
            b = torch.zeros(F, device=dev)
            fl = 2.0 * n * H * H * F * ci * 9
            line = f"{H:3d}^2 x{n:2d} Cin={ci:3d}->64 "
            for fmt in fmts:
                wp = torch.empty(L.packed_floats(ci, F, fmt), device=dev)
                t = L.PackTable(buf.device, fmt); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
                dt = timeit(lambda: L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=n, H=H, W=H, OH=H, OW=H, Cin=ci, Cout=F, slope=0.01, wp_format=fmt))
                dm = timeit(lambda: L.conv3x3(L.View(buf, 0, ci), wp, None, L.View(out), N=n, H=H, W=H, OH=H, OW=H, Cin=ci, Cout=F, mask=L.View(mask), mask_slope=0.01, wp_format=fmt))
                line += f"| fmt{fmt}: {dt*1e6:7.1f} us {fl/dt/1e12:6.1f} TF/s alg  (mask {dm*1e6:7.1f} us) "
            print(line, flush=True)

if __name__ == "__main__":
    main()
