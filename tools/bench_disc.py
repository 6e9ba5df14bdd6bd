#!/usr/bin/env python3
"""Per-layer timing of the Markovian discriminator's convolutions (forward, data gradient, weight gradient) at the
headline geometry (batch 32, 256x256 HR input), straight through the C ABI."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
from bench_conv import timeit
N = int(os.environ.get("N", 32))
layers = [(1, 16, 1, 256), (16, 16, 2, 256), (16, 32, 1, 128), (32, 32, 2, 128), (32, 32, 1, 64), (32, 32, 2, 64),
          (32, 64, 1, 32), (64, 64, 2, 32), (64, 1, 1, 16)]
tot = [0.0, 0.0, 0.0]
for ci, co, s, h in layers:
    oh = (h + s - 1) // s
    x = torch.randn(N, h, h, ci, device="cuda"); y = torch.empty(N, oh, oh, co, device="cuda"); dy = torch.randn(N, oh, oh, co, device="cuda")
    dx = torch.empty(N, h, h, ci, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.05
    wp = torch.empty(L.packed_floats(ci, co), device="cuda"); wpt = torch.empty(L.packed_floats(co, ci), device="cuda")
    t = L.PackTable(x.device); t.add(w, wp, M=co, k_off=0, k_len=ci, K_total=ci); t.add(w, wpt, M=ci, k_off=0, k_len=co, K_total=co, transpose=True); t.run()
    b = torch.zeros(co, device="cuda"); dw = torch.empty_like(w); db = torch.empty(co, device="cuda")
    f = timeit(lambda: L.conv3x3(L.View(x), wp, b, L.View(y), N=N, H=h, W=h, OH=oh, OW=oh, Cin=ci, Cout=co, stride=s, in_slope=0.2), iters=20)
    if s == 1:
        d = timeit(lambda: L.conv3x3(L.View(dy), wpt, None, L.View(dx), N=N, H=oh, W=oh, OH=h, OW=h, Cin=co, Cout=ci), iters=20)
    else:
        d = timeit(lambda: L.conv3x3(L.View(dy), wpt, None, L.View(dx), N=N, H=oh, W=oh, OH=h, OW=h, Cin=co, Cout=ci, in_mode=L.IN_ZERO_UPSAMPLE), iters=20)
    g = timeit(lambda: L.conv3x3_wgrad(L.View(x), L.View(dy), dw, db, N=N, H=h, W=h, OH=oh, OW=oh, Cin=ci, Cout=co, stride=s, in_slope=0.2), iters=20)
    byt = 4 * N * (h * h * ci + oh * oh * co)
    print(f"L {ci:3d}->{co:3d} s{s} @{h:3d}: fwd {f*1e6:7.1f} us  dgrad {d*1e6:7.1f} us  wgrad {g*1e6:7.1f} us   (min HBM time {byt/6e12*1e6:6.1f} us)")
    tot[0] += f; tot[1] += d; tot[2] += g
print(f"total: fwd {tot[0]*1e3:.2f} ms  dgrad {tot[1]*1e3:.2f} ms  wgrad {tot[2]*1e3:.2f} ms")
