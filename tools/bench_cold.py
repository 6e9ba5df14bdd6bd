#!/usr/bin/env python3
"""Does the wino42 conv slow down when its input comes from HBM instead of L2 / Infinity Cache?  The same launch on ONE input
buffer (warm) and rotating over NB buffers that together exceed the 256 MB Infinity Cache (cold)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
N, H, F = 32, 64, 64
NB = int(os.environ.get("NB", 6))
fmt = int(os.environ.get("FMT", 6))
bufs = [torch.randn(N, H, H, 5 * F, device="cuda") for _ in range(NB)]
outs = [torch.empty(N, H, H, F, device="cuda") for _ in range(NB)]
for ci in (64, 192, 320):
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    b = torch.zeros(F, device="cuda")
    wp = torch.empty(L.packed_floats(ci, F, fmt), device="cuda")
    t = L.PackTable(bufs[0].device, fmt); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
    res = []
    for nb in (1, NB):
        def run(i):
            L.conv3x3(L.View(bufs[i % nb], 0, ci), wp, b, L.View(outs[i % nb]), N=N, H=H, W=H, OH=H, OW=H, Cin=ci, Cout=F, slope=0.01, wp_format=fmt)
        for i in range(12): run(i)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(60): run(i)
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 60 * 1e3)
    print(f"fmt{fmt} Cin={ci:3d}: warm {res[0]:7.1f} us   cold ({NB} buffers) {res[1]:7.1f} us", flush=True)
