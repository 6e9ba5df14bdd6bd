#!/usr/bin/env python3
"""The 16-bit weight gradient of a dense block (5 convs batched, BASELINE configs[4]'s trunk geometry) -- timing of the shipped library
or of an ablation build (SRK_LIB_PATH; -DWH_NO_DMA / -DWH_NO_MFMA give wrong results by design)."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = importlib.import_module("super-resolution_amd")._lib
N, H, F = 8, 128, 64
bufs = [torch.randn(N, H, H, 5 * F, device="cuda").half() for _ in range(4)]
Es = [torch.randn(N, H, H, 5 * F, device="cuda").half() for _ in range(4)]
sets = []
fl = 0.0
for D, E in zip(bufs, Es):
    probs = []
    for k in range(1, 6):
        ci = k * F
        probs.append(dict(x=L.View(D, 0, ci), dy=L.View(E, (5 - k) * F, F), dw=torch.empty(F, ci, 3, 3, device="cuda"),
                          db=torch.empty(F, device="cuda"), Cin=ci, Cout=F))
    sets.append(probs)
fl = sum(2.0 * N * H * H * F * k * F * 9 for k in range(1, 6))
run = lambda i: L.conv3x3_wgrad_batched(sets[i % 4], N=N, H=H, W=H, OH=H, OW=H, precision=3)
for i in range(8): run(i)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(40): run(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 40 * 1e3
print(f"{os.environ.get('SRK_LIB_PATH', 'shipped').split('/')[-1]}: dense-block weight gradient incl. reduce {us:.1f} us = {fl / us / 1e6:.0f} TF/s")
