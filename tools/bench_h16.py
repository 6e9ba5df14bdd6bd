#!/usr/bin/env python3
"""Micro-benchmark of the 16-bit-storage conv kernel (wp_format 7 / 8) on BASELINE configs[4]'s dense-block shapes
(8 x 128 x 128, Cin = 64 .. 320 -> 64), warm (one buffer) and cold (NB buffers in rotation, as in the training step), both
tile forms.  FMT=7|8, N, HW, NB from the environment."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
N, H, F = int(os.environ.get("N", 8)), int(os.environ.get("HW", 128)), 64
NB = int(os.environ.get("NB", 8))
fmt = int(os.environ.get("FMT", 7))
dt = {7: torch.float16, 8: torch.bfloat16}[fmt]


class PW:
    def __init__(self, t, fmt):
        self.t, self.fmt = t, fmt

    def data_ptr(self):
        return self.t.data_ptr()


bufs = [torch.randn(N, H, H, 5 * F, device="cuda").to(dt) for _ in range(NB)]
outs = [torch.empty(N, H, H, 5 * F, device="cuda", dtype=dt) for _ in range(NB)]
for mt in (2, 4):
    L.lib().srk_debug_set_h16_mt(mt)
    for ci in (64, 128, 192, 256, 320):
        w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
        b = torch.zeros(F, device="cuda")
        wp = torch.empty(L.packed_floats(ci, F, fmt), device="cuda")
        t = L.PackTable(bufs[0].device, fmt); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
        pw = PW(wp, fmt)
        res = []
        for nb in (1, NB):
            def run(i):
                L.conv3x3(L.View(bufs[i % nb], 0, ci), pw, b, L.View(outs[i % nb], 64, F), N=N, H=H, W=H, OH=H, OW=H, Cin=ci, Cout=F, slope=0.01)
            for i in range(12): run(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(80): run(i)
            e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / 80 * 1e3)
        fl = 2.0 * N * H * H * F * ci * 9
        print(f"fmt{fmt} MT={mt} Cin={ci:3d}: warm {res[0]:7.1f} us = {fl / res[0] / 1e6:7.1f} TF/s   cold ({NB} buffers) {res[1]:7.1f} us = "
              f"{fl / res[1] / 1e6:7.1f} TF/s", flush=True)
