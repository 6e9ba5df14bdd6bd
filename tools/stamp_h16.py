#!/usr/bin/env python3
"""Diagnostic: phase stamps of the 16-bit-storage conv workgroups (make -C super-resolution_amd/csrc stamp;
   SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so python tools/stamp_h16.py).  N, HW, MT, FMT from the environment."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, F = int(os.environ.get("N", 8)), int(os.environ.get("HW", 128)), 64
fmt = int(os.environ.get("FMT", 7))
mt = int(os.environ.get("MT", 4))
dt = {7: torch.float16, 8: torch.bfloat16}[fmt]
lib.srk_debug_set_h16_mt(mt)


class PW:
    def __init__(self, t, fmt):
        self.t, self.fmt = t, fmt

    def data_ptr(self):
        return self.t.data_ptr()


NB = 8
bufs = [torch.randn(N, H, H, 5 * F, device="cuda").to(dt) for _ in range(NB)]
outs = [torch.empty(N, H, H, 5 * F, device="cuda", dtype=dt) for _ in range(NB)]
for ci in (64, 192, 320):
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    wp = torch.zeros(L.packed_floats(ci, F, fmt), device="cuda")
    t = L.PackTable(bufs[0].device, fmt); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
    b = torch.zeros(F, device="cuda")
    stamps = torch.zeros(8192 * 16, dtype=torch.int64, device="cuda")
    lib.srk_debug_set_h16_stamps(ctypes.c_void_p(stamps.data_ptr()))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    REPS = 40
    for it in range(REPS + 8):
        if it == 8:
            e0.record()
        L.conv3x3(L.View(bufs[it % NB], 0, ci), PW(wp, fmt), b, L.View(outs[it % NB], 64, F), N=N, H=H, W=H, OH=H, OW=H, Cin=ci, Cout=F, slope=0.01)
    e1.record()
    torch.cuda.synchronize()
    print("Cin=%d MT=%d: %.1f us per launch (events, stamped build, %d buffers in rotation)" % (ci, mt, e0.elapsed_time(e1) / REPS * 1e3, NB))
    nwg = N * ((H + 4 * mt - 1) // (4 * mt)) * ((H + 31) // 32)
    raw = stamps.cpu().view(-1, 16)[:nwg].double()
    s = raw[:, :8] * 0.01
    t0 = s[:, 0].min()
    names = ["wave 0 starts", "stage 0 DMA issued", "stage 0 in LDS", "main loop done", "epilogue issued", "stores drained"]
    for k in range(6):
        col = s[:, k] - t0
        print(f"   {names[k]:22s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
    nq = ci // 32
    cyc = raw[:, 8 + 3] - raw[:, 8 + 2]; us = (raw[:, 3] - raw[:, 2]) * 0.01
    print("   main loop: %.0f shader cycles per stage (ideal %d), shader clock %.3f GHz, %.2f us per stage"
          % ((cyc / nq).median(), 18 * mt * 2 * 32, (cyc / us).median() * 1e-3, (us / nq).median()))
    print("   wave 0: its share of stage 0 issued at median %.2f us" % (s[:, 6] - t0).median())
    d = s[:, 1:6] - s[:, 0:5]
    print("   per-workgroup phase medians: setup+issue %.2f | wait stage 0 %.2f | main loop %.2f | epilogue %.2f | drain %.2f us" % tuple(d.median(0).values.tolist()))
