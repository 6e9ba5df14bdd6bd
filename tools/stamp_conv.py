#!/usr/bin/env python3
"""Diagnostic: where does a conv workgroup spend its time?  Needs a -DSRK_STAMP build (tools/stamp_conv.sh)."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, W, F = 16, 64, 64, 64
for ci in (8, 64, 320):
    buf = torch.randn(N, H, W, 320, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    wp = torch.empty(L.packed_floats(ci, F), device="cuda")
    t = L.PackTable(buf.device); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
    b = torch.zeros(F, device="cuda")
    stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
    lib.srk_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    for _ in range(3):
        L.conv3x3(L.View(buf, 0, ci), wp, b, L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, slope=0.01)
    torch.cuda.synchronize()
    s = stamps.cpu().view(-1, 16)[:N * 16].double() * 0.01   # us
    t0 = s[:, 0].min()
    names = ["start", "loads issued", "chunk0 in LDS", "main loop done", "epilogue done"]
    print(f"Cin={ci}: blocks={s.shape[0]}")
    raw = stamps.cpu().view(-1, 16)[:N * 16].double()
    cyc = (raw[:, 6] - raw[:, 5]); us = (raw[:, 3] - raw[:, 2]) * 0.01
    print(f"   main loop: {cyc.median():.0f} shader cycles in {us.median():.2f} us -> in-kernel clock {cyc.median()/us.median()/1e3:.3f} GHz; "
          f"{cyc.median()/(ci//8)/144:.1f} cycles per MFMA")
    seg = raw[:, 8:14].median(0).values / max(ci // 8, 1)
    print("   per-chunk cycles: load-issue %.0f | taps0-3 %.0f | lds-store %.0f | taps4-7 %.0f | barrier %.0f | tap8+copy(+loop) %.0f" % tuple(seg[[0,1,2,3,4,5]].tolist()))
    for k in range(5):
        col = s[:, k] - t0
        print(f"   {names[k]:16s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
