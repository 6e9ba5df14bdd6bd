#!/usr/bin/env python3
"""Diagnostic: phase stamps of the Winograd conv workgroups.  Needs a -DSRK_STAMP build:
   hipcc ... -DSRK_STAMP into super-resolution_amd/csrc/build_stamp/libsrk_stamp.so (every .hip of the Makefile)
   SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so python tools/stamp_wino.py"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, W, F = int(os.environ.get("N", 16)), 64, 64, 64
FMT = int(os.environ.get("FMT", 3))          # 3: F(2,3) kernel, 5: F(4,3) kernel
for ci in (64, 320):
    buf = torch.randn(N, H, W, 320, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    wp = torch.empty(L.packed_floats(ci, F, FMT), device="cuda")
    t = L.PackTable(buf.device, FMT); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
    b = torch.zeros(F, device="cuda")
    stamps = torch.zeros(2 * 4096 * 16, dtype=torch.int64, device="cuda")
    lib.srk_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    EPI = os.environ.get("EPI", "bias")       # bias | mask | res2: what the fused epilogue reads besides the accumulators
    mk = torch.randn(N, H, W, 320, device="cuda"); r2 = torch.randn(N, H, W, F, device="cuda")
    kw = dict(slope=0.01)
    if EPI == "mask":
        kw = dict(mask=L.View(mk, 64, F), mask_slope=0.01)
    elif EPI == "res2":
        kw = dict(alpha=0.02, r1=L.View(mk, 0, F), beta1=0.1, r2=L.View(r2), beta2=1.0)
    for _ in range(int(os.environ.get("REPS", 3))):
        L.conv3x3(L.View(buf, 0, ci), wp, (None if EPI == "mask" else b), L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, wp_format=FMT, **kw)
    torch.cuda.synchronize()
    nwg = N * 16 if FMT == 3 else N * 8
    s = stamps.cpu().view(-1, 16)[:nwg].double() * 0.01   # us (s_memrealtime, 100 MHz)
    t0 = s[:, 0].min()
    names = ["wave 0 starts", "before 1st barrier", "chunk 0 in LDS", "main loop done", "epilogue done"]
    print(f"Cin={ci}: workgroups={s.shape[0]}")
    for k in range(5):
        col = s[:, k] - t0
        print(f"   {names[k]:20s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
    d = s[:, 1:5] - s[:, 0:4]
    if FMT == 5:
        raw = stamps.cpu().view(-1, 16)[:nwg].double()
        cyc = raw[:, 6] - raw[:, 5]; us = (raw[:, 3] - raw[:, 2]) * 0.01
        nq = ci // 8
        print("   main loop: %.0f shader cycles per chunk (ideal 9216), shader clock %.3f GHz (s_memtime / s_memrealtime), %.2f us per chunk"
              % ((cyc / nq).median(), (cyc / us).median() * 1e-3, (us / nq).median()))
    if FMT == 5:
        allst = stamps.cpu().view(2, 4096, 16)
        for wname, wi in (("wave 0 (MFMA first)", 0), ("wave 4 (transform first)", 1)):
            seg = allst[wi, :nwg, 8:12].double() / (ci // 8)
            print("   %-26s cycles per chunk: LDS reads+waits %5.0f | 72 MFMAs + DMA pieces %6.0f | transforms %5.0f | barrier %5.0f | sum %6.0f"
                  % ((wname,) + tuple(seg.median(0).values.tolist()) + (seg.sum(1).median().item(),)))
    print("   per-workgroup phase medians: setup %.2f | wait chunk0 %.2f | main loop %.2f | epilogue %.2f us" % tuple(d.median(0).values.tolist()))
