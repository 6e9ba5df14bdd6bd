#!/usr/bin/env python3
"""Diagnostic: phase stamps of the F(2x4,3x3) conv workgroups (make -C super-resolution_amd/csrc stamp;
   SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so python tools/stamp_w42.py)"""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, W, F = int(os.environ.get("N", 32)), 64, 64, 64
for ci in (64, 320):
    buf = torch.randn(N, H, W, 320, device="cuda"); out = torch.empty(N, H, W, F, device="cuda")
    w = torch.randn(F, ci, 3, 3, device="cuda") * 0.02
    wp = torch.empty(L.packed_floats(ci, F, 6), device="cuda")
    t = L.PackTable(buf.device, 6); t.add(w, wp, M=F, k_off=0, k_len=ci, K_total=ci); t.run()
    b = torch.zeros(F, device="cuda")
    stamps = torch.zeros((4096 + 4096 * 4) * 16, dtype=torch.int64, device="cuda")
    lib.srk_debug_set_w42_stamps(ctypes.c_void_p(stamps.data_ptr()))
    EPI = os.environ.get("EPI", "bias")
    mk = torch.randn(N, H, W, 320, device="cuda")
    kw = dict(slope=0.01) if EPI == "bias" else dict(mask=L.View(mk, 64, F), mask_slope=0.01)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    REPS = int(os.environ.get("REPS", 30))
    for it in range(REPS + 5):
        if it == 5:
            e0.record()
        L.conv3x3(L.View(buf, 0, ci), wp, (None if EPI == "mask" else b), L.View(out), N=N, H=H, W=W, OH=H, OW=W, Cin=ci, Cout=F, wp_format=6, **kw)
    e1.record()
    torch.cuda.synchronize()
    print("Cin=%d: %.1f us per launch (events, stamped build)" % (ci, e0.elapsed_time(e1) / REPS * 1e3))
    nwg = N * 8
    raw = stamps.cpu().view(-1, 16)[:nwg].double()
    s = raw[:, :8] * 0.01
    t0 = s[:, 0].min()
    names = ["wave 0 starts", "DMA + weights issued", "chunk 0 in LDS", "main loop done", "exchange done", "epilogue done"]
    print(f"Cin={ci}: workgroups={nwg}")
    for k in range(6):
        col = s[:, k] - t0
        print(f"   {names[k]:22s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
    cyc = raw[:, 8 + 3] - raw[:, 8 + 2]; us = (raw[:, 3] - raw[:, 2]) * 0.01
    nq = ci // 8
    print("   main loop: %.0f shader cycles per chunk (ideal 6144), shader clock %.3f GHz, %.2f us per chunk"
          % ((cyc / nq).median(), (cyc / us).median() * 1e-3, (us / nq).median()))
    d = s[:, 1:6] - s[:, 0:5]
    print("   per-workgroup phase medians: setup %.2f | wait chunk0 %.2f | main loop %.2f | exchange %.2f | epilogue %.2f us" % tuple(d.median(0).values.tolist()))
    seg = stamps.cpu().view(-1, 16)[4096:4096 + nwg * 4, :9].double() / nq
    for wv in range(4):
        m = seg[wv::4].median(0).values.tolist()
        print("   wave %d cycles per chunk: phases (e,mt) 00 %4.0f | 01 %4.0f | 10 %4.0f | 11 %4.0f | 20 %4.0f | 21 %4.0f | 30 %4.0f | barrier %4.0f | 31+DMA %4.0f | sum %5.0f (ideal 768 per phase)"
              % tuple([wv] + m + [sum(m)]))
