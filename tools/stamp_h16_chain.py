#!/usr/bin/env python3
"""Diagnostic: per-conv phase stamps of the chain form of the 16-bit conv (make -C super-resolution_amd/csrc stamp;
   SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so python tools/stamp_h16_chain.py).  BWD=1 for the data-gradient pattern."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "debug"))
import chain_check as cc
L = cc.L
lib = L.lib()
N, H = int(os.environ.get("N", 8)), int(os.environ.get("HW", 128))
bw = bool(int(os.environ.get("BWD", 0)))
blocks = [cc.make_block(N, H, H, 100 + i, bw) for i in range(6)]
nwg = N * ((H + 15) // 16) * ((H + 31) // 32)
stamps = torch.zeros(nwg * 64, dtype=torch.int64, device="cuda")
lib.srk_debug_set_h16_stamps(ctypes.c_void_p(stamps.data_ptr()))
lib.srk_debug_set_h16_chain(1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(26):
    if it == 6:
        e0.record()
    L.conv3x3_seq(blocks[it % 6][2])
e1.record(); torch.cuda.synchronize()
print("%s chain at N=%d %dx%d: %.1f us per launch (events, stamped build)" % ("bwd" if bw else "fwd", N, H, H, e0.elapsed_time(e1) / 20 * 1e3))
s = stamps.cpu().view(nwg, 8, 8).double() * 0.01
t0 = s[:, 0, 0].min()
print("conv | start (min med max) | main loop done (med) | epilogue issued (med) | main us | epilogue us | flag wait begins (med) | waited us (med, max) | previous conv published (med)")
for c in range(5):
    st, ml, ep = s[:, c, 0] - t0, s[:, c, 1] - t0, s[:, c, 2] - t0
    line = "%d    | %6.2f %6.2f %6.2f | %6.2f | %6.2f | %5.2f | %5.2f |" % (c, st.min(), st.median(), st.max(), ml.median(), ep.median(), (ml - st).median(), (ep - ml).median())
    if s[:, c, 3].max() > 0:          # (16x16x32 form: the epilogue's set-up -- argument loads, bias -- ends here)
        line += " setup %4.2f |" % (s[:, c, 3] - t0 - ml).median()
    if c > 0:
        w0, w1 = s[:, c, 4] - t0, s[:, c, 5] - t0
        line += " %6.2f | %5.2f %5.2f | %6.2f" % (w0.median(), (w1 - w0).median(), (w1 - w0).max(), (s[:, c, 6] - t0).median())
    print(line)
