#!/usr/bin/env python3
"""Diagnostic: phase stamps of the 2-D Winograd weight-gradient workgroups on the dense-block batch
   (make -C super-resolution_amd/csrc stamp;
    SRK_LIB_PATH=super-resolution_amd/csrc/build_stamp/libsrk_stamp.so python tools/stamp_w22.py).
   Stamps sit outside the tile loop only (a chained node inside it breaks the hand-placed MFMA slots)."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, W, F = int(os.environ.get("N", 32)), 64, 64, 64
buf = torch.randn(N, H, W, 5 * F, device="cuda")
E = torch.randn(N, H, W, 5 * F, device="cuda")
probs = []
for k in range(1, 6):
    ci = k * F
    probs.append(dict(x=L.View(buf, 0, ci), dy=L.View(E, (5 - k) * F, F), dw=torch.empty(F, ci, 3, 3, device="cuda"),
                      db=torch.empty(F, device="cuda"), Cin=ci, Cout=F))
stamps = torch.zeros(4096 * 16, dtype=torch.int64, device="cuda")
lib.srk_debug_set_w22_stamps(ctypes.c_void_p(stamps.data_ptr()))
REPS = int(os.environ.get("REPS", 30))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(REPS + 5):
    if it == 5:
        e0.record()
    L.conv3x3_wgrad_batched(probs, N=N, H=H, W=W, OH=H, OW=W)
e1.record()
torch.cuda.synchronize()
print("batched DRB wgrad + reduce: %.1f us per call (events, stamped build)" % (e0.elapsed_time(e1) / REPS * 1e3))
raw = stamps.cpu().view(-1, 16).double()
nwg = int((raw[:, 0] > 0).sum())
raw = raw[:nwg]
s = raw[:, :4] * 0.01
t0 = s[:, 0].min()
total_tiles = N * (H // 8) * (W // 16)
tiles = -(-total_tiles // (nwg // 15))
print(f"workgroups={nwg}, tiles per workgroup <= {tiles}")
for k, name in enumerate(["wave 0 starts", "prologue done (tile 0 in LDS, first operands)", "tile loop done", "partials written"]):
    col = s[:, k] - t0
    print(f"   {name:46s} min {col.min():7.2f}  median {col.median():7.2f}  max {col.max():7.2f} us")
cyc = raw[:, 8 + 2] - raw[:, 8 + 1]; us = s[:, 2] - s[:, 1]
print("   tile loop: %.0f shader cycles per tile (ideal 16384 = 256 MFMAs x 64), shader clock %.3f GHz, %.2f us per tile"
      % ((cyc / tiles).median(), (cyc / us).median() * 1e-3, (us / tiles).median()))
d = s[:, 1:4] - s[:, 0:3]
print("   per-workgroup medians: prologue %.2f | tile loop %.2f | transform + store %.2f us" % tuple(d.median(0).values.tolist()))
