#!/usr/bin/env python3
"""Diagnostic (needs the -DSRK_STAMP build, see tools/stamp_wino.py): where do the waves of the Winograd weight-gradient
kernel spend their cycles?  Per workgroup (wave 0): tile bookkeeping | k-steps | barrier wait, in shader cycles."""
import ctypes, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
lib = L.lib()
N, H, W, F = int(os.environ.get("N", 32)), 64, 64, 64
buf = torch.randn(N, H, W, 5 * F, device="cuda"); E = torch.randn(N, H, W, 5 * F, device="cuda")
probs = [dict(x=L.View(buf, 0, k * F), dy=L.View(E, (5 - k) * F, F), dw=torch.empty(F, k * F, 3, 3, device="cuda"), db=torch.empty(F, device="cuda"),
              Cin=k * F, Cout=F) for k in range(1, 6)]
st = torch.zeros(4096 * 32, dtype=torch.int64, device="cuda")
lib.srk_debug_set_wstamps(ctypes.c_void_p(st.data_ptr()))
for _ in range(5):
    L.conv3x3_wgrad_batched(probs, N=N, H=H, W=W, OH=H, OW=W)
torch.cuda.synchronize()
s = st.cpu().view(-1, 8, 4).double()
s = s[s.sum((1, 2)) > 0]
tiles = N * 16 * 4 / 17.0
print(f"workgroups {s.shape[0]}, ~{tiles:.0f} tiles each; ideal MFMA time per tile (2 waves/SIMD share the pipe): 12288 cycles")
m = s.median(0).values / tiles
for w in range(8):
    print("wave %d (tile a=%d b=%d, pixel half %d): bookkeeping %5.0f | k-steps %6.0f | barrier wait %5.0f | total %6.0f" %
          (w, w & 1, (w >> 1) & 1, w >> 2, m[w, 0], m[w, 1], m[w, 2], m[w, :3].sum()))
