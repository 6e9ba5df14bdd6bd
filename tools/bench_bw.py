#!/usr/bin/env python3
"""Bandwidth-bound pieces against the HBM roof (PixelShuffle reported as HBM GB/s over algorithmic bytes
2*N*4F*H*W*sizeof(T) per stage; models.py:89) and the small-channel convolutions of srk_conv_small.hip over their algorithmic
bytes (input + output once).  Steady state: every case runs ~0.3 s back to back after a warm-up."""
import importlib, os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib


def timeit(fn, target_s=0.3):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fn(); e1.record(); torch.cuda.synchronize()
    iters = max(10, int(target_s / max(e0.elapsed_time(e1) * 1e-3, 1e-6)))
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


res = {}
N, F = int(os.environ.get("N", 16)), 64
for stage, H in ((0, 64), (1, 128)):
    x = torch.randn(N, H, H, 4 * F, device="cuda"); y = torch.empty(N, 2 * H, 2 * H, F, device="cuda")
    nbytes = 2.0 * N * 4 * F * H * H * 4
    dt = timeit(lambda: L.pixel_shuffle_fwd(x, y, N, H, H, F))
    res[f"pixel_shuffle_fwd stage{stage} (N={N}, {H}x{H}x{4*F} -> {2*H}x{2*H}x{F})"] = (dt, nbytes)
    dt = timeit(lambda: L.pixel_shuffle_bwd(y, x, N, H, H, F))
    res[f"pixel_shuffle_bwd stage{stage}"] = (dt, nbytes)
# small-channel convs at the GAN step's shapes (batch 32, 256x256)
Nb, Hh = 32, 256
for ci, co in ((1, 16), (16, 1), (64, 1), (1, 64)):
    x = torch.randn(Nb, Hh, Hh, ci, device="cuda"); y = torch.empty(Nb, Hh, Hh, co, device="cuda")
    w = torch.randn(co, ci, 3, 3, device="cuda") * 0.1
    wp = torch.empty(L.packed_floats(ci, co), device="cuda")
    t = L.PackTable(x.device); t.add(w, wp, M=co, k_off=0, k_len=ci, K_total=ci); t.run()
    b = torch.zeros(co, device="cuda")
    dt = timeit(lambda: L.conv3x3(L.View(x), wp, b, L.View(y), N=Nb, H=Hh, W=Hh, OH=Hh, OW=Hh, Cin=ci, Cout=co))
    res[f"conv3x3 {ci}->{co} at {Hh}x{Hh}, N={Nb} (srk_conv_small.hip)"] = (dt, float(Nb * Hh * Hh * (ci + co) * 4))
print("%-90s %10s %10s %8s" % ("case", "us", "GB/s", "of 8TB/s"))
out = {}
for k, (dt, nb) in res.items():
    print("%-90s %10.1f %10.0f %8.3f" % (k, dt * 1e6, nb / dt / 1e9, nb / dt / 8e12))
    out[k] = {"us": dt * 1e6, "algorithmic_bytes": nb, "GBps": nb / dt / 1e9, "frac_of_8TBps": nb / dt / 8e12}
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
