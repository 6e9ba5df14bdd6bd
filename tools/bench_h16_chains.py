#!/usr/bin/env python3
"""Does running the batch halves as TWO dependency chains on two streams pay for the 16-bit conv kernel?  One dense block's forward
chain (five convs, Cin = 64 .. 320 -> 64, each reading the slices the previous ones wrote) at 8 x 128 x 128:
  (a) one stream, full batch, 16-row tiles (the default form: one workgroup per CU),
  (b) one stream, full batch, shared-CU form (8-row tiles, two workgroups per CU),
  (c) two streams, half a batch each, shared-CU form.
Reports us per full-batch dense block (5 convs) and the TFLOP/s over its 144.5 GFLOP."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
N, H, F = int(os.environ.get("N", 8)), int(os.environ.get("HW", 128)), 64
fmt = 7
dt = torch.float16
REPS = 40


class PW:
    def __init__(self, t, fmt):
        self.t, self.fmt = t, fmt

    def data_ptr(self):
        return self.t.data_ptr()


wps = []
for k in range(1, 6):
    w = torch.randn(F, k * F, 3, 3, device="cuda") * 0.02
    wp = torch.zeros(L.packed_floats(k * F, F, fmt), device="cuda")
    t = L.PackTable(w.device, fmt); t.add(w, wp, M=F, k_off=0, k_len=k * F, K_total=k * F); t.run()
    wps.append(PW(wp, fmt))
b = torch.zeros(F, device="cuda")
NB = 6
bufs = [torch.randn(N, H, H, 6 * F, device="cuda").to(dt) for _ in range(NB)]


def block(buf, n):      # (one library call per block: the host must not become the bottleneck of the two-stream mode)
    L.conv3x3_seq([(L.View(buf, 0, k * F), wps[k - 1], b, L.View(buf, k * F, F), dict(N=n, H=H, W=H, OH=H, OW=H, Cin=k * F, Cout=F, slope=0.01))
                   for k in range(1, 6)])


def run(mode):
    if mode == "a":
        L.lib().srk_debug_set_h16_mt(4)
    else:
        L.lib().srk_debug_set_h16_mt(1)
    s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
    halves = [(x[:N // 2], x[N // 2:]) for x in bufs]

    def one(i):
        if mode in ("a", "b"):
            block(bufs[i % NB], N)
        else:
            ha, hb = halves[i % NB]
            with torch.cuda.stream(s0):
                block(ha, N // 2)
            with torch.cuda.stream(s1):
                block(hb, N // 2)
    for i in range(6):
        one(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if mode == "c":
        s0.wait_event(e0); s1.wait_event(e0)
    for i in range(REPS):
        one(i)
    if mode == "c":
        torch.cuda.current_stream().wait_stream(s0); torch.cuda.current_stream().wait_stream(s1)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / REPS * 1e3
    fl = sum(2.0 * N * H * H * F * k * F * 9 for k in range(1, 6))
    print(f"mode {mode}: {us:7.1f} us per dense block = {us / 5:6.1f} us per conv, {fl / us / 1e6:7.1f} TFLOP/s", flush=True)


for m in ("a", "b", "c", "a", "b", "c"):
    run(m)
