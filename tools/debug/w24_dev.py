#!/usr/bin/env python3
"""Development check of the wino24 weight-gradient kernel (form 2) against the wino22 row-owner form (form 1) and float64: several
geometries incl. ragged ones and the unshuffled dy, then timing of the dense-block batch at N = 32 / 16."""
import importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools"))
L = importlib.import_module("super-resolution_amd")._lib
from bench_conv import timeit
torch.manual_seed(0)


def ref(x, dy):      # float64 weight / bias gradient of a 3 x 3 stride-1 pad-1 conv (NCHW)
    xd, dyd = x.double(), dy.double()
    w = torch.zeros(dy.shape[1], x.shape[1], 3, 3, dtype=torch.float64, device=x.device, requires_grad=True)
    y = torch.nn.functional.conv2d(xd, w, padding=1)
    y.backward(dyd)
    return w.grad, dyd.sum((0, 2, 3))


bad = 0
for (n, ci, co, h, w) in ((2, 64, 64, 16, 16), (2, 64, 64, 15, 17), (1, 128, 128, 7, 9), (3, 64, 192, 1, 1), (1, 96, 64, 33, 3), (2, 320, 64, 64, 64), (4, 64, 64, 40, 72)):
    x = torch.randn(n, ci, h, w, device="cuda"); dy = torch.randn(n, co, h, w, device="cuda")
    gw, gb = ref(x, dy)
    xn, dn = x.permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
    for form in (1, 2):
        L.lib().srk_debug_set_wgrad_w22_form(form)
        L.poison_lds()
        dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda"); db = torch.full((co,), float("nan"), device="cuda")
        L.conv3x3_wgrad(L.View(xn), L.View(dn), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co)
        torch.cuda.synchronize()
        ew = ((dw.double() - gw).abs().max() / gw.abs().max()).item(); eb = ((db.double() - gb).abs().max() / gb.abs().max()).item()
        print(f"N={n} {ci}->{co} {h}x{w} form {form}: dw err {ew:.2e}  db err {eb:.2e}", flush=True)
        bad += (not ew < 1e-4) or (not eb < 1e-4)
# unshuffled dy (upsampling conv)
n, F_, h, w = 2, 64, 8, 12
x = torch.randn(n, F_, h, w, device="cuda"); g = torch.randn(n, F_, 2 * h, 2 * w, device="cuda")
dy = torch.nn.functional.pixel_unshuffle(g, 2)            # (n, 4F, h, w): channel c*4 + i*2 + j
gw, gb = ref(x, dy)
for form in (1, 2):
    L.lib().srk_debug_set_wgrad_w22_form(form)
    L.poison_lds()
    dw = torch.full((4 * F_, F_, 3, 3), float("nan"), device="cuda"); db = torch.full((4 * F_,), float("nan"), device="cuda")
    L.conv3x3_wgrad(L.View(x.permute(0, 2, 3, 1).contiguous()), L.View(g.permute(0, 2, 3, 1).contiguous()), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_,
                    dy_mode=L.IN_UNSHUFFLE)
    ew = ((dw.double() - gw).abs().max() / gw.abs().max()).item(); eb = ((db.double() - gb).abs().max() / gb.abs().max()).item()
    print(f"unshuffle form {form}: dw err {ew:.2e}  db err {eb:.2e}", flush=True)
    bad += (not ew < 1e-4) or (not eb < 1e-4)
print("MISMATCH" if bad else "results agree", flush=True)
if bad and not os.environ.get("TIME_ANYWAY"):
    sys.exit(1)
F = 64
for N in (32, 16):
    buf = torch.randn(N, 64, 64, 5 * F, device="cuda"); E = torch.randn(N, 64, 64, 5 * F, device="cuda")
    probs, fl = [], 0.0
    for k in range(1, 6):
        ci = k * F
        probs.append(dict(x=L.View(buf, 0, ci), dy=L.View(E, (5 - k) * F, F), dw=torch.empty(F, ci, 3, 3, device="cuda"), db=torch.empty(F, device="cuda"), Cin=ci, Cout=F))
        fl += 2.0 * N * 64 * 64 * F * ci * 9
    for rnd in range(2):
        for form in (1, 2):
            L.lib().srk_debug_set_wgrad_w22_form(form)
            dt = timeit(lambda: L.conv3x3_wgrad_batched(probs, N=N, H=64, W=64, OH=64, OW=64), iters=30, warm=8)
            print(f"batched DRB wgrad N={N} form {form}: {dt*1e6:8.1f} us  {fl/dt/1e12:6.1f} TF/s algorithmic", flush=True)
sys.exit(1 if bad else 0)
