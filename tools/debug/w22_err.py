#!/usr/bin/env python3
"""Debug: wino22 weight gradient vs a CPU reference, per tap."""
import importlib, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
L = importlib.import_module("super-resolution_amd")._lib
torch.manual_seed(0)
ci, co, h, w, n = [int(v) for v in os.environ.get("SHAPE", "64,64,16,16,1").split(",")]
x = torch.randn(n, ci, h, w); wt = (torch.randn(co, ci, 3, 3) * 0.05).requires_grad_(True)
b = torch.zeros(co, requires_grad=True)
y = F.conv2d(x, wt, b, padding=1)
g = torch.randn_like(y)
if os.environ.get("DELTA"):
    g.zero_(); r, c = [int(v) for v in os.environ["DELTA"].split(",")]; g[0, :, r, c] = 1.0
y.backward(g)
xd = x.permute(0, 2, 3, 1).contiguous().cuda(); gd = g.permute(0, 2, 3, 1).contiguous().cuda()
dw = torch.full((co, ci, 3, 3), float("nan"), device="cuda"); db = torch.full((co,), float("nan"), device="cuda")
L.conv3x3_wgrad(L.View(xd), L.View(gd), dw, db, N=n, H=h, W=w, OH=h, OW=w, Cin=ci, Cout=co)
torch.cuda.synchronize()
dw = dw.cpu(); ref = wt.grad
print("max |ref|", ref.abs().max().item(), "max err", (dw - ref).abs().max().item(), "bias err", (db.cpu() - b.grad).abs().max().item())
for t in range(9):
    e = (dw[:, :, t // 3, t % 3] - ref[:, :, t // 3, t % 3]).abs()
    print("tap", t, "max err %.4g" % e.max().item(), " ratio dw/ref median %.4g" % (dw[:, :, t // 3, t % 3] / ref[:, :, t // 3, t % 3]).median().item(),
          " bad couts", sorted(set((e > 1e-3).nonzero()[:, 0].tolist()))[:6], " bad cins", sorted(set((e > 1e-3).nonzero()[:, 1].tolist()))[:6])
