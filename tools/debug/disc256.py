import importlib, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import esrgan_oracle as O
sr = importlib.import_module("super-resolution_amd")
H = int(os.environ.get("HW", 256))
for chans in ([16], [16, 32], [16, 32, 32], [16, 32, 32, 64]):
    D = sr.Markovian_Discriminator((1, H, H), chans).cuda()
    sd = O.closed_form_fill({k: v.cpu() for k, v in D.state_dict().items()}, gain=2.0)
    D.load_state_dict(sd)
    _, gt = O.jet_images(2, 1, H, H, 31, 4)
    if os.environ.get("DENSE", "0") == "1":
        _, gen = O.jet_images(2, 1, H, H, 32, 4)
        eps = torch.rand(2, 1, 1, 1, generator=torch.Generator().manual_seed(9))
        gt = eps * gt + (1 - eps) * (gen * 0.7 + 0.05)
    x = gt.clone().requires_grad_(True)
    yo = O.discriminator_forward(sd, x, channels=tuple(chans))
    go = torch.autograd.grad(yo.sum(), x)[0]
    xg = gt.cuda().requires_grad_(True)
    y = D(xg, None)
    g = torch.autograd.grad(y, xg, grad_outputs=torch.ones_like(y), create_graph=(os.environ.get("CG", "0") == "1"))[0].detach().cpu()
    err = (g - go).abs()
    idx = err.view(-1).argmax().item()
    n, r = divmod(idx, H * H); i, j = divmod(r, H)
    print(chans, "fwd rel", ((y.detach().cpu() - yo).abs().max() / yo.abs().max()).item(), "grad rel", (err.max() / go.abs().max()).item(),
          "at n,i,j", n, i, j, "nbad", int((err > 1e-3 * go.abs().max()).sum()))
    bad = (err[0, 0] > 1e-3 * go.abs().max())
    if bad.any():
        rows = bad.any(1).nonzero().flatten(); cols = bad.any(0).nonzero().flatten()
        print("   bad rows", rows.min().item(), rows.max().item(), "cols", cols.min().item(), cols.max().item(), "count img0", int(bad.sum()))
