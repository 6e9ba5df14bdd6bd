"""Which Python lines launch the small ATen kernels (fill / add / mul / where ...) of one GAN iteration.

The conv / weight-gradient launches go through the C-ABI; what is left on the torch side are the loss heads' glue and
autograd's own bookkeeping.  This lists, per (aten op, innermost package frame), the launches of ONE iteration so the
latency-bound discriminator path can be cleaned of avoidable tiny launches.   Usage: python tools/debug/aten_sites.py
"""
import collections
import importlib
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
train = importlib.import_module("super-resolution_amd.train")

st = train.Stepper(workload="gan", res_blocks=23, device=torch.device("cuda"), hr=256, factor=4)
g = torch.Generator().manual_seed(0)
hr = (10 * torch.rand(32, 1, 256, 256, generator=g) * (torch.rand(32, 1, 256, 256, generator=g) < 0.1)).cuda()
lr = torch.nn.functional.avg_pool2d(hr, 4) * 16
for _ in range(3):
    st.step(lr, hr)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    st.step(lr, hr)
    torch.cuda.synchronize()

WATCH = ("aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::where", "aten::copy_",
         "aten::sum", "aten::mean", "aten::sigmoid", "aten::log_sigmoid_forward", "aten::gt", "aten::lt", "aten::ge", "aten::neg",
         "aten::sub", "aten::div", "aten::ones_like", "aten::zeros_like", "aten::zeros", "aten::cat", "aten::clone",
         "aten::contiguous", "aten::_foreach_add_", "aten::_fused_adam_")
sites = collections.Counter()
for ev in prof.events():
    if ev.name not in WATCH:
        continue
    where = "<autograd / no package frame>"
    for fr in ev.stack:
        if "super-resolution_amd" in fr or "bench.py" in fr:
            where = fr.split("super-resolution_amd/")[-1]
            break
    sites[(ev.name, where)] += 1
tot = collections.Counter()
for (name, where), n in sites.items():
    tot[name] += n
print("per-op totals of one iteration:", dict(tot.most_common()))
for (name, where), n in sites.most_common(70):
    print(f"{n:5d}  {name:28s} {where}")
