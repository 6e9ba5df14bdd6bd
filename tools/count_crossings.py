#!/usr/bin/env python3
"""Python -> C crossings per training iteration: every call into libsrk.so, by entry point (a proxy around the ctypes handle).
WORKLOAD=gan|g_only, BATCH from the environment."""
import collections, importlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("super-resolution_amd")._lib
train = importlib.import_module("super-resolution_amd.train")
real = L.lib()
counts = collections.Counter()


class Proxy:
    def __getattr__(self, name):
        fn = getattr(real, name)

        def wrapped(*a):
            counts[name] += 1
            return fn(*a)
        return wrapped


L._lib = Proxy()
wl = os.environ.get("WORKLOAD", "gan")
B = int(os.environ.get("BATCH", 32 if wl == "gan" else 16))
st = train.Stepper(workload=wl, res_blocks=23, device=torch.device("cuda"), hr=256, factor=4)
g = torch.Generator().manual_seed(0)
hr = (10 * torch.rand(B, 1, 256, 256, generator=g) * (torch.rand(B, 1, 256, 256, generator=g) < 0.1)).cuda()
lr = torch.nn.functional.avg_pool2d(hr, 4) * 16
for _ in range(3):
    st.step(lr, hr)
torch.cuda.synchronize()
counts.clear()
N = 4
for _ in range(N):
    st.step(lr, hr)
torch.cuda.synchronize()
tot = sum(counts.values())
print(f"{wl}: {tot / N:.0f} Python->C crossings per iteration")
for k, v in counts.most_common():
    print(f"   {k:44s} {v / N:8.1f}")
