#!/usr/bin/env python3
"""Per-iteration wall times of a bench workload (synchronised after every iteration): finds iterations that stall on the host.
  WORKLOAD=c4|gan|g_only ITERS=40 python3 tools/debug/step_times.py"""
import importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sr = importlib.import_module("super-resolution_amd")
train = importlib.import_module("super-resolution_amd.train")
L = sr._lib
import bench
wl = os.environ.get("WORKLOAD", "c4")
w = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
st = train.Stepper(workload=w["step"], res_blocks=23, device=dev, hr=w["hr"], factor=4, channels=w["channels"], distributed=False)
st.generator._engine.precision = w["precision"]
lr, hr = bench.synth_batch(w["batch"], dev, 1234, w["channels"], w["hr"])
ts = []
for it in range(int(os.environ.get("ITERS", 40))):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.step(lr, hr)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(wl, "ms per iteration:", " ".join(f"{t:.1f}" for t in ts))
s = sorted(ts[5:])
print(f"behind the first five: median {s[len(s)//2]:.2f}  min {s[0]:.2f}  max {s[-1]:.2f}; chain protocol {L.chain_stats()}")
