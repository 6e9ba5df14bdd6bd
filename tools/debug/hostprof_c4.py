#!/usr/bin/env python3
"""Host time per iteration of the configs[4] step (how far ahead of the GPU the host runs: 30 ms of kernels, ~1500 launches):
issue time of 10 iterations without synchronising (host time per step) vs with the final drain, then a cProfile of 5 iterations."""
import cProfile, pstats, importlib, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
train = importlib.import_module("super-resolution_amd.train")
import bench
w = bench.WORKLOADS["c4"]
dev = torch.device("cuda", 0)
st = train.Stepper(workload=w["step"], res_blocks=23, device=dev, hr=w["hr"], factor=4, channels=w["channels"], distributed=False)
st.generator._engine.precision = w["precision"]
lr, hr = bench.synth_batch(w["batch"], dev, 1234, w["channels"], w["hr"])
for _ in range(4): st.step(lr, hr)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): st.step(lr, hr)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time per step {(t1-t0)/10*1e3:.1f} ms; with the final drain {(t2-t0)/10*1e3:.1f} ms", flush=True)
pr = cProfile.Profile(); pr.enable()
for _ in range(5): st.step(lr, hr)
torch.cuda.synchronize()
pr.disable()
ps = pstats.Stats(pr); ps.sort_stats("tottime").print_stats(22)
