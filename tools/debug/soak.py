#!/usr/bin/env python3
"""Soak of the headline iteration (BASELINE configs[2]: full G + D step, batch 32, 64 -> 256, 23 RRDBs, fp32): ITERS whole iterations
over four rotating synthetic batches, once with the chain forms (a dense block = one persistent launch) and once conv by conv.
Both runs start from the same seed; the F(2x4,3x3) chain kernel runs the same arithmetic in the same order as five launches, so
every loss of every iteration and every weight at the end must agree BIT FOR BIT.  Also printed: the chain protocol's counters
(launches, faults, recoveries) -- a soak without a single bounded wait running out -- and ms per iteration of both runs.
  python3 tools/debug/soak.py            (ITERS=60 RES_BLOCKS=23 BATCH=32 by default; WORKLOAD=gan | g_only)"""
import hashlib, importlib, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sr = importlib.import_module("super-resolution_amd")
train = importlib.import_module("super-resolution_amd.train")
L = sr._lib
import bench                                                          # synth_batch: the bench's synthetic jets

ITERS = int(os.environ.get("ITERS", 60))
RB = int(os.environ.get("RES_BLOCKS", 23))
BATCH = int(os.environ.get("BATCH", 32))
WL = os.environ.get("WORKLOAD", "gan")
dev = torch.device("cuda", 0)


def run(chain):
    L.lib().srk_debug_set_w42_chain(1 if chain else 0)
    torch.manual_seed(0)
    st = train.Stepper(workload=bench.WORKLOADS[WL]["step"], res_blocks=RB, device=dev, hr=256, factor=4, channels=1, distributed=False)
    batches = [bench.synth_batch(BATCH, dev, 77 + i, 1, 256) for i in range(4)]
    torch.manual_seed(1)                                              # the gradient penalty's epsilons
    s0 = L.chain_stats()
    trace = []
    t0 = None
    for it in range(ITERS):
        if it == min(5, ITERS - 1):                                   # (the first iterations pay allocations and page-in)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        out = st.step(*batches[it % 4])
        row = [out["g_loss"].reshape(-1)[:1]] + [v.reshape(-1)[:1] for _, v in sorted(out.get("d_loss", {}).items())]
        trace.append(torch.cat([r.float() for r in row]))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    s1 = L.chain_stats()
    h = hashlib.sha256()
    for _, p in sorted(st.generator.named_parameters()):
        h.update(p.detach().cpu().numpy().tobytes())
    for k in sorted(st.discriminators):
        for _, p in sorted(st.discriminators[k].named_parameters()):
            h.update(p.detach().cpu().numpy().tobytes())
    tr = torch.stack([t.cpu() for t in trace])
    return tr, h.hexdigest(), dt / max(1, ITERS - min(5, ITERS - 1)) * 1e3, {k: s1[k] - s0[k] for k in s1 if isinstance(s1[k], int)}


if __name__ == "__main__":
    res = {}
    for chain in (True, False):
        res[chain] = run(chain)
        tr, digest, ms, stats = res[chain]
        print(f"chain forms {'on ' if chain else 'off'}: {ITERS} iterations, {ms:.2f} ms each behind the first five; losses finite: {bool(torch.isfinite(tr).all())}; "
              f"first / last g_loss {tr[0, 0].item():.6f} / {tr[-1, 0].item():.6f}; weights sha256 {digest[:16]}; protocol counters {stats}", flush=True)
    a, b = res[True], res[False]
    same_losses = bool((a[0].view(torch.int32) == b[0].view(torch.int32)).all())
    print(f"losses of all {ITERS} iterations bit-identical: {same_losses}; weights after {ITERS} iterations bit-identical: {a[1] == b[1]}")
    sys.exit(0 if same_losses and a[1] == b[1] and bool(torch.isfinite(a[0]).all()) else 1)
