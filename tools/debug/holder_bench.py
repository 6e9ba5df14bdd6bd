"""what does a collective's kernel beside the chain forms cost?  (One GPU: RCCL launches no kernel for a 1-rank
all-reduce, so the all-reduce kernel is stood in for by srk_debug_hold_cus: k workgroups that each hold a CU -- 4 waves, 64 KB of LDS:
neither kind of chain workgroup fits beside one -- for t microseconds, no peer needed.)  The holder is launched where engine.backward
issues a gradient bucket's all-reduce (25 points per generator backward: tail | RRDB 22..0 | conv1), on a stream of its own that waits for
the issue point of the main and the weight-gradient streams exactly as engine._reduce_bucket's issue stream does, and the main stream joins
it where _finish_reduce would wait for the collectives.
Full GAN iteration (batch 32, 64^2 -> 256^2, R = 23), chain forms on / off x holder off / (k CUs, t us): ms per iteration, interleaved rounds."""
import importlib
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import esrgan_oracle as O  # noqa: E402  (inputs only)

L = importlib.import_module("super-resolution_amd._lib")
train = importlib.import_module("super-resolution_amd.train")
R = int(os.environ.get("R", "23"))
STEPS = int(os.environ.get("STEPS", "8"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))
HOLDERS = [None] + [tuple(int(v) for v in h.split("x")) for h in os.environ.get("HOLDERS", "8x50,16x100,32x200,32x400").split(",")]

lr, hr = O.jet_images(32, 1, 256, 256, 21, 4)
lr, hr = lr.cuda(), hr.cuda()
torch.manual_seed(0)
st = train.Stepper(workload="gan", res_blocks=R, filters=64, device=torch.device("cuda"), hr=256, factor=4, res_scale=0.2)
eng = st.generator._engine
hold_stream = torch.cuda.Stream()
state = {"holder": None, "launched": 0}


def reduce_bucket(key):
    h = state["holder"]
    if h is None:
        return
    main = torch.cuda.current_stream()
    hold_stream.wait_stream(main)
    if eng._side is not None and eng.overlap_wgrad:
        hold_stream.wait_stream(eng._side)
    L.check(L.lib().srk_debug_hold_cus(h[0], h[1], hold_stream.cuda_stream), "srk_debug_hold_cus")
    state["launched"] += 1


def finish_reduce():
    eng._join_side()
    torch.cuda.current_stream().wait_stream(hold_stream)


eng._sync = True                 # (switches the bucket points of engine.backward on; no process group is touched: both hooks are replaced)
eng._grad_scale = 1.0
eng._reduce_bucket = reduce_bucket
eng._finish_reduce = finish_reduce


def run(chain, holder):
    L.lib().srk_debug_set_w42_chain(1 if chain else 0)
    state["holder"] = holder
    for _ in range(2):
        st.step(lr, hr)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        st.step(lr, hr)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / STEPS * 1e3


res = {}
for rnd in range(ROUNDS):
    for chain in (True, False):
        for h in HOLDERS:
            res.setdefault((chain, h), []).append(run(chain, h))
print(f"full GAN iteration, batch 32, R = {R}; holder bursts at the {state['launched'] // max(1, (ROUNDS * (STEPS + 2) * 2 * (len(HOLDERS) - 1)))} bucket points of a backward")
print("holder (CUs x us) | chain forms ON: ms/iter (median, min) | chain forms OFF: ms/iter (median, min) | ON - OFF")
for h in HOLDERS:
    a, b = res[(True, h)], res[(False, h)]
    print(f"{'none' if h is None else f'{h[0]} x {h[1]}':>17} | {statistics.median(a):7.2f} {min(a):7.2f} | {statistics.median(b):7.2f} {min(b):7.2f} | {statistics.median(a) - statistics.median(b):+.2f}")
st_ = L.chain_stats()
print("chain stats:", st_)
