"""Data-parallel exactness of the GAN iteration on the GPU: 2 ranks x B images == 1 process x 2B images.

The reference is single-process (esrgan.py:173); its losses couple the images of a batch through batch statistics
(esrgan.py:487 batch-mean image, :507-508 / :578-579 relativistic batch means, :623 the d_threshold gate).  The build's
data-parallel path exchanges those statistics (train._AllReduceMean) and averages the weight gradients (engine buckets,
issued inside backward with async all-reduce; Stepper._sync_grads for the discriminators), and must therefore reproduce the
single-process iteration on the concatenated batch.  Here two rank processes share the one GPU of the test box through
torch.distributed's gloo backend on CUDA tensors (RCCL refuses two ranks on one device; the collectives' semantics are the
same), running the real engine.backward / Stepper code with _sync on; the parent runs the single-process iteration.
"""
import importlib
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import esrgan_oracle as O  # noqa: E402  (checker only: closed-form weights and jet images)

pytestmark = pytest.mark.gpu

B_RANK, HR, FACTOR, R, F = 3, 32, 2, 2, 16
G_KEYS = ("conv1.weight", "res_blocks.0.dense_blocks.1.b3.0.weight", "res_blocks.1.dense_blocks.2.b5.0.weight", "conv2.bias",
          "upsampling.0.weight", "conv3.0.weight", "conv3.2.weight")


def _stepper(distributed):
    train = importlib.import_module("super-resolution_amd.train")
    st = train.Stepper(workload="gan", res_blocks=R, filters=F, device=torch.device("cuda", 0), hr=HR, factor=FACTOR, res_scale=0.1,
                       distributed=distributed, exact_dp=True)
    st.generator.load_state_dict(O.closed_form_fill({k: v.cpu() for k, v in st.generator.state_dict().items()}))
    for k, D in st.discriminators.items():
        D.load_state_dict(O.closed_form_fill({n: v.cpu() for n, v in D.state_dict().items()}, gain=2.0 + k))
    return st


def _inputs(world):
    lr, hr = O.jet_images(B_RANK * world, 1, HR, HR, 21, FACTOR)
    eps = torch.rand(2, B_RANK * world, 1, 1, 1, generator=torch.Generator().manual_seed(5))
    return lr, hr, eps


def _one_iteration(st, lr, hr, eps):
    """G phase + both D phases of one iteration without the optimizer steps: losses, gradients and the gate values."""
    out = {}
    loss_G, generated, gt, parts = st.g_phase_loss(lr, hr)
    loss_G.backward()                       # generator gradients: bucketed all-reduce inside engine.backward when distributed
    named = dict(st.generator.named_parameters())
    out["g_loss"] = loss_G.detach().cpu()
    out["g_grads"] = {k: named[k].grad.detach().cpu().clone() for k in G_KEYS}
    out["parts"] = {k: {n: v.cpu() for n, v in p.items()} for k, p in parts.items()}
    lr_gt = [lr, lr ** st.scaling_power]
    for k, D in st.discriminators.items():
        D.zero_grad()
        loss_D, gp = st.d_phase_loss(k, gt[k], generated[k].detach(), eps[k], cond=lr_gt[k])
        loss_D.backward()
        st._sync_grads(D)
        out[f"d_loss{k}"] = loss_D.detach().cpu()
        out[f"d_grads{k}"] = {n: q.grad.detach().cpu().clone() for n, q in D.named_parameters()}
    return out


def _worker(rank, world, port, outdir, schedule):
    import torch.distributed as dist
    os.environ["SRK_DP_SCHEDULE"] = schedule          # read when the Stepper / engine are built
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        st = _stepper(True)
        assert st.generator._engine._sync and st.exact_dp
        # "overlap" (default since round 3): weight gradients on the side stream with the bucket all-reduces issued from the issue
        # stream, the two discriminators on two streams, the D phase beside the generator's backward -- all under data parallelism
        assert st.generator._engine.overlap_wgrad == (schedule == "overlap") and (st._d_streams is not None) == (schedule == "overlap")
        lr, hr, eps = _inputs(world)
        sl = slice(rank * B_RANK, (rank + 1) * B_RANK)
        res = _one_iteration(st, lr[sl].cuda(), hr[sl].cuda(), eps[:, sl].cuda())
        # then one full gan_step (optimizer steps, gate exchange, NaN probe) on the same shard
        step = st.gan_step(lr[sl].cuda(), hr[sl].cuda(), epsilons=eps[:, sl].cuda())
        res["gate"] = {k: v.cpu() for k, v in st.last_gate.items()}
        res["nan_probe"] = step["nan_probe"].cpu()
        res["conv1_after"] = st.generator.conv1.weight.detach().cpu().clone()
        res["d0_after"] = st.discriminators[0].model[0].weight.detach().cpu().clone()
        # a NaN on rank 1 only must reach rank 0's probe
        bad = torch.full_like(lr[sl], float("nan")) if rank == 1 else lr[sl]
        step = st.gan_step(bad.cuda(), hr[sl].cuda(), epsilons=eps[:, sl].cuda())
        res["nan_probe_poisoned"] = step["nan_probe"].cpu()
        torch.save(res, os.path.join(outdir, f"rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def _rel(a, b):
    # references that are analytically zero (the final D bias: the relativistic loss is invariant to it) -> absolute tolerance
    d = (a - b).abs().max()
    return 0.0 if d < 1e-6 else (d / b.abs().max().clamp_min(1e-4)).item()


@pytest.mark.parametrize("schedule", ["overlap", "serial"])
def test_two_ranks_equal_one_process_on_the_whole_batch(tmp_path, schedule):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    port = 23000 + (os.getpid() % 4000) + (1 if schedule == "serial" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), schedule)) for r in range(world)]
    for p in procs:
        p.start()
    # single process on the whole batch, meanwhile
    st = _stepper(False)
    lr, hr, eps = _inputs(world)
    ref = _one_iteration(st, lr.cuda(), hr.cuda(), eps.cuda())
    step = st.gan_step(lr.cuda(), hr.cuda(), epsilons=eps.cuda())
    ref_gate = {k: v.cpu() for k, v in st.last_gate.items()}
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
    ranks = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt")) for r in range(world)]

    # losses: the single-process loss is the mean over ranks of the rank losses (every batch statistic was exchanged)
    g_mean = sum(r["g_loss"] for r in ranks) / world
    assert _rel(g_mean, ref["g_loss"]) < 2e-5
    for k in (0, 1):
        d_mean = sum(r[f"d_loss{k}"] for r in ranks) / world
        assert _rel(d_mean.reshape(1), ref[f"d_loss{k}"].reshape(1)) < 2e-5
        # the batch-mean pixel term is a global statistic: identical on every rank and equal to the single-process value
        for r in ranks:
            assert _rel(r["parts"][k]["pixel"].reshape(1), ref["parts"][k]["pixel"].reshape(1)) < 2e-5
    # gradients: identical on both ranks (they were all-reduced) and equal to the single-process gradients
    for key in G_KEYS:
        assert torch.equal(ranks[0]["g_grads"][key], ranks[1]["g_grads"][key]), key
        assert _rel(ranks[0]["g_grads"][key], ref["g_grads"][key]) < 5e-4, key
    for k in (0, 1):
        for n, g in ref[f"d_grads{k}"].items():
            assert torch.equal(ranks[0][f"d_grads{k}"][n], ranks[1][f"d_grads{k}"][n]), (k, n)
            assert _rel(ranks[0][f"d_grads{k}"][n], g) < 5e-4, (k, n)
    # the d_threshold gate sees the same value on every rank = the single-process loss_D; replicas stay identical after the step
    for k in (0, 1):
        assert torch.equal(ranks[0]["gate"][k], ranks[1]["gate"][k])
        assert _rel(ranks[0]["gate"][k], ref_gate[k]) < 2e-5
    assert torch.equal(ranks[0]["conv1_after"], ranks[1]["conv1_after"]) and torch.equal(ranks[0]["d0_after"], ranks[1]["d0_after"])
    upd_ref = st.generator.conv1.weight.detach().cpu() - O.closed_form_fill({"conv1.weight": st.generator.conv1.weight.detach().cpu()})["conv1.weight"]
    assert torch.isfinite(ranks[0]["nan_probe"]).all() and torch.isfinite(ranks[1]["nan_probe"]).all() and upd_ref.abs().max() > 0
    # NaN guard (esrgan.py:645-648) on an all-reduced flag: rank 1's NaN input makes the probe NaN on rank 0 too
    assert torch.isnan(ranks[0]["nan_probe_poisoned"]).all() and torch.isnan(ranks[1]["nan_probe_poisoned"]).all()


# ---------------------------------------------------------------- BASELINE configs[4] under data parallelism (fp16 storage, loss scaling)
C4_KEYS = ("conv1.weight", "res_blocks.0.dense_blocks.1.b3.0.weight", "res_blocks.0.dense_blocks.2.b5.0.bias", "upsampling.0.weight", "conv3.2.weight")


def _c4_stepper(distributed):
    train = importlib.import_module("super-resolution_amd.train")
    st = train.Stepper(workload="g_only", res_blocks=1, filters=64, device=torch.device("cuda", 0), hr=64, factor=2, res_scale=0.2, channels=3,
                       distributed=distributed)
    st.generator._engine.precision = "fp16"
    st.generator.load_state_dict(O.closed_form_fill({k: v.cpu() for k, v in st.generator.state_dict().items()}))
    return st


def _c4_inputs(world):
    g = torch.Generator().manual_seed(11)
    hr = torch.rand(2 * world, 3, 64, 64, generator=g)
    return torch.nn.functional.avg_pool2d(hr, 2), hr


def _c4_run(st, lr, hr):
    res = {}
    out = st.step(lr.cuda(), hr.cuda())                       # warm-up iteration: forward, scaled backward (+ all-reduce), unscale, Adam
    named = dict(st.generator.named_parameters())
    res["loss"] = out["g_loss"].detach().cpu()
    res["scale"] = st._grad_scaler.get_scale()
    # p.grad still holds the (loss-scaled) gradients the optimizer consumed
    res["grads"] = {k: (named[k].grad.detach() / res["scale"]).cpu().clone() for k in C4_KEYS}
    out = st.step(lr.cuda(), hr.cuda())
    res["loss2"] = out["g_loss"].detach().cpu()
    res["after"] = {k: named[k].detach().cpu().clone() for k in C4_KEYS}
    return res


def _c4_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        st = _c4_stepper(True)
        assert st.generator._engine._sync
        lr, hr = _c4_inputs(world)
        sl = slice(2 * rank, 2 * rank + 2)
        torch.save(_c4_run(st, lr[sl], hr[sl]), os.path.join(outdir, f"c4rank{rank}.pt"))
    finally:
        dist.destroy_process_group()


def test_fp16_storage_warmup_two_ranks_equal_one_process(tmp_path):
    """configs[4]'s path (16-bit activation storage, dynamic loss scaling) as two data-parallel ranks against one process on the whole
    batch: the all-reduced (still loss-scaled) gradients are identical on both ranks, the unscaled gradients and the weights after two
    Adam steps agree with the single process within the storage format's rounding, and both ranks keep the same loss scale."""
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    port = 27000 + (os.getpid() % 3000)
    procs = [ctx.Process(target=_c4_worker, args=(r, world, port, str(tmp_path))) for r in range(world)]
    for p in procs:
        p.start()
    st = _c4_stepper(False)
    lr, hr = _c4_inputs(world)
    ref = _c4_run(st, lr, hr)
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, f"rank process exit code {p.exitcode}"
    ranks = [torch.load(os.path.join(str(tmp_path), f"c4rank{r}.pt")) for r in range(world)]
    assert ranks[0]["scale"] == ranks[1]["scale"] == ref["scale"] == 65536.0
    assert _rel((ranks[0]["loss"] + ranks[1]["loss"]).reshape(1) / 2, ref["loss"].reshape(1)) < 1e-3
    assert _rel((ranks[0]["loss2"] + ranks[1]["loss2"]).reshape(1) / 2, ref["loss2"].reshape(1)) < 1e-3
    for k in C4_KEYS:
        assert torch.equal(ranks[0]["grads"][k], ranks[1]["grads"][k]), k
        assert torch.equal(ranks[0]["after"][k], ranks[1]["after"][k]), k
        # two half batches in fp16 storage vs the whole batch: different rounding of the 16-bit gradient buffers, same gradient
        assert _rel(ranks[0]["grads"][k], ref["grads"][k]) < 2e-2, k
        upd, upd_ref = ranks[0]["after"][k] - O_fill(st, k), ref["after"][k] - O_fill(st, k)
        assert ((upd - upd_ref).abs().mean() / upd_ref.abs().mean().clamp_min(1e-12)).item() < 0.1, k


def O_fill(st, key):
    """the closed-form initial value of a generator tensor (what both runs started from)"""
    sd = O.closed_form_fill({k: v.cpu() for k, v in st.generator.state_dict().items()})
    return sd[key]
