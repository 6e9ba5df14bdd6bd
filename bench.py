#!/usr/bin/env python3
"""Benchmark of the MI355X ESRGAN training hot path.

    python bench.py --gpus N --steps K --warmup W [--workload g_only|gan] [--batch B]

A "step" is one pass of the hot path over one synthetic batch of jet images (64x64 LR -> 256x256 HR,
GeneratorRRDB(1, 64, 23, num_upsample=2)):
  gan    : (default; BASELINE.json's metric "G+D step") full iteration, esrgan.py:457-626: G phase through two
           patch discriminators + D phase with relativistic BCE and gradient penalty, three Adam steps, batch 32/GPU
           (BASELINE configs[2]; configs[3] = the same on 8 GPUs)
  g_only : the reference's warm-up iteration (esrgan.py:416-439): G forward, L1, backward, Adam, batch 16/GPU
           (BASELINE configs[1])
  c4     : BASELINE configs[4]: the same warm-up iteration on GeneratorRRDB(3, 64, 23, num_upsample=2), 3x128x128 -> 3x512x512
           photographic-style images, batch 8/GPU, fp16 path: activations and gradient buffers STORED in fp16, fp16 MFMA operands,
           fp32 accumulate, fp32 master weights, dynamic loss scaling (--precision bf16s: the same in bf16, no loss scaling).
The default run (gan, one GPU) appends short g_only and c4 measurements as "configs": {...} to its JSON line, so one record
carries all three single-GPU configurations of BASELINE.json.
Inputs are generated on the GPU before the timed region.  N > 1: ``python bench.py --gpus N`` starts its N ranks itself
(``torch.distributed.run`` as a child process, before anything touches the GPU); under an external
``python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`` it takes RANK / LOCAL_RANK / WORLD_SIZE as
given.  One process per GPU, weak scaling (fixed per-GPU batch), gradients averaged with RCCL all-reduce overlapped with the
backward pass.  Rank 0 prints ONE JSON line.
"""
import argparse
import importlib
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HR = 256
FACTOR = 4
F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA (the 5 PF headline figure includes 2:1 sparsity)
# per workload: image channels, HR extent, per-GPU batch, the Stepper's workload, default precision
WORKLOADS = {"gan": dict(channels=1, hr=256, batch=32, step="gan", precision="f32"),
             "g_only": dict(channels=1, hr=256, batch=16, step="g_only", precision="f32"),
             "c4": dict(channels=3, hr=512, batch=8, step="g_only", precision="fp16")}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("SRK_WORKLOAD", "gan"), choices=["g_only", "gan", "c4"],
                    help="gan = BASELINE.json's metric (full G+D iteration, configs[2]); g_only = warm-up iteration (configs[1]); "
                         "c4 = configs[4] (3-channel 128->512, bf16 MFMA path, batch 8)")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default 16 for g_only, 32 for gan)")
    ap.add_argument("--res-blocks", type=int, default=23)
    ap.add_argument("--precision", default=os.environ.get("SRK_PRECISION", ""), choices=["", "f32", "bf16x3", "bf16", "fp16", "bf16s"],
                    help="f32 (default for gan / g_only, the headline): exact-fp32 MFMA everywhere.  bf16x3: opt-in split-bf16 MFMA "
                         "mode.  fp16 (default for c4) / bf16s: 16-bit activation storage + MFMA operands, fp32 accumulate, fp32 master "
                         "weights (fp16: with loss scaling).  bf16: bf16 operands on fp32 storage (round 2's stand-in)")
    ap.add_argument("--no-configs", action="store_true", help="skip the g_only / c4 sub-records of the default run")
    ap.add_argument("--no-alt", action="store_true", help="skip the extra opt-in bf16x3 measurement and the full-size parity leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="do not bracket conv launches with events")
    return ap.parse_args()


def synth_batch(n, device, seed, channels=1, hr_px=HR):
    """SURVEY 8(d).  Jets (1 channel): hr = 10*U(0,1)*Bernoulli(0.1); lr = SumPool2d(4)(hr) (datasets.py:227,247).
    Photographic config (3 channels): hr = U(0,1), lr = avg_pool(4)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    if channels == 1:
        hr = 10.0 * torch.rand(n, 1, hr_px, hr_px, generator=g) * (torch.rand(n, 1, hr_px, hr_px, generator=g) < 0.1).float()
        hr = hr.to(device)
        lr = 16.0 * torch.nn.functional.avg_pool2d(hr, FACTOR)
    else:
        hr = torch.rand(n, channels, hr_px, hr_px, generator=g).to(device)
        lr = torch.nn.functional.avg_pool2d(hr, FACTOR)
    return lr.contiguous(), hr.contiguous()


def host_cores():
    """CPU cores this process may really use: affinity mask capped by the cgroup CPU quota (the GPU box exposes
    256 logical CPUs but a 16-CPU quota; oversubscribing it makes the baseline ~10x slower than it is)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def cpu_baseline(res_blocks, workload):
    """Times the CPU oracle (a port of the reference's step, oracle/esrgan_oracle.py) on this box's host cores
    on a bounded sample (batch 4 jets / batch 1 photographic image, 1 warm-up + up to 4 timed iterations, ~10-30 s); baseline only."""
    from oracle import esrgan_oracle as O
    wl = WORKLOADS[workload]
    ch, hr_px, step = wl["channels"], wl["hr"], wl["step"]
    n = 4 if ch == 1 else 1
    cores = host_cores()
    torch.set_num_threads(cores)
    sd = O.default_init_generator(0, channels=ch, filters=64, num_res_blocks=res_blocks, num_upsample=2)
    params = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    opt = torch.optim.Adam([p for p in params.values() if p.requires_grad], lr=2e-4, betas=(0.9, 0.999))
    if ch == 1:
        lr, hr = O.jet_images(n, 1, hr_px, hr_px, 1234, FACTOR)
    else:
        lr, hr = [t.cpu() for t in synth_batch(n, "cpu", 1234, ch, hr_px)]
    dsd, optD = [], []
    if step == "gan":
        g = torch.Generator().manual_seed(1)
        for k in range(2):
            d = {}
            for name, shp in O.discriminator_state_shapes().items():
                fan = shp[1] * 9 if len(shp) == 4 else 16 * 9
                d[name] = ((torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan)).requires_grad_(True)
            dsd.append(d)
            optD.append(torch.optim.Adam(list(d.values()), lr=2e-4, betas=(0.9, 0.999)))
    times = []
    for it in range(5):
        t0 = time.perf_counter()
        opt.zero_grad()
        y, srs = O.generator_forward(params, lr, res_blocks, 2, 0.2, training=True)
        if step == "gan":                             # esrgan.py:457-626 restated (oracle/esrgan_oracle.py)
            lG, _ = O.g_phase_loss([y, srs], hr, lr, dsd, FACTOR)
            lG.backward()
            opt.step()
            for k in range(2):
                optD[k].zero_grad()
                lD, _ = O.d_phase_loss(dsd[k], hr, [y, srs][k].detach(), torch.rand(n, 1, 1, 1), 0.01)
                lD.backward()
                if lD.item() > 0.001:
                    optD[k].step()
        else:
            O.warmup_loss(y, hr).backward()
            opt.step()
        times.append(time.perf_counter() - t0)
        print(f"[cpu_baseline] iter {it}: {times[-1]:.1f} s on {cores} threads", file=sys.stderr, flush=True)
        if it >= 1 and sum(times) > 25.0:
            break
    timed = times[1:]
    dt = sum(timed) / len(timed)
    return {"value": n * hr_px * hr_px / dt, "unit": "HR-px/s", "cores": cores, "kind": "port",
            "sample": ("full G+D iteration (two patch discriminators, relativistic BCE, gradient penalty, 3x Adam)" if step == "gan"
                       else "G-only step (fwd+L1+bwd+Adam)") +
                      f" of the same generator, batch {n} of {ch}x{hr_px // FACTOR}x{hr_px // FACTOR}, 1 warm-up + {len(timed)} timed iters, "
                      f"{dt*1e3:.0f} ms/iter, torch CPU fp32"}


def full_size_parity(sr, res_blocks, dev, workload):
    """Generator forward at the workload's full architecture (F=64, R=23, 4x) on 2 jet images (1 photographic image for c4),
    HIP path vs the CPU fp32 oracle on identical weights/inputs: max |diff| / max |ref| per precision mode, with the tolerance
    each mode is held to (f32 / bf16x3: BASELINE's 1e-3; 16-bit activation storage: fp16 4e-3, bf16 3e-2 -- one rounding per stored
    activation along 351 convolutions, the mixed-precision bar of configs[4])."""
    from oracle import esrgan_oracle as O
    wl = WORKLOADS[workload]
    ch, hr_px = wl["channels"], wl["hr"]
    sd = O.default_init_generator(0, channels=ch, filters=64, num_res_blocks=res_blocks, num_upsample=2)
    if ch == 1:
        lr, hr = O.jet_images(2, 1, hr_px, hr_px, 4321, FACTOR)
    else:
        lr, hr = synth_batch(1, "cpu", 4321, ch, hr_px)
    with torch.no_grad():
        ref, _ = O.generator_forward(sd, lr, res_blocks, 2, 0.2, training=True)
    gen = sr.GeneratorRRDB(ch, filters=64, num_res_blocks=res_blocks, num_upsample=2).to(dev)
    gen.load_state_dict(sd)
    out = {}
    modes = ("f32", "bf16x3") if wl["precision"] == "f32" else ("f32", "fp16", "bf16s")
    for mode in modes:
        gen._engine.precision = mode
        with torch.no_grad():
            y = gen(lr.to(dev)).cpu()
        out[mode] = float((y - ref).abs().max() / ref.abs().max())
    out["tolerance"] = {"f32": 1e-3, "bf16x3": 1e-3, "bf16": 3e-2, "fp16": 4e-3, "bf16s": 3e-2}
    out["ok"] = all(out[m] < out["tolerance"][m] for m in modes)
    out["what"] = ("GeneratorRRDB(%d,64,%d,num_upsample=2) forward, %s -> %s, max-abs relative error vs the CPU fp32 oracle"
                   % (ch, res_blocks, "x".join(map(str, lr.shape)), "x".join(map(str, ref.shape))))
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: run `torch.distributed.run` with N ranks on this node as a child
    process (never an exec: this process stays a plain parent that has not initialised the GPU) and return its exit code."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as plain `python bench.py --gpus N`: start the N ranks ourselves, BEFORE anything touches the GPU
        # (importing torch does not), as children of this process; rank 0's JSON line passes through on stdout.
        raise SystemExit(self_launch(args.gpus))
    backend = os.environ.get("SRK_DIST_BACKEND", "nccl")     # "gloo": rehearsal of N ranks sharing one GPU (RCCL refuses that)
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
        if world > torch.cuda.device_count():
            # several ranks on ONE GPU: a chain kernel needs every CU of the device for itself (csrc/srk_chain.h keeps one in flight per
            # PROCESS); two processes' chain kernels would each hold part of the CUs and run into the bounded wait
            os.environ.setdefault("SRK_W42_CHAIN", "0")
            os.environ.setdefault("SRK_H16_CHAIN", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    sr = importlib.import_module("super-resolution_amd")
    from importlib import import_module
    train = import_module("super-resolution_amd.train")
    L = sr._lib

    dist = None
    comm_info = None               # (N > 1: backend / world size as the communicator reports them, in the JSON line as "comm")
    force_dist = os.environ.get("SRK_FORCE_DIST", "0") == "1"      # rehearse the N>1 code path with a 1-rank RCCL group
    if world > 1 or force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        # RCCL prints a version banner on STDOUT when its communicator comes up (measured: 5 lines in front of the JSON line of a
        # 1-rank rehearsal); route C-level stdout to stderr while the group and its first collective initialise
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
            warm = torch.ones(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
            # what the communicator itself says: after a SUM of ones every rank holds the number of ranks that took part
            comm_info = {"backend": dist.get_backend() + (" (RCCL)" if backend == "nccl" else ""), "world_size": dist.get_world_size(),
                         "ranks_counted_by_all_reduce": int(warm.item()),
                         "devices_visible": torch.cuda.device_count(), "rank0_device": torch.cuda.get_device_name(dev)}
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    wl = WORKLOADS[args.workload]
    precision = args.precision or wl["precision"]
    hr_px, channels = wl["hr"], wl["channels"]
    batch = args.batch or wl["batch"]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(workload, precision, batch, steps, warmup, probe):
        """W untimed + K timed steps of one workload; returns (seconds for the K steps [max over ranks], per-kernel event times of the
        first timed step or None, the stepper, the batch)."""
        w = WORKLOADS[workload]
        torch.manual_seed(0)                                  # identical replicas on every rank
        stepper = train.Stepper(workload=w["step"], res_blocks=args.res_blocks, device=dev, hr=w["hr"], factor=FACTOR, channels=w["channels"],
                                distributed=(world > 1 or force_dist))
        stepper.generator._engine.precision = precision
        lr_img, hr_img = synth_batch(batch, dev, 1234 + rank, w["channels"], w["hr"])
        for _ in range(warmup):
            stepper.step(lr_img, hr_img)
        barrier()
        # Per-launch HIP events bracket every conv / wgrad launch during the FIRST timed step only: bracketing all of them
        # costs ~4.5 % of the step (measured 132.4 vs 126.5 ms), one step in K keeps the perturbation of `value` below
        # 0.5 % while still timing >1000 launches live inside the timed region.
        timed_probe = 1 if probe else 0
        t0 = time.perf_counter()
        for it in range(steps):
            if it < timed_probe and it == 0:
                L.KernelTimer.start()
            stepper.step(lr_img, hr_img)
            if it + 1 == timed_probe:
                L.KernelTimer.active = False           # stop recording; elapsed times are read after the region
        barrier()
        dt = time.perf_counter() - t0
        ktimes = L.KernelTimer.stop() if probe else None
        tmax = torch.tensor([dt], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        return tmax.item(), ktimes, stepper, (lr_img, hr_img)

    dt, ktimes, stepper, (lr_img, hr_img) = measure(args.workload, precision, batch, args.steps, args.warmup, not args.no_kernel_timing)

    # ---- extra, outside the headline: the opt-in split-bf16 mode on the same workload (every rank takes part)
    alt = None
    if not args.no_alt and precision == "f32":
        stepper.generator._engine.precision = "bf16x3"
        for _ in range(2):
            stepper.step(lr_img, hr_img)
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            stepper.step(lr_img, hr_img)
        barrier()
        adt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(adt, op=dist.ReduceOp.MAX)
        adt = adt.item()
        stepper.generator._engine.precision = "f32"
        alt = {"precision": "bf16x3: fwd/dgrad/wgrad convs as 3 bf16 MFMAs per product (operands split hi+lo, fp32 accumulate); "
                            "opt-in, NOT the headline", "value": world * batch * hr_px * hr_px * args.steps / adt, "unit": "HR-px/s",
               "ms_per_step": adt / args.steps * 1e3}
    # ... and the 16-bit-storage mode on the same workload (generator activations / gradients in fp16 with loss scaling, fp16 MFMAs,
    # fp32 accumulate / master weights / discriminators): opt-in mixed precision, NOT the headline
    if not args.no_alt and precision == "f32":
        stepper.generator._engine.precision = "fp16"
        for _ in range(3):
            stepper.step(lr_img, hr_img)
        barrier()
        t1 = time.perf_counter()
        hsteps = max(3, min(args.steps, 10))
        for _ in range(hsteps):
            stepper.step(lr_img, hr_img)
        barrier()
        hdt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(hdt, op=dist.ReduceOp.MAX)
        hdt = hdt.item()
        stepper.generator._engine.precision = "f32"
        alt["fp16_storage_mode"] = {"precision": "fp16: generator activations and gradient buffers stored in fp16, fp16 MFMA operands, fp32 accumulate, "
                                                 "fp32 master weights, dynamic loss scaling; discriminators fp32; opt-in, NOT the headline",
                                    "value": world * batch * hr_px * hr_px * hsteps / hdt, "unit": "HR-px/s", "ms_per_step": hdt / hsteps * 1e3}
    ms = dt / args.steps * 1e3
    value = world * batch * hr_px * hr_px * args.steps / dt
    del stepper, lr_img, hr_img

    def roofline_of(ktimes, step_ms):
        if not ktimes:
            return None
        dom = max(ktimes.items(), key=lambda kv: kv[1]["ms"])
        name, st = dom
        ach = st["flops"] / (st["ms"] * 1e-3) / 1e12
        peak_tflops = BF16_MFMA_PEAK_TFLOPS if ("bf16" in name or "h16" in name) else F32_MFMA_PEAK_TFLOPS
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:   # HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this
                # same command (tools/traffic_from_pmc.py; FETCH_SIZE x2 on gfx950), not collectable in-process
                tj = json.load(open(tpath))          # (rocprofv3 does not demangle the _Float16 kernels: those are keyed by base name)
                traffic = (tj.get(name) or tj.get(name.split("<")[0]) or {}).get("hbm_bytes_per_launch")
                if traffic is not None:
                    traffic_source = ("profiles/traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                      "command on an earlier box (PMC counters cannot be read in-process); not measured in this run")
            except Exception:
                traffic = None
        # `frac` is the fraction of the matrix pipe's peak that the kernel SUSTAINS: executed multiply-adds / time / peak.
        # The Winograd kernels execute fewer multiply-adds than the convolution's algorithmic count -- F(2,3) along W:
        # 4 instead of 6 per output pair (2/3); F(4,3): 6 instead of 12 per output quad (1/2) -- so the algorithmic-equivalent
        # rate (what a direct kernel would need to match the time) is reported beside it and may exceed the peak.
        # F(2x4,3x3) (wino42): 24 instead of 72 per 2x4 output patch (1/3)
        if "wino42" in name:
            fac, what = 1.0 / 3.0, "2-D Winograd F(2x4, 3x3): F(4,3) along W times F(2,3) along H"
        elif "wino24" in name:
            fac, what = 1.0 / 3.0, "2-D transposed Winograd F(2,3) along H x F(4,3) along W"
        elif "wino22" in name:
            fac, what = 4.0 / 9.0, "2-D transposed Winograd F(2,3) x F(2,3)"
        else:
            fac, what = (0.5, "Winograd F(4,3) along W") if "wino4" in name else ((2.0 / 3.0, "Winograd F(2,3) along W") if "wino" in name else (1.0, "direct"))
        if "bf16x3" in name:      # template argument TERMS: 3 bf16 MFMAs per product (split operands) or 1 (plain bf16 operands)
            terms = 3 if name.rstrip(">+reduce").rstrip(">").endswith("3") else 1
            fac, what = float(terms), ("direct, %d bf16 MFMA%s per product, fp32 accumulate" % (terms, "s" if terms > 1 else ""))
        if "h16" in name:
            what = "direct, 16-bit activation storage, %s MFMA operands, fp32 accumulate" % ("fp16" if "_Float16" in name else "bf16")
        return {"bound": "mfma", "kernel": name, "achieved": round(ach * fac, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                "frac": round(ach * fac / peak_tflops, 4), "traffic": traffic, "traffic_source": traffic_source,
                "launches": st["n"], "avg_us": round(st["ms"] * 1e3 / st["n"], 2),
                "avg_gflop_per_launch": round(st["flops"] / st["n"] / 1e9, 3),
                "algorithm": what, "executed_over_algorithmic": round(fac, 4),
                "algorithmic_tflops": round(ach, 2), "algorithmic_over_peak": round(ach / peak_tflops, 4),
                "probe": "HIP events around every conv/wgrad launch of the first timed step, on the launching stream",
                # Sum of the bracketed launches of the PROBED step over the AVERAGE step: the probed step runs without stream overlap and
                # with event overhead, so this can exceed 1 -- it is not the conv share of an average step
                "probed_step_conv_ms_over_avg_step_ms": round(sum(v["ms"] for v in ktimes.values()) / step_ms, 4),
                "by_kernel": {k: {"launches": v["n"], "ms": round(v["ms"], 3),
                                  "algorithmic_tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)}
                              for k, v in sorted(ktimes.items(), key=lambda kv: -kv[1]["ms"])[:(40 if L.KernelTimer.detail else 6)]}}

    DTYPES = {"f32": "f32", "bf16x3": "bf16x3 (split-bf16 MFMA operands, fp32 accumulate)",
              "bf16": "bf16 (MFMA operands; fp32 accumulate, fp32 master weights and fp32 activations in HBM)",
              "fp16": "fp16 (activations and gradient buffers stored in fp16, fp16 MFMA operands; fp32 accumulate, fp32 master weights, "
                      "dynamic loss scaling)",
              "bf16s": "bf16 (activations and gradient buffers stored in bf16, bf16 MFMA operands; fp32 accumulate, fp32 master weights)"}

    def config_of(workload, batch):
        w = WORKLOADS[workload]
        ch, hp = w["channels"], w["hr"]
        return {"workload": f"{workload}: GeneratorRRDB({ch},64,{args.res_blocks},num_upsample=2) "
                            f"{hp // FACTOR}x{hp // FACTOR}->{hp}x{hp}, "
                            f"batch {batch}/GPU" + (", 2x Markovian_Discriminator[16,32,32,64], relativistic BCE + GP"
                                                    if w["step"] == "gan" else ", L1 + Adam"),
                "global_batch": batch * world, "per_gpu_batch": batch, "parallelism": f"dp{world}",
                "generator_train_gflop_per_image": 971.4 if ch == 1 else 3887.6,
                "inputs": "the same synthetic batch every step (resident in HBM); tools/soak.py with a fresh batch per "
                          "iteration measures the same ms/iter, and the d_threshold gate stays open on it"}

    STEP_NAMES = {"gan": "G+D step", "g_only": "G-only warm-up step", "c4": "G-only warm-up step, configs[4]"}

    # ---- the other single-GPU configurations of BASELINE.json beside the headline (default run only): short measurements AFTER the
    # headline's timed region, each with its own roofline leg, so that one driver-run record carries configs[1], [2] and [4]
    subs = None
    if args.workload == "gan" and world == 1 and not args.no_configs and not args.precision and not args.batch:
        subs = {}
        for sub in ("g_only", "c4"):
            w = WORKLOADS[sub]
            ssteps = max(3, min(args.steps, 8))
            sdt, skt, sst, sbatch = measure(sub, w["precision"], w["batch"], ssteps, 3, not args.no_kernel_timing)
            del sst, sbatch          # (the allocator keeps its blocks: handing them back makes the next workload pay hipMalloc again)
            sms = sdt / ssteps * 1e3
            subs[sub] = {"metric": f"HR-pixels/s + ms/iter ({STEP_NAMES[sub]})", "value": w["batch"] * w["hr"] * w["hr"] * ssteps / sdt,
                         "unit": "HR-px/s", "ms_per_step": sms, "steps": ssteps, "warmup": 3, "dtype": DTYPES[w["precision"]],
                         "config": config_of(sub, w["batch"]), "roofline": roofline_of(skt, sms)}

    if rank == 0:
        roof = roofline_of(ktimes, ms)
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(args.res_blocks, args.workload)
        parity = None
        if not args.no_alt and world == 1:
            parity = full_size_parity(sr, args.res_blocks, dev, args.workload)
            if subs is not None:
                subs["c4"]["full_size_parity"] = full_size_parity(sr, args.res_blocks, dev, "c4")
        img = "64->256 jet images" if channels == 1 else "3-channel 128->512 images"
        out = {
            "metric": f"HR-pixels/s + ms/iter ({STEP_NAMES[args.workload]}), {img}",
            "value": value, "unit": "HR-px/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPES[precision], "data": "synthetic",
            "config": config_of(args.workload, batch),
            "roofline": roof, "cpu_baseline": cpu, "split_bf16_mode": alt, "full_size_parity": parity,
        }
        if subs is not None:
            out["configs"] = subs
        if comm_info is not None:
            out["comm"] = comm_info
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
