"""Training steps of the reference's hot loop (esrgan.py:399-626) on the MI355X modules.

``Stepper`` owns the replicas (generator, discriminators, Adam states) of ONE rank and runs one iteration:
  * ``g_only``: the warm-up iteration, esrgan.py:416-439 -- ``L1(G(lr), hr)``, backward, ``optimizer_G.step()``
  * ``gan``   : G phase (esrgan.py:457-555) + D phase (esrgan.py:561-626) at the reference's default flags
                (relativistic average BCE, hr/lr pixel terms, both "views" def/pow -> two discriminators,
                gradient penalty lambda_reg, d_threshold gate)
Data parallelism is one process per GPU; weight gradients are averaged with RCCL all-reduce
(``torch.distributed`` backend "nccl"): the generator's are bucketed per RRDB and launched while the backward
convolutions are still running (engine.py), the discriminators' (0.36 MB each) go as one flat buffer.
With ``exact_dp`` the batch-coupled statistics of the loss (relativistic means, batch-mean
image of the pixel loss, the d_threshold decision) are exchanged too, so N ranks x B images reproduce the
single-process step on N*B images.
"""
import math
import os

import contextlib
import torch
import torch.nn as nn

from . import _lib as L
from . import losses, models, ops, optim

EPS = 1e-7   # esrgan.py:319


_STAT_GROUP = None


def _stat_group():
    """A process group of its own (all ranks) for the small exchanges: batch statistics of the losses, discriminator gradients, the
    d_threshold gate.  torch.distributed runs the collectives of ONE group in host call order on that group's stream: on the default
    group a statistic exchange of the D phase, which runs beside the generator's backward on the discriminator streams, would queue
    behind every gradient bucket of that backward -- i.e. the D phase would wait for the end of the backward it is meant to overlap.
    Same host order of the calls on every rank for this group too, so nothing new can deadlock."""
    global _STAT_GROUP
    import torch.distributed as dist
    world = dist.distributed_c10d._get_default_group()
    if _STAT_GROUP is None or _STAT_GROUP[0] is not world:        # (a new default group after destroy / init: a new statistics group)
        _STAT_GROUP = (world, dist.new_group())
    return _STAT_GROUP[1]


class _AllReduceMean(torch.autograd.Function):
    """y = mean over ranks of x.  d(loss_total)/dx = mean over ranks of dy (loss_total = mean of rank losses)."""

    @staticmethod
    def forward(ctx, x):
        import torch.distributed as dist
        y = x.clone()
        dist.all_reduce(y, op=dist.ReduceOp.SUM, group=_stat_group())
        return y / dist.get_world_size()

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g.clone()
        dist.all_reduce(g, op=dist.ReduceOp.SUM, group=_stat_group())
        return g / dist.get_world_size()


def _ranks_share_a_gpu(device) -> bool:
    """True if two ranks of the job run on the same physical GPU (same host, same device UUID / PCI address)."""
    import socket
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or device is None or torch.device(device).type != "cuda":
        return False
    pr = torch.cuda.get_device_properties(torch.device(device))
    ident = getattr(pr, "uuid", None)
    ident = str(ident) if ident is not None else "/".join(str(getattr(pr, k, "?")) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id"))
    mine = (socket.gethostname(), ident)
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine, group=_stat_group())
    return len(set(everyone)) < len(everyone)


def _uses(stream, *tensors):
    """Tell the caching allocator that ``tensors`` (allocated on another stream) are read or written by work queued on ``stream``:
    a block freed on its home stream is then not handed out again before that work has run.  Every tensor that crosses between the
    main stream and the discriminator streams goes through here, so the schedule does not rest on the joins alone."""
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(stream)


class Stepper:
    def __init__(self, workload="g_only", res_blocks=23, device=None, hr=256, factor=4, distributed=False, channels=1,
                 filters=64, res_scale=0.2, lr=2e-4, betas=(0.9, 0.999), d_channels=(16, 32, 32, 64), lambdas=(0.2, 1.0),
                 lambda_hr=1.0, lambda_adv=0.01, lambda_lr=0.1, lambda_reg=0.01, d_threshold=0.001, scaling_power=1.0,
                 exact_dp=True, hr_shape=None, lr_g=0.0, lr_d=0.0, weight_decay=0.0, multiplier=1.0, num_final_layer_res=0,
                 uniform_init=False, lambda_nnz=0.0, lambda_mask=0.0, lambda_hit=0.0, lambda_hist=0.0, hit_threshold=0.5, sigma=500.0, conditional=False, drop_rate=0.0, discriminator="patch", relativistic=True,
                 use_transposed_conv=False, fully_tconv_upsample=False):
        self.workload = workload
        self.drop_rate = drop_rate
        self.relativistic = relativistic
        self.conditional = conditional
        # optional physics heads of the G phase (esrgan.py:522-547); the histogram head needs set_hist_binedges() first
        self.lambda_nnz, self.lambda_mask, self.lambda_hit, self.lambda_hist = lambda_nnz, lambda_mask, lambda_hit, lambda_hist
        self.hit_threshold, self.sigma = hit_threshold, sigma
        self.histograms, self.criterion_hist = {}, {}
        self.device = device
        self.distributed = distributed
        self.exact_dp = exact_dp and distributed
        self.factor = factor
        self.lambdas = tuple(lambdas)
        self.lambda_hr, self.lambda_adv, self.lambda_lr, self.lambda_reg = lambda_hr, lambda_adv, lambda_lr, lambda_reg
        self.d_threshold = d_threshold
        self.scaling_power = scaling_power
        hr_shape = hr_shape or (hr, hr)
        self.generator = models.GeneratorRRDB(channels, filters=filters, num_res_blocks=res_blocks,
                                              num_upsample=int(math.log2(factor)), res_scale=res_scale,
                                              power=scaling_power, multiplier=multiplier, num_final_layer_res=num_final_layer_res,
                                              uniform_init=uniform_init, drop_rate=drop_rate, use_transposed_conv=use_transposed_conv,
                                              fully_tconv_upsample=fully_tconv_upsample).to(device)
        # esrgan.py:299,305: Adam(lr_g or lr), Adam(lr_d or lr); optim.Adam = torch.optim.Adam's arithmetic in ONE launch over a pointer
        # table (ATen's fused multi-tensor form: 20 launches, 1.5 ms for the generator's 702 tensors); SRK_TORCH_ADAM=1 restores ATen's
        Adam = (lambda ps, **kw: torch.optim.Adam(ps, fused=True, **kw)) if os.environ.get("SRK_TORCH_ADAM", "0") == "1" else optim.Adam
        self._Adam = Adam
        self.optimizer_G = Adam([p for p in self.generator.parameters() if p.requires_grad],
                                lr=lr_g if lr_g > 0 else lr, betas=betas, weight_decay=weight_decay)
        self.criterion_pixel = nn.L1Loss()
        self.criterion_GAN = nn.BCEWithLogitsLoss()
        self.mse = nn.MSELoss()
        self.pool = models.SumPool2d(factor)
        self.discriminators, self.optimizer_D = {}, {}
        if workload == "gan":
            for k in range(2):
                if self.lambdas[k] > 0:
                    if conditional:       # esrgan.py:213-219
                        D = models.Conditional_Discriminator(input_shape=(channels, *hr_shape), channels=list(d_channels),
                                                             num_upsample=int(math.log2(factor))).to(device)
                    elif discriminator == "standard":     # esrgan.py:205-207
                        D = models.Standard_Discriminator(input_shape=(channels, *hr_shape), channels=list(d_channels)).to(device)
                    else:
                        D = models.Markovian_Discriminator(input_shape=(channels, *hr_shape), channels=list(d_channels)).to(device)
                    self.discriminators[k] = D
                    self.optimizer_D[k] = self._Adam(D.parameters(), lr=lr_d if lr_d > 0 else lr, betas=betas)
        if distributed and not self.generator.modulewise:
            self.generator._engine.enable_grad_sync()      # (module-wise generators: gradients go through _sync_grads)
        if distributed:
            _stat_group()          # (created by every rank at the same point)
            # wait bound of the chain forms (include/srk.h): a collective's kernel holds CUs while a peer rank is late -- seconds in a
            # first iteration -- and a tile behind it has to wait that out; giving up after 50 ms would end the job (see step())
            if "SRK_CHAIN_WAIT_MS" not in os.environ and torch.device(device).type == "cuda":
                L.lib().srk_chain_set_wait_us(30_000_000)
            self.shared_gpu = _ranks_share_a_gpu(device)
            if self.shared_gpu:
                # The chain forms need the whole GPU: two ranks' 256-tile persistent launches would split the CUs and both give up at
                # their census, every time.  (Rehearsals of several ranks on one card: tests/test_dp_gpu.py, bench.py with gloo.)
                L.lib().srk_debug_set_h16_chain(0)
                L.lib().srk_debug_set_w42_chain(0)
        self.last = {}
        self._grad_scaler = None      # fp16 activation storage (engine.precision == "fp16"): dynamic loss scaling, created on first use
        # The two discriminators run on two streams (133.4 vs 135.3 ms per iteration) and the D phase beside the generator's backward
        # (-1.8 ms): SRK_D_STREAMS=0 / SRK_D_OVERLAP=0 switch them off.  Since round 3 also under data parallelism: the statistic
        # exchanges inside the D losses are then issued from the discriminator streams, which is legal for the same reason as the
        # bucket all-reduces of engine._reduce_bucket -- torch.distributed enqueues the collectives of a process group in HOST call
        # order (identical on every rank) on the group's own stream, whatever stream they were issued from.  SRK_DP_SCHEDULE=serial
        # keeps a distributed run on one stream (round 2's conservative schedule).
        self._d_streams = None
        self._d_overlap = os.environ.get("SRK_D_OVERLAP", "1") != "0"
        mode = os.environ.get("SRK_D_STREAMS", "1")
        dp_serial = distributed and os.environ.get("SRK_DP_SCHEDULE", "overlap") == "serial"
        if mode != "0" and (mode == "2" or not dp_serial) and torch.cuda.is_available() and len(self.discriminators) == 2:
            self._d_streams = {k: torch.cuda.Stream() for k in self.discriminators}

    # ------------------------------------------------------------------ helpers
    def set_hist_binedges(self, k, binedges):
        """esrgan.py:454-455: bin edges of the energy histogram of view k (0 = def, 1 = pow)."""
        self.histograms[k] = losses.DiffableHistogram(binedges, sigma=self.sigma).to(self.device)
        self.criterion_hist[k] = losses.KLD_hist(torch.as_tensor(binedges)).to(self.device)

    def _gmean(self, t):
        """Batch statistic that the reference takes over the whole batch: mean over ranks of the local value."""
        return _AllReduceMean.apply(t) if self.exact_dp else t

    def _sync_grads(self, module):
        if not self.distributed:
            return
        import torch.distributed as dist
        grads = [p.grad for p in module.parameters() if p.grad is not None]
        flat = torch.cat([g.reshape(-1) for g in grads])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_stat_group())
        flat /= dist.get_world_size()
        # one multi-tensor copy back (18 tensors per discriminator: 18 launches otherwise, on the path between two host syncs)
        torch._foreach_copy_(grads, [c.view_as(g) for c, g in zip(flat.split([g.numel() for g in grads]), grads)])

    def _scaler(self):
        """Dynamic loss scaling for the generator's backward when its activations / gradients are stored in fp16 (BASELINE configs[4]):
        the gradient buffers hold S x the true gradients (an L1 loss over 8 x 3 x 512 x 512 outputs starts at 1.6e-7 per element, below
        fp16's normal range), the fp32 weight gradients come out scaled and fused Adam divides by S; a step whose gradients
        overflowed is skipped and S halved (torch.amp.GradScaler: device-side, no host sync).  None in every other mode."""
        if getattr(self.generator, "modulewise", False) or self.generator._engine.precision != "fp16":
            return None
        if self._grad_scaler is None:
            self._grad_scaler = torch.amp.GradScaler("cuda", init_scale=2.0 ** 16)
        return self._grad_scaler

    def _backward_and_step_G(self, loss):
        sc = self._scaler()
        if sc is None:
            loss.backward()
            if self.generator.modulewise:
                self._sync_grads(self.generator)
            self.optimizer_G.step()
            return
        sc.scale(loss).backward()
        sc.step(self.optimizer_G)
        sc.update()

    def step(self, imgs_lr, imgs_hr):
        """One iteration.  A dense block's convolutions may go out as ONE persistent launch (the chain forms, include/srk.h), which needs
        every workgroup of that launch resident at once.  If one gave up -- a foreign process on the GPU, a kernel holding CUs beyond
        the wait bound -- its results are garbage in activation buffers, every optimizer step since has skipped itself on the device, and
        the first library call that notices raises ChainTimeout: recover (device-wide wait, fault cleared, chain forms rested) and run
        the iteration again, conv by conv.  What iterations in between returned as losses is undefined; weights and optimizer state
        are those of the last good iteration (a discriminator whose step ran beside the failing launch may see this batch twice)."""
        for attempt in range(3):
            try:
                if self.workload == "g_only":
                    return self.warmup_step(imgs_lr, imgs_hr)
                return self.gan_step(imgs_lr, imgs_hr)
            except L.ChainTimeout:
                # data parallel: the other ranks are inside this iteration's collectives; a rank that started over on its own would
                # pair its all-reduces with the wrong ones.  There the wait bound is 30 s (below), so this is a fault, not a busy GPU.
                if attempt == 2 or self.distributed:
                    raise
                self.recover_chain_fault()
        raise AssertionError("unreachable")

    def recover_chain_fault(self):
        """After ChainTimeout: wait for everything queued, clear the fault, drop the half-built state of the interrupted iteration."""
        torch.cuda.synchronize()
        code = L.chain_recover()
        self.chain_recoveries = getattr(self, "chain_recoveries", 0) + 1
        self.optimizer_G.zero_grad(set_to_none=True)
        for opt in self.optimizer_D.values():
            opt.zero_grad(set_to_none=True)
        eng = getattr(self.generator, "_engine", None)
        if eng is not None and hasattr(eng, "abandon_iteration"):
            eng.abandon_iteration()
        return code

    def loss_scalars(self, out):
        """Python floats of one iteration's losses under the reference's loss_dict names (esrgan.py:355), fetched
        with ONE device->host copy instead of ~20 ``.item()`` calls (esrgan.py:632-645)."""
        z = torch.zeros((), device=self.device)
        parts = out.get("parts", {})
        d = out.get("d_loss", {})

        def g(k, name):
            return parts[k][name].reshape(()) if k in parts and name in parts[k] else z
        vec = torch.stack([d.get(0, z).reshape(()), d.get(1, z).reshape(()), out["g_loss"].reshape(()), g(0, "tot"), g(1, "tot"),
                           g(0, "adv"), g(1, "adv"), g(0, "pixel"), g(1, "pixel"), g(0, "lr"), g(1, "lr"),
                           g(0, "hist"), g(1, "hist"), g(0, "nnz"), g(1, "nnz"), g(0, "mask"), g(1, "mask"),
                           g(0, "hit"), g(1, "hit"),
                           (out["nan_probe"].reshape(()) if out.get("nan_probe") is not None else z)]).tolist()
        names = ['d_loss_def', 'd_loss_pow', 'g_loss', 'def_loss', 'pow_loss', 'adv_loss', 'adv_loss_pow', 'pixel_loss',
                 'pixel_loss_pow', 'lr_loss', 'lr_loss_pow', 'hist_loss', 'hist_loss_pow', 'nnz_loss', 'nnz_loss_pow',
                 'mask_loss', 'mask_loss_pow', 'hit_loss', 'hit_loss_pow']
        out_d = dict(zip(names, vec))
        out_d["nan_probe"] = vec[-1]       # data parallel: sum of all ranks' losses (NaN on every rank if any rank's is); else 0
        for name in ('wasser_loss', 'wasser_loss_pow', 'wasser_dist', 'wasser_dist_pow'):   # heads this build does not implement
            out_d[name] = 0.0
        return out_d

    # ------------------------------------------------------------------ warm-up iteration
    def warmup_step(self, imgs_lr, imgs_hr):
        """esrgan.py:416-427."""
        self.optimizer_G.zero_grad(set_to_none=True)
        gen_hr = self.generator(imgs_lr)
        loss_pixel = self.criterion_pixel(gen_hr, imgs_hr)
        self._backward_and_step_G(loss_pixel)
        self.last = {"g_loss": loss_pixel.detach()}
        return self.last

    # ------------------------------------------------------------------ full GAN iteration
    def g_phase_loss(self, imgs_lr, imgs_hr):
        """esrgan.py:466-552 at the default flags.  Returns (loss_G, generated list, ground_truth list, parts)."""
        p = self.scaling_power
        ground_truth = [imgs_hr, imgs_hr ** p]
        ground_truth_lr = [imgs_lr, imgs_lr ** p]
        # (single-process runs: the two views' discriminator passes and losses on two streams, as in the D phase; autograd runs
        # each view's backward on the stream of its forward)
        two_streams = self._d_streams is not None and all(self.lambdas[k] > 0 for k in range(2))
        main = torch.cuda.current_stream() if two_streams else None
        # D(real) needs the ground truth only: on the discriminator streams it runs beside the generator's forward
        early_real = {}
        if two_streams and self._d_overlap and not L.KernelTimer.active:
            have_gt = main.record_event()
            for k in range(2):
                self._d_streams[k].wait_event(have_gt)
                _uses(self._d_streams[k], ground_truth[k], ground_truth_lr[k])
                with torch.cuda.stream(self._d_streams[k]), torch.no_grad():
                    early_real[k] = self.discriminators[k](ground_truth[k], ground_truth_lr[k])
        generated = [self.generator(imgs_lr), self.generator.srs]
        gen_lr = self.pool(generated[0])
        generated_lr = [gen_lr, gen_lr ** p]
        loss_G = torch.zeros(1, device=imgs_lr.device)
        parts = {}
        tots = {}
        for k in range(2):
            if self.lambdas[k] <= 0:
                continue
            D = self.discriminators[k]
            if two_streams:
                self._d_streams[k].wait_stream(main)
                _uses(self._d_streams[k], generated[k], generated_lr[k], ground_truth[k], ground_truth_lr[k], imgs_lr, imgs_hr)
            ctx = torch.cuda.stream(self._d_streams[k]) if two_streams else contextlib.nullcontext()
            with ctx:
                loss_pixel = self.criterion_pixel(self._gmean(generated[k].mean(0))[None, ...], self._gmean(ground_truth[k].mean(0))[None, ...])
                loss_lr_pixel = self.criterion_pixel(generated_lr[k], ground_truth_lr[k])
                # gradients deposited on D's weights here are discarded by optimizer_D.zero_grad() (esrgan.py:568):
                # do not compute them
                for q in D.parameters():
                    q.requires_grad_(False)
                if k in early_real:
                    pred_real = early_real[k]
                else:
                    with torch.no_grad():
                        pred_real = D(ground_truth[k], ground_truth_lr[k])
                pred_fake = D(generated[k], generated_lr[k])
                for q in D.parameters():
                    q.requires_grad_(True)
                valid = torch.ones_like(pred_real)
                fake = torch.zeros_like(pred_real)
                if self.relativistic:      # esrgan.py:498-508
                    loss_GAN = .5 * (self.criterion_GAN(EPS + pred_fake - self._gmean(pred_real.mean(0, keepdim=True)), valid) +
                                     self.criterion_GAN(EPS + pred_real - self._gmean(pred_fake.mean(0, keepdim=True)), fake))
                else:                      # esrgan.py:509-510
                    loss_GAN = self.criterion_GAN(EPS + pred_fake, valid)
                tot = self.lambda_hr * loss_pixel + self.lambda_adv * loss_GAN + self.lambda_lr * loss_lr_pixel
                parts[k] = dict(pixel=loss_pixel.detach(), lr=loss_lr_pixel.detach(), adv=loss_GAN.detach())
                # optional physics heads: one fused HIP pass each (csrc/srk_loss.hip) instead of 3-6 HR-sized ATen ops.
                # Under data parallelism these are per-rank means of per-rank batches (like the reference under DDP would be).
                if self.lambda_nnz > 0:                                                    # esrgan.py:522-525
                    loss_nnz = self.mse(losses.soft_count(generated[k], 0.0, 50000.0), losses.hard_count(ground_truth[k], 0.0))
                    tot = tot + self.lambda_nnz * loss_nnz
                    parts[k]["nnz"] = loss_nnz.detach()
                if self.lambda_mask > 0:                                                   # esrgan.py:526-529
                    loss_mask = losses.mask_l1(generated[k], ground_truth[k])
                    tot = tot + self.lambda_mask * loss_mask
                    parts[k]["mask"] = loss_mask.detach()
                if self.lambda_hist > 0:                                                   # esrgan.py:530-538
                    if k not in self.histograms:
                        raise RuntimeError("lambda_hist > 0 needs set_hist_binedges(k, edges) first (esrgan.py:441-456)")
                    gen_hist = self.histograms[k].forward_positive(generated[k])
                    real_hist = self.histograms[k].forward_positive(ground_truth[k])
                    loss_hist = self.criterion_hist[k](gen_hist, real_hist)
                    tot = tot + self.lambda_hist * loss_hist
                    parts[k]["hist"] = loss_hist.detach()
                if self.lambda_hit > 0:                                                    # esrgan.py:543-547
                    gen_hit = losses.get_hitogram(generated[k], self.factor, self.hit_threshold, self.sigma)
                    target = losses.get_hitogram(ground_truth[k], self.factor, self.hit_threshold, self.sigma)
                    loss_hit = self.mse(gen_hit, target)
                    tot = tot + self.lambda_hit * loss_hit
                    parts[k]["hit"] = loss_hit.detach()
                parts[k]["tot"] = tot.detach()
                tots[k] = tot
        if two_streams:
            for k in tots:
                main.wait_stream(self._d_streams[k])
                _uses(main, tots[k], *parts[k].values())
        for k in sorted(tots):
            loss_G = loss_G + self.lambdas[k] * tots[k]
        return loss_G, generated, ground_truth, parts

    def d_phase_loss(self, k, gt, gen_detached, epsilon=None, cond=None):
        """esrgan.py:569-606 for discriminator k.  ``epsilon``: (B,1,1,1) interpolation factors or None -> drawn here.
        ``cond``: the LR ground truth, second argument of all three D calls (only the conditional discriminator reads it)."""
        D = self.discriminators[k]
        pred_real = D(gt, cond)
        pred_fake = D(gen_detached, cond)
        valid = torch.ones_like(pred_real)
        fake = torch.zeros_like(pred_real)
        if self.relativistic:          # esrgan.py:576-583
            loss_real = self.criterion_GAN(EPS + pred_real - self._gmean(pred_fake.mean(0, keepdim=True)), valid)
            loss_fake = self.criterion_GAN(EPS + pred_fake - self._gmean(pred_real.mean(0, keepdim=True)), fake)
        else:                          # esrgan.py:584-586
            loss_real = self.criterion_GAN(EPS + pred_real, valid)
            loss_fake = self.criterion_GAN(EPS + pred_fake, fake)
        loss_D = (loss_real + loss_fake) / 2
        gp = None
        if self.lambda_reg > 0:
            # gradient penalty (esrgan.py:596-606): D's input gradient at a random point on the segment between each real
            # image and its generated counterpart should have unit L2 norm; lambda_reg / 2 * mean((|g| - 1)^2).  The graph of
            # that gradient is kept (create_graph): loss_D.backward() differentiates through it into D's weights.
            n_img = gt.shape[0]
            if epsilon is None:
                epsilon = torch.rand(n_img, 1, 1, 1, device=gt.device)
            x_hat = (epsilon * gt + (1 - epsilon) * gen_detached).requires_grad_(True)
            # (only the INPUT gradient of this pass is ever asked for -- pred_hat enters the loss through grad_hat alone --: its nodes are
            # built to skip their weight gradients, ops.input_grad_only)
            with ops.input_grad_only():
                pred_hat = D(x_hat, cond)
            (grad_hat,) = torch.autograd.grad(pred_hat, x_hat, grad_outputs=valid, create_graph=True, retain_graph=True)
            grad_norm = grad_hat.reshape(n_img, -1).norm(2, dim=1)
            gp = (grad_norm - 1).square().mean() * (self.lambda_reg / 2)
            loss_D = loss_D + gp
        return loss_D, gp

    def gan_step(self, imgs_lr, imgs_hr, epsilons=None, update_g=True, update_d=True):
        # ---- generator (esrgan.py:416,457-555)
        self.optimizer_G.zero_grad(set_to_none=True)
        ground_truth_lr = [imgs_lr, imgs_lr ** self.scaling_power]
        pre_backward = None
        if update_g:
            loss_G, generated, ground_truth, parts = self.g_phase_loss(imgs_lr, imgs_hr)
            # The D phase (below) needs nothing the generator's backward produces: the discriminators see the pre-update generator
            # OUTPUT and their own weights, which the G phase only reads.  On the two discriminator streams it therefore waits for
            # THIS point of the main stream, not for its end, and runs beside the generator's backward, whose one-workgroup-per-CU
            # launches leave every launch gap and tail idle (single-process runs; SRK_D_OVERLAP=0: after it).  Stream order keeps it
            # behind the G phase's own discriminator passes, which ran on the same streams.
            # (not while bench.py brackets every launch with events: the per-kernel times of the probed step must not overlap)
            if self._d_streams is not None and self._d_overlap and update_d and not L.KernelTimer.active:
                pre_backward = torch.cuda.current_stream().record_event()
            self._backward_and_step_G(loss_G)
            # drop the 702 gradient views now, while the GPU is busy with the D phase: at the top of the next iteration this
            # loop would sit on the host's critical path right after the discriminator gate's sync
            self.optimizer_G.zero_grad(set_to_none=True)
        else:       # the D phase still needs the generator output (esrgan.py:457 skips only the G update)
            with torch.no_grad():
                generated = [self.generator(imgs_lr), self.generator.srs]
            ground_truth = [imgs_hr, imgs_hr ** self.scaling_power]
            loss_G, parts = torch.zeros(1, device=imgs_lr.device), {}
        # ---- discriminators (esrgan.py:561-626); they see the pre-update generator output
        loss_D_tot = {}
        nan_probe = None
        self.last_gate = {}
        # The two discriminators are independent until their gradient exchange / gates: each runs its forward / loss / backward on
        # its own stream -- their layers are short, latency-bound launches --, joined before anything is exchanged or read.
        d_items = list(self.discriminators.items()) if update_d else []
        two_streams = self._d_streams is not None and len(d_items) == 2
        main = torch.cuda.current_stream() if two_streams else None
        losses_D = {}
        for k, D in d_items:
            if two_streams:
                if pre_backward is not None:
                    self._d_streams[k].wait_event(pre_backward)
                else:
                    self._d_streams[k].wait_stream(main)
                _uses(self._d_streams[k], ground_truth[k], generated[k], ground_truth_lr[k], *(() if epsilons is None else (epsilons[k],)))
            with (torch.cuda.stream(self._d_streams[k]) if two_streams else contextlib.nullcontext()):
                self.optimizer_D[k].zero_grad(set_to_none=True)
                loss_D, gp = self.d_phase_loss(k, ground_truth[k], generated[k].detach(), None if epsilons is None else epsilons[k],
                                               cond=ground_truth_lr[k])
                loss_D.backward()
                losses_D[k] = loss_D
        if two_streams:
            for k, D in d_items:
                main.wait_stream(self._d_streams[k])
                _uses(main, losses_D[k], *[q.grad for q in D.parameters() if q.grad is not None])
        for k, D in d_items:
            loss_D = losses_D[k]
            self._sync_grads(D)          # (collectives stay on the main stream, in the same order on every rank)
            gate = loss_D.detach().reshape(1)
            if self.distributed:
                # Every rank must take the same branch, or the replicas' weights and Adam states drift apart: the
                # gate is always the mean over ranks (= the single-process loss_D when exact_dp).  The same exchange carries
                # this rank's generator loss, so a NaN on ANY rank turns the probe NaN on EVERY rank and the guard of
                # esrgan.py:645-648 fires everywhere in the same iteration instead of leaving the others in a collective.
                pair = _AllReduceMean.apply(torch.cat([gate, loss_G.detach().reshape(1)]))
                gate, nan_probe = pair[:1], (pair.sum() if nan_probe is None else nan_probe + pair.sum())
            if gate.item() > self.d_threshold:            # host sync, as in the reference (esrgan.py:623)
                self.optimizer_D[k].step()
            self.last_gate[k] = gate
            loss_D_tot[k] = loss_D.detach()
        if self.distributed and nan_probe is None:        # no discriminator ran this iteration: exchange the probe on its own
            nan_probe = _AllReduceMean.apply(loss_G.detach().reshape(1)).sum()
        self.last = {"g_loss": loss_G.detach(), "d_loss": loss_D_tot, "parts": parts, "nan_probe": nan_probe}
        return self.last
