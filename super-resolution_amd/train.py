"""Training steps of the reference's hot loop (esrgan.py:399-626) on the MI355X modules.

``Stepper`` owns the replicas (generator, discriminators, Adam states) of ONE rank and runs one iteration:
  * ``g_only``: the warm-up iteration, esrgan.py:416-439 -- ``L1(G(lr), hr)``, backward, ``optimizer_G.step()``
  * ``gan``   : G phase (esrgan.py:457-555) + D phase (esrgan.py:561-626) at the default flags
Data parallelism is one process per GPU; gradients are averaged with RCCL all-reduce (``torch.distributed``
backend "nccl"), bucketed per RRDB and launched while the backward convolutions are still running.
"""
import torch
import torch.nn as nn

from . import models


class Stepper:
    def __init__(self, workload="g_only", res_blocks=23, device=None, hr=256, factor=4, distributed=False, channels=1,
                 filters=64, res_scale=0.2, lr=2e-4, betas=(0.9, 0.999), d_channels=(16, 32, 32, 64), lambdas=(0.2, 1.0),
                 lambda_hr=1.0, lambda_adv=0.01, lambda_lr=0.1, lambda_reg=0.01, d_threshold=0.001, scaling_power=1.0):
        import math
        self.workload = workload
        self.device = device
        self.distributed = distributed
        self.factor = factor
        self.generator = models.GeneratorRRDB(channels, filters=filters, num_res_blocks=res_blocks,
                                              num_upsample=int(math.log2(factor)), res_scale=res_scale).to(device)
        self.optimizer_G = torch.optim.Adam(self.generator.parameters(), lr=lr, betas=betas, fused=True)
        self.criterion_pixel = nn.L1Loss()
        if distributed:
            self.generator._engine.enable_grad_sync()
        self.last = {}

    def step(self, imgs_lr, imgs_hr):
        if self.workload == "g_only":
            return self._warmup_step(imgs_lr, imgs_hr)
        raise NotImplementedError(self.workload)

    def _warmup_step(self, imgs_lr, imgs_hr):
        """esrgan.py:416-427."""
        self.optimizer_G.zero_grad(set_to_none=True)
        gen_hr = self.generator(imgs_lr)
        loss_pixel = self.criterion_pixel(gen_hr, imgs_hr)
        loss_pixel.backward()
        self.optimizer_G.step()
        self.last = {"g_loss": loss_pixel.detach()}
        return self.last
