"""torch.optim.Adam as ONE kernel launch per step (csrc/srk_optim.hip through srk_adam_step).

The reference steps three ``torch.optim.Adam`` instances per iteration (esrgan.py:299,305; steps at 427,487,623).  This class keeps
their hyper-parameters, ``param_groups`` and ``state_dict`` layout (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter: a checkpoint
written by either loads into the other) and ATen's fused-Adam arithmetic, including ``torch.amp.GradScaler``'s contract for fused
optimizers (``grad_scale`` / ``found_inf`` device tensors: the gradients are unscaled inside the update and a step with non-finite
gradients is skipped without a host round trip).  One difference from ATen's fused form under a GradScaler: ``p.grad`` is NOT written back
unscaled (the update divides in registers), so code that reads gradients after ``scaler.step`` -- norm logging, clipping -- sees them
multiplied by the loss scale; call ``scaler.unscale_(opt)`` first if it needs true gradients (the step then sees grad_scale = 1).
A step is also skipped, on the device, while a fault of the chain kernels is pending (include/srk.h: srk_adam_count_step).
GPU only, fp32 parameters, no amsgrad / maximize: anything else raises."""
import torch

from . import _lib as L


class Adam(torch.optim.Optimizer):
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= weight_decay:
            raise ValueError("invalid Adam hyper-parameter")
        if not (0.5 < betas[0] < 1.0 and 0.0 <= betas[1] < 1.0):
            raise ValueError(f"betas {betas}: 0.5 < beta1 < 1, 0 <= beta2 < 1 expected")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._tables = {}
        self.table_builds = 0          # diagnostic: how often a pointer table had to be (re)built

    def _init_group(self, gi, group):
        """exp_avg / exp_avg_sq of a group's parameters as slices of two flat buffers (one memset each instead of two per tensor); the
        step counter is ONE fp32 device scalar per group, which every parameter's state["step"] refers to"""
        ps = [p for p in group["params"] if p.requires_grad]
        fresh = [p for p in ps if "exp_avg" not in self.state[p]]
        if fresh:
            dev = fresh[0].device
            flat_m = torch.zeros(sum(p.numel() for p in fresh), dtype=torch.float32, device=dev)
            flat_v = torch.zeros_like(flat_m)
            step = group.get("_srk_step")
            if step is None:
                step = torch.zeros((), dtype=torch.float32, device=dev)
            off = 0
            for p in fresh:
                n = p.numel()
                self.state[p].update(step=step, exp_avg=flat_m[off:off + n].view_as(p), exp_avg_sq=flat_v[off:off + n].view_as(p))
                off += n
            group["_srk_step"] = step
        if "_srk_step" not in group:          # (state came in through load_state_dict: one counter per group again)
            st = [self.state[p]["step"] for p in ps if "step" in self.state[p]]
            group["_srk_step"] = torch.as_tensor(float(st[0]) if st else 0.0, dtype=torch.float32, device=ps[0].device).clone()
            for p in ps:
                self.state[p]["step"] = group["_srk_step"]
        return ps

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        grad_scale, found_inf = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        for gi, group in enumerate(self.param_groups):
            cache = self._tables.get(gi)
            if cache is None or cache["params"] is not group["params"] or cache["nparams"] != len(group["params"]):
                ps_all = self._init_group(gi, group)
                cache = self._tables[gi] = dict(params=group["params"], nparams=len(group["params"]), all=ps_all)
            # per step the host looks at the gradients only (one pointer per tensor: the generator has 702); parameters and moments do not
            # move.  A table is kept per set of gradient addresses (the allocator cycles through a few): rebuilding one costs ~1 ms of host time
            live = [p for p in cache["all"] if p.grad is not None]
            if not live:
                continue
            # (key = parameter AND gradient addresses of the live rows: a different live subset whose gradients land on the same blocks, or a
            # parameter whose storage was replaced by .data = / .to(), must not reuse a table that points elsewhere)
            gptr = tuple([(p.data_ptr(), p.grad.data_ptr()) for p in live])
            tabs = cache.setdefault("tabs", {})
            ent = tabs.get(gptr)
            if ent is None or ent[0] is not cache["all"] or ent[1] != len(live):
                rows = []
                for p, (_, gp) in zip(live, gptr):
                    g = p.grad
                    if not p.is_cuda or p.dtype != torch.float32 or g.is_sparse or not p.is_contiguous():
                        raise RuntimeError("super-resolution_amd.optim.Adam: contiguous fp32 CUDA parameters with dense gradients only (no CPU fallback)")
                    if not g.is_contiguous() or g.dtype != torch.float32 or g.numel() != p.numel():
                        raise RuntimeError("super-resolution_amd.optim.Adam: contiguous fp32 gradients of the parameter's size expected")
                    st = self.state[p]
                    rows.append((p.data_ptr(), gp, st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel()))
                if len(tabs) >= 8:
                    tabs.pop(next(iter(tabs)))
                # (addresses only: a table that held the gradient TENSORS would keep them alive, and the allocator could never hand the same
                # block back -- every step would see new addresses)
                ent = tabs[gptr] = (cache["all"], len(live), L.AdamTable(live[0].device, rows))
                self.table_builds += 1
            step = group["_srk_step"]
            # the number of THIS update; a skipped step does not count (ATen: _foreach_add_(steps, 1) ... _foreach_sub_(steps, found_inf)).
            # ONE decision per step, taken on the device: GradScaler's found_inf, or a pending fault of the chain kernels (srk.h: a
            # dense-block launch that gave up leaves garbage in the gradient buffers) -> skip word, which the update reads as found_inf
            skip = group.get("_srk_skip")
            if skip is None:
                skip = group["_srk_skip"] = torch.zeros((), dtype=torch.float32, device=step.device)
            L.adam_count_step(step, None if found_inf is None else found_inf.to(torch.float32).reshape(()), skip)
            b1, b2 = group["betas"]
            ent[2].run(lr=float(group["lr"]), beta1=float(b1), beta2=float(b2), eps=float(group["eps"]),
                               weight_decay=float(group["weight_decay"]), step=step,
                               grad_scale=None if grad_scale is None else grad_scale.to(torch.float32).reshape(()),
                               found_inf=skip)
        return loss

    def state_dict(self):
        sd = super().state_dict()
        sd["param_groups"] = [{k: v for k, v in g.items() if k not in ("_srk_step", "_srk_skip")} for g in sd["param_groups"]]
        # one step tensor per parameter, as torch.optim.Adam writes it (it increments every entry of the list: a shared one would count double)
        sd["state"] = {k: dict(v, step=v["step"].detach().clone()) if "step" in v else dict(v) for k, v in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        for g in self.param_groups:
            g.pop("_srk_step", None)
            g.pop("_srk_skip", None)
        self._tables = {}
