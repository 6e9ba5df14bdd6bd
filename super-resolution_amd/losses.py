"""The generator phase's optional physics loss heads (esrgan.py:522-547) on the fused HIP kernels of csrc/srk_loss.hip.

Same names and argument meaning as the reference helpers they replace::

    softgreater(x, val, sigma=5000, delta=0)          utils.py:259-261
    nnz_mask(x, sigma=5e4)                            utils.py:271-272
    get_hitogram(t, factor, threshold=.1, sig=80)     utils.py:264-268
    DiffableHistogram(bins, min, max, sigma, batchwise)  models.py:308-342
    KLD_hist(binedges)                                utils.py:90-113

plus the fused forms the training step uses, which never materialise an HR-sized intermediate::

    soft_count(x, val, sigma)   == softgreater(x, val, sigma).sum(1).sum(1).sum(1)          esrgan.py:523
    hard_count(x, val)          == (x > val).sum(1).sum(1).sum(1).float()                   esrgan.py:524
    mask_l1(a, b, sigma)        == L1Loss()(nnz_mask(a, sigma), nnz_mask(b, sigma))         esrgan.py:527-529
    soft_hist_positive(x, centers, delta, sigma) == DiffableHistogram(...)(x[x > 0])        esrgan.py:533-536

All are differentiable w.r.t. their first argument (what the reference differentiates: the generator output); they
need CUDA tensors and fail loudly otherwise -- there is no PyTorch fallback.
"""
import numpy as np
import torch
import torch.nn as nn

from . import _lib as L


def _req(x):
    if not x.is_cuda:
        raise RuntimeError("super-resolution_amd.losses runs on the HIP kernels of libsrk.so: got a CPU tensor (no CPU fallback)")
    return x.contiguous().float()


def _st():
    return L.stream_ptr()


class _Sigmoid(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift):
        x = _req(x)
        y = torch.empty_like(x)
        L.check(L.lib().srk_sigmoid_fwd(x.data_ptr(), y.data_ptr(), x.numel(), scale, shift, _st()), "srk_sigmoid_fwd")
        ctx.save_for_backward(y)
        ctx.scale = scale
        return y

    @staticmethod
    def backward(ctx, gy):
        y, = ctx.saved_tensors
        gy = gy.contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().srk_sigmoid_bwd(y.data_ptr(), gy.data_ptr(), dx.data_ptr(), y.numel(), ctx.scale, _st()), "srk_sigmoid_bwd")
        return dx, None, None


def softgreater(x, val, sigma=5000, delta=0):
    """Differentiable version of ``x > val``: sigmoid(sigma * (x - val + delta))  (utils.py:259-261)."""
    return _Sigmoid.apply(x, float(sigma), float(sigma) * (float(delta) - float(val)))


def nnz_mask(x, sigma=5e4):
    """sigmoid(sigma * x)  (utils.py:271-272)."""
    return _Sigmoid.apply(x, float(sigma), 0.0)


class _SoftCount(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, val, sigma):
        x = _req(x)
        B = x.shape[0]
        per = x.numel() // B
        out = torch.empty(B, dtype=torch.float32, device=x.device)
        ws = L.loss_workspace(x.device)
        L.check(L.lib().srk_soft_count_fwd(x.data_ptr(), out.data_ptr(), B, per, sigma, val, 0, ws.data_ptr(), ws.numel(), _st()),
                "srk_soft_count_fwd")
        ctx.save_for_backward(x)
        ctx.val, ctx.sigma = val, sigma
        return out

    @staticmethod
    def backward(ctx, g):
        x, = ctx.saved_tensors
        B = x.shape[0]
        dx = torch.empty_like(x)
        g = g.contiguous().float()
        L.check(L.lib().srk_soft_count_bwd(x.data_ptr(), g.data_ptr(), dx.data_ptr(), B, x.numel() // B, ctx.sigma, ctx.val, _st()),
                "srk_soft_count_bwd")
        return dx, None, None


def soft_count(x, val=0.0, sigma=50000.0):
    """Per-image soft count of entries above ``val``: [B,...] -> [B]  (esrgan.py:523)."""
    return _SoftCount.apply(x, float(val), float(sigma))


def hard_count(x, val=0.0):
    """Per-image count of entries above ``val`` as float: [B,...] -> [B]  (esrgan.py:524).  Not differentiable."""
    x = _req(x.detach())
    B = x.shape[0]
    out = torch.empty(B, dtype=torch.float32, device=x.device)
    ws = L.loss_workspace(x.device)
    L.check(L.lib().srk_soft_count_fwd(x.data_ptr(), out.data_ptr(), B, x.numel() // B, 1.0, float(val), 1, ws.data_ptr(), ws.numel(), _st()),
            "srk_soft_count_fwd")
    return out


class _MaskL1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b, sigma):
        a, b = _req(a), _req(b)
        if a.shape != b.shape:
            raise ValueError(f"mask_l1: shapes differ {tuple(a.shape)} vs {tuple(b.shape)}")
        out = torch.empty(1, dtype=torch.float32, device=a.device)
        ws = L.loss_workspace(a.device)
        L.check(L.lib().srk_mask_l1_fwd(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), sigma, ws.data_ptr(), ws.numel(), _st()),
                "srk_mask_l1_fwd")
        ctx.save_for_backward(a, b)
        ctx.sigma = sigma
        return out.reshape(())

    @staticmethod
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        g = g.contiguous().float().reshape(1)
        L.check(L.lib().srk_mask_l1_bwd(a.data_ptr(), b.data_ptr(), g.data_ptr(), da.data_ptr(), a.numel(), ctx.sigma, _st()),
                "srk_mask_l1_bwd")
        return da, None, None


def mask_l1(a, b, sigma=5e4):
    """mean |sigmoid(sigma a) - sigmoid(sigma b)|, differentiable w.r.t. ``a``  (esrgan.py:527-529)."""
    return _MaskL1.apply(a, b.detach(), float(sigma))


class _Hitogram(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t, factor, thr, sig):
        t = _req(t)
        if t.dim() != 4:
            raise ValueError("get_hitogram expects [B, C, H, W]")
        B, Cc, H, W = t.shape
        out = torch.empty(factor, factor, dtype=torch.float32, device=t.device)
        ws = L.loss_workspace(t.device)
        L.check(L.lib().srk_hitogram_fwd(t.data_ptr(), out.data_ptr(), B * Cc, H, W, factor, thr, sig, ws.data_ptr(), ws.numel(), _st()),
                "srk_hitogram_fwd")
        ctx.save_for_backward(t)
        ctx.args = (factor, thr, sig)
        return out

    @staticmethod
    def backward(ctx, g):
        t, = ctx.saved_tensors
        factor, thr, sig = ctx.args
        B, Cc, H, W = t.shape
        dt = torch.empty_like(t)
        g = g.contiguous().float()
        L.check(L.lib().srk_hitogram_bwd(t.data_ptr(), g.data_ptr(), dt.data_ptr(), B * Cc, H, W, factor, thr, sig, _st()), "srk_hitogram_bwd")
        return dt, None, None, None


def get_hitogram(t, factor, threshold=.1, sig=80):
    """[B,C,H,W] -> [factor, factor]: mean soft hit probability per position inside the factor x factor super-pixel
    (utils.py:264-268; ``sig <= 0``: plain mean)."""
    return _Hitogram.apply(t, int(factor), float(threshold), float(sig))


class _SoftHist(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, centers, delta, sigma, positive_only):
        x = _req(x)
        K = centers.numel()
        out = torch.empty(1, K, dtype=torch.float32, device=x.device)
        ws = L.loss_workspace(x.device)
        L.check(L.lib().srk_soft_hist_fwd(x.data_ptr(), x.numel(), centers.data_ptr(), delta.data_ptr(), K, sigma, positive_only,
                                          out.data_ptr(), ws.data_ptr(), ws.numel(), _st()), "srk_soft_hist_fwd")
        ctx.save_for_backward(x, centers, delta)
        ctx.sigma, ctx.pos = sigma, positive_only
        return out

    @staticmethod
    def backward(ctx, g):
        x, centers, delta = ctx.saved_tensors
        dx = torch.empty_like(x)
        g = g.contiguous().float()
        L.check(L.lib().srk_soft_hist_bwd(x.data_ptr(), x.numel(), centers.data_ptr(), delta.data_ptr(), centers.numel(), ctx.sigma, ctx.pos,
                                          g.data_ptr(), dx.data_ptr(), _st()), "srk_soft_hist_bwd")
        return dx, None, None, None, None


def soft_hist_positive(x, centers, delta, sigma):
    """Soft histogram [1, K] of the POSITIVE entries of ``x`` (== DiffableHistogram(...)(x[x > 0]), esrgan.py:533-536) in
    one pass: the boolean-index gather of the reference is folded into the kernel as a predicate."""
    return _SoftHist.apply(x, centers.contiguous().float().reshape(-1), delta.contiguous().float().reshape(-1), float(sigma), 1)


def soft_hist_all(x, centers, delta, sigma):
    """Soft histogram [1, K] of all entries of ``x`` (DiffableHistogram.forward, models.py:335-340)."""
    return _SoftHist.apply(x, centers.contiguous().float().reshape(-1), delta.contiguous().float().reshape(-1), float(sigma), 0)


class DiffableHistogram(nn.Module):
    """Soft (sigmoid-edged) histogram, models.py:308-342.  ``bins``: an int (that many equal bins over [min, max]; like the
    reference this yields bins - 1 centres) or a sequence of bin edges.  ``forward(x)`` histograms ALL entries of x like the
    reference; the training step feeds it ``x[x > 0]`` -- :meth:`forward_positive` on the full tensor gives the same result
    without the gather.  The counting itself is the fused kernel behind soft_hist_all / soft_hist_positive."""

    def __init__(self, bins, min=0, max=1, sigma=25, batchwise=False):
        super().__init__()
        self.sigma, self.batchwise = sigma, batchwise
        if isinstance(bins, int):
            width = (max - min) / bins
            self.bins = torch.Tensor(bins)
            self.delta = torch.full((bins - 1,), float(width))
            self.centers = float(min) + width * (torch.arange(bins - 1, dtype=torch.float32) + 0.5)
        else:
            edges = np.asarray(bins)
            self.bins = torch.as_tensor(edges.astype(np.float32))
            self.delta = torch.as_tensor(np.diff(edges.astype(np.float64))).float().unsqueeze(0)
            self.centers = self.bins[:-1] + self.delta / 2

    def to(self, device):                      # plain attributes, not buffers (as in the reference): moved by hand
        self.centers, self.delta = self.centers.to(device), self.delta.to(device)
        return self

    def forward(self, x):
        if self.batchwise and x.dim() == 4 and len(x) != 1:
            return torch.cat([self.forward(img.reshape(-1)) for img in x], 0)
        return soft_hist_all(x.reshape(-1), self.centers, self.delta, self.sigma)

    def forward_positive(self, x):
        return soft_hist_positive(x, self.centers, self.delta, self.sigma)


class KLD_hist(nn.Module):
    """KL(p || q) between two histograms given as bin COUNTS, every bin weighted by its width and the result divided by the
    mean width (utils.py:90-113).  q gets +1e-6 per bin before normalising so that empty bins stay finite.  K-sized tensors:
    plain torch ops."""

    def __init__(self, binedges):
        super().__init__()
        edges = torch.as_tensor(binedges)
        self.binsizes = (edges[1:] - edges[:-1]).float()
        self.binmean = self.binsizes.mean()
        self.kldiv = nn.KLDivLoss(reduction='sum')

    def to(self, device):
        self.binsizes = self.binsizes.to(device)
        return self

    def forward(self, q_entries, p_entries):
        p = p_entries * self.binsizes / p_entries.sum().float()
        log_q = ((q_entries + 1e-6) * self.binsizes / q_entries.sum().float()).log()
        return self.kldiv(log_q, p) / self.binmean
