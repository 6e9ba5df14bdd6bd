"""Single-convolution autograd nodes on NHWC tensors, closed under differentiation.

The discriminator's gradient penalty (esrgan.py:596-606) differentiates *through* a backward pass, so the
three kernels (forward conv, data-gradient, weight-gradient) are exposed as three ``autograd.Function``s whose
backward methods are written in terms of each other:

    ConvPre (x,w,b)      z  = conv(lrelu_s(x), w) + b
    ConvDgrad(dz,w,x)    dx = conv^T(dz, w) * lrelu_s'(x)
    ConvWgrad(x,dz)      dw = sum dz (x) lrelu_s(x),  db = sum dz

``s`` (``in_slope``) is the LeakyReLU applied to the *input* while it is staged: the discriminator is chained on
pre-activations, conv(lrelu(z_prev)), so every node only needs its own input and output
(discriminator_block, models.py:140-146).  Weights arrive as canonical OIHW tensors and are packed per call.
"""
import os

import threading

import torch

from . import _lib as L
from ._lib import View

# stride-2 data gradients as PixelShuffle convs on dy (SRK_S2_PS_DGRAD=0: the zero-upsample form of srk_conv3x3)
S2_PS_DGRAD = os.environ.get("SRK_S2_PS_DGRAD", "1") != "0"


def _pack(w: torch.Tensor, transpose: bool):
    """OIHW weight -> packed fragment order (forward, or transposed + tap-flipped for the data gradient).
    Always packs from the current values: ``Parameter._version`` cannot be used as a cache key (fused Adam updates
    parameters without bumping it), so callers that reuse weights pass pre-packed tensors explicitly (PackedConvs)."""
    co, ci = w.shape[:2]
    wd = w.detach().contiguous()
    K, M = (co, ci) if transpose else (ci, co)
    dst = torch.empty(L.packed_floats(K, M), dtype=torch.float32, device=w.device)
    t = L.PackTable(w.device)
    t.add(wd, dst, M=M, k_off=0, k_len=K, K_total=K, transpose=transpose)
    t.run()
    return dst


class PackedConvs:
    """Forward and data-gradient packings of a list of conv weights, each refreshed with ONE launch.  A module calls
    ``refresh()`` at the top of every forward (weights cannot change between a forward and its backward), so one
    discriminator pass costs 2 pack launches instead of one per conv call."""

    def __init__(self, convs):
        self.convs = list(convs)
        self._sig = None

    def _build(self):
        dev = self.convs[0].weight.device
        self.fwd, self.bwd = [], []
        self.tab_f, self.tab_b = L.PackTable(dev), L.PackTable(dev)
        self.tab_s2 = {}            # fmt -> table of the stride-2 layers' data-gradient weights in PixelShuffle-conv form
        for c in self.convs:
            co, ci = c.weight.shape[:2]
            f = torch.empty(L.packed_floats(ci, co), dtype=torch.float32, device=dev)
            b = torch.empty(L.packed_floats(co, ci), dtype=torch.float32, device=dev)
            self.tab_f.add(c.weight.data, f, M=co, k_off=0, k_len=ci, K_total=ci)
            self.tab_b.add(c.weight.data, b, M=ci, k_off=0, k_len=co, K_total=co, transpose=True)
            self.fwd.append(f)
            self.bwd.append(b)
            if S2_PS_DGRAD and tuple(c.stride) == (2, 2) and ci % 4 == 0:
                # data gradient of the stride-2 layer = a stride-1 conv on dy with 4 * ci outputs, PixelShuffled into dx (srk.h,
                # srk_pack_entry.transpose == 2): no zero-upsampled input, full 64-wide output tiles -> the Winograd kernels apply
                fmt = 6 if (co % 8 == 0 and (4 * ci) % 64 == 0 and os.environ.get("SRK_WINOGRAD", "1") != "0") else 0
                p2 = torch.empty(L.packed_floats(co, 4 * ci, fmt), dtype=torch.float32, device=dev)
                self.tab_s2.setdefault(fmt, L.PackTable(dev, fmt)).add(c.weight.data, p2, M=4 * ci, k_off=0, k_len=co, K_total=co, transpose=2)
                b.s2pack = (p2, fmt)
        self.tab_f.finalize()
        self.tab_b.finalize()
        for t in self.tab_s2.values():
            t.finalize()

    def refresh(self, need_bwd: bool):
        if not self.convs:
            self.fwd, self.bwd = [], []
            return
        sig = tuple(c.weight.data_ptr() for c in self.convs)
        if sig != self._sig:
            self._build()
            self._sig = sig
        self.tab_f.run()
        if need_bwd:
            self.tab_b.run()
            for t in self.tab_s2.values():
                t.run()


def _out_hw(h, w, stride):
    return ((h - 1) // stride + 1, (w - 1) // stride + 1)


def _require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("super-resolution_amd: convolution kernels only run on a ROCm GPU tensor (no CPU fallback)")


def conv_pre_raw(x, w, b, stride, in_slope, wp=None):
    _require_gpu(x)
    N, H, W, Ci = x.shape
    Co = w.shape[0]
    OH, OW = _out_hw(H, W, stride)
    y = torch.empty(N, OH, OW, Co, dtype=torch.float32, device=x.device)
    L.conv3x3(View(x), wp if wp is not None else _pack(w, False), None if b is None else b.detach().contiguous(), View(y), N=N, H=H, W=W, OH=OH, OW=OW,
              Cin=Ci, Cout=Co, stride=stride, in_slope=in_slope)
    return y


def _dgrad_call(dz, w, x_or_none, stride, in_slope, H, W, wpt, dx):
    """the srk_conv3x3 call (x view, packed weights, bias, y view, kwargs) of one data gradient dx = conv^T(dz, w) * lrelu'(x)"""
    N, OH, OW, Co = dz.shape
    Ci = w.shape[1]
    mask = View(x_or_none) if (x_or_none is not None and in_slope != 1.0) else None
    s2 = getattr(wpt, "s2pack", None)
    if stride == 1:
        return (View(dz), wpt, None, View(dx), dict(N=N, H=H, W=W, OH=H, OW=W, Cin=Co, Cout=Ci, mask=mask, mask_slope=in_slope))
    if s2 is not None and H == 2 * OH and W == 2 * OW:
        # stride-2 layer, even extent: conv on dy with 4 * Ci outputs stored through the PixelShuffle epilogue (the mask is read
        # at the shuffled position, i.e. at the dx pixel)
        return (View(dz), s2[0], None, View(dx), dict(N=N, H=OH, W=OW, OH=OH, OW=OW, Cin=Co, Cout=4 * Ci, ps_out=True,
                                                     mask=mask, mask_slope=in_slope, wp_format=s2[1]))
    return (View(dz), wpt, None, View(dx), dict(N=N, H=OH, W=OW, OH=H, OW=W, Cin=Co, Cout=Ci,
                                                 in_mode=L.IN_ZERO_UPSAMPLE, mask=mask, mask_slope=in_slope))


def conv_dgrad_raw(dz, w, x_or_none, stride, in_slope, H, W, wpt=None):
    _require_gpu(dz)
    wpt = wpt if wpt is not None else _pack(w, True)
    dx = torch.empty(dz.shape[0], H, W, w.shape[1], dtype=torch.float32, device=dz.device)
    xv, wp, b, yv, kw = _dgrad_call(dz, w, x_or_none, stride, in_slope, H, W, wpt, dx)
    L.conv3x3(xv, wp, b, yv, **kw)
    return dx


def conv_wgrad_raw(x, dz, stride, in_slope, want_bias=True):
    _require_gpu(x)
    N, H, W, Ci = x.shape
    _, OH, OW, Co = dz.shape
    dw = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device=x.device)
    db = torch.empty(Co, dtype=torch.float32, device=x.device) if want_bias else None
    L.conv3x3_wgrad(View(x), View(dz), dw, db, N=N, H=H, W=W, OH=OH, OW=OW, Cin=Ci, Cout=Co, stride=stride, in_slope=in_slope)
    return dw, db


class ConvPre(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, in_slope, wp=None, wpt=None):
        x = x.contiguous()
        ctx.set_materialize_grads(False)        # an unused output gradient stays None instead of a zero-filled tensor
        ctx.save_for_backward(x, w)
        ctx.stride, ctx.in_slope, ctx.has_b = stride, in_slope, b is not None
        ctx.wp, ctx.wpt = wp, wpt
        ctx.want_pg = _want_param_grads()
        return conv_pre_raw(x, w, b, stride, in_slope, wp)

    @staticmethod
    def backward(ctx, dz):
        if dz is None:
            return None, None, None, None, None, None, None
        x, w = ctx.saved_tensors
        dz = dz.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ConvDgrad.apply(dz, w, x, ctx.stride, ctx.in_slope, ctx.wp, ctx.wpt)
        if ctx.want_pg and (ctx.needs_input_grad[1] or (ctx.has_b and ctx.needs_input_grad[2])):
            dw, db = ConvWgrad.apply(x, dz, ctx.stride, ctx.in_slope)
            if not ctx.has_b:
                db = None
        return dx, dw, db, None, None, None, None


class ConvDgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, dz, w, x, stride, in_slope, wp=None, wpt=None):
        dz = dz.contiguous()
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(dz, w, x)
        ctx.stride, ctx.in_slope = stride, in_slope
        ctx.wp = wp
        return conv_dgrad_raw(dz, w, x, stride, in_slope, x.shape[1], x.shape[2], wpt)

    @staticmethod
    def backward(ctx, gdx):
        if gdx is None:
            return None, None, None, None, None, None, None
        dz, w, x = ctx.saved_tensors
        s = ctx.in_slope
        t = gdx.contiguous()
        if s != 1.0:                                  # lrelu' is piecewise constant: no gradient to x
            _require_gpu(t)
            scaled = torch.empty_like(t)
            L.lrelu_grad_mul(x, t, scaled, s)         # = torch.where(x > 0, t, t * s) in one pass, no bool tensor (x: saved contiguous)
            t = scaled
        g_dz = g_w = None
        if ctx.needs_input_grad[0]:
            g_dz = ConvPre.apply(t, w, None, ctx.stride, 1.0, ctx.wp, None)
        if ctx.needs_input_grad[1]:
            g_w, _ = ConvWgrad.apply(t, dz, ctx.stride, 1.0)
        return g_dz, g_w, None, None, None, None, None


class ConvWgrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, dz, stride, in_slope):
        x, dz = x.contiguous(), dz.contiguous()
        ctx.set_materialize_grads(False)        # the gradient penalty's double backward reaches dw only: ggb stays None
        ctx.save_for_backward(x, dz)
        ctx.stride, ctx.in_slope = stride, in_slope
        dw, db = conv_wgrad_raw(x, dz, stride, in_slope)
        return dw, db

    @staticmethod
    def backward(ctx, ggw, ggb):
        if ggw is None and ggb is None:
            return None, None, None, None
        x, dz = ctx.saved_tensors
        g_x = g_dz = None
        if ctx.needs_input_grad[0] and ggw is not None:
            g_x = ConvDgrad.apply(dz, ggw, x, ctx.stride, ctx.in_slope)
        if ctx.needs_input_grad[1]:
            if ggw is not None:
                g_dz = ConvPre.apply(x, ggw, ggb, ctx.stride, ctx.in_slope)
            elif ggb is not None:
                g_dz = ggb.view(1, 1, 1, -1).expand_as(dz)
        return g_x, g_dz, None, None


# The gradient penalty's INNER torch.autograd.grad(pred_hat, x_hat, create_graph=True) (esrgan.py:602-605) wants the input gradient only,
# but a custom Function's ctx.needs_input_grad is static: the backward below would also launch every layer's weight gradient and throw
# it away (18 launches per iteration).  train.Stepper.d_phase_loss therefore builds the x_hat FORWARD under input_grad_only(): the nodes
# created inside it carry the decision themselves (ctx.want_pg), so nothing global is read while some other backward -- a second
# Stepper, another host thread, a re-entrant backward through the same Functions -- is running on the autograd threads.  (Round 3 used a
# module-global flag around the inner grad call: any backward running concurrently silently lost its weight gradients.)
_tls = threading.local()


class input_grad_only:
    """Forward passes built inside this context produce nodes whose backward returns the INPUT gradient only (no weight / bias
    gradients).  Thread-local, captured at forward time."""

    def __enter__(self):
        self._prev = getattr(_tls, "param_grads", True)
        _tls.param_grads = False
        return self

    def __exit__(self, *exc):
        _tls.param_grads = self._prev
        return False


def _want_param_grads() -> bool:
    return getattr(_tls, "param_grads", True)


class ChainPre(torch.autograd.Function):
    """z_L = conv_L(lrelu_sL(... conv_1(lrelu_s1(x)) ...)): a whole chain of pre-activation convolutions (the patch discriminator,
    models.py:149-174) as ONE autograd node whose forward hands all its launches to the library in one call (srk_conv3x3_seq).
    The backward walks the chain with ConvDgrad / ConvWgrad, the differentiable nodes above, so the gradient penalty's double
    backward (esrgan.py:596-606) works as before.  (The intermediate pre-activations are saved detached: what ConvPre per layer would
    add -- a gradient of a WEIGHT gradient w.r.t. the activations of earlier layers -- nothing in the reference's iteration asks for.)
    ``meta``: per layer (stride, in_slope, packed forward weights, packed data-gradient weights or None); ``wb``: w_1, b_1, w_2, ..."""

    @staticmethod
    def forward(ctx, x, meta, *wb):
        _require_gpu(x)
        x = x.contiguous()
        ctx.set_materialize_grads(False)
        n_l = len(meta)
        zs, calls = [x], []
        N, H, W, Ci = x.shape
        for l, (stride, in_slope, wp, _) in enumerate(meta):
            w, b = wb[2 * l], wb[2 * l + 1]
            Co = w.shape[0]
            OH, OW = _out_hw(H, W, stride)
            y = torch.empty(N, OH, OW, Co, dtype=torch.float32, device=x.device)
            calls.append((View(zs[-1]), wp if wp is not None else _pack(w, False), None if b is None else b.detach().contiguous(), View(y),
                          dict(N=N, H=H, W=W, OH=OH, OW=OW, Cin=Ci, Cout=Co, stride=stride, in_slope=in_slope)))
            zs.append(y)
            H, W, Ci = OH, OW, Co
        L.conv3x3_seq(calls)
        ctx.meta = meta
        ctx.want_pg = _want_param_grads()
        ctx.save_for_backward(*zs[:-1], *[wb[2 * l] for l in range(n_l)])
        ctx.has_b = [wb[2 * l + 1] is not None for l in range(n_l)]
        return zs[-1]

    @staticmethod
    def backward(ctx, dz):
        n_l = len(ctx.meta)
        if dz is None:
            return (None,) * (2 + 2 * n_l)
        saved = ctx.saved_tensors
        zs, ws = saved[:n_l], saved[n_l:]
        dz = dz.contiguous()
        grads = [None] * (2 * n_l)
        want_w = [ctx.want_pg and (ctx.needs_input_grad[2 + 2 * l] or (ctx.has_b[l] and ctx.needs_input_grad[3 + 2 * l]))
                  for l in range(n_l)]
        if not torch.is_grad_enabled():
            # first-order backward (nothing will differentiate THIS pass): no autograd nodes needed, so the whole chain of data
            # gradients goes to the library in one call and all weight gradients in a second one
            _require_gpu(dz)
            dzs, calls = [None] * n_l + [dz], []
            first = 0 if ctx.needs_input_grad[0] else 1
            for l in range(n_l - 1, first - 1, -1):
                stride, in_slope, wp, wpt = ctx.meta[l]
                _, H, W, Ci = zs[l].shape
                dzs[l] = torch.empty(zs[l].shape, dtype=torch.float32, device=dz.device)
                calls.append(_dgrad_call(dzs[l + 1], ws[l], zs[l], stride, in_slope, H, W, wpt if wpt is not None else _pack(ws[l], True), dzs[l]))
            if calls:
                L.conv3x3_seq(calls)
            wcalls = []
            for l in range(n_l):
                if want_w[l]:
                    stride, in_slope, _, _ = ctx.meta[l]
                    Co, Ci = ws[l].shape[:2]
                    dw = torch.empty(Co, Ci, 3, 3, dtype=torch.float32, device=dz.device)
                    db = torch.empty(Co, dtype=torch.float32, device=dz.device) if ctx.has_b[l] else None
                    N, H, W, _ = zs[l].shape
                    _, OH, OW, _ = dzs[l + 1].shape
                    wcalls.append((View(zs[l]), View(dzs[l + 1]), dw, db, dict(N=N, H=H, W=W, OH=OH, OW=OW, Cin=Ci, Cout=Co, stride=stride,
                                                                              in_slope=in_slope)))
                    grads[2 * l], grads[2 * l + 1] = dw, db
            if wcalls:
                L.conv3x3_wgrad_seq(wcalls)
            return (dzs[0] if ctx.needs_input_grad[0] else None, None) + tuple(grads)
        for l in range(n_l - 1, -1, -1):
            stride, in_slope, wp, wpt = ctx.meta[l]
            if want_w[l]:
                dw, db = ConvWgrad.apply(zs[l], dz, stride, in_slope)
                grads[2 * l], grads[2 * l + 1] = dw, (db if ctx.has_b[l] else None)
            if l > 0 or ctx.needs_input_grad[0]:
                dz = ConvDgrad.apply(dz, ws[l], zs[l], stride, in_slope, wp, wpt)
        return (dz if ctx.needs_input_grad[0] else None, None) + tuple(grads)


def conv_pre(x, w, b, stride=1, in_slope=1.0, wp=None, wpt=None):
    """z = conv3x3(lrelu_{in_slope}(x), w, stride, pad 1) + b on an NHWC tensor; differentiable twice.
    ``wp`` / ``wpt``: optional pre-packed forward / data-gradient weights (PackedConvs) of ``w``."""
    return ConvPre.apply(x, w, b, stride, in_slope, wp, wpt)


class _ToNHWC(torch.autograd.Function):
    """NCHW -> NHWC copy whose gradient is again a CONTIGUOUS tensor (plain permute().contiguous() hands autograd a
    permuted view back, and the reference's ``gradients.view(batch_size, -1)`` of the gradient penalty, esrgan.py:604,
    needs a contiguous input gradient like nn.Conv2d's).  The pair is closed under differentiation (double backward)."""

    @staticmethod
    def forward(ctx, x):
        return x.permute(0, 2, 3, 1).contiguous()

    @staticmethod
    def backward(ctx, g):
        return _ToNCHW.apply(g)


class _ToNCHW(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.permute(0, 3, 1, 2).contiguous()

    @staticmethod
    def backward(ctx, g):
        return _ToNHWC.apply(g)


def to_nhwc(x: torch.Tensor) -> torch.Tensor:
    """NCHW -> NHWC (free view when C == 1: the two layouts coincide)."""
    n, c, h, w = x.shape
    if c == 1:
        return x.reshape(n, h, w, 1)
    return _ToNHWC.apply(x)


def to_nchw(x: torch.Tensor) -> torch.Tensor:
    n, h, w, c = x.shape
    if c == 1:
        return x.reshape(n, 1, h, w)
    return _ToNCHW.apply(x)


class _SumPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, k):
        _require_gpu(x)
        x = x.contiguous()
        n, c, h, w = x.shape
        ctx.k, ctx.shape = k, x.shape
        y = torch.empty(n, c, h // k, w // k, dtype=torch.float32, device=x.device)
        L.sum_pool_fwd(x, y, n * c, h, w, k)
        return y

    @staticmethod
    def backward(ctx, dy):
        n, c, h, w = ctx.shape
        dx = torch.empty(n, c, h, w, dtype=torch.float32, device=dy.device)
        L.sum_pool_bwd(dy.contiguous(), dx, n * c, h, w, ctx.k)
        return dx, None


def sum_pool(x, k):
    return _SumPool.apply(x, k)
