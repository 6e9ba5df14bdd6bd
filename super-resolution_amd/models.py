"""Drop-in for the class surface of the reference's ``models.py`` on MI355X.

Same constructor signatures, attributes (``.thres``, ``.srs``, ``.power``, ``.multiplier``, ``.output_shape``),
``nn.Module`` protocol and ``state_dict`` keys as /root/reference/models.py, so ``from models import *`` users
(esrgan.py:17, evaluation/eval.py:11, evaluation/demo.py:17) and ``.pth`` checkpoints interchange both ways.
Parameters stay canonical OIHW fp32 ``nn.Conv2d`` weights (``weight_reset``/``uniform_reset``/``plot_grad_flow``
keep working); the arithmetic runs in the gfx950 kernels of ``libsrk.so``.  There is no CPU fallback: a non-GPU
input raises.
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .engine import GeneratorEngine, generator_raw


class Conv3x3(nn.Conv2d):
    """``nn.Conv2d(cin, cout, 3, stride, 1)`` whose forward is the srk kernel (NCHW in / NCHW out).

    Inside GeneratorRRDB / Markovian_Discriminator the parent runs fused NHWC pipelines and never calls this
    forward; it exists so that any other composition of the reference's building blocks still runs on the HIP
    path (and stays twice differentiable)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True):
        if kernel_size != 3 or padding != 1 or stride not in (1, 2):
            raise NotImplementedError("only the reference's 3x3 / pad 1 / stride 1|2 convolutions are implemented")
        super().__init__(in_channels, out_channels, 3, stride, 1, bias=bias)

    def forward(self, x):
        y = ops.conv_pre(ops.to_nhwc(x.float()), self.weight, self.bias, self.stride[0], 1.0)
        return ops.to_nchw(y)


class DenseResidualBlock(nn.Module):
    """models.py:9-41.  Parameter container; GeneratorRRDB runs it concat-free (engine.py) unless drop_rate > 0, where the
    module-wise forward below is used (HIP convolutions, torch.cat / Dropout2d glue: a non-default branch)."""

    def __init__(self, filters, res_scale=0.2, drop_rate=0):
        super().__init__()
        self.res_scale = res_scale

        def block(in_features, non_linearity=True):
            layers = [Conv3x3(in_features, filters, 3, 1, 1, bias=True)]
            if non_linearity:
                layers += [nn.LeakyReLU()]
            return nn.Sequential(*layers)

        self.b1 = block(in_features=1 * filters)
        self.b2 = block(in_features=2 * filters)
        self.b3 = block(in_features=3 * filters)
        self.b4 = block(in_features=4 * filters)
        self.b5 = block(in_features=5 * filters, non_linearity=False)
        self.drop = drop_rate > 0
        if self.drop:
            self.drop1 = nn.Dropout2d(drop_rate)
        self.blocks = [self.b1, self.b2, self.b3, self.b4, self.b5]

    def forward(self, x):
        """Module-wise form (used by the drop_rate > 0 and ConvTranspose2d variants only): conv k sees x and the outputs of
        convs 1..k-1 on the channel axis; the last conv's output, after the optional channel dropout, is the residual."""
        feats = [x]
        for conv_k in self.blocks:
            feats.append(conv_k(feats[0] if len(feats) == 1 else torch.cat(feats, 1)))
        residual = self.drop1(feats[-1]) if self.drop else feats[-1]
        return torch.add(x, residual, alpha=self.res_scale)


class ResidualInResidualDenseBlock(nn.Module):
    """models.py:44-53 (inner blocks keep res_scale 0.2: the reference does not forward its argument)."""

    def __init__(self, filters, res_scale=0.2, drop_rate=0):
        super().__init__()
        self.res_scale = res_scale
        self.dense_blocks = nn.Sequential(
            DenseResidualBlock(filters, drop_rate=drop_rate), DenseResidualBlock(filters, drop_rate=drop_rate),
            DenseResidualBlock(filters, drop_rate=drop_rate))

    def forward(self, x):
        return torch.add(x, self.dense_blocks(x), alpha=self.res_scale)


class GeneratorRRDB(nn.Module):
    """models.py:56-135, default upsampling branch (conv F->4F, LeakyReLU, PixelShuffle(2))."""

    def __init__(self, channels=1, filters=64, num_res_blocks=10, num_upsample=1, power=1, multiplier=1, drop_rate=0,
                 res_scale=0.2, use_transposed_conv=False, fully_tconv_upsample=False, num_final_layer_res=0, uniform_init=False):
        super().__init__()
        if filters % 8 != 0:
            raise NotImplementedError("filters must be a multiple of 8 (8-channel K chunks of the MFMA kernels)")
        self.channels, self.filters, self.num_upsample = channels, filters, num_upsample
        self.num_final_layer_res = num_final_layer_res
        self.drop_rate = drop_rate
        self.conv1 = Conv3x3(channels, filters, kernel_size=3, stride=1, padding=1)
        self.res_blocks = nn.Sequential(*[ResidualInResidualDenseBlock(filters, res_scale=res_scale, drop_rate=drop_rate)
                                          for _ in range(num_res_blocks)])
        self.conv2 = Conv3x3(filters, filters, kernel_size=3, stride=1, padding=1)
        # Upsampling stages (models.py:69-90).  Default: conv F->4F + LeakyReLU + PixelShuffle(2), fused into the HIP engine.
        # The two non-default variants put a ConvTranspose2d(F, F, 2, stride 2) + LeakyReLU at every odd stage
        # (use_transposed_conv) or at every stage (fully_tconv_upsample): those layers are plain PyTorch modules (SURVEY 8b:
        # non-default branches may stay on PyTorch ops) and the generator then runs module by module, every 3x3 conv still
        # on the HIP kernels, so checkpoints trained with these flags load and run.
        kinds = ["tconv" if (fully_tconv_upsample and not use_transposed_conv) or (use_transposed_conv and u % 2 == 1) else "shuffle"
                 for u in range(num_upsample)]
        upsample_layers = []
        for kind in kinds:
            if kind == "shuffle":
                upsample_layers += [Conv3x3(filters, filters * 4, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(),
                                    nn.PixelShuffle(upscale_factor=2)]
            else:
                upsample_layers += [nn.ConvTranspose2d(filters, filters, kernel_size=2, stride=2, padding=0), nn.LeakyReLU()]
        self.upsampling = nn.Sequential(*upsample_layers)
        self.modulewise = drop_rate > 0 or "tconv" in kinds      # these run _forward_modulewise instead of the fused engine
        if num_final_layer_res > 0:
            self.res_blocks_final = nn.Sequential(*[ResidualInResidualDenseBlock(filters, res_scale=res_scale, drop_rate=drop_rate)
                                                    for _ in range(num_final_layer_res)])
        self.conv3 = nn.Sequential(Conv3x3(filters, filters, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(),
                                   Conv3x3(filters, channels, kernel_size=3, stride=1, padding=1))
        self.thres = 0
        self.power = nn.Parameter(torch.Tensor([power]), False)
        self.multiplier = nn.Parameter(torch.Tensor([multiplier]), False)
        self._scalars = None
        self.register_load_state_dict_post_hook(_drop_scalar_cache)     # a module-level function: the module stays picklable
        self._engine = GeneratorEngine(self)
        if uniform_init:
            self.init_conv2d()

    def init_conv2d(self):
        # models.py:108-112: children() is not recursive -> only conv1 / conv2 are touched
        for c in self.children():
            if isinstance(c, nn.Conv2d):
                nn.init.xavier_uniform_(c.weight)
                nn.init.constant_(c.bias, 0.)

    def _power_multiplier(self):
        """Host copies of the two frozen scalars (the reference syncs on them every forward, models.py:116,133).  In training
        mode they are cached -- two device->host syncs per forward would drain the launch queue of the hot loop -- and re-read
        when the Parameters were replaced or written through autograd-visible ops, on load_state_dict / .to() / .train() /
        .eval(), and on refresh_scalars(); in eval mode they are read on every forward.  (A bare ``gen.power.data.fill_(p)``
        bumps no version counter: follow it with gen.train(), gen.eval() or gen.refresh_scalars().)"""
        key = (self.power.data_ptr(), self.power._version, self.multiplier.data_ptr(), self.multiplier._version)
        if self._scalars is None or self._scalars[0] != key or not self.training:
            self._scalars = (key, float(self.power.item()), float(self.multiplier.item()))
        return self._scalars[1], self._scalars[2]

    def refresh_scalars(self):
        self._scalars = None

    def train(self, mode=True):
        self._scalars = None
        return super().train(mode)

    def _apply(self, fn, *args, **kwargs):
        self._scalars = None
        return super()._apply(fn, *args, **kwargs)

    def out(self, x, pow=1.0):
        lambd = float(self.thres) ** float(pow)
        if self.training:
            return F.hardshrink(x, lambd=lambd)
        return F.hardshrink(F.relu(x), lambd=lambd)

    def _forward_modulewise(self, x):
        """models.py:123-132 module by module (every Conv3x3 still runs the HIP kernel); used for drop_rate > 0 and for the
        ConvTranspose2d upsampling variants."""
        if not x.is_cuda:
            raise RuntimeError("super-resolution_amd: the generator hot path only runs on a ROCm GPU tensor (no CPU fallback)")
        out1 = self.conv1(x)
        out = self.res_blocks(out1)
        out2 = self.conv2(out)
        out = torch.add(out1, out2)
        out = self.upsampling(out)
        if self.num_final_layer_res > 0:
            out3 = self.res_blocks_final(out)
            out = torch.add(out3, out)
        return self.conv3(out)

    def forward(self, x):
        power, mult = self._power_multiplier()
        if power != 1.0 or mult != 1.0:
            x = self.multiplier * (x ** self.power)
        if self.modulewise:
            out = self._forward_modulewise(x)        # Dropout2d in the dense blocks / ConvTranspose2d stages: no slot in the fused engine
        else:
            out = generator_raw(self._engine, x)
        if mult != 1.0:
            out = out / self.multiplier
        self.srs = self.out(out, power)
        if power != 1.0:
            out = F.relu(out) ** (1 / self.power)
        return self.out(out)


def _drop_scalar_cache(module, incompatible_keys):
    module._scalars = None


def discriminator_block(in_filters, out_filters, stride=(1, 2)):
    """models.py:140-146."""
    return [Conv3x3(in_filters, out_filters, kernel_size=3, stride=stride[0], padding=1), nn.LeakyReLU(0.2, inplace=True),
            Conv3x3(out_filters, out_filters, kernel_size=3, stride=stride[1], padding=1), nn.LeakyReLU(0.2, inplace=True)]


class Markovian_Discriminator(nn.Module):
    """PatchGAN discriminator, models.py:149-174.  Runs as a pre-activation NHWC chain
    z_l = conv(lrelu_0.2(z_{l-1})) on the srk kernels; twice differentiable w.r.t. the input (gradient penalty)."""

    def __init__(self, input_shape, channels=[16, 32, 32, 64]):
        super().__init__()
        self.channels, self.input_shape = channels, input_shape
        c_img, h, w = input_shape
        widths = [c_img] + list(channels)
        layers = []
        for cin, cout in zip(widths[:-1], widths[1:]):      # one (stride 1, stride 2) block per entry: the patch grid halves (ceil)
            layers += discriminator_block(cin, cout)
            h, w = -(-h // 2), -(-w // 2)
        layers.append(Conv3x3(widths[-1], 1, kernel_size=3, stride=1, padding=1))
        self.output_shape = (1, h, w)
        self.model = nn.Sequential(*layers)
        self._packed = None

    def forward(self, img, *args):
        convs = [m for m in self.model if isinstance(m, nn.Conv2d)]
        if self._packed is None or len(self._packed.convs) != len(convs):
            self._packed = ops.PackedConvs(convs)
        need_bwd = torch.is_grad_enabled() and (img.requires_grad or any(c.weight.requires_grad for c in convs))
        self._packed.refresh(need_bwd)
        z = ops.to_nhwc(img.float())
        # the whole chain as ONE autograd node: all nine launches of a pass from one library call (ops.ChainPre)
        meta, wb = [], []
        in_slope, ci = 1.0, 0
        for m in self.model:
            if isinstance(m, nn.Conv2d):
                meta.append((m.stride[0], in_slope, self._packed.fwd[ci], self._packed.bwd[ci] if need_bwd else None))
                wb += [m.weight, m.bias]
                ci += 1
                in_slope = 1.0
            elif isinstance(m, nn.LeakyReLU):
                in_slope = m.negative_slope        # applied while the next conv stages its input
            else:
                raise RuntimeError("unexpected layer in Markovian_Discriminator.model")
        return ops.to_nchw(ops.ChainPre.apply(z, meta, *wb))


class Standard_Discriminator(Markovian_Discriminator):
    """models.py:177-186: patch trunk without its last conv, plus a two-layer FC head (PyTorch Linear)."""

    def __init__(self, input_shape, channels):
        super().__init__(input_shape, channels)
        self.model = self.model[:-1]
        self.fc = nn.Sequential(nn.Linear(self.channels[-1] * self.output_shape[-2] * self.output_shape[-1], 1024), nn.ReLU(),
                                nn.Linear(1024, 1))
        self.output_shape = (1,)

    def forward(self, img, *args):
        z = ops.to_nhwc(img.float())
        in_slope = 1.0
        for m in self.model:
            if isinstance(m, nn.Conv2d):
                z = ops.conv_pre(z, m.weight, m.bias, m.stride[0], in_slope)
                in_slope = 1.0
            else:
                in_slope = m.negative_slope
        z = F.leaky_relu(z, in_slope)                 # trailing LeakyReLU of the last block
        return self.fc(ops.to_nchw(z).reshape(img.shape[0], -1))


def _run_preact_chain(seq, packed, z, in_slope, need_bwd):
    """z_l = conv(lrelu(z_{l-1})) over an nn.Sequential of Conv3x3 / LeakyReLU; returns (z, pending LeakyReLU slope)."""
    ci = 0
    for m in seq:
        if isinstance(m, nn.Conv2d):
            z = ops.conv_pre(z, m.weight, m.bias, m.stride[0], in_slope, packed.fwd[ci], packed.bwd[ci] if need_bwd else None)
            ci += 1
            in_slope = 1.0
        elif isinstance(m, nn.LeakyReLU):
            in_slope = m.negative_slope
        else:
            raise RuntimeError("unexpected layer in a discriminator conv chain")
    return z, in_slope


class Conditional_Discriminator(nn.Module):
    """models.py:189-223: the HR image runs through ``num_upsample`` stride-(1,2) blocks (``model_hr``), the LR condition
    through as many stride-(1,1) blocks (``model_c``); both land on the LR grid, are concatenated on the channel axis and
    finished by ``endmodel``.  Same pre-activation NHWC chains as Markovian_Discriminator (the LeakyReLU that closes both
    branches is applied while the first endmodel conv stages the concatenation); twice differentiable w.r.t. ``img``."""

    def __init__(self, input_shape, channels=[32, 64, 128, 256], num_upsample=3):
        super().__init__()
        self.channels, self.input_shape = channels, input_shape
        c_img, h, w = input_shape
        # block i < num_upsample: one stride-2 block on the image branch and one stride-1 block on the condition branch;
        # block num_upsample takes the concatenation (2x channels); every block halves the patch grid (ceil)
        branches = {"hr": [], "c": [], "end": []}
        widths = [c_img] + list(channels)
        for i, (cin, cout) in enumerate(zip(widths[:-1], widths[1:])):
            if i < num_upsample:
                branches["hr"] += discriminator_block(cin, cout)
                branches["c"] += discriminator_block(cin, cout, stride=(1, 1))
            else:
                branches["end"] += discriminator_block(2 * cin if i == num_upsample else cin, cout)
            h, w = -(-h // 2), -(-w // 2)
        branches["end"].append(Conv3x3(widths[-1], 1, kernel_size=3, stride=1, padding=1))
        self.output_shape = (1, h, w)
        self.model_hr = nn.Sequential(*branches["hr"])
        self.model_c = nn.Sequential(*branches["c"])
        self.endmodel = nn.Sequential(*branches["end"])
        self._packed = None

    def forward(self, img, cond):
        seqs = (self.model_hr, self.model_c, self.endmodel)
        convs = [[m for m in seq if isinstance(m, nn.Conv2d)] for seq in seqs]
        if self._packed is None:
            self._packed = [ops.PackedConvs(c) for c in convs]
        need_bwd = torch.is_grad_enabled() and (img.requires_grad or cond.requires_grad or
                                                any(c.weight.requires_grad for cs in convs for c in cs))
        for pk in self._packed:
            pk.refresh(need_bwd)
        zh, sh = _run_preact_chain(self.model_hr, self._packed[0], ops.to_nhwc(img.float()), 1.0, need_bwd)
        zc, sc = _run_preact_chain(self.model_c, self._packed[1], ops.to_nhwc(cond.float()), 1.0, need_bwd)
        if sh != sc:
            raise RuntimeError("Conditional_Discriminator: branches end in different activations")
        z, s = _run_preact_chain(self.endmodel, self._packed[2], torch.cat([zh, zc], 3), sh, need_bwd)
        return ops.to_nchw(z)


class SumPool2d(nn.Module):
    """models.py:297-305: k*k * AvgPool2d(k)."""

    def __init__(self, k=4, stride=None):
        super().__init__()
        if stride is not None and stride != k:
            raise NotImplementedError("SumPool2d with stride != k is not used by the reference's train path")
        self.k = k
        self.kernel_size = k * k

    def forward(self, x):
        return ops.sum_pool(x, self.k)


def weight_reset(m):
    """models.py:375-378."""
    if isinstance(m, nn.Conv2d) or isinstance(m, nn.Linear):
        m.reset_parameters()


def uniform_reset(m):
    """models.py:380-384."""
    if isinstance(m, nn.Conv2d):
        nn.init.xavier_uniform_(m.weight)
        nn.init.constant_(m.bias, 0.)
