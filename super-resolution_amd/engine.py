"""Host-side execution engine of the RRDB generator on the gfx950 kernels.

The reference runs ``GeneratorRRDB.forward`` (models.py:120-135) as ~1000 ATen ops with a ``torch.cat`` per
dense-block conv.  Here the whole generator is ONE autograd node: activations live in NHWC "dense buffers"
([N,H,W,5F] per DenseResidualBlock; conv k reads the channel prefix [0,kF) and writes slice k, so the
concatenation never exists), every conv is one ``srk_conv3x3`` launch with its elementwise neighbours fused,
and the backward pass is the mirror image: a second dense buffer holds dy5..dy1 in reverse slice order, so the
data-gradient of block input slice m is again a growing-prefix conv (no read-modify-write accumulation), with
LeakyReLU' fused as a mask in the epilogue.

Nothing here falls back to PyTorch convolutions: without libsrk.so / a GPU tensor it raises.
"""
import os
from typing import Dict, List, Optional

import torch

from . import _lib as L
from ._lib import View

G_SLOPE = 0.01            # nn.LeakyReLU() default slope (models.py:21,88,98)
INNER_RES_SCALE = 0.2     # DenseResidualBlock.res_scale; RRDB never forwards its own (models.py:49)


def _empty(*shape, device, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=device)


# precision modes with 16-BIT ACTIVATION STORAGE (BASELINE configs[4]): every activation / gradient buffer of the generator is
# fp16 ("fp16": with loss scaling, train.Stepper) or bf16 ("bf16s"); weights stay fp32 Parameters (packed into 16-bit fragments
# every step), accumulation is fp32, weight gradients come out in fp32.  Kernels: csrc/srk_conv_h16.hip, srk_wgrad_h16.hip.
H16_DTYPE = {"fp16": torch.float16, "bf16s": torch.bfloat16}


class _Flat:
    """Bump allocator over one flat fp32 tensor (16-byte aligned slices)."""

    def __init__(self):
        self.sizes = []

    def reserve(self, n):
        assert n % 4 == 0
        self.sizes.append(n)
        return len(self.sizes) - 1

    def materialize(self, device):
        total = sum(self.sizes)
        self.buf = torch.zeros(max(total, 4), dtype=torch.float32, device=device)
        self.slices = []
        off = 0
        for n in self.sizes:
            self.slices.append(self.buf[off:off + n])
            off += n


class _PackedW:
    """A packed weight slice plus the fragment format it was packed in (read by _lib.conv3x3)."""
    __slots__ = ("t", "fmt")

    def __init__(self, t, fmt):
        self.t, self.fmt = t, fmt

    def data_ptr(self):
        return self.t.data_ptr()


class DrbPack:
    """Packed weights of one DenseResidualBlock: forward convs k=1..5 and backward convs m=4..0."""

    def __init__(self, filters):
        self.F = filters
        self.fwd = [None] * 6     # fwd[k] flat index, k = 1..5
        self.bwd = [None] * 5     # bwd[m] flat index, m = 0..4


class GeneratorEngine:
    """Owns packed weights and runs forward/backward for one GeneratorRRDB module instance."""

    def __init__(self, gen):
        self.gen = gen
        self._sig = None
        self._sync = False
        self._grad_scale = 1.0
        # "f32": exact-fp32 MFMA everywhere (default).  "bf16x3": forward / data-gradient convs whose K is a multiple
        # of 16 run as 3 split-bf16 MFMAs per product (fp32 accumulate, ~2^-16 operand precision); opt-in.
        self.precision = os.environ.get("SRK_PRECISION", "f32")
        # 16-bit storage: LeakyReLU' masks of the data-gradient convolutions from sign bits the forward convolutions write (SRK_SIGN_BITS=0:
        # from the forward activations themselves)
        self.sign_bits = os.environ.get("SRK_SIGN_BITS", "1") != "0"
        self._signs_bytes = {}
        self.use_graphs = os.environ.get("SRK_GRAPHS", "0") == "1"     # hipGraph replay of forward / backward (_GraphSet)
        self._graphs = {}
        self._side = None            # second HIP stream: weight-gradient kernels overlap the data-gradient chain
        self._issue = None           # third stream, carries nothing but the issue points of the bucket all-reduces (_reduce_bucket)
        # weight gradients on the side stream: -0.95 ms per GAN iteration (133.2 -> 132.3, three same-box pairs).  Default: on, also
        # under data parallelism (round 3; SRK_DP_SCHEDULE=serial keeps a distributed run on one stream); SRK_OVERLAP_WGRAD=0 | 1
        # forces either.
        self._overlap_env = os.environ.get("SRK_OVERLAP_WGRAD")
        self._dp_serial = os.environ.get("SRK_DP_SCHEDULE", "overlap") == "serial"

    @property
    def act_dtype(self):
        """storage type of the activation / gradient buffers"""
        return H16_DTYPE.get(self.precision, torch.float32)

    def _wprec(self):
        """srk_wgrad_args.precision of this mode (0 exact fp32, 1 split-bf16, 2 bf16 operands, 3 / 4 fp16 / bf16 storage)"""
        if self.precision in H16_DTYPE:
            return L.WGRAD_PRECISION_OF_DTYPE[H16_DTYPE[self.precision]]
        return {"bf16x3": 1, "bf16": 2}.get(self.precision, 0) if self.gen.filters % 8 == 0 else 0

    @property
    def overlap_wgrad(self) -> bool:
        if self._overlap_env is not None:
            return self._overlap_env != "0"
        # (not while bench.py brackets every launch with events -- the per-kernel times must not overlap -- nor with hipGraph replay)
        return not (self._sync and self._dp_serial) and not self.use_graphs and not L.KernelTimer.active

    def __getstate__(self):
        """Pickling / torch.save(module) / multiprocessing spawn: everything but the module reference and the flags is a cache
        (packed weights, pack tables, graphs, streams) that is rebuilt on the first forward."""
        keep = ("gen", "_sync", "_grad_scale", "precision", "use_graphs", "_overlap_env", "_dp_serial", "sign_bits")
        st = {k: self.__dict__[k] for k in keep}
        st.update(_sig=None, _graphs={}, _side=None, _issue=None, _signs_bytes={})
        return st

    # ------------------------------------------------------------------ data-parallel gradient exchange
    def enable_grad_sync(self, enabled: bool = True):
        """Average weight gradients across ranks (RCCL all-reduce via torch.distributed) inside backward().
        Gradients are produced into one flat buffer; each bucket (tail convs, every RRDB, conv1) is reduced with an
        asynchronous all-reduce as soon as its last wgrad has been launched, so the exchange of RRDB i overlaps the
        backward convolutions of RRDBs i-1..0."""
        self._sync = enabled
        self._grad_scale = 1.0
        if enabled:
            import torch.distributed as dist
            self._grad_scale = 1.0 / dist.get_world_size()   # folded into the wgrad reduction; the exchange is a SUM

    def _alloc_grads(self, device):
        ps = self.params()
        sizes = [p.numel() for p in ps]
        flat = torch.empty(sum(sizes), dtype=torch.float32, device=device)
        grads, off, offs = {}, 0, []
        for p, n in zip(ps, sizes):
            grads[p] = flat[off:off + n].view_as(p)
            offs.append(off)
            off += n
        offs.append(off)
        R = len(self.gen.res_blocks)
        # param-index ranges: conv1 | RRDB 0..R-1 (30 tensors each) | tail (conv2, upsampling, final RRDBs, conv3)
        self._bucket = {"conv1": (offs[0], offs[2]), "tail": (offs[2 + 30 * R], offs[-1])}
        for i in range(R):
            self._bucket[i] = (offs[2 + 30 * i], offs[2 + 30 * (i + 1)])
        self._flat_grad = flat
        self._works = []
        return grads

    def _reduce_bucket(self, key):
        """Start the all-reduce of one gradient bucket.  Its weight gradients were launched on the main stream and (with the
        weight-gradient side stream on) on the side stream: the collective is issued from a third, otherwise empty "issue" stream that
        waits for THIS point of both, so neither the data-gradient chain nor the weight-gradient stream ever waits for a collective.
        torch.distributed enqueues every collective of a process group on the group's own stream in HOST call order -- which is the
        same on every rank (same code, same shapes) -- and makes it wait for the stream it was issued from: the order of collectives
        on the communicator does not depend on the issuing streams, and a collective only ever waits for work that was queued before
        it, so the schedule cannot introduce a cycle."""
        if not self._sync:
            return
        import torch.distributed as dist
        a, b = self._bucket[key]
        if self._side is not None and self.overlap_wgrad:
            if self._issue is None:
                self._issue = torch.cuda.Stream()
            self._issue.wait_stream(torch.cuda.current_stream())
            self._issue.wait_stream(self._side)
            self._flat_grad.record_stream(self._issue)
            with torch.cuda.stream(self._issue):
                self._works.append(dist.all_reduce(self._flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))
            return
        self._works.append(dist.all_reduce(self._flat_grad[a:b], op=dist.ReduceOp.SUM, async_op=True))

    def abandon_iteration(self):
        """After a chain fault (train.Stepper.recover_chain_fault; the device is idle): forget what the interrupted iteration left behind.
        Every buffer of an iteration lives in its autograd node; what the engine itself holds are the handles of bucket all-reduces
        and the per-geometry sign-bit decisions, which depend on which kernel form a sequence takes (the chain forms now rest)."""
        self._works = []
        self._signs_bytes = {}
        self._graphs = {}

    def _finish_reduce(self):
        self._join_side()
        for w in self._works:          # (the current = main stream waits for every bucket)
            w.wait()
        self._works = []

    # ------------------------------------------------------------------ parameter bookkeeping
    def _conv_modules(self):
        g = self.gen
        convs = [("conv1", g.conv1)]
        for i, rr in enumerate(g.res_blocks):
            for j, d in enumerate(rr.dense_blocks):
                for k in range(1, 6):
                    convs.append((f"res_blocks.{i}.dense_blocks.{j}.b{k}", getattr(d, f"b{k}")[0]))
        convs.append(("conv2", g.conv2))
        for u in range(g.num_upsample):
            convs.append((f"upsampling.{3*u}", g.upsampling[3 * u]))
        if g.num_final_layer_res > 0:
            for i, rr in enumerate(g.res_blocks_final):
                for j, d in enumerate(rr.dense_blocks):
                    for k in range(1, 6):
                        convs.append((f"res_blocks_final.{i}.dense_blocks.{j}.b{k}", getattr(d, f"b{k}")[0]))
        convs.append(("conv3.0", g.conv3[0]))
        convs.append(("conv3.2", g.conv3[2]))
        return convs

    def params(self) -> List[torch.nn.Parameter]:
        # called twice per iteration (forward + backward) on the host's critical path right after the D-gate sync: the module
        # list is fixed after construction, and Parameters are read from the modules' dicts (nn.Module.__getattr__ is slow)
        mods = getattr(self, "_conv_mods", None)
        if mods is None:
            mods = self._conv_mods = [m for _, m in self._conv_modules()]
        ps = []
        for m in mods:
            pd = m._parameters
            ps.append(pd["weight"]); ps.append(pd["bias"])
        return ps

    def _build_tables(self, device, geo):
        g = self.gen
        gN, gH, gW = geo                      # LR input extent of the forward that triggered the build
        F_, C_ = g.filters, g.channels
        self.flat_f, self.flat_b = _Flat(), _Flat()
        # one table per (direction, fragment format)
        self.tab_f = {f: L.PackTable(device, f) for f in (0, 1, 3, 5, 6, 7, 8)}
        self.tab_b = {f: L.PackTable(device, f) for f in (0, 1, 3, 5, 6, 7, 8)}
        self.fmt_f, self.fmt_b = {}, {}      # flat index -> fragment format of that packed conv
        jobs_f, jobs_b = [], []   # deferred (need materialized dst)
        bf = self.precision in ("bf16x3", "bf16")
        h16 = self.precision in H16_DTYPE          # 16-bit storage: ONE format for every conv, image-side channels zero-padded to 32
        hfmt = L.FMT_OF_DTYPE[H16_DTYPE[self.precision]] if h16 else 0
        if h16 and F_ % 32 != 0:
            raise NotImplementedError("the 16-bit-storage modes need filters % 32 == 0")
        self._built_precision = self.precision

        def kpad(k):
            return (k + 31) // 32 * 32 if h16 else k       # the 16-bit conv kernel stages 32 input channels at a time

        # exact-fp32 mode: stride-1 convs with 64-multiple outputs run the Winograd F(2,3)-along-W kernel (2/3 of the MFMAs)
        wino = (not bf) and os.environ.get("SRK_WINOGRAD", "1") != "0"

        wino4 = wino and os.environ.get("SRK_WINOGRAD4", "1") != "0"
        # the same tiles through the 2-D F(2x4, 3x3) kernel (a third instead of half of the MFMAs); SRK_WINOGRAD42=0: F(4,3)
        w4fmt = 6 if os.environ.get("SRK_WINOGRAD42", "1") != "0" else 5

        def fmt_of(K, M, up=1):
            """fragment format of a conv with K inputs, M outputs running at `up` x the LR resolution"""
            if h16:
                return hfmt
            if bf:
                return 1 if (K % 16 == 0 and M >= 16) else 0
            if not (wino and L.wino_eligible(K, M)):
                return 0
            # F(4,3) works on 32 x 16 tiles, one 8-wave workgroup per CU: worth it when those tiles fill the chip and do not
            # pad the image more than the 16 x 16 tiles of the F(2,3) kernel would
            return w4fmt if (wino4 and self._wino4_ok(geo, up, M // 64)) else 3

        def simple(name, conv, ps=False, need_bwd=True, up=1):
            co, ci = conv.weight.shape[:2]
            ff = fmt_of(ci, co, up) if (h16 or not ps or (co // 4) % 4 == 0) else 0
            fi = self.flat_f.reserve(L.packed_floats(kpad(ci), co, ff))
            self.fmt_f[fi] = ff
            jobs_f.append((conv.weight, fi, dict(M=co, k_off=0, k_len=ci, K_total=kpad(ci), ps=ps)))
            bi = None
            if need_bwd:
                fb = fmt_of(co, ci, up) if (h16 or not ps or (co // 4) % 16 == 0) else 0
                if fb in (3, 5, 6) and conv.stride[0] != 1:
                    fb = 0
                bi = self.flat_b.reserve(L.packed_floats(kpad(co), ci, fb))
                self.fmt_b[bi] = fb
                jobs_b.append((conv.weight, bi, dict(M=ci, k_off=0, k_len=co, K_total=kpad(co), transpose=True, ps=ps)))
            return fi, bi

        def drb(d, s5, up=1):
            p = DrbPack(F_)
            for k in range(1, 6):
                w = getattr(d, f"b{k}")[0].weight
                ff = fmt_of(k * F_, F_, up)
                p.fwd[k] = self.flat_f.reserve(L.packed_floats(k * F_, F_, ff))
                self.fmt_f[p.fwd[k]] = ff
                jobs_f.append((w, p.fwd[k], dict(M=F_, k_off=0, k_len=k * F_, K_total=k * F_)))
            for m in range(0, 5):
                K = (5 - m) * F_
                fb = fmt_of(K, F_, up) if (F_ % 16 == 0 or not bf) else 0
                p.bwd[m] = self.flat_b.reserve(L.packed_floats(K, F_, fb))
                self.fmt_b[p.bwd[m]] = fb
                for k in range(5, m, -1):     # input slice (5-k) of the dy buffer carries dy_k
                    w = getattr(d, f"b{k}")[0].weight
                    jobs_b.append((w, p.bwd[m], dict(M=F_, k_off=(5 - k) * F_, k_len=F_, K_total=K, transpose=True,
                                                     c_begin=m * F_, scale=(s5 if k == 5 else 1.0))))
            p.s5 = s5
            return p

        self.idx = {}
        self.idx["conv1"] = simple("conv1", g.conv1)
        self.drbs = []
        for rr in g.res_blocks:
            self.drbs.append([drb(d, INNER_RES_SCALE * (rr.res_scale if j == 2 else 1.0)) for j, d in enumerate(rr.dense_blocks)])
        self.idx["conv2"] = simple("conv2", g.conv2)
        for u in range(g.num_upsample):
            self.idx[f"up{u}"] = simple(f"up{u}", g.upsampling[3 * u], ps=True, up=2 ** u)
        hr_up = 2 ** g.num_upsample
        self.drbs_final = []
        if g.num_final_layer_res > 0:
            for rr in g.res_blocks_final:
                self.drbs_final.append([drb(d, INNER_RES_SCALE * (rr.res_scale if j == 2 else 1.0), up=hr_up) for j, d in enumerate(rr.dense_blocks)])
        self.idx["conv3.0"] = simple("conv3.0", g.conv3[0], up=hr_up)
        self.idx["conv3.2"] = simple("conv3.2", g.conv3[2], up=hr_up)

        self.flat_f.materialize(device)
        self.flat_b.materialize(device)
        for w, fi, kw in jobs_f:
            self.tab_f[self.fmt_f[fi]].add(w.data, self.flat_f.slices[fi], **kw)
        for w, bi, kw in jobs_b:
            self.tab_b[self.fmt_b[bi]].add(w.data, self.flat_b.slices[bi], **kw)
        for t in list(self.tab_f.values()) + list(self.tab_b.values()):
            t.finalize()

    @staticmethod
    def _wino4_ok(geo, up, mtiles):
        """THE rule for F(4,3) vs F(2,3): a conv at `up` x the LR extent with `mtiles` 64-channel output tiles runs the F(4,3)
        kernel (32 x 16 pixel tiles, one 8-wave workgroup per CU) when those tiles fill the chip (>= 200 workgroups) and pad the
        image no more than the 16-row tiles of the F(2,3) kernel would."""
        gN, gH, gW = geo
        h, w = gH * up, gW * up
        if gN * ((h + 31) // 32) * ((w + 15) // 16) * mtiles >= 200 and ((h + 31) // 32) * 32 == ((h + 15) // 16) * 16:
            return True
        # the F(2x4,3x3) kernel also has a 16-row-tile form (picked by its launcher): enough when THOSE tiles fill the chip
        return os.environ.get("SRK_WINOGRAD42", "1") != "0" and gN * ((h + 15) // 16) * ((w + 15) // 16) * mtiles >= 200

    def _wino4_levels(self, geo, mtiles=1):
        """which resolution levels (1x, 2x, ... of the LR extent) run the F(4,3) kernel for a conv with `mtiles` output tiles"""
        return tuple(self._wino4_ok(geo, 2 ** u, mtiles) for u in range(self.gen.num_upsample + 1))

    def _ensure_packed(self, need_bwd: bool, geo=None):
        """(Re)pack the weights.  The canonical OIHW Parameters stay the source of truth (optimizer steps,
        load_state_dict, weight_reset all write them); packing is one kernel launch over the whole table, so it
        is simply redone on every forward (fwd table) / backward (bwd table) instead of tracking versions."""
        ps = self.params()
        dev = ps[0].device
        geo = geo or getattr(self, "_geo", (1, 16, 16))
        self._geo = geo
        # the tables depend on the geometry only through the F(4,3)-vs-F(2,3) decisions: per resolution level, for the 64-output
        # convs (1 output tile) and the 4F PixelShuffle convs (4 tiles) -- the same rule fmt_of applies (_wino4_ok)
        sig = (dev, self.precision, self._wino4_levels(geo, 1), self._wino4_levels(geo, 4), tuple(p.data_ptr() for p in ps))
        if sig != self._sig:
            self._build_tables(dev, geo)
            self._sig = sig
        if need_bwd:
            for t in self.tab_b.values():
                t.run()
            return
        for t in self.tab_f.values():
            t.run()
        # packed-order biases of the PixelShuffle convs (o' = ij*F + c  <->  o = 4c + ij)
        self.ps_bias = {}
        for u in range(self.gen.num_upsample):
            b = self.gen.upsampling[3 * u].bias.data
            self.ps_bias[u] = b.view(-1, 4).t().contiguous().view(-1)

    def wf(self, i):          # wp_format: 0 fp32, 1 split-bf16, 2 plain bf16 (same packing as 1)
        f = self.fmt_f[i]
        return _PackedW(self.flat_f.slices[i], 2 if (f == 1 and self.precision == "bf16") else f)

    def wb(self, i):
        f = self.fmt_b[i]
        return _PackedW(self.flat_b.slices[i], 2 if (f == 1 and self.precision == "bf16") else f)

    # ------------------------------------------------------------------ building blocks
    def _drb_forward(self, d, pk: DrbPack, D, out: View, geo, outer_x: Optional[View], rs: float, save: bool = False):
        """One DenseResidualBlock on dense buffer D (slice 0 = block input).  ``outer_x`` is the RRDB input
        for the third block (its conv5 epilogue also applies ``*res_scale + x``, models.py:53).  ``save``: convs 1-4 also write the SIGN
        BITS of their outputs (srk_conv_args.signs: 1 MB instead of the 16.8 / 33.5 MB slice) where the launches offer them; the
        block's data-gradient convolutions then take their LeakyReLU' masks from those.  (Whether they do is decided once per geometry
        and precision: the chain switches are process-wide settings, not something that changes between iterations.)"""
        N, H, W = geo
        F_ = pk.F
        calls = []          # the block's five convolutions go to the library in ONE call (srk_conv3x3_seq)
        for k in range(1, 5):
            calls.append((View(D, 0, k * F_), self.wf(pk.fwd[k]), getattr(d, f"b{k}")[0].bias.data, View(D, k * F_, F_),
                          dict(N=N, H=H, W=W, OH=H, OW=W, Cin=k * F_, Cout=F_, slope=G_SLOPE)))
        b5 = d.b5[0].bias.data
        if outer_x is None:
            calls.append((View(D, 0, 5 * F_), self.wf(pk.fwd[5]), b5, out, dict(N=N, H=H, W=W, OH=H, OW=W, Cin=5 * F_, Cout=F_,
                          alpha=INNER_RES_SCALE, r1=View(D, 0, F_), beta1=1.0)))
        else:
            calls.append((View(D, 0, 5 * F_), self.wf(pk.fwd[5]), b5, out, dict(N=N, H=H, W=W, OH=H, OW=W, Cin=5 * F_, Cout=F_,
                          alpha=INNER_RES_SCALE * rs, r1=View(D, 0, F_), beta1=rs, r2=outer_x, beta2=1.0)))
        # sign bits of the outputs of convs 1-4 for the block's data-gradient convolutions (1 MB per conv instead of the 16.8 / 33.5 MB
        # slice), where this sequence's launches offer them (16-bit storage: always; fp32: when it goes out as a chain kernel)
        signs, tag = None, 0
        if save and self.sign_bits and not self.use_graphs:
            # (decided once per geometry, precision and DISPATCH GENERATION: _lib.dispatch_gen moves whenever a switch that changes the
            # kernel form of a launch is thrown -- debug setters, a recovered chain fault)
            key = ("f", N, H, W, F_, self.precision, calls[0][1].fmt, L.dispatch_gen)
            ent = self._signs_bytes.get(key)
            if ent is None:
                ent = self._signs_bytes[key] = L.conv_seq_signs(calls)
            nb, tag = ent
            if nb > 0:
                signs = torch.empty(4, nb, dtype=torch.uint8, device=D.device)
                for k in range(4):
                    calls[k][4]["signs_out"] = signs[k]
        # the layout tag travels with the bits: the data-gradient sequence reads them only if ITS launches use the same layout
        D._srk_signs, D._srk_signs_tag = signs, tag
        L.conv3x3_seq(calls)

    def _drb_backward(self, d, pk: DrbPack, D, E, gx_out: View, geo, beta_self: float, outer_g: Optional[View], grads: Dict):
        """Backward of one DenseResidualBlock.  E slice 0 holds the (unscaled) gradient of the block output; slices
        1..4 receive dy4..dy1.  Writes the block-input gradient to ``gx_out``."""
        N, H, W = geo
        F_ = pk.F
        calls = []          # the five data-gradient convolutions in ONE library call
        signs = getattr(D, "_srk_signs", None)        # written by the block's forward convolutions
        for use_signs in ((True, False) if signs is not None else (False,)):
            calls = []
            for m in range(4, 0, -1):
                K = (5 - m) * F_
                kw = dict(N=N, H=H, W=W, OH=H, OW=W, Cin=K, Cout=F_, mask_slope=G_SLOPE)
                if use_signs:
                    kw["mask_signs"] = signs[m - 1]
                else:
                    kw["mask"] = View(D, m * F_, F_)
                calls.append((View(E, 0, K), self.wb(pk.bwd[m]), None, View(E, K, F_), kw))
            if not use_signs:
                break
            # (fp32: only the chain form of THIS sequence reads sign bits; decided once per geometry)
            key = ("b", N, H, W, F_, self.precision, calls[0][1].fmt, L.dispatch_gen)
            ent = self._signs_bytes.get(key)
            if ent is None:
                ent = self._signs_bytes[key] = L.conv_seq_signs(calls + [(View(E, 0, 5 * F_), self.wb(pk.bwd[0]), None, gx_out,
                                                                          dict(N=N, H=H, W=W, OH=H, OW=W, Cin=5 * F_, Cout=F_))])
            if ent[0] > 0 and ent[1] == getattr(D, "_srk_signs_tag", 0) and ent[0] == signs.shape[1]:
                break
        calls.append((View(E, 0, 5 * F_), self.wb(pk.bwd[0]), None, gx_out, dict(N=N, H=H, W=W, OH=H, OW=W, Cin=5 * F_, Cout=F_,
                      r1=View(E, 0, F_), beta1=beta_self, r2=outer_g, beta2=1.0)))
        L.conv3x3_seq(calls)
        # weight gradients: conv k reads D[0:kF), its dy is E slice (5-k) (k=5: slice 0 scaled by s5).
        # One batched launch for the five convs (15 chunks of 64x64x9).
        probs = []
        for k in range(1, 6):
            conv = getattr(d, f"b{k}")[0]
            probs.append(dict(x=View(D, 0, k * F_), dy=View(E, (5 - k) * F_, F_), dw=grads[conv.weight], db=grads[conv.bias],
                              Cin=k * F_, Cout=F_, scale=(pk.s5 if k == 5 else 1.0) * self._grad_scale))
        wprec = self._wprec()
        self._on_side(lambda: L.conv3x3_wgrad_batched(probs, N=N, H=H, W=W, OH=H, OW=W, precision=wprec), (D, E))

    def _on_side(self, fn, tensors):
        """Run ``fn`` (weight-gradient launches) on the side stream, ordered after everything already queued on the
        current stream.  The data-gradient chain on the main stream never waits for it, so the two kernels share
        the CUs and fill each other's prologue / epilogue / tail bubbles.  ``tensors`` are kept alive for the side
        stream (caching-allocator ``record_stream``)."""
        if not self.overlap_wgrad:
            fn()
            return
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream()
        ev = torch.cuda.Event()
        ev.record(main)
        self._side.wait_event(ev)
        with torch.cuda.stream(self._side):
            fn()
        for t in tensors:
            t.record_stream(self._side)

    def _join_side(self):
        if self._side is not None and self.overlap_wgrad:
            torch.cuda.current_stream().wait_stream(self._side)

    def _rrdb_chain_forward(self, rrdbs, packs, x0: torch.Tensor, geo, save: bool):
        """Runs a chain of RRDBs.  x0: dense buffer [N,H,W,5F] whose slice 0 already holds the chain input.
        Returns (list of dense buffers, output tensor [N,H,W,F])."""
        N, H, W = geo
        F_ = self.gen.filters
        dev = x0.device
        bufs = [x0]
        cur = x0
        n_r = len(rrdbs)
        for i, rr in enumerate(rrdbs):
            first = cur
            for j, d in enumerate(rr.dense_blocks):
                last = (i == n_r - 1 and j == 2)
                if last:
                    nxt = _empty(N, H, W, F_, device=dev, dtype=self.act_dtype)
                    out = View(nxt)
                else:
                    nxt = _empty(N, H, W, 5 * F_, device=dev, dtype=self.act_dtype)
                    out = View(nxt, 0, F_)
                self._drb_forward(d, packs[i][j], cur, out, geo, View(first, 0, F_) if j == 2 else None, rr.res_scale, save)
                if not last:
                    bufs.append(nxt)
                cur = nxt
        return bufs, cur

    def _rrdb_chain_backward(self, rrdbs, packs, bufs, g_out: torch.Tensor, geo, grads, sync=False):
        """g_out: [N,H,W,F] gradient of the chain output.  Returns gradient of the chain input [N,H,W,F]."""
        N, H, W = geo
        F_ = self.gen.filters
        dev = g_out.device
        n_r = len(rrdbs)
        # E buffer of block (i,2) must stay alive until block (i,0) finishes (it is the outer residual G)
        E_next = _empty(N, H, W, 5 * F_, device=dev, dtype=self.act_dtype)
        E_next[..., :F_].copy_(g_out)
        result = None
        for i in range(n_r - 1, -1, -1):
            rr = rrdbs[i]
            E_outer = E_next
            E_cur = E_next
            for j in (2, 1, 0):
                D = bufs[3 * i + j]
                final = (i == 0 and j == 0)
                if final:
                    result = _empty(N, H, W, F_, device=dev, dtype=self.act_dtype)
                    gx = View(result)
                else:
                    E_prev = _empty(N, H, W, 5 * F_, device=dev, dtype=self.act_dtype)
                    gx = View(E_prev, 0, F_)
                self._drb_backward(rr.dense_blocks[j], packs[i][j], D, E_cur, gx, geo,
                                   beta_self=(rr.res_scale if j == 2 else 1.0),
                                   outer_g=(View(E_outer, 0, F_) if j == 0 else None), grads=grads)
                if not final:
                    E_cur = E_prev
            if sync:
                self._reduce_bucket(i)
            E_next = E_cur
        return result

    # ------------------------------------------------------------------ whole generator
    def forward(self, x: torch.Tensor, need_grad: bool):
        """x: NCHW fp32 CUDA (already power/multiplier-scaled).  Returns (raw NCHW output of conv3, saved)."""
        g = self.gen
        if not x.is_cuda:
            raise RuntimeError("super-resolution_amd: the generator hot path only runs on a ROCm GPU tensor "
                               "(no CPU fallback; use oracle/ for a CPU check in tests)")
        N, C_, H, W = x.shape
        self._ensure_packed(need_bwd=False, geo=(N, H, W))
        assert C_ == g.channels
        F_ = g.filters
        dev = x.device
        x = x.contiguous().float()
        if C_ == 1:
            x_nhwc = x.view(N, H, W, 1)
        else:
            x_nhwc = _empty(N, H, W, C_, device=dev)
            L.nchw_to_nhwc(x, View(x_nhwc), N, C_, H, W)
        geo = (N, H, W)
        h16 = self.precision in H16_DTYPE
        cin1 = C_
        if h16:
            # 16-bit storage: the image enters zero-padded to one 32-channel stage (conv1's packed weights are padded alike)
            cin1 = 32
            x_pad = torch.zeros(N, H, W, cin1, dtype=self.act_dtype, device=dev)
            x_pad[..., :C_] = x_nhwc
            x_nhwc = x_pad
        D0 = _empty(N, H, W, 5 * F_, device=dev, dtype=self.act_dtype)
        # conv1 -> slice 0 of the first dense buffer (= out1, also the trunk skip; models.py:123)
        L.conv3x3(View(x_nhwc), self.wf(self.idx["conv1"][0]), g.conv1.bias.data, View(D0, 0, F_), N=N, H=H, W=W, OH=H, OW=W,
                  Cin=cin1, Cout=F_)
        if len(g.res_blocks) > 0:
            bufs, trunk = self._rrdb_chain_forward(list(g.res_blocks), self.drbs, D0, geo, need_grad)
            trunk_v = View(trunk)
        else:
            bufs, trunk = [D0], D0
            trunk_v = View(D0, 0, F_)
        # conv2 + trunk skip (models.py:125-126)
        feat = _empty(N, H, W, F_, device=dev, dtype=self.act_dtype)
        L.conv3x3(trunk_v, self.wf(self.idx["conv2"][0]), g.conv2.bias.data, View(feat), N=N, H=H, W=W, OH=H, OW=W, Cin=F_, Cout=F_,
                  r1=View(D0, 0, F_), beta1=1.0)
        # upsampling: conv F->4F + LeakyReLU + PixelShuffle(2) fused into the store (models.py:86-90)
        ups = []
        cur, h, w = feat, H, W
        for u in range(g.num_upsample):
            last_up = (u == g.num_upsample - 1)
            if last_up and g.num_final_layer_res > 0:
                nxt = _empty(N, 2 * h, 2 * w, 5 * F_, device=dev, dtype=self.act_dtype)   # doubles as first dense buffer of the final RRDBs
                yv = View(nxt, 0, F_)
            else:
                nxt = _empty(N, 2 * h, 2 * w, F_, device=dev, dtype=self.act_dtype)
                yv = View(nxt)
            L.conv3x3(View(cur) if cur.shape[3] == F_ else View(cur, 0, F_), self.wf(self.idx[f"up{u}"][0]), self.ps_bias[u], yv,
                      N=N, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=4 * F_, ps_out=True, slope=G_SLOPE)
            ups.append(nxt)
            cur, h, w = nxt, 2 * h, 2 * w
        fin = None
        if g.num_final_layer_res > 0:
            if g.num_upsample == 0:
                raise NotImplementedError("num_final_layer_res > 0 requires num_upsample >= 1 in this build")
            fbufs, fout = self._rrdb_chain_forward(list(g.res_blocks_final), self.drbs_final, cur, (N, h, w), need_grad)
            pre3 = _empty(N, h, w, F_, device=dev, dtype=self.act_dtype)
            torch.add(fout, cur[..., :F_], out=pre3)      # out = out3 + out (models.py:130)
            fin = (fbufs, pre3)
            cur_v = View(pre3)
        else:
            cur_v = View(cur)
        h3 = _empty(N, h, w, F_, device=dev, dtype=self.act_dtype)
        L.conv3x3(cur_v, self.wf(self.idx["conv3.0"][0]), g.conv3[0].bias.data, View(h3), N=N, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=F_,
                  slope=G_SLOPE)
        out_nhwc = _empty(N, h, w, C_, device=dev)
        L.conv3x3(View(h3), self.wf(self.idx["conv3.2"][0]), g.conv3[2].bias.data, View(out_nhwc), N=N, H=h, W=w, OH=h, OW=w,
                  Cin=F_, Cout=C_, flags=(L.CONV_OUT_F32 if h16 else 0))        # (16-bit modes: the last conv writes the fp32 image)
        if C_ == 1:
            out = out_nhwc.view(N, 1, h, w)
        else:
            out = _empty(N, C_, h, w, device=dev)
            L.nhwc_to_nchw(View(out_nhwc), out, N, C_, h, w)
        saved = None
        if need_grad:
            saved = dict(x=x_nhwc, bufs=bufs, trunk=trunk, feat=feat, ups=ups, fin=fin, h3=h3, geo=geo, hr=(h, w))
        return out, saved

    def backward(self, saved, g_out: torch.Tensor, need_input_grad: bool):
        """g_out: NCHW gradient of the raw conv3 output.  Returns (dx NCHW or None, {param: grad})."""
        g = self.gen
        N, H, W = saved["geo"]
        self._ensure_packed(need_bwd=True, geo=(N, H, W))
        h, w = saved["hr"]
        F_, C_ = g.filters, g.channels
        dev = g_out.device
        g_out = g_out.contiguous().float()
        h16 = self.precision in H16_DTYPE
        if h16:
            # 16-bit storage: the (loss-scaled) output gradient enters zero-padded to one 32-channel stage
            go = torch.zeros(N, h, w, 32, dtype=self.act_dtype, device=dev)
            go[..., :C_] = g_out.permute(0, 2, 3, 1)
            cgo = 32
        elif C_ == 1:
            go = g_out.view(N, h, w, 1)
            cgo = C_
        else:
            go = _empty(N, h, w, C_, device=dev)
            L.nchw_to_nhwc(g_out, View(go), N, C_, h, w)
            cgo = C_
        grads = self._alloc_grads(dev)
        ups, h3 = saved["ups"], saved["h3"]
        fin = saved["fin"]
        pre3_in = fin[1] if fin is not None else (ups[-1] if ups else saved["feat"])
        pre3_v = View(pre3_in) if pre3_in.shape[3] == F_ else View(pre3_in, 0, F_)

        # conv3.2 (F -> C) then conv3.0 (F -> F, LeakyReLU)
        c32, c30 = g.conv3[2], g.conv3[0]
        L.conv3x3_wgrad(View(h3), View(go), grads[c32.weight], grads[c32.bias], N=N, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=C_, scale=self._grad_scale,
                        precision=(self._wprec() if h16 else 0))
        g_h3 = _empty(N, h, w, F_, device=dev, dtype=self.act_dtype)
        L.conv3x3(View(go), self.wb(self.idx["conv3.2"][1]), None, View(g_h3), N=N, H=h, W=w, OH=h, OW=w, Cin=cgo, Cout=F_,
                  mask=View(h3), mask_slope=G_SLOPE)
        L.conv3x3_wgrad(pre3_v, View(g_h3), grads[c30.weight], grads[c30.bias], N=N, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=F_, scale=self._grad_scale, precision=self._wprec())
        g_cur = _empty(N, h, w, F_, device=dev, dtype=self.act_dtype)
        # the tensor feeding conv3.0 is a LeakyReLU output (last upsample stage) unless final RRDBs / no upsampling sit between
        mask_v = None
        if fin is None and ups:
            mask_v = View(ups[-1])
        L.conv3x3(View(g_h3), self.wb(self.idx["conv3.0"][1]), None, View(g_cur), N=N, H=h, W=w, OH=h, OW=w, Cin=F_, Cout=F_,
                  mask=mask_v, mask_slope=G_SLOPE)
        if fin is not None:
            fbufs, _ = fin
            g_chain = self._rrdb_chain_backward(list(g.res_blocks_final), self.drbs_final, fbufs, g_cur, (N, h, w), grads)
            # out = out3 + out: both paths; then LeakyReLU' of the upsample output
            up_last = ups[-1]
            g_sum = g_chain + g_cur
            g_cur = (g_sum * torch.where(up_last[..., :F_] > 0, 1.0, G_SLOPE).to(g_sum.dtype)).contiguous()

        # upsampling stages, last to first
        hh, ww = h, w
        for u in range(g.num_upsample - 1, -1, -1):
            conv = g.upsampling[3 * u]
            hh, ww = hh // 2, ww // 2
            xin = ups[u - 1] if u > 0 else saved["feat"]
            xin_v = View(xin) if xin.shape[3] == F_ else View(xin, 0, F_)
            L.conv3x3_wgrad(xin_v, View(g_cur), grads[conv.weight], grads[conv.bias], N=N, H=hh, W=ww, OH=hh, OW=ww, Cin=F_, Cout=4 * F_,
                            dy_mode=L.IN_UNSHUFFLE, scale=self._grad_scale, precision=self._wprec())
            g_prev = _empty(N, hh, ww, F_, device=dev, dtype=self.act_dtype)
            L.conv3x3(View(g_cur), self.wb(self.idx[f"up{u}"][1]), None, View(g_prev), N=N, H=hh, W=ww, OH=hh, OW=ww, Cin=4 * F_, Cout=F_,
                      in_mode=L.IN_UNSHUFFLE, mask=(View(ups[u - 1]) if u > 0 else None), mask_slope=G_SLOPE)
            g_cur = g_prev
        g_feat = g_cur                                   # gradient of out1 + out2
        # conv2
        trunk = saved["trunk"]
        bufs = saved["bufs"]
        D0 = bufs[0]
        trunk_v = View(trunk) if trunk.shape[3] == F_ else View(trunk, 0, F_)
        L.conv3x3_wgrad(trunk_v, View(g_feat), grads[g.conv2.weight], grads[g.conv2.bias], N=N, H=H, W=W, OH=H, OW=W, Cin=F_, Cout=F_, scale=self._grad_scale, precision=self._wprec())
        g_trunk = _empty(N, H, W, F_, device=dev, dtype=self.act_dtype)
        L.conv3x3(View(g_feat), self.wb(self.idx["conv2"][1]), None, View(g_trunk), N=N, H=H, W=W, OH=H, OW=W, Cin=F_, Cout=F_)
        self._reduce_bucket("tail")
        if len(g.res_blocks) > 0:
            g_out1 = self._rrdb_chain_backward(list(g.res_blocks), self.drbs, bufs, g_trunk, (N, H, W), grads, sync=True)
        else:
            g_out1 = g_trunk
        g_out1 = g_out1 + g_feat                         # trunk skip (models.py:126)
        # conv1
        L.conv3x3_wgrad(View(saved["x"]), View(g_out1), grads[g.conv1.weight], grads[g.conv1.bias], N=N, H=H, W=W, OH=H, OW=W,
                        Cin=C_, Cout=F_, scale=self._grad_scale, precision=(self._wprec() if h16 else 0))
        self._reduce_bucket("conv1")
        dx = None
        if need_input_grad:
            dxn = _empty(N, H, W, C_, device=dev)
            L.conv3x3(View(g_out1), self.wb(self.idx["conv1"][1]), None, View(dxn), N=N, H=H, W=W, OH=H, OW=W, Cin=F_, Cout=C_,
                      flags=(L.CONV_OUT_F32 if h16 else 0))
            if C_ == 1:
                dx = dxn.view(N, 1, H, W)
            else:
                dx = _empty(N, C_, H, W, device=dev)
                L.nhwc_to_nchw(View(dxn), dx, N, C_, H, W)
        self._finish_reduce()
        return dx, grads


class _GraphSet:
    """hipGraph capture of the generator's forward and backward launch sequences for one (shape, mode).

    engine.forward / engine.backward are pure launch sequences on the current stream (no host sync, every buffer from
    the torch allocator), i.e. ~1050 + ~1500 kernel nodes per iteration whose Python/ctypes launch cost (~10 us each)
    is the gap between kernels.  Captured once (after the usual side-stream warm-up), replayed with static input /
    output / activation / gradient buffers; weight packing is inside the graphs, so optimizer updates are picked up."""

    def __init__(self, engine, x, need_grad, x_req):
        self.engine = engine
        dev = x.device
        self.x_static = x.detach().clone()
        self.need_grad, self.x_req = need_grad, x_req
        cur = torch.cuda.current_stream()
        s = torch.cuda.Stream()
        s.wait_stream(cur)
        with torch.cuda.stream(s):                       # warm-up: builds pack tables, workspaces, allocator pools
            for _ in range(2):
                out, saved = engine.forward(self.x_static, need_grad)
                if need_grad:
                    engine.backward(saved, torch.zeros_like(out), x_req)
        cur.wait_stream(s)
        torch.cuda.synchronize()
        self.g_fwd = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.g_fwd):
            self.out_static, self.saved = engine.forward(self.x_static, need_grad)
        self.g_bwd = None
        if need_grad:
            self.gout_static = torch.zeros_like(self.out_static)
            self.g_bwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool()):
                self.dx_static, self.grads = engine.backward(self.saved, self.gout_static, x_req)
        torch.cuda.synchronize()

    def forward(self, x):
        self.x_static.copy_(x)
        self.g_fwd.replay()
        return self.out_static

    def backward(self, g_out):
        self.gout_static.copy_(g_out)
        self.g_bwd.replay()
        return self.dx_static, self.grads


def _graphs_enabled(engine):
    """Graph replay is used when asked for (SRK_GRAPHS=1 / engine.use_graphs) and nothing needs eager launches:
    no per-launch kernel timing, no in-backward RCCL exchange, no side stream."""
    return engine.use_graphs and not L.KernelTimer.active and not engine._sync and not engine.overlap_wgrad


class _GeneratorFn(torch.autograd.Function):
    """raw = conv3(upsample(conv1(x) + conv2(RRDB^R(conv1(x))))) as one autograd node."""

    @staticmethod
    def forward(ctx, engine: GeneratorEngine, x, *params):
        need = any(ctx.needs_input_grad)
        ctx.engine = engine
        ctx.x_req = x.requires_grad
        ctx.graph = None
        if _graphs_enabled(engine) and x.is_cuda:
            key = (tuple(x.shape), need, x.requires_grad, engine.precision, tuple(p.data_ptr() for p in params))
            gs = engine._graphs.get(key)
            if gs is None:
                if len(engine._graphs) >= 4:
                    engine._graphs.clear()
                gs = engine._graphs[key] = _GraphSet(engine, x.detach().contiguous().float(), need, x.requires_grad)
            ctx.graph = gs
            ctx.saved = True if need else None
            return gs.forward(x.detach())
        out, saved = engine.forward(x.detach(), need_grad=need)
        ctx.saved = saved
        return out

    @staticmethod
    def backward(ctx, g_out):
        if ctx.saved is None:
            raise RuntimeError("generator backward called but forward ran without grad")
        eng = ctx.engine
        if ctx.graph is not None:
            dx, grads = ctx.graph.backward(g_out)
        else:
            dx, grads = eng.backward(ctx.saved, g_out, ctx.x_req)
        ctx.saved = None
        return (None, dx) + tuple(grads[p] for p in eng.params())


def generator_raw(engine: GeneratorEngine, x: torch.Tensor) -> torch.Tensor:
    return _GeneratorFn.apply(engine, x, *engine.params())
