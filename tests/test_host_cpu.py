"""CPU: host-side logic of the drop-in (no kernel launches): class surface, state_dict contract, C-ABI exports,
loud failure without a GPU, and the world-size-2 data-parallel plumbing on gloo."""
import ctypes
import importlib
import os
import re
import sys

import pytest
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import esrgan_oracle as O  # noqa: E402


def test_abi_library_exports_every_declared_symbol(srk):
    hdr = open(os.path.join(ROOT, "include", "srk.h")).read()
    declared = set(re.findall(r"\b(srk_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"srk_status"}
    lib = srk._lib.lib()
    assert declared == set(srk._lib.EXPORTS), declared ^ set(srk._lib.EXPORTS)
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.srk_version() == 100
    assert lib.srk_strerror(-4).decode().startswith("workspace")
    assert lib.srk_packed_floats(64, 64) == 8 * 9 * 2 * 64 * 4
    assert lib.srk_packed_floats(1, 1) == 2 * 9 * 2 * 32 * 4      # K rounds up to 16 (both fragment formats share buffers)


def test_wino42_packed_buffer_covers_the_kernels_unconditional_prefetch(srk):
    """srk_conv_w42.hip prefetches the weights of (chunk q + 1, channel pair 0) in every chunk, also behind the last one, through
    the scalar offset of raw buffer loads (not range-checked): byte offsets up to (2 nq + 1) * sB_ep with nq = ceil(K / 8) 8-channel
    chunks and sB_ep = 24 positions x 2 k-halves x CoutP x 8 bytes.  The packed buffer must cover them (mapped bytes, never used)."""
    lib = srk._lib.lib()
    for K in (8, 16, 24, 64, 72, 128, 320):
        for M in (64, 128, 256):
            Mp = (M + 31) // 32 * 32
            nq = (K + 7) // 8
            highest = (2 * nq + 1) * 24 * 2 * Mp * 8
            assert lib.srk_packed_floats_wino42(K, M) * 4 >= highest, (K, M)


def test_struct_layouts_match_header(srk, tmp_path):
    """ctypes mirrors == what a C compiler makes of include/srk.h (sizes and a few offsets)."""
    import subprocess
    L = srk._lib
    src = tmp_path / "layout.c"
    src.write_text("""#include <stdio.h>
#include <stddef.h>
#include "srk.h"
int main(void){
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(srk_conv_args), sizeof(srk_wgrad_args), sizeof(srk_pack_entry),
         offsetof(srk_conv_args, in_slope), offsetof(srk_conv_args, wp), offsetof(srk_conv_args, mask_slope),
         offsetof(srk_wgrad_args, workspace_bytes), offsetof(srk_pack_entry, elem_begin));
  return 0; }""")
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    vals = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    assert vals == [ctypes.sizeof(L.ConvArgs), ctypes.sizeof(L.WgradArgs), ctypes.sizeof(L.PackEntry),
                    L.ConvArgs.in_slope.offset, L.ConvArgs.wp.offset, L.ConvArgs.mask_slope.offset,
                    L.WgradArgs.workspace_bytes.offset, L.PackEntry.elem_begin.offset]


def test_generator_surface_and_state_dict(srk, golden_dir):
    g = srk.GeneratorRRDB(1, 32, 2)
    keys = open(os.path.join(golden_dir, "G5_state_keys.txt")).read().split()
    assert list(g.state_dict().keys()) == keys
    shapes = O.generator_state_shapes(1, 32, 2, 1)
    for k, v in g.state_dict().items():
        assert tuple(v.shape) == tuple(shapes[k]) and v.dtype == torch.float32
    assert g.thres == 0 and not g.power.requires_grad and not g.multiplier.requires_grad
    n = sum(p.numel() for p in g.parameters())
    assert n == 886499 + 2 - 2 or n == 886499          # cfg0 has 886,499 parameters (incl. power, multiplier)
    big = srk.GeneratorRRDB(1, filters=64, num_res_blocks=23, num_upsample=2)
    assert sum(p.numel() for p in big.parameters()) == 38546819
    assert sum(isinstance(m, nn.Conv2d) for m in big.modules()) == 351
    # default ctor == reference defaults (models.py:58)
    d = srk.GeneratorRRDB()
    assert (d.channels, d.filters, len(d.res_blocks), d.num_upsample) == (1, 64, 10, 1)
    # reference checkpoints load (same keys) and apply()-style resets work on nn.Conv2d subclasses
    g2 = srk.GeneratorRRDB(1, 32, 2)
    g2.load_state_dict(O.closed_form_fill(g.state_dict()))
    g2.apply(srk.weight_reset)
    g2.apply(srk.uniform_reset)
    assert float(g2.conv3[2].bias.detach().abs().sum()) == 0.0


def test_uniform_init_only_touches_conv1_conv2(srk):
    torch.manual_seed(0)
    g = srk.GeneratorRRDB(1, 16, 1, uniform_init=True)
    assert float(g.conv1.bias.abs().sum()) == 0.0 and float(g.conv2.bias.abs().sum()) == 0.0
    assert float(g.conv3[0].bias.abs().sum()) > 0.0
    assert float(g.res_blocks[0].dense_blocks[0].b1[0].bias.abs().sum()) > 0.0


def test_discriminator_surface(srk, golden_dir):
    D = srk.Markovian_Discriminator((1, 256, 256), [16, 32, 32, 64])
    assert D.output_shape == (1, 16, 16)
    assert sum(p.numel() for p in D.parameters()) == 90865
    assert list(D.state_dict().keys()) == open(os.path.join(golden_dir, "G7_state_keys.txt")).read().split()
    assert srk.Markovian_Discriminator((1, 80, 80)).output_shape == (1, 5, 5)
    assert srk.Markovian_Discriminator((1, 75, 75)).output_shape == (1, 5, 5)
    S = srk.Standard_Discriminator((1, 32, 32), [16, 32, 32, 64])
    assert S.output_shape == (1,)


def test_unsupported_branches_raise(srk):
    with pytest.raises(NotImplementedError):
        srk.Conv3x3(3, 3, 5, 1, 2)
    # drop_rate > 0 is supported (module-wise branch): same state_dict keys, Dropout2d modules hold no parameters
    g = srk.GeneratorRRDB(1, 16, 1, drop_rate=0.1)
    assert list(g.state_dict().keys()) == list(srk.GeneratorRRDB(1, 16, 1).state_dict().keys())
    assert sum(isinstance(m, torch.nn.Dropout2d) for m in g.modules()) == 3
    with pytest.raises(RuntimeError, match="GPU"):
        g(torch.rand(1, 1, 8, 8))


def test_no_cpu_fallback(srk):
    g = srk.GeneratorRRDB(1, 16, 1)
    with pytest.raises(RuntimeError, match="GPU"):
        g(torch.rand(1, 1, 8, 8))
    D = srk.Markovian_Discriminator((1, 16, 16))
    with pytest.raises(RuntimeError, match="GPU"):
        D(torch.rand(1, 1, 16, 16))
    with pytest.raises(RuntimeError, match="GPU"):
        srk.SumPool2d(2)(torch.rand(1, 1, 4, 4))
    C = srk.Conditional_Discriminator((1, 16, 16), [8, 16], num_upsample=1)
    with pytest.raises(RuntimeError, match="GPU"):
        C(torch.rand(1, 1, 16, 16), torch.rand(1, 1, 8, 8))
    # optional loss heads and the jet decode: HIP kernels only, no torch fallback
    x = torch.rand(2, 1, 8, 8)
    for fn in (lambda: srk.losses.soft_count(x), lambda: srk.losses.mask_l1(x, x), lambda: srk.losses.get_hitogram(x, 2),
               lambda: srk.losses.softgreater(x, 0.1), lambda: srk.losses.nnz_mask(x),
               lambda: srk.losses.DiffableHistogram([0.0, 0.5, 1.0]).forward_positive(x),
               lambda: srk.datasets.extract_batch(torch.zeros(2, 9), 4, 4)):
        with pytest.raises(RuntimeError):
            fn()


def test_conditional_discriminator_host_contract(srk):
    """ctor signature, output_shape and state_dict keys of models.py:189-223 (checked against the reference's key list)."""
    D = srk.Conditional_Discriminator((1, 32, 32), [8, 16, 16, 32], num_upsample=2)
    assert tuple(D.output_shape) == (1, 2, 2)
    keys = open(os.path.join(ROOT, "tests", "golden", "G13_state_keys.txt")).read().split()
    assert list(D.state_dict().keys()) == keys
    assert [tuple(v.shape) for k, v in D.state_dict().items() if k.startswith("endmodel.0.")] == [(16, 32, 3, 3), (16,)]


def test_dataset_host_logic(srk):
    """array-backed jet datasets: __len__/__getitem__ hand out raw rows; cutters match datasets.py:170-201 on CPU tensors."""
    import numpy as np
    rows = np.arange(3 * 9, dtype=np.float32).reshape(3, 9)
    ds = srk.datasets.SparseJetDataset(rows, etaBins=4, phiBins=4, factor=2, amount=2)
    assert len(ds) == 2 and torch.equal(ds[1]["rows"], torch.from_numpy(rows[1]))
    x = torch.tensor([[[0.1, 2.0], [3.0, 0.5]]])
    assert torch.equal(srk.datasets.Cutter(thres=1.0)(x), torch.tensor([[[0.0, 2.0], [3.0, 0.0]]]))
    assert torch.equal(srk.datasets.Cutter(amount=1)(x), torch.tensor([[[0.0, 0.0], [3.0, 0.0]]]))
    with pytest.raises(NotImplementedError):
        srk.datasets.Cutter(1.0, 2)
    with pytest.raises(NotImplementedError):
        srk.datasets.get_dataset("h5", "x.h5", 8, 8)


def test_parser_accepts_float_values_for_float_flags():
    """the reference declares the loss weights etc. with type=float (esrgan.py:58-120) even where default.json holds an int"""
    import importlib
    es = importlib.import_module("super-resolution_amd.esrgan")
    opt = es.get_parser(["--lambda_nnz", "1e-7", "--lambda_hit", "0.5", "--scaling_power", "0.3", "--sigma", "12.5", "--factor", "4",
                         "--d_channels", "8", "16", "--relativistic", "false"])
    assert opt.lambda_nnz == 1e-7 and opt.lambda_hit == 0.5 and opt.scaling_power == 0.3 and opt.sigma == 12.5
    assert opt.factor == 4 and isinstance(opt.factor, int) and opt.d_channels == [8, 16] and opt.relativistic is False
    assert es.get_parser([]).lambda_hr == 1 and es.get_parser([]).warmup_batches == 500


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "super-resolution_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src, fn


def test_bucket_layout_of_gradient_exchange(srk):
    g = srk.GeneratorRRDB(1, 16, 3, num_upsample=1)
    eng = g._engine
    ps = eng.params()
    assert len(ps) == 2 + 3 * 30 + 2 + 2 + 4
    grads = eng._alloc_grads(torch.device("cpu"))
    tot = sum(p.numel() for p in ps)
    assert eng._flat_grad.numel() == tot
    spans = [eng._bucket["conv1"]] + [eng._bucket[i] for i in range(3)] + [eng._bucket["tail"]]
    assert spans[0][0] == 0 and spans[-1][1] == tot
    for a, b in zip(spans[:-1], spans[1:]):
        assert a[1] == b[0]
    rr1 = [p for p in g.res_blocks[1].parameters()]
    a, b = eng._bucket[1]
    assert sum(p.numel() for p in rr1) == b - a
    assert grads[rr1[0]].data_ptr() == eng._flat_grad[a:].data_ptr()


def _dp_worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sr = importlib.import_module("super-resolution_amd")
    torch.manual_seed(0)
    g = sr.GeneratorRRDB(1, 16, 2, num_upsample=1)
    eng = g._engine
    eng.enable_grad_sync()
    assert eng._grad_scale == 1.0 / world
    grads = eng._alloc_grads(torch.device("cpu"))
    eng._flat_grad.copy_(torch.arange(eng._flat_grad.numel(), dtype=torch.float32) * (rank + 1))
    eng._reduce_bucket("tail")
    for i in (1, 0):
        eng._reduce_bucket(i)
    eng._reduce_bucket("conv1")
    eng._finish_reduce()
    expect = torch.arange(eng._flat_grad.numel(), dtype=torch.float32) * sum(r + 1 for r in range(world))
    ok = torch.equal(eng._flat_grad, expect) and torch.equal(grads[g.conv1.weight].flatten(), expect[:g.conv1.weight.numel()])
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_data_parallel_bucket_exchange_gloo_world2():
    """world_size-2 rehearsal of the gradient exchange on CPU (gloo): every bucket is summed across ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _armean_worker(rank, world, port, q):
    import importlib
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    train = importlib.import_module("super-resolution_amd.train")
    x = torch.tensor([1.0 + rank, 10.0 * (rank + 1)], requires_grad=True)
    y = train._AllReduceMean.apply(x)                         # batch statistic over all ranks
    w = torch.tensor([2.0 + rank, -1.0 - rank])
    (y * w).sum().backward()
    ok_f = torch.allclose(y.detach(), torch.tensor([1.5, 15.0]))
    ok_b = torch.allclose(x.grad, torch.tensor([2.5, -1.5]))   # mean over ranks of the upstream gradients
    q.put((rank, bool(ok_f and ok_b)))
    dist.destroy_process_group()


def test_exact_dp_statistic_exchange_gloo_world2():
    """batch-coupled statistics (relativistic means, batch-mean image) use a differentiable all-reduce-mean
    so that, after the gradient averaging, N ranks x B images equal one process on N*B images."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_armean_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_reference_option_sets_go_through_the_parser(golden_dir, tmp_path):
    """The reference's own option data (options/default.json, options/constant_args.json; tools/make_golden_options.py) through
    get_parser / options / the namedtuple path of hyper_search.py:104: every key is declared with the reference's default, json
    overlays work (also from an info.json), unknown bookkeeping keys ride along, and a namedtuple is accepted by train()'s
    option handling (esrgan.py:164-167)."""
    import importlib
    import json
    from collections import namedtuple
    es = importlib.import_module("super-resolution_amd.esrgan")
    ref = json.load(open(os.path.join(golden_dir, "ref_options.json")))
    dflt, const = ref["default"], ref["constant_args"]
    opt = es.get_parser([])
    for k, v in dflt.items():
        if k == "default":
            continue
        assert hasattr(opt, k), k
        assert getattr(opt, k) == v, (k, getattr(opt, k), v)
    # constant_args.json as --default overlay; an explicit flag wins; an info.json ("argument": {...}) works too
    f1 = tmp_path / "const.json"; f1.write_text(json.dumps(const))
    o1 = es.get_parser(["--default", str(f1), "--batch_size", "16"])
    assert o1.n_epochs == 20 and o1.sample_interval == -1 and o1.warmup_batches == 0 and o1.batch_size == 16 and o1.n_cpu == 0
    f2 = tmp_path / "info.json"; f2.write_text(json.dumps({"argument": {**const, "n_histograms": 1}, "epochs": 3}))
    o2 = es.get_parser(["--default", str(f2)])
    assert o2.residual_blocks == 10 and o2.n_histograms == 1
    # the reference's own flags that this build ignores or rejects parse with the reference's syntax
    o3 = es.get_parser(["-N", "100", "--wait", "hist", "2e4", "--set_zero_def", "hr", "lr", "--eval_modes", "E_1", "meanimg", "--n_hardest", "5",
                        "--noise_factor", "0.1", "--E_thres", "0.5", "--save_info", "false"])
    assert o3.N == 100 and o3.wait == ["hist", "2e4"] and o3.set_zero_def == ["hr", "lr"] and o3.n_hardest == 5 and o3.noise_factor == 0.1
    assert o3.save_info is False
    with pytest.raises(NotImplementedError):
        es._check_supported(es._complete(o3))
    # hyper_search.py:96-104: a namedtuple of a FEW options plus its own bookkeeping keys
    args = {**const, "name": "hs0", "n_batches": 10, "n_validations": 2, "n_histograms": -1, "metric_results": []}
    nt = namedtuple("arguments", args.keys())(*args.values())
    full = es._complete(nt)
    assert es._opt_dict(nt)["metric_results"] == [] and full.lambda_reg == dflt["lambda_reg"] and full.n_batches == 10 and full.batch_size == 4
    es._check_supported(full)
    assert es.options(foo=1).foo == 1


def test_transposed_conv_variants_keep_the_reference_state_dict(srk, golden_dir):
    """models.py:69-83: --use_transposed_conv / --fully_transposed_conv generators have the reference's state_dict keys and
    shapes (G16 key lists written from the imported reference), so its checkpoints load; they survive pickling like any module."""
    import pickle
    lines = dict(l.split(": ", 1) for l in open(os.path.join(golden_dir, "G16_state_keys.txt")).read().strip().split("\n"))
    for tag, kw in (("tc", dict(use_transposed_conv=True)), ("full", dict(fully_tconv_upsample=True))):
        g = srk.GeneratorRRDB(1, 16, 1, num_upsample=2, res_scale=0.1, **kw)
        assert list(g.state_dict().keys()) == lines[tag].split() and g.modulewise
        shapes = O.generator_state_shapes(1, 16, 1, 2, **kw)
        assert {k: tuple(v.shape) for k, v in g.state_dict().items()} == shapes
        g2 = pickle.loads(pickle.dumps(g))
        assert list(g2.state_dict().keys()) == lines[tag].split() and g2._engine.gen is g2
    assert not srk.GeneratorRRDB(1, 16, 1).modulewise


def test_inline_asm_mfma_hazards_of_the_wino42_kernel():
    """The F(2x4,3x3) conv kernel issues its MFMAs as inline assembly (register classes spelled out), which the compiler's
    hazard recogniser does not see: the code of the SHIPPED object must keep two wait states between a VALU write and an MFMA
    read, and its hand-counted `s_waitcnt vmcnt(12 | 6)` must sit behind exactly [halo DMA piece, 12 weight loads | 6 ring DMA
    instructions].  Checked in the four one-conv kernels (864 MFMAs) and the two chain kernels, whose K loop exists twice (576); the
    2-D Winograd weight-gradient kernels (builtin MFMAs, but a packed transform in inline assembly) get the VALU -> MFMA check too."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_w42_hazards.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 violation(s)" in r.stdout and "1440 v_mfma" in r.stdout
    # (4 x 256 in the wino22 kernels, 2 x 192 in wino24 -- its MFMAs are inline assembly too)
    assert "1408 v_mfma instructions in the wino22 / wino24 kernels" in r.stdout and "violation(s)" in r.stdout and r.stdout.count(": 0 violation(s)") == 2


def test_unknown_and_ignored_options_are_reported_once(capsys):
    """A misspelt option must not silently train with the default; an IGNORED option given a value says that it is ignored."""
    es = importlib.import_module("super-resolution_amd.esrgan")
    es._warned.clear()
    o = es._complete(es.options(lamda_hist=0.5, n_cpu=4, n_histograms=3))
    assert o.lamda_hist == 0.5 and o.lambda_hist == es.DEFAULTS["lambda_hist"]
    out = capsys.readouterr().out
    assert "unknown option 'lamda_hist'" in out and "option 'n_cpu' = 4" in out and "n_histograms" not in out
    es._complete(es.options(lamda_hist=0.5))
    assert capsys.readouterr().out == ""          # once


def test_chain_epoch_wrap_arithmetic(srk):
    """csrc/srk_chain.h: a chain kernel's tile flags hold `epoch + k + 1` in 32 bits, never reset per launch, and a waiter tests
    (int)(flag - target) >= 0.  That is only right while every live value is within 2^31 of every target: srk_chain_epoch_plan zeroes the
    flags before the epoch would pass 2^30.  Pure host arithmetic, no GPU: replay launches across the wrap with randomly used tiles and
    check that a flag a tile left behind ANY number of launches ago never reads as "published" for a later launch, and that what the
    launch itself publishes always does."""
    import ctypes as C
    import random
    lib = srk._lib.lib()
    eo, rs = C.c_uint(0), C.c_int(0)

    def plan(epoch, n):
        assert lib.srk_chain_epoch_plan(epoch, n, C.byref(eo), C.byref(rs)) == 0
        return eo.value, rs.value
    assert plan(0, 5) == (0, 0)
    assert plan((1 << 30) - 5, 5) == (0, 1)            # epoch + n would reach 2^30
    assert plan((1 << 30) - 6, 5) == ((1 << 30) - 6, 0)
    assert lib.srk_chain_epoch_plan(0, 9, C.byref(eo), C.byref(rs)) != 0      # more convs than a launch holds

    def ready(flag, target):          # the kernels' test, in 32-bit arithmetic
        d = (flag - target) & 0xffffffff
        return d < 0x80000000
    rnd = random.Random(7)
    flags = [0] * 1024
    epoch = (1 << 30) - 4000
    resets = 0
    for launch in range(3000):
        n, tiles = rnd.randint(2, 8), rnd.choice([1, 7, 64, 256, 1000])
        epoch, reset = plan(epoch, n)
        if reset:
            flags = [0] * 1024
            resets += 1
        assert epoch + n <= (1 << 30)
        for k in range(1, n):         # conv k waits for its neighbours' conv k - 1: target = epoch + k
            tgt = (epoch + k) & 0xffffffff
            # whatever earlier launches left in ANY flag is not mistaken for this launch's conv k - 1 ...
            assert not any(ready(f, tgt) for f in flags[:tiles]), (launch, k)
        for t in range(tiles):        # ... and what this launch publishes satisfies every wait of this launch
            for k in range(n - 1):
                flags[t] = (epoch + k + 1) & 0xffffffff
                assert ready(flags[t], (epoch + k + 1) & 0xffffffff)
        epoch += n
    assert resets >= 1
