"""ctypes binding of the C-ABI library ``libsrk.so`` (include/srk.h).

The binding is deliberately thin: it converts tensors to raw device pointers, fills the plain-C
argument structs and turns a non-zero ``srk_status`` into ``RuntimeError``.  It never falls back to
another implementation: if the library is missing the import of the compute path fails loudly.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRK_LIB_PATH", os.path.join(_HERE, "libsrk.so"))   # override: diagnostic builds only

IN_PLAIN, IN_UNSHUFFLE, IN_ZERO_UPSAMPLE = 0, 1, 2

EXPORTS = [
    "srk_conv3x3_signs_bytes", "srk_conv3x3_seq_signs_bytes", "srk_conv3x3_seq_signs_tag", "srk_adam_plan", "srk_adam_step", "srk_adam_step_small", "srk_conv3x3", "srk_conv3x3_seq", "srk_conv3x3_seq_kernel_name", "srk_debug_set_h16_chain", "srk_debug_set_h16_chain_m16", "srk_debug_set_w42_chain", "srk_conv3x3_kernel_name", "srk_debug_set_conv_small", "srk_debug_set_wino42_nmt", "srk_debug_set_wgrad_w22_form", "srk_conv3x3_wgrad", "srk_conv3x3_wgrad_workspace", "srk_conv3x3_wgrad_batched",
    "srk_conv3x3_wgrad_batched_workspace", "srk_conv3x3_wgrad_seq", "srk_conv3x3_wgrad_kernel_name", "srk_pack_plan", "srk_pack_weights",
    "srk_pack_weights_bf16x3", "srk_pack_weights_h16", "srk_packed_floats_h16", "srk_debug_set_h16_mt", "srk_conv3x3_bf16x3_supported", "srk_packed_floats", "srk_packed_floats_wino", "srk_packed_floats_wino4", "srk_packed_floats_wino42", "srk_pixel_shuffle_fwd", "srk_pixel_shuffle_bwd", "srk_nchw_to_nhwc", "srk_nhwc_to_nchw",
    "srk_sum_pool_fwd", "srk_sum_pool_bwd", "srk_workspace_bytes", "srk_conv3x3_fwd", "srk_conv3x3_dgrad", "srk_conv3x3_wgrad_flat",
    "srk_loss_workspace_bytes", "srk_sigmoid_fwd", "srk_sigmoid_bwd", "srk_lrelu_grad_mul", "srk_soft_count_fwd", "srk_soft_count_bwd",
    "srk_mask_l1_fwd", "srk_mask_l1_bwd", "srk_hitogram_fwd", "srk_hitogram_bwd", "srk_soft_hist_fwd", "srk_soft_hist_bwd",
    "srk_jet_extract", "srk_strerror", "srk_version",
    "srk_chain_recover", "srk_chain_stats", "srk_chain_epoch_plan", "srk_debug_chain_set", "srk_chain_set_wait_us", "srk_debug_chain_inject_fault",
    "srk_debug_hold_cus", "srk_debug_poison_lds", "srk_adam_count_step", "srk_debug_chain_inject_fault_async", "srk_debug_chain_skew",
]
ERR_CHAIN_TIMEOUT = -6
OP_CONV_FWD, OP_CONV_DGRAD, OP_CONV_WGRAD = 0, 1, 2

_fp = C.c_void_p


class ConvArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32), ("stride", C.c_int32), ("in_mode", C.c_int32), ("ps_out", C.c_int32),
        ("x", _fp), ("x_ldc", C.c_int32), ("x_coff", C.c_int32), ("in_slope", C.c_float),
        ("wp", _fp), ("bias", _fp),
        ("y", _fp), ("y_ldc", C.c_int32), ("y_coff", C.c_int32),
        ("alpha", C.c_float),
        ("r1", _fp), ("r1_ldc", C.c_int32), ("r1_coff", C.c_int32), ("beta1", C.c_float),
        ("r2", _fp), ("r2_ldc", C.c_int32), ("r2_coff", C.c_int32), ("beta2", C.c_float),
        ("slope", C.c_float),
        ("mask", _fp), ("m_ldc", C.c_int32), ("m_coff", C.c_int32), ("mask_slope", C.c_float),
        ("wp_format", C.c_int32), ("flags", C.c_int32),
        ("signs", _fp),
    ]


class WgradArgs(C.Structure):
    _fields_ = [
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("OH", C.c_int32), ("OW", C.c_int32),
        ("Cin", C.c_int32), ("Cout", C.c_int32), ("stride", C.c_int32), ("dy_mode", C.c_int32),
        ("x", _fp), ("x_ldc", C.c_int32), ("x_coff", C.c_int32), ("in_slope", C.c_float),
        ("dy", _fp), ("dy_ldc", C.c_int32), ("dy_coff", C.c_int32),
        ("dw", _fp), ("db", _fp), ("scale", C.c_float), ("accumulate", C.c_int32),
        ("workspace", _fp), ("workspace_bytes", C.c_size_t), ("precision", C.c_int32),
    ]


class PackEntry(C.Structure):
    _fields_ = [
        ("src", _fp), ("dst", _fp), ("src_cout", C.c_int32), ("src_cin", C.c_int32), ("transpose", C.c_int32),
        ("c_begin", C.c_int32), ("M", C.c_int32), ("k_off", C.c_int32), ("k_len", C.c_int32), ("K_total", C.c_int32),
        ("ps", C.c_int32), ("scale", C.c_float), ("fmt", C.c_int32), ("elem_begin", C.c_int64),
    ]


class AdamEntry(C.Structure):
    _fields_ = [("p", _fp), ("g", _fp), ("m", _fp), ("v", _fp), ("n", C.c_int64), ("chunk_begin", C.c_int64)]


_lib = None
dispatch_gen = 0


def _bumping(fn):
    def call(*a):
        global dispatch_gen
        dispatch_gen += 1
        return fn(*a)
    return call


def lib():
    """Load libsrk.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C super-resolution_amd/csrc`). There is no CPU/PyTorch fallback for the hot path.")
        L = C.CDLL(LIB_PATH)
        L.srk_strerror.restype = C.c_char_p
        L.srk_strerror.argtypes = [C.c_int]
        L.srk_version.restype = C.c_int
        L.srk_packed_floats.restype = C.c_size_t
        L.srk_packed_floats.argtypes = [C.c_int, C.c_int]
        L.srk_packed_floats_wino.restype = C.c_size_t
        L.srk_packed_floats_wino.argtypes = [C.c_int, C.c_int]
        L.srk_packed_floats_wino4.restype = C.c_size_t
        L.srk_packed_floats_wino4.argtypes = [C.c_int, C.c_int]
        L.srk_packed_floats_wino42.restype = C.c_size_t
        L.srk_packed_floats_wino42.argtypes = [C.c_int, C.c_int]
        L.srk_packed_floats_h16.restype = C.c_size_t
        L.srk_packed_floats_h16.argtypes = [C.c_int, C.c_int]
        L.srk_debug_set_h16_mt.argtypes = [C.c_int]
        L.srk_pack_weights_h16.argtypes = [_fp, C.c_int, C.c_int64, C.c_int, _fp]
        L.srk_conv3x3.argtypes = [C.POINTER(ConvArgs), _fp]
        L.srk_conv3x3_signs_bytes.restype = C.c_size_t
        L.srk_conv3x3_signs_bytes.argtypes = [C.POINTER(ConvArgs)]
        L.srk_conv3x3_seq_signs_bytes.restype = C.c_size_t
        L.srk_conv3x3_seq_signs_bytes.argtypes = [C.POINTER(ConvArgs), C.c_int]
        L.srk_adam_plan.argtypes = [C.POINTER(AdamEntry), C.c_int, C.POINTER(C.c_int64)]
        L.srk_adam_step.argtypes = [_fp, C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp, _fp]
        L.srk_adam_step_small.argtypes = [C.POINTER(AdamEntry), C.c_int, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _fp, _fp, _fp, _fp]
        L.srk_conv3x3_seq.argtypes = [C.POINTER(ConvArgs), C.c_int, _fp]
        L.srk_conv3x3_seq_kernel_name.argtypes = [C.POINTER(ConvArgs), C.c_int, C.c_char_p, C.c_size_t]
        L.srk_debug_set_h16_chain.argtypes = [C.c_int]
        L.srk_debug_set_w42_chain.argtypes = [C.c_int]
        L.srk_debug_set_conv_small.argtypes = [C.c_int]
        L.srk_debug_set_wino42_nmt.argtypes = [C.c_int]
        L.srk_debug_set_wgrad_w22_form.argtypes = [C.c_int]
        L.srk_conv3x3_kernel_name.argtypes = [C.POINTER(ConvArgs), C.c_char_p, C.c_size_t]
        L.srk_conv3x3_wgrad.argtypes = [C.POINTER(WgradArgs), _fp]
        L.srk_conv3x3_wgrad_workspace.argtypes = [C.POINTER(WgradArgs), C.POINTER(C.c_size_t)]
        L.srk_conv3x3_wgrad_batched.argtypes = [C.POINTER(WgradArgs), C.c_int, _fp]
        L.srk_conv3x3_wgrad_batched_workspace.argtypes = [C.POINTER(WgradArgs), C.c_int, C.POINTER(C.c_size_t)]
        L.srk_conv3x3_wgrad_kernel_name.argtypes = [C.POINTER(WgradArgs), C.c_int, C.c_char_p, C.c_size_t]
        L.srk_conv3x3_wgrad_seq.argtypes = [C.POINTER(WgradArgs), C.c_int, _fp]
        L.srk_pack_plan.argtypes = [C.POINTER(PackEntry), C.c_int, C.POINTER(C.c_int64)]
        L.srk_pack_weights.argtypes = [_fp, C.c_int, C.c_int64, _fp]
        L.srk_pack_weights_bf16x3.argtypes = [_fp, C.c_int, C.c_int64, _fp]
        L.srk_conv3x3_bf16x3_supported.argtypes = [C.POINTER(ConvArgs)]
        for name in ("srk_pixel_shuffle_fwd", "srk_pixel_shuffle_bwd"):
            getattr(L, name).argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.srk_nchw_to_nhwc.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.srk_nhwc_to_nchw.argtypes = [_fp, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        for name in ("srk_sum_pool_fwd", "srk_sum_pool_bwd"):
            getattr(L, name).argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.srk_workspace_bytes.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_size_t)]
        L.srk_conv3x3_fwd.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_float, _fp, C.c_float, C.c_int, C.c_int, _fp, C.c_size_t, _fp]
        L.srk_conv3x3_dgrad.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, _fp]
        L.srk_conv3x3_wgrad_flat.argtypes = [_fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, C.c_int,
                                             C.c_int, C.c_float, C.c_int, C.c_int, C.c_int, _fp, C.c_size_t, _fp]
        L.srk_jet_extract.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _fp, _fp]
        L.srk_loss_workspace_bytes.argtypes = [C.POINTER(C.c_size_t)]
        L.srk_sigmoid_fwd.argtypes = [_fp, _fp, C.c_long, C.c_float, C.c_float, _fp]
        L.srk_lrelu_grad_mul.argtypes = [_fp, _fp, _fp, C.c_long, C.c_float, _fp]
        L.srk_sigmoid_bwd.argtypes = [_fp, _fp, _fp, C.c_long, C.c_float, _fp]
        L.srk_soft_count_fwd.argtypes = [_fp, _fp, C.c_int, C.c_long, C.c_float, C.c_float, C.c_int, _fp, C.c_size_t, _fp]
        L.srk_soft_count_bwd.argtypes = [_fp, _fp, _fp, C.c_int, C.c_long, C.c_float, C.c_float, _fp]
        L.srk_mask_l1_fwd.argtypes = [_fp, _fp, _fp, C.c_long, C.c_float, _fp, C.c_size_t, _fp]
        L.srk_mask_l1_bwd.argtypes = [_fp, _fp, _fp, _fp, C.c_long, C.c_float, _fp]
        L.srk_hitogram_fwd.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp, C.c_size_t, _fp]
        L.srk_hitogram_bwd.argtypes = [_fp, _fp, _fp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _fp]
        L.srk_soft_hist_fwd.argtypes = [_fp, C.c_long, _fp, _fp, C.c_int, C.c_float, C.c_int, _fp, _fp, C.c_size_t, _fp]
        L.srk_soft_hist_bwd.argtypes = [_fp, C.c_long, _fp, _fp, C.c_int, C.c_float, C.c_int, _fp, _fp, _fp]
        L.srk_conv3x3_seq_signs_tag.argtypes = [C.POINTER(ConvArgs), C.c_int]
        L.srk_chain_recover.argtypes = []
        L.srk_chain_stats.argtypes = [C.POINTER(C.c_ulonglong), C.POINTER(C.c_ulonglong), C.POINTER(C.c_int), C.POINTER(C.c_long)]
        L.srk_chain_epoch_plan.argtypes = [C.c_uint, C.c_int, C.POINTER(C.c_uint), C.POINTER(C.c_int)]
        L.srk_debug_chain_set.argtypes = [C.c_uint, C.c_long]
        L.srk_chain_set_wait_us.argtypes = [C.c_uint]
        L.srk_debug_chain_inject_fault.argtypes = [C.c_uint]
        L.srk_debug_chain_inject_fault_async.argtypes = [_fp]
        L.srk_debug_chain_skew.argtypes = [C.c_int, C.c_uint, C.c_uint]
        L.srk_debug_hold_cus.argtypes = [C.c_int, C.c_int, _fp]
        L.srk_debug_poison_lds.argtypes = [_fp]
        L.srk_adam_count_step.argtypes = [_fp, _fp, _fp, _fp]
        # whatever changes which kernel form a launch takes bumps dispatch_gen: callers that cache a dispatch-dependent answer (the engine's
        # sign-bit decisions) key it with the generation
        L.srk_debug_set_h16_chain_m16.argtypes = [C.c_int]
        for name in ("srk_debug_set_h16_mt", "srk_debug_set_h16_chain", "srk_debug_set_h16_chain_m16", "srk_debug_set_w42_chain", "srk_debug_set_wino42_nmt",
                     "srk_chain_recover", "srk_debug_chain_set"):
            setattr(L, name, _bumping(getattr(L, name)))
        _lib = L
    return _lib


class ChainTimeout(RuntimeError):
    """SRK_ERR_CHAIN_TIMEOUT: a chain launch (one persistent kernel per dense-block sequence) gave up -- a tile's neighbour did not
    publish within the wait bound (it was not resident).  The call that raised launched nothing; optimizer steps skip themselves on the
    device while the fault is pending.  chain_recover(), then repeat the iteration (train.Stepper.step does both)."""


def check(status: int, what: str):
    if status == ERR_CHAIN_TIMEOUT:
        raise ChainTimeout(f"{what}: {lib().srk_strerror(status).decode()} (status {status})")
    if status != 0:
        raise RuntimeError(f"{what} failed: {lib().srk_strerror(status).decode()} (status {status})")


def chain_recover() -> int:
    """srk_chain_recover: waits for the device, clears a pending chain fault, rests the chain forms; returns the fault code (0: none)."""
    rc = lib().srk_chain_recover()
    if rc < 0:
        check(rc, "srk_chain_recover")
    return rc


_POISON = os.environ.get("SRK_POISON_LDS", "0") == "1"      # test runs: NaN patterns into every CU's LDS in front of every conv / weight-gradient launch


def _poison():
    if _POISON:
        poison_lds()


def poison_lds():
    """test aid (srk_debug_poison_lds): NaN bit patterns into the whole LDS of every CU, on the current stream"""
    check(lib().srk_debug_poison_lds(stream_ptr()), "srk_debug_poison_lds")


def chain_stats() -> dict:
    a, b, c, d = C.c_ulonglong(0), C.c_ulonglong(0), C.c_int(0), C.c_long(0)
    check(lib().srk_chain_stats(C.byref(a), C.byref(b), C.byref(c), C.byref(d)), "srk_chain_stats")
    return {"launches": a.value, "resets": b.value, "strikes": c.value, "off_calls": d.value}


def adam_count_step(step, found_inf, skip_out):
    """srk_adam_count_step: *step += 1 unless *found_inf != 0 or a chain fault is pending; *skip_out = 1 / 0 accordingly"""
    check(lib().srk_adam_count_step(step.data_ptr(), ptr(found_inf), skip_out.data_ptr(), stream_ptr()), "srk_adam_count_step")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t):
    return None if t is None else t.data_ptr()


CONV_OUT_F32 = 1
CONV_WRITE_SIGNS, CONV_MASK_SIGNS = 2, 4
FMT_OF_DTYPE = {torch.float16: 7, torch.bfloat16: 8}       # wp_format / pack fmt of the 16-bit-storage kernels
WGRAD_PRECISION_OF_DTYPE = {torch.float16: 3, torch.bfloat16: 4}


class View:
    """An NHWC channel-slice view ``base[pixel*ldc + coff + c]`` of a contiguous [N,H,W,ldc] tensor: fp32, or fp16 / bf16 for the
    16-bit-storage kernels (wp_format 7 / 8; ldc / coff count elements)."""
    __slots__ = ("t", "ldc", "coff", "C")

    def __init__(self, t: torch.Tensor, coff: int = 0, C_: int = None):
        assert t.is_cuda and t.dtype in (torch.float32, torch.float16, torch.bfloat16) and t.is_contiguous() and t.dim() == 4, \
            "NHWC contiguous CUDA tensor (fp32 / fp16 / bf16) expected"
        self.t = t
        self.ldc = t.shape[3]
        self.coff = coff
        self.C = (self.ldc - coff) if C_ is None else C_
        assert 0 <= coff and coff + self.C <= self.ldc


class KernelTimer:
    """Optional per-launch timing of the conv kernels with HIP events on the launching stream (bench.py's
    roofline leg).  Off by default; when on, every srk_conv3x3 / srk_conv3x3_wgrad call is bracketed by two
    events and attributed to the kernel template the C side dispatches to (same names rocprofv3 reports)."""
    active = False
    detail = os.environ.get("SRK_KT_DETAIL", "0") == "1"
    records = []

    @classmethod
    def start(cls):
        cls.records = []
        cls.active = True

    @classmethod
    def stop(cls):
        cls.active = False
        torch.cuda.synchronize()
        out = {}
        for name, flops, e0, e1 in cls.records:
            st = out.setdefault(name, {"ms": 0.0, "flops": 0.0, "n": 0})
            st["ms"] += e0.elapsed_time(e1)
            st["flops"] += flops
            st["n"] += 1
        cls.records = []
        return out

    @classmethod
    def bracket(cls, name, flops):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cls.records.append((name, flops, e0, e1))
        return e0, e1


def _conv_kernel_name(a) -> str:
    """Kernel the C side dispatches this conv launch to (srk_conv3x3_kernel_name: the rules live in srk_conv.hip only)."""
    buf = C.create_string_buffer(96)
    check(lib().srk_conv3x3_kernel_name(C.byref(a), buf, 96), "srk_conv3x3_kernel_name")
    return buf.value.decode()


def _fill_conv_args(a, x: View, wp, bias, y: View, *, N, H, W, OH, OW, Cin, Cout, stride=1, in_mode=IN_PLAIN,
                    ps_out=False, alpha=1.0, r1: View = None, beta1=0.0, r2: View = None, beta2=0.0, slope=1.0,
                    mask: View = None, mask_slope=1.0, in_slope=1.0, wp_format=0, flags=0, signs_out=None, mask_signs=None):
    """signs_out: uint8 tensor of conv_signs_bytes(...) bytes that receives the sign bits of y (srk_conv_args.signs, SRK_CONV_WRITE_SIGNS);
    mask_signs: such a tensor used as the LeakyReLU' mask (SRK_CONV_MASK_SIGNS; mask_slope applies)"""
    a.in_slope = in_slope
    a.wp_format = getattr(wp, "fmt", wp_format)
    if signs_out is not None:
        flags |= CONV_WRITE_SIGNS
        a.signs = signs_out.data_ptr()
    if mask_signs is not None:
        flags |= CONV_MASK_SIGNS
        a.signs = mask_signs.data_ptr()
        a.mask_slope = mask_slope
    a.flags = flags
    if a.wp_format in (7, 8):
        want = torch.float16 if a.wp_format == 7 else torch.bfloat16
        for v in (x, r1, r2, mask) + (() if flags & CONV_OUT_F32 else (y,)):
            if v is not None and v.t.dtype != want:
                raise ValueError(f"wp_format {a.wp_format}: {want} views expected, got {v.t.dtype}")
    elif any(v is not None and v.t.dtype != torch.float32 for v in (x, y, r1, r2, mask)):
        raise ValueError("fp32 views expected (16-bit views go with wp_format 7 / 8)")
    a.N, a.H, a.W, a.OH, a.OW, a.Cin, a.Cout = N, H, W, OH, OW, Cin, Cout
    a.stride, a.in_mode, a.ps_out = stride, in_mode, int(ps_out)
    a.x, a.x_ldc, a.x_coff = x.t.data_ptr(), x.ldc, x.coff
    a.wp, a.bias = wp.data_ptr(), ptr(bias)
    a.y, a.y_ldc, a.y_coff = y.t.data_ptr(), y.ldc, y.coff
    a.alpha = alpha
    if r1 is not None:
        a.r1, a.r1_ldc, a.r1_coff, a.beta1 = r1.t.data_ptr(), r1.ldc, r1.coff, beta1
    if r2 is not None:
        a.r2, a.r2_ldc, a.r2_coff, a.beta2 = r2.t.data_ptr(), r2.ldc, r2.coff, beta2
    a.slope = slope
    if mask is not None:
        a.mask, a.m_ldc, a.m_coff, a.mask_slope = mask.t.data_ptr(), mask.ldc, mask.coff, mask_slope


def conv_signs_bytes(x: View, wp, bias, y: View, **kw) -> int:
    """bytes of the sign-bit buffer the conv with these arguments writes / reads (0: its launch does not offer sign bits)"""
    a = ConvArgs()
    _fill_conv_args(a, x, wp, bias, y, **kw)
    return int(lib().srk_conv3x3_signs_bytes(C.byref(a)))


def conv_seq_signs_bytes(calls) -> int:
    """bytes of ONE conv's sign-bit buffer when the sequence `calls` (as for conv3x3_seq) offers sign bits, else 0"""
    n = len(calls)
    arr = (ConvArgs * n)()
    for a, (x, wp, bias, y, kw) in zip(arr, calls):
        _fill_conv_args(a, x, wp, bias, y, **kw)
    return int(lib().srk_conv3x3_seq_signs_bytes(arr, n))


def conv_seq_signs(calls):
    """(bytes of ONE conv's sign-bit buffer, layout tag) of the sequence `calls`; (0, 0) if its launches offer no sign bits.  Bits written
    under one tag may only be read by a sequence reporting the same tag (srk_conv3x3_seq_signs_tag)."""
    n = len(calls)
    arr = (ConvArgs * n)()
    for a, (x, wp, bias, y, kw) in zip(arr, calls):
        _fill_conv_args(a, x, wp, bias, y, **kw)
    nb = int(lib().srk_conv3x3_seq_signs_bytes(arr, n))
    return (nb, int(lib().srk_conv3x3_seq_signs_tag(arr, n))) if nb else (0, 0)


def conv3x3(x: View, wp: torch.Tensor, bias, y: View, **kw):
    a = ConvArgs()
    _fill_conv_args(a, x, wp, bias, y, **kw)
    if KernelTimer.active:
        name = _conv_kernel_name(a)
        if KernelTimer.detail:       # diagnostic split by problem shape and by what the fused epilogue reads
            name += f" Cin={a.Cin} Cout={a.Cout} {a.OH}x{a.OW} epi={'b' if bias is not None else ''}{'r' if a.r1 else ''}{'R' if a.r2 else ''}{'m' if a.mask else ''}"
        e0, e1 = KernelTimer.bracket(name, 2.0 * a.N * a.OH * a.OW * a.Cout * a.Cin * 9)
        e0.record()
        check(lib().srk_conv3x3(C.byref(a), stream_ptr()), "srk_conv3x3")
        e1.record()
        return
    _poison()
    check(lib().srk_conv3x3(C.byref(a), stream_ptr()), "srk_conv3x3")


def conv3x3_seq(calls):
    """``calls``: list of (x, wp, bias, y, kwargs) as for conv3x3 -- launched back to back on the current stream by ONE C call
    (srk_conv3x3_seq: a dense block's five forward or five data-gradient convolutions; with 16-bit storage they may go out as ONE
    persistent kernel, the chain form).  While bench.py brackets launches with events, a sequence that is not one kernel goes conv by
    conv, so that every launch keeps its own time."""
    n = len(calls)
    arr = (ConvArgs * n)()
    for a, (x, wp, bias, y, kw) in zip(arr, calls):
        _fill_conv_args(a, x, wp, bias, y, **kw)
    if KernelTimer.active:
        buf = C.create_string_buffer(96)
        check(lib().srk_conv3x3_seq_kernel_name(arr, n, buf, 96), "srk_conv3x3_seq_kernel_name")
        name = buf.value.decode()
        if not name:
            for x, wp, bias, y, kw in calls:
                conv3x3(x, wp, bias, y, **kw)
            return
        if KernelTimer.detail:
            name += f" n={n} Cin={arr[0].Cin}..{arr[n - 1].Cin} {'m' if arr[0].mask else 'b'}"
        e0, e1 = KernelTimer.bracket(name, sum(2.0 * a.N * a.OH * a.OW * a.Cout * a.Cin * 9 for a in arr))
        e0.record()
        check(lib().srk_conv3x3_seq(arr, n, stream_ptr()), "srk_conv3x3_seq")
        e1.record()
        return
    _poison()
    check(lib().srk_conv3x3_seq(arr, n, stream_ptr()), "srk_conv3x3_seq")


_ws_cache = {}
_wgrad_ws_bytes = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    key = (device.index, torch.cuda.current_stream().cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def _wgrad_kernel_name(arr, n) -> str:
    """Kernel the C side dispatches this weight-gradient launch to (srk_conv3x3_wgrad_kernel_name: the rules live there)."""
    buf = C.create_string_buffer(96)
    check(lib().srk_conv3x3_wgrad_kernel_name(arr, n, buf, 96), "srk_conv3x3_wgrad_kernel_name")
    return buf.value.decode()


def _fill_wgrad_args(a, x: View, dy: View, dw: torch.Tensor, db, *, N, H, W, OH, OW, Cin, Cout, stride=1, dy_mode=IN_PLAIN,
                     scale=1.0, accumulate=False, in_slope=1.0, precision=0):
    """fills one WgradArgs (all but the workspace fields); returns the workspace bytes the launch needs"""
    a.in_slope = in_slope
    a.precision = precision
    a.N, a.H, a.W, a.OH, a.OW, a.Cin, a.Cout, a.stride, a.dy_mode = N, H, W, OH, OW, Cin, Cout, stride, dy_mode
    a.x, a.x_ldc, a.x_coff = x.t.data_ptr(), x.ldc, x.coff
    a.dy, a.dy_ldc, a.dy_coff = dy.t.data_ptr(), dy.ldc, dy.coff
    a.dw, a.db, a.scale, a.accumulate = dw.data_ptr(), ptr(db), scale, int(accumulate)
    key = (N, H, W, OH, OW, stride, dy_mode, precision, ((Cin, Cout),), x.ldc, dy.ldc)      # (size by geometry: one query per shape)
    need = _wgrad_ws_bytes.get(key)
    if need is None:
        nbytes = C.c_size_t(0)
        check(lib().srk_conv3x3_wgrad_workspace(C.byref(a), C.byref(nbytes)), "srk_conv3x3_wgrad_workspace")
        need = _wgrad_ws_bytes[key] = nbytes.value
    return need


def conv3x3_wgrad(x: View, dy: View, dw: torch.Tensor, db, **kw):
    a = WgradArgs()
    need = _fill_wgrad_args(a, x, dy, dw, db, **kw)
    ws = _workspace(need, x.t.device)
    a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    if KernelTimer.active:
        e0, e1 = KernelTimer.bracket(_wgrad_kernel_name(C.byref(a), 1) + "+reduce", 2.0 * a.N * a.OH * a.OW * a.Cout * a.Cin * 9)
        e0.record()
        check(lib().srk_conv3x3_wgrad(C.byref(a), stream_ptr()), "srk_conv3x3_wgrad")
        e1.record()
        return
    _poison()
    check(lib().srk_conv3x3_wgrad(C.byref(a), stream_ptr()), "srk_conv3x3_wgrad")


def conv3x3_wgrad_seq(calls):
    """``calls``: list of (x, dy, dw, db, kwargs) as for conv3x3_wgrad: independent problems of any geometry (a discriminator's
    layers) launched back to back by ONE C call (srk_conv3x3_wgrad_seq); one by one while bench.py brackets launches with events."""
    if KernelTimer.active:
        for x, dy, dw, db, kw in calls:
            conv3x3_wgrad(x, dy, dw, db, **kw)
        return
    n = len(calls)
    arr = (WgradArgs * n)()
    need = 0
    for a, (x, dy, dw, db, kw) in zip(arr, calls):
        need = max(need, _fill_wgrad_args(a, x, dy, dw, db, **kw))
    ws = _workspace(need, calls[0][0].t.device)
    for a in arr:
        a.workspace, a.workspace_bytes = ws.data_ptr(), ws.numel()
    check(lib().srk_conv3x3_wgrad_seq(arr, n, stream_ptr()), "srk_conv3x3_wgrad_seq")


def conv3x3_wgrad_batched(problems, *, N, H, W, OH, OW, stride=1, dy_mode=IN_PLAIN, precision=0):
    """problems: list of dicts(x=View, dy=View, dw=Tensor, db=Tensor|None, Cin, Cout, scale, accumulate, in_slope)."""
    n = len(problems)
    arr = (WgradArgs * n)()
    flops = 0.0
    for a, p in zip(arr, problems):
        a.N, a.H, a.W, a.OH, a.OW, a.stride, a.dy_mode = N, H, W, OH, OW, stride, dy_mode
        a.Cin, a.Cout = p["Cin"], p["Cout"]
        x, dy = p["x"], p["dy"]
        a.x, a.x_ldc, a.x_coff, a.in_slope = x.t.data_ptr(), x.ldc, x.coff, p.get("in_slope", 1.0)
        a.dy, a.dy_ldc, a.dy_coff = dy.t.data_ptr(), dy.ldc, dy.coff
        a.dw, a.db = p["dw"].data_ptr(), ptr(p.get("db"))
        a.scale, a.accumulate = p.get("scale", 1.0), int(p.get("accumulate", False))
        a.precision = precision
        flops += 2.0 * N * OH * OW * a.Cout * a.Cin * 9
    # (the workspace size depends on the geometry only: one query per shape, not one more C call per launch)
    key = (N, H, W, OH, OW, stride, dy_mode, precision, tuple((a.Cin, a.Cout) for a in arr))
    need = _wgrad_ws_bytes.get(key)
    if need is None:
        nbytes = C.c_size_t(0)
        check(lib().srk_conv3x3_wgrad_batched_workspace(arr, n, C.byref(nbytes)), "srk_conv3x3_wgrad_batched_workspace")
        need = _wgrad_ws_bytes[key] = nbytes.value
    ws = _workspace(need, problems[0]["x"].t.device)
    arr[0].workspace, arr[0].workspace_bytes = ws.data_ptr(), ws.numel()
    if KernelTimer.active:
        e0, e1 = KernelTimer.bracket(_wgrad_kernel_name(arr, n) + "+reduce", flops)
        e0.record()
        check(lib().srk_conv3x3_wgrad_batched(arr, n, stream_ptr()), "srk_conv3x3_wgrad_batched")
        e1.record()
        return
    _poison()
    check(lib().srk_conv3x3_wgrad_batched(arr, n, stream_ptr()), "srk_conv3x3_wgrad_batched")


def workspace_bytes(op: int, N, H, W, Cin, Cout, dtype=0) -> int:
    n = C.c_size_t(0)
    check(lib().srk_workspace_bytes(op, N, H, W, Cin, Cout, dtype, C.byref(n)), "srk_workspace_bytes")
    return n.value


def conv3x3_fwd_flat(x: "View", w, bias, y: "View", *, N, H, W, Cin, Cout, stride=1, slope=1.0, residual=None, res_scale=1.0, ps=0):
    """srk_conv3x3_fwd: canonical OIHW weights, packed into a scratch workspace inside the call."""
    ws = _workspace(workspace_bytes(OP_CONV_FWD, N, H, W, Cin, Cout), w.device)
    check(lib().srk_conv3x3_fwd(x.t.data_ptr(), x.ldc, x.coff, Cin, w.data_ptr(), ptr(bias), y.t.data_ptr(), y.ldc, y.coff, Cout,
                                N, H, W, stride, slope, ptr(residual), res_scale, ps, 0, ws.data_ptr(), ws.numel(), stream_ptr()),
          "srk_conv3x3_fwd")


def conv3x3_dgrad_flat(dy: "View", w, dx: "View", *, N, H, W, Cin, Cout, stride=1, ps=0):
    """srk_conv3x3_dgrad: gradient w.r.t. the conv input (H, W = input extent)."""
    ws = _workspace(workspace_bytes(OP_CONV_DGRAD, N, H, W, Cin, Cout), w.device)
    check(lib().srk_conv3x3_dgrad(dy.t.data_ptr(), dy.ldc, dy.coff, Cout, w.data_ptr(), dx.t.data_ptr(), dx.ldc, dx.coff, Cin,
                                  N, H, W, stride, ps, 0, ws.data_ptr(), ws.numel(), stream_ptr()), "srk_conv3x3_dgrad")


def conv3x3_wgrad_flat(x: "View", dy: "View", dw, db, *, N, H, W, Cin, Cout, stride=1, scale=1.0, accumulate=False, ps=0):
    """srk_conv3x3_wgrad_flat: weight / bias gradient into canonical OIHW tensors (H, W = extent of the conv input)."""
    ws = _workspace(workspace_bytes(OP_CONV_WGRAD, N, H, W, Cin, Cout), dw.device)
    check(lib().srk_conv3x3_wgrad_flat(x.t.data_ptr(), x.ldc, x.coff, Cin, dy.t.data_ptr(), dy.ldc, dy.coff, Cout, dw.data_ptr(), ptr(db),
                                       N, H, W, stride, scale, int(accumulate), ps, 0, ws.data_ptr(), ws.numel(), stream_ptr()),
          "srk_conv3x3_wgrad_flat")


def loss_workspace(device) -> torch.Tensor:
    n = C.c_size_t(0)
    check(lib().srk_loss_workspace_bytes(C.byref(n)), "srk_loss_workspace_bytes")
    return _workspace(n.value, device)


def packed_floats(K: int, M: int, fmt: int = 0) -> int:
    """floats of the packed buffer for a (K inputs, M outputs) conv; fmt 3 (Winograd) carries 12 taps instead of 9."""
    if fmt in (7, 8):
        return lib().srk_packed_floats_h16(K, M)
    if fmt == 6:
        return lib().srk_packed_floats_wino42(K, M)
    if fmt == 5:
        return lib().srk_packed_floats_wino4(K, M)
    return lib().srk_packed_floats_wino(K, M) if fmt == 3 else lib().srk_packed_floats(K, M)


def wino_eligible(K: int, M: int) -> bool:
    """Static part of srk_conv3x3's wp_format == 3 contract (stride 1 and a plain / unshuffle input are the caller's)."""
    return K % 8 == 0 and M % 64 == 0


class PackTable:
    """A batch of weight-packing jobs executed by ONE kernel launch (srk_pack_weights)."""

    def __init__(self, device, fmt=0):
        self.device = device
        self.fmt = fmt               # 0: fp32 fragments, 1: split-bf16, 3 / 5 / 6: Winograd F(2,3) / F(4,3) / F(2x4,3x3) fp32 fragments
        self.entries = []
        self._dev = None
        self._total = 0

    def add(self, src: torch.Tensor, dst: torch.Tensor, *, M, k_off, k_len, K_total, transpose=False, c_begin=0, ps=False, scale=1.0):
        assert src.dim() == 4 and src.shape[2] == 3 and src.shape[3] == 3 and src.is_contiguous()
        e = PackEntry()
        e.src, e.dst = src.data_ptr(), dst.data_ptr()
        e.src_cout, e.src_cin = src.shape[0], src.shape[1]
        e.transpose, e.c_begin, e.M, e.k_off, e.k_len, e.K_total = int(transpose), c_begin, M, k_off, k_len, K_total
        e.ps, e.scale, e.fmt = int(ps), scale, self.fmt
        self.entries.append(e)
        self._dev = None

    def finalize(self):
        n = len(self.entries)
        if n == 0:
            return
        arr = (PackEntry * n)(*self.entries)
        total = C.c_int64(0)
        check(lib().srk_pack_plan(arr, n, C.byref(total)), "srk_pack_plan")
        self._total = total.value
        raw = bytes(arr)
        self._dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        self._n = n

    def run(self):
        if not self.entries:
            return
        if self._dev is None:
            self.finalize()
        if self.fmt in (7, 8):
            check(lib().srk_pack_weights_h16(self._dev.data_ptr(), self._n, self._total, self.fmt, stream_ptr()), "srk_pack_weights_h16")
            return
        fn = lib().srk_pack_weights_bf16x3 if self.fmt == 1 else lib().srk_pack_weights
        check(fn(self._dev.data_ptr(), self._n, self._total, stream_ptr()), "srk_pack_weights")


def pixel_shuffle_fwd(x, y, N, H, W, C_):
    check(lib().srk_pixel_shuffle_fwd(x.data_ptr(), y.data_ptr(), N, H, W, C_, stream_ptr()), "srk_pixel_shuffle_fwd")


def pixel_shuffle_bwd(dy, dx, N, H, W, C_):
    check(lib().srk_pixel_shuffle_bwd(dy.data_ptr(), dx.data_ptr(), N, H, W, C_, stream_ptr()), "srk_pixel_shuffle_bwd")


def nchw_to_nhwc(x, y: View, N, C_, H, W):
    check(lib().srk_nchw_to_nhwc(x.data_ptr(), y.t.data_ptr(), y.ldc, y.coff, N, C_, H, W, stream_ptr()), "srk_nchw_to_nhwc")


def nhwc_to_nchw(x: View, y, N, C_, H, W):
    check(lib().srk_nhwc_to_nchw(x.t.data_ptr(), x.ldc, x.coff, y.data_ptr(), N, C_, H, W, stream_ptr()), "srk_nhwc_to_nchw")


def sum_pool_fwd(x, y, NC, H, W, k):
    check(lib().srk_sum_pool_fwd(x.data_ptr(), y.data_ptr(), NC, H, W, k, stream_ptr()), "srk_sum_pool_fwd")


def sum_pool_bwd(dy, dx, NC, H, W, k):
    check(lib().srk_sum_pool_bwd(dy.data_ptr(), dx.data_ptr(), NC, H, W, k, stream_ptr()), "srk_sum_pool_bwd")


def lrelu_grad_mul(x, g, out, slope):
    """out = g * LeakyReLU'(x) in one pass (srk.h: srk_lrelu_grad_mul); x, g, out contiguous fp32 of equal size."""
    if not (x.is_contiguous() and g.is_contiguous() and out.is_contiguous()) or x.numel() != g.numel() or out.numel() != g.numel():
        raise ValueError("lrelu_grad_mul: contiguous tensors of equal size expected")
    check(lib().srk_lrelu_grad_mul(x.data_ptr(), g.data_ptr(), out.data_ptr(), g.numel(), float(slope), stream_ptr()), "srk_lrelu_grad_mul")


class AdamTable:
    """The pointer table of one srk_adam_step launch: rows (param, grad, exp_avg, exp_avg_sq addresses, numel) of fp32 tensors.  Up to 64
    rows travel in the kernel arguments; a larger table is copied to the device once, through pinned memory on the launching stream."""

    def __init__(self, device, rows):
        n = len(rows)
        arr = (AdamEntry * n)()
        for e, (p, g, m, v, numel) in zip(arr, rows):
            e.p, e.g, e.m, e.v, e.n = p, g, m, v, numel
        total = C.c_int64(0)
        check(lib().srk_adam_plan(arr, n, C.byref(total)), "srk_adam_plan")
        self._host, self._n, self._total, self._dev = arr, n, total.value, None
        if n > 64:
            self._pin = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).pin_memory()
            self._dev = torch.empty(self._pin.numel(), dtype=torch.uint8, device=device)
            self._dev.copy_(self._pin, non_blocking=True)

    def run(self, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=None, found_inf=None):
        """step: fp32 device scalar holding the step number of THIS update"""
        if self._dev is None:
            check(lib().srk_adam_step_small(self._host, self._n, self._total, lr, beta1, beta2, eps, weight_decay, step.data_ptr(),
                                            ptr(grad_scale), ptr(found_inf), stream_ptr()), "srk_adam_step_small")
            return
        check(lib().srk_adam_step(self._dev.data_ptr(), self._n, self._total, lr, beta1, beta2, eps, weight_decay, step.data_ptr(),
                                  ptr(grad_scale), ptr(found_inf), stream_ptr()), "srk_adam_step")
