"""Helpers for the GPU parity tests: NHWC plumbing around the raw C-ABI calls."""
import importlib
import torch

L = importlib.import_module("super-resolution_amd")._lib


def nhwc(x_nchw: torch.Tensor, ldc=None, coff=0, fill=0.0):
    """NCHW cpu -> [N,H,W,ldc] cuda with the tensor placed at channel offset coff."""
    n, c, h, w = x_nchw.shape
    ldc = ldc or c
    buf = torch.full((n, h, w, ldc), fill, dtype=torch.float32)
    buf[..., coff:coff + c] = x_nchw.permute(0, 2, 3, 1)
    return buf.cuda().contiguous()


def nchw(buf_nhwc: torch.Tensor, coff=0, c=None):
    c = c or buf_nhwc.shape[3] - coff
    return buf_nhwc[..., coff:coff + c].permute(0, 3, 1, 2).contiguous().cpu()


def pack_fwd(w_oihw: torch.Tensor, ps=False, scale=1.0, fmt=0):
    """OIHW cpu weight -> packed cuda tensor for the forward conv (fmt 0: direct fp32, 3: Winograd fp32)."""
    co, ci = w_oihw.shape[:2]
    src = w_oihw.contiguous().cuda()
    dst = torch.empty(L.packed_floats(ci, co, fmt), dtype=torch.float32, device="cuda")
    t = L.PackTable(src.device, fmt)
    t.add(src, dst, M=co, k_off=0, k_len=ci, K_total=ci, ps=ps, scale=scale)
    t.run()
    torch.cuda.synchronize()
    return dst, src


def pack_bwd(w_oihw: torch.Tensor, ps=False, scale=1.0, c_begin=0, c_len=None, fmt=0):
    """OIHW cpu weight -> packed cuda tensor of the data-gradient conv (K = Cout, M = Cin slice)."""
    co, ci = w_oihw.shape[:2]
    c_len = c_len or ci
    src = w_oihw.contiguous().cuda()
    dst = torch.empty(L.packed_floats(co, c_len, fmt), dtype=torch.float32, device="cuda")
    t = L.PackTable(src.device, fmt)
    t.add(src, dst, M=c_len, k_off=0, k_len=co, K_total=co, transpose=True, c_begin=c_begin, ps=ps, scale=scale)
    t.run()
    torch.cuda.synchronize()
    return dst, src


def rel_err(a: torch.Tensor, b: torch.Tensor):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()
