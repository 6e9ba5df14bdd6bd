#!/usr/bin/env python3
"""Data-gradient chain of a dense block (FMT=7|8: 16-bit storage, 8 x 128 x 128; FMT=6: fp32 F(2x4,3x3), 32 x 64 x 64) with the LeakyReLU'
masks read from the forward activations vs from sign bits, and the forward chain with / without writing the bits."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import chain_check as cc
L = cc.L
N, H = (32, 64) if cc.fmt == 6 else (8, 128)


def timed(blocks):
    for i in range(6): L.conv3x3_seq(blocks[i % len(blocks)])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(60): L.conv3x3_seq(blocks[i % len(blocks)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 60 * 1e3


cc.set_chain(1)
fw = [cc.make_block(N, H, H, 100 + i, False) for i in range(6)]
bw = [cc.make_block(N, H, H, 200 + i, True) for i in range(6)]
nb = L.conv_seq_signs_bytes(fw[0][2])
assert nb > 0, "this build offers no sign bits for this format"
sg = [torch.zeros(4, nb, dtype=torch.uint8, device="cuda") for _ in range(6)]
fw_plain = [b[2] for b in fw]
fw_signs = [[(x, wp, b_, y, dict(kw, signs_out=sg[i][k]) if k < 4 else kw) for k, (x, wp, b_, y, kw) in enumerate(b[2])] for i, b in enumerate(fw)]
bw_mask = [b[2] for b in bw]
bw_signs = []
for i, b in enumerate(bw):
    cs = []
    for k, (x, wp, b_, y, kw) in enumerate(b[2]):
        kw = dict(kw)
        if k < 4:
            kw.pop("mask"); kw["mask_signs"] = sg[i][k]
        cs.append((x, wp, b_, y, kw))
    bw_signs.append(cs)
for rep in range(2):
    print("forward chain: plain %.1f us, writing sign bits %.1f us | data-gradient chain: mask tensors %.1f us, sign bits %.1f us"
          % (timed(fw_plain), timed(fw_signs), timed(bw_mask), timed(bw_signs)), flush=True)
