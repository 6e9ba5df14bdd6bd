#!/usr/bin/env python3
"""G16: the ConvTranspose2d upsampling variants of GeneratorRRDB (models.py:69-83; --use_transposed_conv /
--fully_transposed_conv) from the *imported* reference (build container only): checks oracle == reference and stores inputs,
outputs, the input gradient and two weight gradients, plus the state_dict key lists.  Weights are closed-form."""
import os, sys
sys.dont_write_bytecode = True
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = "/root/reference"
if not os.path.isdir(REF):
    sys.exit("reference not present; goldens can only be regenerated in the build container")
sys.path.insert(0, REF)
import models as ref  # noqa: E402
from oracle import esrgan_oracle as O  # noqa: E402

out = {}
keys_txt = []
for tag, kw, U in (("tc", dict(use_transposed_conv=True), 2), ("full", dict(fully_tconv_upsample=True), 2)):
    torch.manual_seed(0)
    g = ref.GeneratorRRDB(1, 16, 1, num_upsample=U, res_scale=0.1, **kw)
    sd = O.closed_form_fill({k: v.clone() for k, v in g.state_dict().items()})
    g.load_state_dict(sd)
    g.train()
    assert {k: tuple(v.shape) for k, v in sd.items()} == O.generator_state_shapes(1, 16, 1, U, **kw), tag
    lr, _ = O.jet_images(2, 1, 8 * 2 ** U, 8 * 2 ** U, 5, 2 ** U)
    x = lr.clone().requires_grad_(True)
    y = g(x)
    tgt = torch.rand(y.shape, generator=torch.Generator().manual_seed(6))
    (y - tgt).abs().mean().backward()
    sdo = {k: (v.clone().requires_grad_(True) if k not in ("power", "multiplier") else v) for k, v in sd.items()}
    xo = lr.clone().requires_grad_(True)
    yo, _ = O.generator_forward(sdo, xo, 1, U, 0.1, training=True, **kw)
    (yo - tgt).abs().mean().backward()
    assert (yo - y).abs().max().item() <= 1e-6 * max(1.0, y.abs().max().item()), tag
    tk = [k for k in sd if k.startswith("upsampling.") and k.endswith("weight")]
    for k in tk + ["conv1.weight"]:
        ref_g = dict(g.named_parameters())[k].grad
        assert (sdo[k].grad - ref_g).abs().max().item() <= 1e-6 * max(1.0, ref_g.abs().max().item()), (tag, k)
        out[f"{tag}.grad.{k}"] = ref_g.numpy()
    out[f"{tag}.lr"], out[f"{tag}.y"], out[f"{tag}.tgt"], out[f"{tag}.dx"] = lr.numpy(), y.detach().numpy(), tgt.numpy(), x.grad.numpy()
    keys_txt.append(tag + ": " + " ".join(g.state_dict().keys()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "G16_tconv_generators.npz"), **out)
open(os.path.join(ROOT, "tests", "golden", "G16_state_keys.txt"), "w").write("\n".join(keys_txt) + "\n")
print("G16 written; oracle == reference for both variants")
