"""Eval-mode inference caller of the generator: the HR/LR L1 part of the reference's ``calculate_metrics``
(evaluation/eval.py:455-494).  The energy-mover's-distance term needs ``energyflow`` and is outside this build.

The whole loop stays on the GPU (the reference moves every batch to the CPU for the pooling and the L1): eval-mode
forward on the HIP kernels under ``no_grad``, ``SumPool2d`` on the HIP kernel, per-image means.
"""
import numpy as np
import torch

from . import models


def calculate_metrics(generator, dataset, device, batch_size=4, n_cpu=0, factor=2, reference_labels=True):
    """Returns ``{'hr_l1': {'mean','std'}, 'lr_l1': {...}}``.

    ``reference_labels=True`` reproduces the reference's key assignment, which zips ``['hr_l1','lr_l1',...]`` with
    ``[lr_similarity, hr_similarity, ...]`` (eval.py:491): the value stored under 'hr_l1' is the LR-space L1 and vice
    versa.  Pass False for self-consistent names."""
    generator.eval()
    loader = torch.utils.data.DataLoader(dataset, batch_size=batch_size, shuffle=False, num_workers=n_cpu)
    pool = models.SumPool2d(factor)
    lr_similarity, hr_similarity = [], []
    with torch.no_grad():
        for imgs in loader:
            imgs_lr = imgs["lr"].to(device).float()
            imgs_hr = imgs["hr"].to(device).float()
            gen_hr = generator(imgs_lr)
            gen_lr = pool(gen_hr)
            lr_similarity.append((gen_lr - imgs_lr).abs().mean((1, 2, 3)))
            hr_similarity.append((gen_hr - imgs_hr).abs().mean((1, 2, 3)))
    lr_similarity = torch.cat(lr_similarity).cpu().numpy()
    hr_similarity = torch.cat(hr_similarity).cpu().numpy()
    names = ["hr_l1", "lr_l1"] if reference_labels else ["lr_l1", "hr_l1"]
    return {n: {"mean": float(np.mean(v)), "std": float(np.std(v))} for n, v in zip(names, [lr_similarity, hr_similarity])}
