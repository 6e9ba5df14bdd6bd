"""MI355X-native ESRGAN hot path (drop-in for lukas-blecher/super-resolution's models.py surface).

Import by string (the directory name carries a hyphen)::

    import importlib; sr = importlib.import_module("super-resolution_amd")
    G = sr.models.GeneratorRRDB(1, filters=64, num_res_blocks=23, num_upsample=2).cuda()
"""
from . import _lib, ops, engine, models, evaluation, losses, datasets  # noqa: F401
from .models import (GeneratorRRDB, Markovian_Discriminator, Standard_Discriminator, Conditional_Discriminator, SumPool2d, DenseResidualBlock,  # noqa: F401
                     ResidualInResidualDenseBlock, Conv3x3, discriminator_block, weight_reset, uniform_reset)

__all__ = ["models", "ops", "engine", "_lib", "losses", "datasets", "GeneratorRRDB", "Markovian_Discriminator", "Standard_Discriminator", "Conditional_Discriminator", "SumPool2d",
           "DenseResidualBlock", "ResidualInResidualDenseBlock", "Conv3x3", "discriminator_block", "weight_reset", "uniform_reset"]
