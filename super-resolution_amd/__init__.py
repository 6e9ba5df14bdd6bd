"""MI355X-native ESRGAN hot path (drop-in for lukas-blecher/super-resolution's models.py surface).

Import by string (the directory name carries a hyphen)::

    import importlib; sr = importlib.import_module("super-resolution_amd")
    G = sr.models.GeneratorRRDB(1, filters=64, num_res_blocks=23, num_upsample=2).cuda()
"""
from . import _lib  # noqa: F401  (ctypes binding of libsrk.so; loads lazily, fails loudly)

__all__ = ["_lib"]
