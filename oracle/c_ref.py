"""ctypes access to oracle/conv_ref.c (TEST INFRASTRUCTURE ONLY)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libconvref.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def _lib():
    if not os.path.exists(_SO):
        build()
    return C.CDLL(_SO)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def conv3x3(x, w, b, stride=1, slope=1.0):
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w, np.float32)
    n, ci, h, wd = x.shape
    co = w.shape[0]
    oh, ow = (h - 1) // stride + 1, (wd - 1) // stride + 1
    y = np.empty((n, co, oh, ow), np.float32)
    bb = None if b is None else np.ascontiguousarray(b, np.float32)
    _lib().ref_conv3x3(_p(x), _p(w), None if bb is None else _p(bb), _p(y), n, ci, h, wd, co, stride, C.c_float(slope))
    return y


def pixel_shuffle2(x):
    x = np.ascontiguousarray(x, np.float32)
    n, c4, h, w = x.shape
    y = np.empty((n, c4 // 4, 2 * h, 2 * w), np.float32)
    _lib().ref_pixel_shuffle2(_p(x), _p(y), n, c4, h, w)
    return y


def sum_pool(x, k):
    x = np.ascontiguousarray(x, np.float32)
    n, c, h, w = x.shape
    y = np.empty((n, c, h // k, w // k), np.float32)
    _lib().ref_sum_pool(_p(x), _p(y), n * c, h, w, k)
    return y
