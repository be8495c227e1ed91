"""Shared plumbing of the four `utils/lib/*/interface.py` shims: the host-pointer builder entry points
(`deftet_*_host` keep the argument lists of the reference's `extern "C" void run(...)` and add an int status) are
looked up once, inputs are checked the way the reference's interfaces check them (an AssertionError for the wrong
dtype), output buffers are plain numpy arrays handed over by pointer."""
import ctypes as c

import numpy as np

from deftet_amd import _lib

I32P = c.POINTER(c.c_int32)
F32P = c.POINTER(c.c_float)


def host_fn(name, argtypes):
    fn = getattr(_lib.load(), name)
    fn.argtypes = argtypes
    fn.restype = c.c_int
    return fn


def call(fn, what, *args):
    _lib.check(fn(*args), what)


def checked(array, dtype):
    """C-contiguous view / copy of `array`; the reference's interfaces assert the dtype instead of converting."""
    assert array.dtype == dtype
    return np.ascontiguousarray(array)


def ptr(array):
    return array.ctypes.data_as(F32P if array.dtype == np.float32 else I32P)


def out_i32(*shape):
    return np.zeros(shape, dtype=np.int32)
