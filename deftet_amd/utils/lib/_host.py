"""Shared loader for the host-pointer builder entry points (`deftet_*_host`), which keep
the argument lists of the reference's `extern "C" void run(...)` functions."""
import ctypes as c

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
