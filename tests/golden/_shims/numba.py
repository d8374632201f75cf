"""Identity-decorator stand-in for numba (quimb/core.py:51 njit wrappers):
kernels run as plain python/numpy.  Only used by tests/golden/make_golden.py."""
import numpy as np

__version__ = "0.0-shim"


def _decorator(*args, **kwargs):
    if len(args) == 1 and callable(args[0]):      # also njit(fn, cache=True), quimb/core.py:50
        return args[0]
    def wrap(fn):
        return fn
    return wrap


njit = jit = vectorize = guvectorize = generated_jit = _decorator
prange = range


class _Types:
    def __getattr__(self, name):
        return self
    def __call__(self, *a, **k):
        return self
    def __getitem__(self, k):
        return self


types = _Types()
float64 = complex128 = int64 = int32 = float32 = complex64 = boolean = uint8 = uint64 = types
typed = types


def set_num_threads(n):
    pass


def get_num_threads():
    return 1


class config:
    NUMBA_NUM_THREADS = 1
