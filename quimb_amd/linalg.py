"""``quimb_amd.linalg`` -- decompositions a backend must *answer* for so that
quimb's split/compress drivers (quimb/tensor/decomp.py:829-1118) keep tensors on
the device between contractions.  They are OUT of the hot-path scope (SURVEY.md
section 2.1): no hand-written kernels here, the calls go to rocSOLVER through
torch on the array's own device storage and are not counted in any metric.
"""

import numpy as np

from .array import Array


def _as_torch(x):
    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    buf = x._buf
    if not hasattr(buf, "view"):
        raise TypeError("quimb_amd.linalg needs device-resident arrays")
    return x, buf[: x.size].view(x.shape)


def _wrap(x, t):
    t = t.contiguous()
    return Array(x._dev, t.reshape(-1), tuple(t.shape), x.dtype if t.dtype == x._buf.dtype else _np_dtype(t))


def _np_dtype(t):
    import torch

    return {torch.float32: "float32", torch.float64: "float64", torch.complex64: "complex64",
            torch.complex128: "complex128"}[t.dtype]


def svd(x, full_matrices=False):
    import torch

    x, t = _as_torch(x)
    u, s, vh = torch.linalg.svd(t, full_matrices=full_matrices)
    return _wrap(x, u), Array(x._dev, s.contiguous().reshape(-1), tuple(s.shape), _np_dtype(s)), _wrap(x, vh)


def qr(x, mode="reduced"):
    import torch

    x, t = _as_torch(x)
    q, r = torch.linalg.qr(t, mode=mode)
    return _wrap(x, q), _wrap(x, r)


def eigh(x):
    import torch

    x, t = _as_torch(x)
    w, v = torch.linalg.eigh(t)
    return Array(x._dev, w.contiguous().reshape(-1), tuple(w.shape), _np_dtype(w)), _wrap(x, v)


def norm(x, ord=None):
    from .ops import norm_fro

    if ord not in (None, "fro", 2):
        raise NotImplementedError("only the Frobenius / vector 2-norm is provided")
    return norm_fro(x)
