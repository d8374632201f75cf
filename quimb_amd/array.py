"""``quimb_amd.Array`` -- the device array class quimb's autoray boundary sees.

Implements the protocol the reference's hot path exercises on its arrays
(SURVEY.md section 8b): ``.shape`` (quimb/tensor/array_ops.py:31), ``.dtype``
(tensor_core.py:2689), ``.ndim``, basic ``__getitem__`` (tensor_core.py:2343),
scalar ``* / - **`` (tensor_core.py:340, 3771, 3813), ``.conj()``, ``.ravel()``
(tensor_core.py:12417), ``.reshape``, ``.astype`` (tensor_core.py:2716),
``.item()``.  The class lives in the top-level module ``quimb_amd`` so that
``autoray.infer_backend`` resolves it to the backend name ``"quimb_amd"``.

All functions are pure (the reference assumes purity,
docs/tensor/tensor-design.ipynb:529): results are new arrays, inputs are never
mutated; ``reshape``/``ravel`` share storage like numpy views of contiguous data.
Arrays are always C-contiguous on the device.
"""

import numbers

import numpy as np

from . import device as _device
from .pairwise import contig_strides, prod

_REAL_OF = {
    np.dtype("float32"): np.dtype("float32"),
    np.dtype("float64"): np.dtype("float64"),
    np.dtype("complex64"): np.dtype("float32"),
    np.dtype("complex128"): np.dtype("float64"),
}
_SUPPORTED = tuple(_REAL_OF)


def _coerce_dtype(dt):
    dt = np.dtype(dt)
    if dt in _REAL_OF:
        return dt
    if dt.kind in "iub":
        return np.dtype("float64")
    if dt == np.dtype("float16"):
        return np.dtype("float32")
    raise TypeError(f"quimb_amd: unsupported dtype {dt}")


class Array:
    __slots__ = ("_dev", "_buf", "shape", "dtype", "__weakref__")
    __array_priority__ = 1000  # make numpy defer to our reflected operators

    def __init__(self, dev, buf, shape, dtype):
        self._dev = dev
        self._buf = buf
        self.shape = tuple(int(d) for d in shape)
        self.dtype = np.dtype(dtype)

    # ---- construction -------------------------------------------------------
    @classmethod
    def from_numpy(cls, x, dtype=None, dev=None):
        dev = dev or _device.default_device()
        x = np.asarray(x)
        dt = _coerce_dtype(dtype if dtype is not None else x.dtype)
        shape = x.shape  # (ascontiguousarray would promote 0-d to 1-d)
        x = np.ascontiguousarray(x, dtype=dt)
        return cls(dev, dev.from_host(x), shape, dt)

    @classmethod
    def empty(cls, shape, dtype, dev=None):
        dev = dev or _device.default_device()
        shape = (shape,) if isinstance(shape, numbers.Integral) else tuple(shape)
        dt = _coerce_dtype(dtype)
        return cls(dev, dev.empty(prod(shape), dt), shape, dt)

    @classmethod
    def full(cls, shape, value, dtype, dev=None):
        out = cls.empty(shape, dtype, dev)
        out._dev.fill(out._buf, out.size, value, out.dtype)
        return out

    # ---- protocol -------------------------------------------------------------
    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return prod(self.shape)

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    @property
    def device(self):
        return self._dev

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of unsized object")
        return self.shape[0]

    def __repr__(self):
        return f"quimb_amd.Array(shape={self.shape}, dtype={self.dtype.name}, device={self._dev.name})"

    def to_numpy(self):
        return self._dev.to_host(self._buf, self.size, self.dtype).reshape(self.shape)

    def __array__(self, dtype=None, copy=None):
        x = self.to_numpy()
        return x.astype(dtype) if dtype is not None else x

    def item(self):
        if self.size != 1:
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        return self.to_numpy().reshape(()).item()

    def __float__(self):
        return float(self.item())

    def __complex__(self):
        return complex(self.item())

    def copy(self):
        return Array(self._dev, self._dev.clone(self._buf), self.shape, self.dtype)

    # ---- views ------------------------------------------------------------------
    def reshape(self, *shape):
        if len(shape) == 1 and not isinstance(shape[0], numbers.Integral):
            shape = tuple(shape[0])
        shape = [int(s) for s in shape]
        if shape.count(-1) > 1:
            raise ValueError("can only specify one unknown dimension")
        if -1 in shape:
            known = prod(s for s in shape if s != -1)
            if known == 0 or self.size % known:
                raise ValueError(f"cannot reshape array of size {self.size} into shape {tuple(shape)}")
            shape[shape.index(-1)] = self.size // known
        if prod(shape) != self.size:
            raise ValueError(f"cannot reshape array of size {self.size} into shape {tuple(shape)}")
        return Array(self._dev, self._buf, shape, self.dtype)

    def ravel(self):
        return self.reshape(self.size)

    flatten = ravel

    def _strided_copy(self, shape, strides, offset):
        out = Array.empty(shape, self.dtype, self._dev)
        if out.size:
            self._dev.permute(out._buf, self._buf, shape, strides, offset, self.dtype)
        return out

    def transpose(self, *axes):
        if len(axes) == 1 and not isinstance(axes[0], numbers.Integral):
            axes = tuple(axes[0]) if axes[0] is not None else ()
        if not axes:
            axes = tuple(reversed(range(self.ndim)))
        axes = tuple(int(a) % self.ndim for a in axes)
        if sorted(axes) != list(range(self.ndim)):
            raise ValueError("axes don't match array")
        if axes == tuple(range(self.ndim)):
            return self
        st = contig_strides(self.shape)
        return self._strided_copy([self.shape[a] for a in axes], [st[a] for a in axes], 0)

    @property
    def T(self):
        return self.transpose()

    def __getitem__(self, key):
        if not isinstance(key, tuple):
            key = (key,)
        if any(k is Ellipsis for k in key):
            i = next(i for i, k in enumerate(key) if k is Ellipsis)
            nfill = self.ndim - sum(1 for k in key if k is not None and k is not Ellipsis)
            key = key[:i] + (slice(None),) * nfill + key[i + 1 :]
        st = contig_strides(self.shape)
        shape, strides, offset, ax = [], [], 0, 0
        for k in key:
            if k is None:
                shape.append(1)
                strides.append(0)
                continue
            if ax >= self.ndim:
                raise IndexError("too many indices for array")
            d = self.shape[ax]
            if isinstance(k, numbers.Integral):
                k = int(k)
                if k < -d or k >= d:
                    raise IndexError(f"index {k} is out of bounds for axis {ax} with size {d}")
                offset += (k % d) * st[ax]
            elif isinstance(k, slice):
                start, stop, step = k.indices(d)
                n = len(range(start, stop, step))
                shape.append(n)
                strides.append(st[ax] * step)
                offset += start * st[ax]
            else:
                raise IndexError("quimb_amd.Array supports basic (int / slice / None / ...) indexing only")
            ax += 1
        for a in range(ax, self.ndim):
            shape.append(self.shape[a])
            strides.append(st[a])
        return self._strided_copy(shape, strides, offset)

    # ---- arithmetic ----------------------------------------------------------------
    def _scaled(self, factor):
        f = complex(factor)
        dt = self.dtype
        if f.imag != 0 and dt.kind != "c":
            dt = np.dtype("complex64") if dt == np.dtype("float32") else np.dtype("complex128")
        out = self.astype(dt, copy=True)
        out._dev.scale(out._buf, out.size, f, dt)
        return out

    def _binary(self, other, op, reflected=False):
        if isinstance(other, numbers.Number):
            if op == "mul":
                return self._scaled(other)
            # a scalar never widens the array's precision (float32 / numpy.float64(2) stays float32, as x * 2.0
            # always has here); a complex scalar makes a real array complex at ITS precision
            sdt = self.dtype
            if isinstance(other, numbers.Complex) and not isinstance(other, numbers.Real) and sdt.kind != "c":
                sdt = np.dtype("complex64") if sdt == np.dtype("float32") else np.dtype("complex128")
            if sdt.kind not in "fc":
                sdt = np.result_type(sdt, type(other)) if not isinstance(other, numbers.Integral) else sdt
            other = Array.full((), other, sdt, self._dev)
        if isinstance(other, np.ndarray):
            other = Array.from_numpy(other, dev=self._dev)
        if not isinstance(other, Array):
            return NotImplemented
        a, b = (other, self) if reflected else (self, other)
        dt = _coerce_dtype(np.result_type(a.dtype, b.dtype))
        a, b = a.astype(dt), b.astype(dt)
        shape = np.broadcast_shapes(a.shape, b.shape)

        def bstrides(x):
            st = contig_strides(x.shape)
            pad = len(shape) - x.ndim
            return [0] * pad + [0 if d == 1 and D != 1 else s for d, D, s in zip(x.shape, shape[pad:], st)]

        out = Array.empty(shape, dt, self._dev)
        if out.size:
            self._dev.binary(out._buf, a._buf, bstrides(a), b._buf, bstrides(b), shape, op, dt)
        return out

    def __mul__(self, other):
        return self._binary(other, "mul")

    def __rmul__(self, other):
        return self._binary(other, "mul", reflected=True)

    def __add__(self, other):
        return self._binary(other, "add")

    def __radd__(self, other):
        return self._binary(other, "add", reflected=True)

    def __sub__(self, other):
        return self._binary(other, "sub")

    def __rsub__(self, other):
        return self._binary(other, "sub", reflected=True)

    def __truediv__(self, other):
        """Elementwise TRUE division (scalar, array, broadcasting): ``x / c`` is computed as x / c, not x * (1 / c)."""
        if self.dtype.kind not in "fc":
            return self.astype("float64") / other
        return self._binary(other, "div")

    def __rtruediv__(self, other):
        if self.dtype.kind not in "fc":
            return other / self.astype("float64")
        return self._binary(other, "div", reflected=True)

    def __abs__(self):
        from . import ops

        return ops.abs(self)

    def __neg__(self):
        return self._scaled(-1.0)

    def __pos__(self):
        return self

    def __pow__(self, p):
        if self.size == 1 and isinstance(p, numbers.Number):
            return Array.from_numpy(np.asarray(self.item() ** p).reshape(self.shape), dev=self._dev)
        if isinstance(p, numbers.Integral):
            from . import ops

            if p >= 0:
                return ops.power(self, int(p))
            return 1.0 / ops.power(self, int(-p))
        return NotImplemented

    def __matmul__(self, other):
        from .ops import matmul

        return matmul(self, other)

    def conj(self):
        if self.dtype.kind != "c":
            return self
        out = Array.empty(self.shape, self.dtype, self._dev)
        self._dev.conj(out._buf, self._buf, self.size, self.dtype)
        return out

    conjugate = conj

    def astype(self, dtype, copy=False):
        dt = _coerce_dtype(dtype)
        if dt == self.dtype:
            return self.copy() if copy else self
        out = Array.empty(self.shape, dt, self._dev)
        self._dev.cast(out._buf, dt, self._buf, self.dtype, self.size)
        return out

    @property
    def real(self):
        if self.dtype.kind != "c":
            return self
        return self.astype(_REAL_OF[self.dtype])

    @property
    def imag(self):
        if self.dtype.kind != "c":
            return Array.full(self.shape, 0.0, self.dtype, self._dev)
        return (self * (-1j)).astype(_REAL_OF[self.dtype])

    def sum(self, axis=None):
        from .ops import sum as _sum

        return _sum(self, axis=axis)


def asarray(x, dtype=None, dev=None):
    """Convert to a device ``Array`` (no copy if already one of matching dtype).
    Mirrors ``quimb.tensor.array_ops.asarray`` (array_ops.py:21-64): anything
    that already is an ``Array`` is left untouched."""
    if isinstance(x, Array):
        return x if dtype is None else x.astype(dtype)
    return Array.from_numpy(x, dtype=dtype, dev=dev)


def to_numpy(x):
    return x.to_numpy() if isinstance(x, Array) else np.asarray(x)
