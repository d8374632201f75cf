"""``TNLinearOperator`` -- a tensor network viewed as a linear operator, with the
operator tensors resident on the device and one cached contraction expression per
call signature.  Mirror of the reference class (quimb/tensor/tensor_core.py:12297-12549):
``_matvec`` reshapes the vector to ``rdims``, runs the cached expression with the
operator tensors as constants (``constants=range(n)``, :12378-12381) and ravels the
result (:12393-12417); ``_matmat`` does the same with one extra column index
(:12427-12443).  This is the hot loop of DMRG's local eigensolve
(quimb/tensor/tn1d/dmrg.py:681-732, ARPACK calling ``A.matvec``).
"""

import numpy as np

from .array import Array, asarray
from .contract import array_contract_expression
from .pairwise import prod


class TNLinearOperator:
    def __init__(self, tensors, left_inds, right_inds, ldims=None, rdims=None, optimize=None, dtype=None,
                 graph=False):
        """``tensors``: sequence of (array, inds) -- or objects with ``.data`` / ``.inds``.
        ``graph=True``: matvecs on device vectors replay ONE recorded hipGraph of the planned steps
         instead of launching them one by one from Python."""
        arrs, inds = [], []
        for t in tensors:
            a, i = (t.data, t.inds) if hasattr(t, "inds") else t
            arrs.append(a)
            inds.append(tuple(i))
        self.left_inds, self.right_inds = tuple(left_inds), tuple(right_inds)
        size = {}
        for a, i in zip(arrs, inds):
            for ix, d in zip(i, np.shape(a) if not isinstance(a, Array) else a.shape):
                size[ix] = int(d)
        self.ldims = tuple(ldims) if ldims is not None else tuple(size[ix] for ix in self.left_inds)
        self.rdims = tuple(rdims) if rdims is not None else tuple(size[ix] for ix in self.right_inds)
        for ix, d in zip(self.right_inds, self.rdims):
            size.setdefault(ix, d)
        self.shape = (prod(self.ldims), prod(self.rdims))
        if dtype is None:
            dtype = np.result_type(*[a.dtype for a in arrs])
        self.dtype = np.dtype(dtype)
        self._arrays = [asarray(a).astype(self.dtype) for a in arrs]  # uploaded once
        self._inds = inds
        self._size = size
        self._optimize = optimize
        self._exprs = {}
        self.is_conj = False
        self._use_graph = bool(graph)
        self._graphed = None

    def _expr(self, ncols):
        ex = self._exprs.get(ncols)
        if ex is None:
            n = len(self._arrays)
            vin = self.right_inds + (("__col__",) if ncols else ())
            out = self.left_inds + (("__col__",) if ncols else ())
            size = dict(self._size)
            if ncols:
                size["__col__"] = ncols
            ex = array_contract_expression(
                list(self._inds) + [vin], out, size_dict=size, optimize=self._optimize, dtype=self.dtype,
                constants={i: self._arrays[i] for i in range(n)}, cache=False,
            )
            self._exprs[ncols] = ex
        return ex

    def _apply(self, x, ncols, borrow=False):
        host = not isinstance(x, Array)
        xin = asarray(x)
        if xin.dtype.kind == "c" and self.dtype.kind != "c":
            # a real operator on complex data (``tn_lo.dot(X)`` with complex X in the reference's own test,
            # tests/test_tensor/test_tensor_core.py:2200-2202): linear, so real and imaginary parts separately
            from . import ops

            re, im = self._apply(ops.real(xin), ncols), self._apply(ops.imag(xin), ncols)
            out = asarray(re).astype(xin.dtype) + asarray(im).astype(xin.dtype) * 1j
            return out.to_numpy() if host else out
        xd = xin.astype(self.dtype)
        xd = xd.reshape(self.rdims + ((ncols,) if ncols else ()))
        if self.is_conj:
            xd = xd.conj()
        if self._use_graph and not ncols and not host and hasattr(xd._dev, "torch"):
            if self._graphed is None:
                from .executor import GraphedContraction

                x0 = Array.full(self.rdims, 0.0, self.dtype, xd._dev)
                expr = self._expr(0)
                ins, where = expr.exec_inputs([x0])       # constants may have been folded: the executed tree's own list
                self._graphed = GraphedContraction(expr.executor, ins)
                self._graph_slot = where[0]
            self._graphed.update(self._graph_slot, xd)
            out = self._graphed.replay()             # the graph's own output buffer, reused by the next replay:
            if not borrow:                           # handed out as it is only to a caller that says it is done with it by then
                out = out.copy()
        else:
            out = self._expr(ncols)(xd)
        if isinstance(out, np.ndarray):
            out = asarray(out)
        if self.is_conj:
            out = out.conj()
        out = out.reshape((self.shape[0], ncols) if ncols else (self.shape[0],))
        return out.to_numpy() if host else out

    def matvec(self, vec):
        return self._apply(vec, 0)

    def matvec_borrow(self, vec):
        """``matvec`` whose result may be a buffer the operator owns, valid (and the caller's to overwrite) until the next
        application -- what an iterative solver that consumes each product before asking for the next one needs; saves
        one pass over the vector per application."""
        return self._apply(vec, 0, borrow=True)

    _matvec = matvec

    def matmat(self, mat):
        ncols = mat.shape[-1]
        return self._apply(mat, ncols)

    _matmat = matmat

    def __matmul__(self, x):
        return self.matvec(x) if len(x.shape) == 1 else self.matmat(x)

    dot = __matmul__

    def conj(self):
        import copy

        new = copy.copy(self)
        new.is_conj = not self.is_conj
        new._graphed = None
        return new

    def _transpose(self):
        """The transposed operator: the same tensors with the roles of ``left_inds`` and ``right_inds`` swapped
        (``TNLinearOperator._transpose``, tensor_core.py:12480-12500)."""
        import copy

        new = copy.copy(self)
        new.left_inds, new.right_inds = self.right_inds, self.left_inds
        new.ldims, new.rdims = self.rdims, self.ldims
        new.shape = (self.shape[1], self.shape[0])
        new._exprs = {}
        new._graphed = None
        return new

    @property
    def T(self):
        return self._transpose()

    def _adjoint(self):
        return self._transpose().conj()

    @property
    def H(self):
        return self._adjoint()

    def rmatvec(self, vec):
        """``A^H @ vec`` (what scipy's ``svds`` / ``lsqr`` ask a LinearOperator for)."""
        return self.H.matvec(vec)

    _rmatvec = rmatvec

    def rmatmat(self, mat):
        return self.H.matmat(mat)

    _rmatmat = rmatmat

    def trace(self):
        """Contract left and right indices pairwise without forming the matrix
        (``TNLinearOperator.trace``, tensor_core.py:12530-12549; the reference reaches it through ``np.trace``)."""
        if self.shape[0] != self.shape[1] or self.ldims != self.rdims:
            raise ValueError("trace needs matching left and right index dimensions")
        from .contract import array_contract

        ren = dict(zip(self.right_inds, self.left_inds))
        inds = [tuple(ren.get(ix, ix) for ix in t) for t in self._inds]
        arrays = [a.conj() for a in self._arrays] if self.is_conj else self._arrays
        out = array_contract(arrays, inds, (), optimize=self._optimize)
        return np.asarray(out.to_numpy() if hasattr(out, "to_numpy") else out).item()

    def aslinearoperator(self):
        """A ``scipy.sparse.linalg.LinearOperator`` view (host vectors in and out) for scipy's iterative solvers."""
        from scipy.sparse.linalg import LinearOperator

        dt = np.dtype(self.dtype)
        return LinearOperator(self.shape, matvec=self.matvec, rmatvec=self.rmatvec, matmat=self.matmat,
                              rmatmat=self.rmatmat, dtype=dt)

    def to_dense(self):
        eye = np.eye(self.shape[1], dtype=self.dtype)
        return self.matmat(eye)
