"""``MicroTree`` -- a whole contraction tree of small tensors executed by ONE kernel launch.

Circuit amplitudes (``Circuit.amplitude``, quimb/tensor/circuit/exact.py:417-501) contract hundreds
of tensors of a few elements each: the reference (and ``TreeExecutor``) pay one dispatch per pairwise
step, which is all the time there is.  ``MicroTree`` compiles the executor's plan into a device-resident
step table (``qamd_micro_step``, include/quimb_amd.h) that one workgroup walks on its own
(csrc/microtree.hip); a grid of workgroups evaluates independent *instances* of the same tree -- e.g.
one output bitstring each, sharing every gate tensor -- in the same launch.

    mt = quimb_amd.MicroTree(tree, "complex64")
    amp = mt(arrays)                                   # one instance -> Array of the output shape
    amps = mt.run_batch([arrays_0, arrays_1, ...])     # (ninst, *output_shape)
"""

import ctypes as C
import os

import numpy as np

from .array import Array, asarray
from .pairwise import prod

MAX_GROUPS = 8
KMAX = 4096


class MicroStepStruct(C.Structure):
    """Mirror of ``qamd_micro_step``."""

    _fields_ = [
        ("a_kind", C.c_int32), ("b_kind", C.c_int32),
        ("a_ref", C.c_int64), ("b_ref", C.c_int64), ("c_off", C.c_int64),
        ("nb", C.c_int32), ("nm", C.c_int32), ("nn", C.c_int32), ("nk", C.c_int32),
        ("B", C.c_uint32), ("M", C.c_uint32), ("N", C.c_uint32), ("K", C.c_uint32),
        ("dim_b", C.c_uint32 * 4), ("dim_m", C.c_uint32 * MAX_GROUPS), ("dim_n", C.c_uint32 * MAX_GROUPS),
        ("dim_k", C.c_uint32 * MAX_GROUPS),
        ("sa_b", C.c_int32 * 4), ("sb_b", C.c_int32 * 4), ("sc_b", C.c_int32 * 4),
        ("sa_m", C.c_int32 * MAX_GROUPS), ("sc_m", C.c_int32 * MAX_GROUPS),
        ("sb_n", C.c_int32 * MAX_GROUPS), ("sc_n", C.c_int32 * MAX_GROUPS),
        ("sa_k", C.c_int32 * MAX_GROUPS), ("sb_k", C.c_int32 * MAX_GROUPS),
    ]


class MicroTree:
    def __init__(self, tree, dtype, max_elems=1 << 16):
        from .executor import TreeExecutor

        old = os.environ.get("QAMD_CHAIN2")
        os.environ["QAMD_CHAIN2"] = "0"        # one plan entry per pairwise step
        try:
            ex = TreeExecutor(tree, dtype)
        finally:
            if old is None:
                del os.environ["QAMD_CHAIN2"]
            else:
                os.environ["QAMD_CHAIN2"] = old
        if tree.nslices != 1:
            raise ValueError("MicroTree: sliced trees are not supported")
        self.tree, self.dtype = tree, np.dtype(dtype)
        self.ninputs = len(tree.inputs)
        self.out_shape = tuple(tree.size_dict[ix] for ix in tree.output)
        self.out_elems = max(prod(self.out_shape), 1)
        steps, arena = [], 0
        loc = {}                                   # node id -> ("in", i) | ("arena", offset)
        for i in range(self.ninputs):
            loc[i] = (0, i)
        for entry in ex.plan:
            if entry[0] != "pair":
                raise ValueError(f"MicroTree: unsupported plan entry {entry[0]!r}")
            _, a, b, res, st = entry
            if st.kind != "gett" or any(st.pre):
                raise ValueError("MicroTree: steps with single-operand preprocessing are not supported")
            sp = st.spec
            ka, kb = (b, a) if st.swapped else (a, b)
            size = prod(st.out_shape)
            if size > max_elems or sp.K > KMAX or len(sp.b) > 4 or max(len(sp.m), len(sp.n), len(sp.k)) > MAX_GROUPS:
                raise ValueError("MicroTree: a step is too large for the single-workgroup kernel")
            d = dict(a=loc[ka], b=loc[kb], spec=sp, size=size)
            if res == ex.root:
                d["c_off"] = -1
            else:
                d["c_off"] = arena
                loc[res] = (1, arena)
                arena += size
            steps.append(d)
        if not steps:
            raise ValueError("MicroTree: nothing to contract")
        if steps[-1]["c_off"] != -1:
            raise ValueError("MicroTree: the root is not produced by the last step")
        self.steps = steps
        self.arena_elems = arena
        self.flops = sum((8 if self.dtype.kind == "c" else 2) * s["spec"].mults for s in steps)
        self._packed = None

    # ---- plan serialisation ---------------------------------------------------------------
    def packed(self):
        """The step table as bytes (array of ``qamd_micro_step``)."""
        if self._packed is None:
            arr = (MicroStepStruct * len(self.steps))()
            for t, s in zip(arr, self.steps):
                sp = s["spec"]
                t.a_kind, t.a_ref = s["a"]
                t.b_kind, t.b_ref = s["b"]
                t.c_off = s["c_off"]
                t.nb, t.nm, t.nn, t.nk = len(sp.b), len(sp.m), len(sp.n), len(sp.k)
                t.B, t.M, t.N, t.K = sp.B, sp.M, sp.N, sp.K
                for i, (d, sa, sb, sc) in enumerate(sp.b):
                    t.dim_b[i], t.sa_b[i], t.sb_b[i], t.sc_b[i] = d, sa, sb, sc
                for i, (d, sa, _, sc) in enumerate(sp.m):
                    t.dim_m[i], t.sa_m[i], t.sc_m[i] = d, sa, sc
                for i, (d, _, sb, sc) in enumerate(sp.n):
                    t.dim_n[i], t.sb_n[i], t.sc_n[i] = d, sb, sc
                for i, (d, sa, sb, _) in enumerate(sp.k):
                    t.dim_k[i], t.sa_k[i], t.sb_k[i] = d, sa, sb
            self._packed = bytes(arr)
        return self._packed

    # ---- execution --------------------------------------------------------------------------
    def _check(self, arrays):
        if len(arrays) != self.ninputs:
            raise ValueError(f"expected {self.ninputs} arrays, got {len(arrays)}")
        xs = [asarray(x).astype(self.dtype) for x in arrays]
        for x, t in zip(xs, self.tree.inputs):
            want = tuple(self.tree.size_dict[ix] for ix in t)
            if x.shape != want:
                raise ValueError(f"array shape {x.shape} does not match indices {t} with sizes {want}")
        return xs

    def bind(self, arrays):
        """Upload / validate the inputs once; the returned object launches with no per-call Python work
        beyond one pointer-table copy."""
        return BoundMicroTree(self, self._check(arrays))

    def run_batch(self, instances):
        """``instances``: sequence of complete input lists, one per instance (generic but slow to set up:
        prefer ``bind(arrays).batch(select)`` when instances differ in a few tensors only).
        Returns an ``Array`` of shape (ninst, *output_shape)."""
        insts = [self._check(arrays) for arrays in instances]
        if not insts:
            raise ValueError("no instances")
        dev = insts[0][0]._dev
        table = np.array([[dev.buffer_address(x._buf) for x in xs] for xs in insts], dtype=np.int64)
        out = Array.empty((len(insts),) + self.out_shape, self.dtype, dev)
        dev.microtree_run(self, table, insts, out._buf)
        return out

    def __call__(self, arrays):
        return self.bind(arrays)()


class BoundMicroTree:
    """A ``MicroTree`` with its inputs resident on the device and their addresses tabulated."""

    def __init__(self, mt, xs):
        self.mt, self.xs = mt, xs
        self.dev = xs[0]._dev
        self.row = np.array([self.dev.buffer_address(x._buf) for x in xs], dtype=np.int64)

    def __call__(self):
        mt = self.mt
        out = Array.empty((1,) + mt.out_shape, mt.dtype, self.dev)
        self.dev.microtree_run(mt, self.row[None, :], self.xs, out._buf)
        return out.reshape(mt.out_shape)

    def batch(self, select):
        """``select``: {input position: (candidate arrays, choice)} with ``choice`` an int array of length
        ninst -- instance i uses ``candidates[choice[i]]`` at that position and the bound tensor everywhere
        else (amplitudes of many bitstrings: position = a qubit's <b| vector, candidates = (<0|, <1|)).
        Returns an ``Array`` of shape (ninst, *output_shape)."""
        mt = self.mt
        ninst = None
        keep = [self.xs]
        cols = {}
        for pos, (cands, choice) in select.items():
            choice = np.asarray(choice, dtype=np.int64)
            if ninst is None:
                ninst = len(choice)
            elif len(choice) != ninst:
                raise ValueError("all choices must have one entry per instance")
            cs = [asarray(c).astype(mt.dtype) for c in cands]
            want = tuple(mt.tree.size_dict[ix] for ix in mt.tree.inputs[pos])
            if any(c.shape != want for c in cs):
                raise ValueError(f"candidate shape mismatch at input {pos}")
            keep.append(cs)
            cols[pos] = np.array([self.dev.buffer_address(c._buf) for c in cs], dtype=np.int64)[choice]
        if not ninst:
            raise ValueError("empty batch")
        table = np.tile(self.row, (ninst, 1))
        for pos, col in cols.items():
            table[:, pos] = col
        out = Array.empty((ninst,) + mt.out_shape, mt.dtype, self.dev)
        self.dev.microtree_run(mt, table, keep, out._buf)
        return out
