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
    """Mirror of ``qamd_micro_step`` (the lowered plan entry)."""

    _fields_ = [
        ("a_kind", C.c_int32), ("b_kind", C.c_int32),
        ("a_ref", C.c_int64), ("b_ref", C.c_int64), ("c_off", C.c_int64),
        ("total", C.c_uint32), ("K", C.c_uint32), ("eoff", C.c_uint32), ("koff", C.c_uint32),
    ]


def _bundle_offsets(groups, col):
    """Element offsets of a bundle (mixed radix over its groups, last group fastest)."""
    off = np.zeros(1, dtype=np.int64)
    for g in groups:
        off = (off[:, None] + (np.arange(g[0], dtype=np.int64) * g[col])[None, :]).reshape(-1)
    return off


class MicroTree:
    def __init__(self, tree, dtype, max_elems=1 << 16):
        from .executor import TreeExecutor

        from .options import get_options

        ex = TreeExecutor(tree, dtype, options=get_options().replace(fuse_pairs=False, fuse_rows=False))     # one plan entry per pairwise step
        if tree.nslices != 1:
            raise ValueError("MicroTree: sliced trees are not supported")
        self.tree, self.dtype = tree, np.dtype(dtype)
        self.ninputs = len(tree.inputs)
        self.out_shape = tuple(tree.size_dict[ix] for ix in tree.output)
        self.out_elems = max(prod(self.out_shape), 1)
        steps = []
        loc = {}                                   # node id -> (0, input index) | (1, arena offset)
        for i in range(self.ninputs):
            loc[i] = (0, i)
        # lifetime-aware arena: an intermediate's slot is recycled after its (single) consumer has run
        # (first-fit over a coalescing free list); the peak footprint decides whether the arena fits in LDS
        free, top = [], 0                          # free: sorted list of (offset, size)

        def alloc(n):
            nonlocal top
            for i, (o, sz) in enumerate(free):
                if sz >= n:
                    if sz == n:
                        free.pop(i)
                    else:
                        free[i] = (o + n, sz - n)
                    return o
            o, top = top, top + n
            return o

        def release(o, n):
            nonlocal top
            free.append((o, n))
            free.sort()
            merged = []
            for off, sz in free:
                if merged and merged[-1][0] + merged[-1][1] == off:
                    merged[-1] = (merged[-1][0], merged[-1][1] + sz)
                else:
                    merged.append((off, sz))
            if merged and merged[-1][0] + merged[-1][1] == top:
                top = merged.pop()[0]
            free[:] = merged

        peak = 0
        sizes = {}
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
                off = alloc(size)                  # allocated BEFORE the operands are released: never aliases them
                d["c_off"] = off
                loc[res] = (1, off)
                sizes[res] = size
                peak = max(peak, top)
            for nid in (ka, kb):                   # every node of a tree is consumed exactly once
                if nid in sizes:
                    release(loc[nid][1], sizes.pop(nid))
            steps.append(d)
        arena = peak
        if not steps:
            raise ValueError("MicroTree: nothing to contract")
        if steps[-1]["c_off"] != -1:
            raise ValueError("MicroTree: the root is not produced by the last step")
        self.steps = steps
        self.arena_elems = arena
        #: the arena fits the LDS budget of one workgroup (QAMD_MICRO_LDS_ARENA_BYTES): dependent steps hand over
        #: through LDS instead of L2 -- the latency mode; large batches may prefer the global arena (more
        #: workgroups resident per CU)
        #: fp32 / complex64 trees carry their INTERMEDIATES in double precision (``qamd_microtree_run_ex`` with
        #: QAMD_MICRO_WIDE; ``Options.micro_wide``, captured here): the arena's elements are then twice as large
        self.wide = bool(get_options().micro_wide) and self.dtype.itemsize in (4, 8) and self.dtype in (
            np.dtype("float32"), np.dtype("complex64"))
        self.arena_itemsize = self.dtype.itemsize * (2 if self.wide else 1)
        self.lds_ok = 0 < arena * self.arena_itemsize <= 112 * 1024
        self.flops = sum((8 if self.dtype.kind == "c" else 2) * s["spec"].mults for s in steps)
        self._packed = None

    # ---- plan serialisation ---------------------------------------------------------------
    def packed(self):
        """The lowered plan: (steps bytes, etab int32 [3 * sum(total)], ktab int32 [2 * sum(K)]) -- every
        result element's (A, B, C) offsets and every k's (A, B) offsets tabulated on the host."""
        if self._packed is None:
            arr = (MicroStepStruct * len(self.steps))()
            etabs, ktabs, eoff, koff = [], [], 0, 0
            for t, s in zip(arr, self.steps):
                sp = s["spec"]
                ob_a, ob_b, ob_c = (_bundle_offsets(sp.b, c) for c in (1, 2, 3))
                om_a, om_c = _bundle_offsets(sp.m, 1), _bundle_offsets(sp.m, 3)
                on_b, on_c = _bundle_offsets(sp.n, 2), _bundle_offsets(sp.n, 3)
                oa = (ob_a[:, None, None] + om_a[None, :, None] + 0 * on_c[None, None, :]).reshape(-1)
                ob = (ob_b[:, None, None] + 0 * om_a[None, :, None] + on_b[None, None, :]).reshape(-1)
                oc = (ob_c[:, None, None] + om_c[None, :, None] + on_c[None, None, :]).reshape(-1)
                etabs.append(np.stack([oa, ob, oc], axis=1))
                ktabs.append(np.stack([_bundle_offsets(sp.k, 1), _bundle_offsets(sp.k, 2)], axis=1))
                t.a_kind, t.a_ref = s["a"]
                t.b_kind, t.b_ref = s["b"]
                t.c_off = s["c_off"]
                t.total, t.K, t.eoff, t.koff = len(oa), sp.K, eoff, koff
                eoff += len(oa)
                koff += sp.K
            etab = np.ascontiguousarray(np.concatenate(etabs), dtype=np.int32).reshape(-1)
            ktab = np.ascontiguousarray(np.concatenate(ktabs), dtype=np.int32).reshape(-1)
            self._packed = (bytes(arr), etab, ktab)
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
