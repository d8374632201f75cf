"""Exact two-sided boundary contraction of an Lx x Ly grid network: the branch decomposition of the
multi-GPU path.

The reference contracts a 2D network from one side or from two sides inwards
(``contract_boundary(sequence=...)``, quimb/tensor/tn2d/core.py:2493-2498: the final step is the
contraction of the two boundary "ladders"), and sums independent sub-contractions serially
(``cut_iter``, quimb/tensor/tensor_core.py:9291-9328; the chunk map-reduce of
quimb/tensor/circuit/exact.py:1999-2018).  Here the two half sweeps are the two BRANCHES of the
contraction tree's root, and the bonds crossing the cut are the indices to slice:

    Z = sum_{x_0 .. x_{Ly-1}}  T[x]  B[x]          T = rows 0 .. c-1 swept down,  B = rows Lx-1 .. c swept up

* no FLOP is added by the decomposition: the site-by-site sweep of a half is the same sequence of steps the
  one-sided sweep executes for those rows, and the join is one dot product of 6^Ly elements;
* slicing the first k cut bonds (x_0 .. x_{k-1}) only touches the cut row of each half: everything above it
  (``hoisted`` below: 3 of the 4 full rows of a 10-row half) does not depend on the slice.  Inside the cut
  row the slices share prefixes -- the tensor after site j depends on (x_0 .. x_j) only -- so they are
  enumerated as nested loops and every prefix is computed once: the slices of one branch add up to exactly
  the work of its unsliced cut row;
* what a rank of the multi-GPU driver evaluates is (branch, contiguous block of slices); the hoisted rows are
  the part ranks of the same branch repeat (``distributed.contract_two_sided`` reports the fraction).
"""

import math

import numpy as np

from .array import Array, asarray
from .contract import array_contract
from .executor import TreeExecutor
from .pathfind import sweep_ssa_2d
from .pairwise import prod
from .tree import ContractionTree


def _bond(a, b):
    shared = [ix for ix in a if ix in b]
    if len(shared) != 1:
        raise ValueError(f"expected exactly one bond between {a} and {b}, found {shared}")
    return shared[0]


class _Half:
    """One branch: rows ``rows`` (far -> near the cut) of the grid, cut bonds ``xs`` (one per column)."""

    def __init__(self, inputs, size_dict, Ly, rows, other_near_row, dtype, k):
        self.Ly, self.rows, self.dtype, self.k = Ly, list(rows), dtype, k
        self.size = size_dict
        site = lambda r, c: tuple(inputs[r * Ly + c])
        near = self.rows[-1]
        self.xs = tuple(_bond(site(near, c), site(other_near_row, c)) for c in range(Ly))
        self.w_pos = [near * Ly + c for c in range(Ly)]
        self.w_inds = [site(near, c) for c in range(Ly)]
        self.hoist_pos = [r * Ly + c for r in self.rows[:-1] for c in range(Ly)]
        self.hoist_ex = None
        self.us = None
        if self.hoist_pos:
            far_near = self.rows[-2]
            self.us = tuple(_bond(site(far_near, c), site(near, c)) for c in range(Ly))
            h_inputs = [tuple(inputs[p]) for p in self.hoist_pos]
            tree = ContractionTree(h_inputs, self.us, size_dict, ssa_path=sweep_ssa_2d(len(self.rows) - 1, Ly))
            self.hoist_ex = TreeExecutor(tree, dtype)
        # remainder of the cut row behind the k sliced sites: {V_{k-1}} u {W_k .. W_{Ly-1}} -> T[x_k ..]
        self._rem = None

    # -- cost model (multiplications), for the driver's report -------------------------------------------
    def hoisted_mults(self):
        return self.hoist_ex.tree.contraction_cost() if self.hoist_ex is not None else 0

    def slab_mults(self, slices):
        """Multiplications of ``slabs(..., slices)``: prefix absorptions (each distinct prefix once) + remainders."""
        k, Ly, size = self.k, self.Ly, self.size
        dims = [size[self.xs[j]] for j in range(k)]
        v_inds = set(self.us) if self.us is not None else set()
        per_prefix = []
        for j in range(k):
            w = set(self.w_inds[j]) - {self.xs[j]}
            per_prefix.append(prod(size[ix] for ix in v_inds | w))
            v_inds = (v_inds | w) - (v_inds & w)
        total, seen = 0, [set() for _ in range(k)]
        for s in slices:
            digits, rem = [], int(s)
            for d in reversed(dims):
                digits.append(rem % d)
                rem //= d
            digits.reverse()
            for j in range(k):
                key = tuple(digits[: j + 1])
                if key not in seen[j]:
                    seen[j].add(key)
                    total += per_prefix[j]
        if k < Ly:
            vin = tuple(sorted(v_inds, key=repr)) if (self.us is not None or k) else None
            ins = ([vin] if vin is not None else []) + [self.w_inds[j] for j in range(k, Ly)]
            tree = ContractionTree(ins, tuple(self.xs[k:]), size, ssa_path=sweep_ssa_2d(1, len(ins)))
            total += tree.contraction_cost() * len(list(slices))
        return total

    # -- evaluation ----------------------------------------------------------------------------------------
    def hoist(self, arrays):
        """(U, log10 exponent): everything of this half that no slice touches."""
        if self.hoist_ex is None:
            return None, 0.0
        return self.hoist_ex([arrays[p] for p in self.hoist_pos], strip_exponent=True)

    def _absorb(self, V, v_inds, j, s_j, arrays):
        """V[.. after site j-1] . W_j[x_j = s_j]  (plain: a prefix tensor of at most Ly absorptions of a
        unit-scale U stays far inside the fp32 range; the remainder strips again)."""
        w = asarray(arrays[self.w_pos[j]])
        winds = self.w_inds[j]
        ax = winds.index(self.xs[j])
        w = w[(slice(None),) * ax + (int(s_j),)]
        winds = winds[:ax] + winds[ax + 1:]
        if V is None:
            return w, winds
        out = tuple(ix for ix in winds if ix not in v_inds) + tuple(ix for ix in v_inds if ix not in winds)
        return array_contract([V, w], [v_inds, winds], out), out

    def _remainder(self, v_inds):
        if self._rem is None or self._rem[0] != v_inds:
            ins = ([tuple(v_inds)] if v_inds is not None else []) + [self.w_inds[j] for j in range(self.k, self.Ly)]
            out = tuple(self.xs[self.k:])
            tree = ContractionTree(ins, out, self.size, ssa_path=sweep_ssa_2d(1, len(ins)))
            self._rem = (v_inds, TreeExecutor(tree, self.dtype))
        return self._rem[1]

    def slabs(self, arrays, U, slices):
        """Yield (slice number, T[x_k ..] , log10 exponent) for the slice numbers in ``slices`` (sorted), sharing
        every common prefix: slice s <-> (x_0 .. x_{k-1}) = digits of s, x_0 the slowest."""
        k, Ly = self.k, self.Ly
        dims = [self.size[self.xs[j]] for j in range(k)]
        stack = []            # stack[j] = (digit s_j, V_j, inds)
        base = (U, self.us)
        for s in slices:
            digits, rem = [], int(s)
            for d in reversed(dims):
                digits.append(rem % d)
                rem //= d
            digits.reverse()
            keep = 0
            while keep < len(stack) and keep < k and stack[keep][0] == digits[keep]:
                keep += 1
            del stack[keep:]
            for j in range(keep, k):
                V, vin = stack[j - 1][1:] if j else base
                V2, vin2 = self._absorb(V, vin, j, digits[j], arrays)
                stack.append((digits[j], V2, vin2))
            V, vin = stack[k - 1][1:] if k else base
            if k == Ly:
                T, e = V, 0.0
            else:
                ex = self._remainder(vin)
                ins = ([V] if V is not None else []) + [asarray(arrays[self.w_pos[j]]) for j in range(k, Ly)]
                T, e = ex(ins, strip_exponent=True)
            yield s, T, e


class TwoSidedContraction:
    """Plan of ``Z = sum_x T[x] B[x]`` for a row-major Lx x Ly grid of tensors (``inputs[r * Ly + c]`` = index
    tuple of site (r, c); closed network, e.g. an amplitude of a PEPS).  ``sliced_cols`` = k: the first k cut
    bonds are sliced (``nslices`` = product of their sizes)."""

    def __init__(self, inputs, size_dict, Lx, Ly, dtype="float32", cut=None, sliced_cols=0):
        if Lx < 2:
            raise ValueError("a two-sided sweep needs at least two rows")
        inputs = [tuple(t) for t in inputs]
        if len(inputs) != Lx * Ly:
            raise ValueError(f"expected {Lx * Ly} site tensors, got {len(inputs)}")
        self.Lx, self.Ly, self.dtype = Lx, Ly, np.dtype(dtype)
        self.cut = Lx // 2 if cut is None else int(cut)
        if not 1 <= self.cut <= Lx - 1:
            raise ValueError("cut must leave at least one row on each side")
        self.k = int(sliced_cols)
        if not 0 <= self.k <= Ly:
            raise ValueError("sliced_cols out of range")
        self.size = dict(size_dict)
        c = self.cut
        self.top = _Half(inputs, self.size, Ly, range(0, c), c, self.dtype, self.k)
        self.bottom = _Half(inputs, self.size, Ly, range(Lx - 1, c - 1, -1), c - 1, self.dtype, self.k)
        assert self.top.xs == self.bottom.xs
        self.nslices = prod(self.size[ix] for ix in self.top.xs[: self.k])
        full = ContractionTree(inputs, (), self.size, ssa_path=sweep_ssa_2d(Lx, Ly))
        self.one_sided_mults = full.contraction_cost()

    def cost_report(self, layout):
        """What a rank layout (``distributed.two_sided_layout``) executes, in multiplications: the useful count
        (= the one-sided sweep of the whole network), the count summed over ranks, and the busiest rank."""
        per_rank = []
        for branch, slices in layout:
            halves = [self.top, self.bottom] if branch == "both" else [self.top if branch == "top" else self.bottom]
            per_rank.append(sum(h.hoisted_mults() + h.slab_mults(slices) for h in halves))
        join = prod(self.size[ix] for ix in self.top.xs)      # the dot products of all slices together
        hoisted = sum((self.top if b == "top" else self.bottom).hoisted_mults() for b, _ in layout if b != "both")
        if any(b == "both" for b, _ in layout):
            hoisted = self.top.hoisted_mults() + self.bottom.hoisted_mults()
        return {"useful_mults": self.one_sided_mults, "executed_mults": sum(per_rank) + join,
                "busiest_rank_mults": max(per_rank), "hoisted_mults_all_ranks": hoisted,
                "inflation": (sum(per_rank) + join) / self.one_sided_mults,
                "ideal_speedup_vs_one_rank": self.one_sided_mults / max(per_rank)}

    def halves(self):
        return {"top": self.top, "bottom": self.bottom}

    def join(self, T, eT, B, eB):
        """<T, B> over the unsliced cut bonds -> (mantissa, log10 exponent)."""
        inds = tuple(self.top.xs[self.k:])
        if not inds:
            m = float(np.asarray(asarray(T).to_numpy()).reshape(-1)[0]) * float(np.asarray(asarray(B).to_numpy()).reshape(-1)[0])
            return m, eT + eB
        out = array_contract([T, B], [inds, inds], (), strip_exponent=True)
        m, e = out
        return float(np.asarray(m.to_numpy() if isinstance(m, Array) else m).reshape(-1)[0]), float(e) + eT + eB

    def __call__(self, arrays, strip_exponent=False):
        """Everything on this device: both halves, every slice (the one-rank form of
        ``distributed.contract_two_sided``)."""
        arrays = [asarray(a) for a in arrays]
        U, eU = self.top.hoist(arrays)
        Ub, eUb = self.bottom.hoist(arrays)
        order = range(self.nslices)
        acc = []
        bot = self.bottom.slabs(arrays, Ub, order)
        for (s, T, eT), (s2, B, eB) in zip(self.top.slabs(arrays, U, order), bot):
            assert s == s2
            acc.append(self.join(T, eT + eU, B, eB + eUb))
        return combine_pairs(acc, strip_exponent)


def combine_pairs(pairs, strip_exponent=False):
    """Sum of ``m * 10**e`` terms given as (mantissa, exponent) pairs, on a common exponent."""
    pairs = [(m, e) for m, e in pairs if m != 0.0 and math.isfinite(e)]
    if not pairs:
        return (0.0, 0.0) if strip_exponent else 0.0
    e_max = max(e for _, e in pairs)
    m = sum(m * 10.0 ** (e - e_max) for m, e in pairs)
    if not strip_exponent:
        return m * 10.0**e_max
    if m == 0.0:
        return 0.0, 0.0
    # the reference's convention for a scalar (tensor_core.py:330-340): mantissa = x / |x|, exponent = log10 |x|
    return math.copysign(1.0, m), e_max + math.log10(abs(m))
