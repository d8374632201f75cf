"""Contraction trees: the description the whole-tree executor consumes.

quimb hands the tree to cotengra (``tn.contract(get="tree")``,
quimb/tensor/tensor_core.py:194-197; ``array_contract_tree``,
quimb/tensor/contraction.py:302-306).  cotengra is not part of this build, so
this module carries the minimal equivalent: (inputs, output, size_dict, path,
sliced_inds) plus the cost model the reference pins -- ``contraction_cost`` =
number of scalar multiplications summed over pairwise steps, ``contraction_width``
= log2 of the largest intermediate (tests/test_tensor/test_tensor_core.py:1199-1205).
A duck-typed cotengra ``ContractionTree`` (``get_path()``, ``inputs``, ``output``,
``size_dict``, ``sliced_inds``) is accepted wherever a tree is.
"""

import math
from collections import Counter

from .pairwise import prod


def linear_to_ssa(path, n):
    ids = list(range(n))
    ssa, nxt = [], n
    for con in path:
        con = tuple(sorted(con, reverse=True))
        picked = [ids.pop(p) for p in con]
        ssa.append(tuple(reversed(picked)))
        ids.append(nxt)
        nxt += 1
    return ssa


def ssa_to_linear(ssa_path, n):
    ids = list(range(n))
    path, nxt = [], n
    for con in ssa_path:
        pos = sorted(ids.index(s) for s in con)
        path.append(tuple(pos))
        for p in reversed(pos):
            ids.pop(p)
        ids.append(nxt)
        nxt += 1
    return path


class ContractionTree:
    """A pairwise contraction order over hashable-labelled tensors."""

    def __init__(self, inputs, output, size_dict, path=None, ssa_path=None, sliced_inds=()):
        self.inputs = tuple(tuple(t) for t in inputs)
        self.output = tuple(output)
        self.size_dict = dict(size_dict)
        n = len(self.inputs)
        if ssa_path is None:
            if path is None:
                raise ValueError("need a path or an ssa_path")
            ssa_path = linear_to_ssa(path, n)
        self.ssa_path = [tuple(c) for c in ssa_path]
        self.sliced_inds = tuple(sliced_inds)
        for ix in self.sliced_inds:
            if ix in self.output:
                raise ValueError(f"cannot slice output index {ix!r}")
        self._build()

    # ---- construction ------------------------------------------------------
    @classmethod
    def from_any(cls, obj, inputs=None, output=None, size_dict=None):
        """Accept our own tree, a cotengra-like tree, or a raw linear path."""
        if isinstance(obj, cls):
            return obj
        if hasattr(obj, "get_path") and hasattr(obj, "size_dict"):
            return cls(
                getattr(obj, "inputs", inputs),
                getattr(obj, "output", output),
                obj.size_dict,
                path=obj.get_path(),
                sliced_inds=tuple(getattr(obj, "sliced_inds", ()) or ()),
            )
        return cls(inputs, output, size_dict, path=obj)

    def _build(self):
        sliced = set(self.sliced_inds)
        terms = {i: tuple(ix for ix in t if ix not in sliced) for i, t in enumerate(self.inputs)}
        counts = Counter()
        for t in terms.values():
            counts.update(set(t))
        out_set = set(self.output)
        self.steps = []  # (operand ssa ids, result ssa id, operand inds, result inds (set-order), mults)
        nxt = len(self.inputs)
        size = self.size_dict
        for con in self.ssa_path:
            ops = [terms.pop(s) for s in con]
            seen = Counter()
            for t in ops:
                seen.update(set(t))
            keep = []
            for t in ops:
                for ix in t:
                    if ix in keep:
                        continue
                    if ix in out_set or counts[ix] > seen[ix]:
                        keep.append(ix)
            for ix, c in seen.items():
                counts[ix] -= c
            for ix in keep:
                counts[ix] += 1
            all_inds = set()
            for t in ops:
                all_inds.update(t)
            mults = prod(size[ix] for ix in all_inds)
            self.steps.append((tuple(con), nxt, tuple(ops), tuple(keep), mults))
            terms[nxt] = tuple(keep)
            nxt += 1
        self.remaining = dict(terms)  # normally a single entry

    # ---- introspection (names follow cotengra / quimb usage) -------------------
    @property
    def nslices(self):
        return prod(self.size_dict[ix] for ix in self.sliced_inds)

    @property
    def N(self):
        return len(self.inputs)

    def get_path(self):
        return ssa_to_linear(self.ssa_path, len(self.inputs))

    def get_ssa_path(self):
        return list(self.ssa_path)

    def contraction_cost(self, per_slice=False):
        """Scalar multiplications of the whole contraction (all slices)."""
        c = sum(s[4] for s in self.steps)
        return c if per_slice else c * self.nslices

    def total_flops(self, dtype="float32"):
        """2 flops per real multiply-add, 8 per complex one."""
        import numpy as np

        return self.contraction_cost() * (8 if np.dtype(dtype).kind == "c" else 2)

    def max_size(self):
        m = max((prod(self.size_dict[ix] for ix in s[3]) for s in self.steps), default=1)
        sliced = set(self.sliced_inds)
        for t in self.inputs:
            m = max(m, prod(self.size_dict[ix] for ix in t if ix not in sliced))
        return m

    def contraction_width(self):
        return math.log2(self.max_size())

    def with_slices(self, sliced_inds):
        return ContractionTree(self.inputs, self.output, self.size_dict, ssa_path=self.ssa_path, sliced_inds=sliced_inds)

    def regrouped(self, max_small=1 << 16, gain=0.75):
        """Re-associate ``(A . W1) . W2`` into ``A . (W1 . W2)`` wherever that lowers the cost.

        Site-by-site absorption orders (``sweep_path_2d``, or any path found on the unsliced network) stay
        optimal while every site tensor has all its legs; once a bond next to two neighbouring sites is
        sliced away, contracting the two small tensors first makes ONE pass over the big operand instead
        of two -- fewer multiplications and a third of the HBM traffic.  The same rewrite is taken where it costs no more
        multiplications (within 2 %) and moves a quarter less data -- two THIN absorptions in a row (product of the two small
        tensors <= 1024 elements with indices of MIXED sizes: MPO tensors, gates; site tensors of a uniform-bond lattice are
        left to the fused-pair kernels, which make the same single pass without the extra product).  Only consecutive steps are
        rewritten, and only where the two steps' multiplications drop below ``gain`` x their old count (the
        caller's tree is otherwise executed as given); the small product is capped at ``max_small`` elements, and the result is the same
        tensor (contraction is associative); ids, inputs and output are unchanged."""
        n = len(self.inputs)
        best = self
        size = self.size_dict
        si = 0
        while si + 1 < len(best.ssa_path):
            ssa = best.ssa_path
            c1, c2 = ssa[si], ssa[si + 1]
            r1 = n + si
            if len(c1) == 2 and len(c2) == 2 and r1 in c2 and c2[0] != c2[1]:
                w2 = c2[0] if c2[1] == r1 else c2[1]
                if w2 < r1:
                    # local evaluation (no tree rebuild): only the two steps' index sets matter.  An index of
                    # W1 u W2 survives into W12 iff A or the pair's final result still carries it.
                    _, _, ops1, keep1, m1 = best.steps[si]
                    _, _, ops2, keep2, m2 = best.steps[si + 1]
                    w2_inds = set(ops2[0] if c2[1] == r1 else ops2[1])
                    final = set(keep2)
                    for (a, w1), (a_inds, w1_inds) in (((c1[0], c1[1]), (ops1[0], ops1[1])),
                                                      ((c1[1], c1[0]), (ops1[1], ops1[0]))):
                        a_set, pair = set(a_inds), set(w1_inds) | w2_inds
                        w12 = pair & (a_set | final)
                        small = prod(size[ix] for ix in w12)
                        cost = prod(size[ix] for ix in pair) + prod(size[ix] for ix in a_set | w12)
                        # ... or where it costs no more multiplications and HALVES the traffic: two thin absorptions in a
                        # row (MPO tensors on a DMRG effective Hamiltonian: K = N = 10 each) read and write the big
                        # tensor twice; one application of the 20 x 20 product reads and writes it once
                        elems_before = prod(size[ix] for ix in a_set) + 2 * prod(size[ix] for ix in keep1) + \
                            prod(size[ix] for ix in keep2)
                        elems_after = prod(size[ix] for ix in a_set) + prod(size[ix] for ix in keep2) + 2 * small
                        thin = (small <= 1024 and cost <= 1.02 * (m1 + m2) and elems_after < 0.75 * elems_before
                                and len({size[ix] for ix in pair}) > 1)    # uniform bonds: the fused-pair kernels' case
                        if small <= max_small and (cost < gain * (m1 + m2) or thin):
                            trial = list(ssa)
                            trial[si], trial[si + 1] = (w1, w2), (a, r1)
                            best = ContractionTree(self.inputs, self.output, self.size_dict, ssa_path=trial,
                                                   sliced_inds=self.sliced_inds)
                            break
            si += 1
        return best

    def death_times(self):
        """For every ssa id: {ind: step at which that index is contracted away}
        (indices surviving to the output never die)."""
        where = {}
        for si, (con, res, ops, keep, _) in enumerate(self.steps):
            kset = set(keep)
            for s, t in zip(con, ops):
                d = where.setdefault(s, {})
                for ix in t:
                    if ix not in kset:
                        d[ix] = si
        # propagate: an index of an intermediate dies when a descendant drops it
        death = {}
        consumer = {}
        for si, (con, res, ops, keep, _) in enumerate(self.steps):
            for s in con:
                consumer[s] = si
        for si in range(len(self.steps) - 1, -1, -1):
            con, res, ops, keep, _ = self.steps[si]
            d = {}
            nxt = consumer.get(res)
            if nxt is not None:
                ncon, nres, nops, nkeep, _ = self.steps[nxt]
                for ix in keep:
                    if ix not in nkeep:
                        d[ix] = nxt
                    elif ix in death.get(nres, {}):
                        d[ix] = death[nres][ix]
            death[res] = d
        return death

    def __repr__(self):
        return (
            f"<ContractionTree N={self.N} cost={self.contraction_cost():.3e} "
            f"width={self.contraction_width():.2f} nslices={self.nslices}>"
        )
