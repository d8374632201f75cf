"""Range slicing: sliced indices without the one-value-per-slice restriction, for ANY contraction tree.

cotengra slices an index by fixing it to each of its values in turn (``tree.sliced_inds``; quimb sums the slices
serially, quimb/tensor/circuit/exact.py:1999-2018, and ``TensorNetwork.cut_iter`` does the same by hand,
quimb/tensor/tensor_core.py:9291-9328).  A contraction is linear in every index RANGE as well: split index ``ix`` of size
``d`` into ``p`` contiguous ranges and the network's value is the sum of the ``p`` networks restricted to one range
each -- the same tree, sizes ``d / p`` along ``ix``.  Where single-value slices multiply the work of every step that
does not carry the sliced index (216 slices of the 10 x 10 D = 6 boundary sweep: 140 x the FLOPs), a range that divides
a FREE index of the dominant steps divides exactly those steps and nothing else: the GEMM-shaped joins of the quadrant
tree split into blocks with 2.6 ... 12.7 % inflation at 2 ... 8 parts (``quimb_amd/quadrants.py`` is this machinery with
the indices chosen by the lattice's geometry; here they are chosen by cost, for any tree).

    parts = find_range_slices(tree, 8)                 # {index: number of ranges}, their product = 8
    rs = RangeSliced(tree, parts)
    value = sum(execute(rs.tree(s), rs.slice_arrays(arrays, s)) for s in range(rs.nslices))

``contract_range_sliced`` runs the slices of this rank and joins the ranks with ONE all-gather.
"""

import numpy as np

from .tree import ContractionTree


def _prime_factors(n):
    f, p = [], 2
    while n > 1:
        while n % p == 0:
            f.append(p)
            n //= p
        p += 1
    return f


def _ranges(d, parts):
    edges = [(d * i) // parts for i in range(parts + 1)]
    return [(edges[i], edges[i + 1]) for i in range(parts)]


def _cost_with_sizes(tree, sizes):
    return ContractionTree(tree.inputs, tree.output, sizes, ssa_path=tree.ssa_path).contraction_cost()


def find_range_slices(tree, nslices, candidates=None):
    """Choose which indices to split into how many ranges so that ``nslices`` independent sub-contractions cover the
    network with the least total work: the prime factors of ``nslices`` (largest first) go, one at a time, to the
    index whose further division costs the fewest multiplications summed over all slices -- ties to the larger index,
    then to equal ranges.  Output indices cannot be split (their ranges would tile the OUTPUT, not sum); an index of an
    already sliced (single-value) tree neither.  Returns ``{index: parts}``."""
    nslices = int(nslices)
    if nslices < 1:
        raise ValueError("need at least one slice")
    if tree.sliced_inds:
        raise ValueError("range slicing starts from an unsliced tree")
    out = set(tree.output)
    if candidates is None:
        candidates = [ix for ix in dict.fromkeys(ix for t in tree.inputs for ix in t) if ix not in out]
    sizes = dict(tree.size_dict)
    parts = {}
    total = 1
    for p in sorted(_prime_factors(nslices), reverse=True):
        best = None
        for ix in candidates:
            k = parts.get(ix, 1) * p
            if k > tree.size_dict[ix]:
                continue
            trial = dict(sizes)
            # per-slice cost of the LARGEST range (the slowest slice), times the number of slices
            trial[ix] = -(-tree.size_dict[ix] // k)
            cost = _cost_with_sizes(tree, trial) * total * p
            key = (cost, 0 if tree.size_dict[ix] % k == 0 else 1, -tree.size_dict[ix], repr(ix))
            if best is None or key < best[0]:
                best = (key, ix, k, trial)
        if best is None:
            raise ValueError(f"{nslices} range slices do not fit the indices of this network")
        _, ix, k, sizes = best
        parts[ix] = k
        total *= p
    return parts


class RangeSliced:
    """A tree plus ``{index: parts}``: slice ``s`` (mixed radix over the split indices, last one fastest) keeps one range
    of every split index."""

    def __init__(self, tree, parts):
        if tree.sliced_inds:
            raise ValueError("range slicing starts from an unsliced tree")
        for ix, k in parts.items():
            if ix in tree.output:
                raise ValueError(f"cannot range-slice output index {ix!r}")
            if not 1 <= int(k) <= tree.size_dict[ix]:
                raise ValueError(f"index {ix!r} of size {tree.size_dict[ix]} cannot be split into {k} ranges")
        self.base = tree
        self.inds = [ix for ix, k in parts.items() if int(k) > 1]
        self.parts = [int(parts[ix]) for ix in self.inds]
        self.nslices = int(np.prod(self.parts)) if self.parts else 1

    def ranges(self, s):
        out, r = {}, int(s)
        if not 0 <= r < self.nslices:
            raise ValueError(f"slice number must lie in [0, {self.nslices})")
        for ix, k in zip(reversed(self.inds), reversed(self.parts)):
            out[ix] = _ranges(self.base.size_dict[ix], k)[r % k]
            r //= k
        return out

    def size_dict(self, s):
        sd = dict(self.base.size_dict)
        for ix, (lo, hi) in self.ranges(s).items():
            sd[ix] = hi - lo
        return sd

    def tree(self, s):
        return ContractionTree(self.base.inputs, self.base.output, self.size_dict(s), ssa_path=self.base.ssa_path)

    def slice_arrays(self, arrays, s):
        rng = self.ranges(s)
        out = []
        for x, t in zip(arrays, self.base.inputs):
            if any(ix in rng for ix in t):
                x = x[tuple(slice(*rng[ix]) if ix in rng else slice(None) for ix in t)]
                if isinstance(x, np.ndarray):
                    x = np.ascontiguousarray(x)
            out.append(x)
        return out

    def cost_report(self):
        one = self.base.contraction_cost()
        per = [self.tree(s).contraction_cost() for s in range(self.nslices)]
        return {"nslices": self.nslices, "indices": len(self.inds), "parts": list(self.parts), "unsliced_mults": one,
                "per_slice_mults": per, "inflation": sum(per) / one, "largest_slice_fraction": max(per) / one}


class RangeSlicedExecutor:
    """Executors of the slices, built once and shared between slices of equal sizes (equal ranges: ONE plan)."""

    def __init__(self, rs, dtype="float32"):
        self.rs, self.dtype = rs, dtype
        self._by_sizes = {}

    def executor(self, s):
        from .executor import TreeExecutor

        key = tuple(sorted((repr(ix), hi - lo) for ix, (lo, hi) in self.rs.ranges(s).items()))
        ex = self._by_sizes.get(key)
        if ex is None:
            ex = self._by_sizes[key] = TreeExecutor(self.rs.tree(s), self.dtype)
        return ex

    def __call__(self, arrays, slices=None, strip_exponent=False):
        """Sum of the given slices (default all).  Scalar outputs with ``strip_exponent`` come back as
        ``(mantissa, exponent)`` floats; otherwise a host array."""
        from .quadrants import combine_pairs

        todo = range(self.rs.nslices) if slices is None else list(slices)
        if strip_exponent:
            if self.rs.base.output:
                raise ValueError("strip_exponent over range slices is implemented for scalar outputs")
            pairs = []
            for s in todo:
                m, e = self.executor(s)(self.rs.slice_arrays(arrays, s), strip_exponent=True)
                pairs.append((m.to_numpy().item(), float(e)))
            return combine_pairs(pairs, strip_exponent=True)
        acc = None
        for s in todo:
            out = self.executor(s)(self.rs.slice_arrays(arrays, s)).to_numpy()
            acc = out.copy() if acc is None else acc + out
        if acc is None:
            acc = np.zeros([self.rs.base.size_dict[ix] for ix in self.rs.base.output], dtype=np.dtype(self.dtype))
        return acc


def contract_range_sliced(rse, arrays, strip_exponent=False, group=None):
    """Rank ``r`` of ``W`` evaluates slices ``{s : s % W == r}`` of ``rse`` (a ``RangeSlicedExecutor``) and ONE
    all-gather joins the ranks: (mantissa, exponent) pairs for scalar outputs under ``strip_exponent``, the partial
    sums themselves otherwise.  Works without a process group (world 1)."""
    import torch
    import torch.distributed as dist

    from .quadrants import combine_pairs

    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    mine = list(range(rank, rse.rs.nslices, world))
    # RCCL ("nccl") moves device memory, gloo host memory
    cdev = torch.device("cuda", torch.cuda.current_device()) if (world > 1 and dist.get_backend(group) == "nccl") else torch.device("cpu")
    if strip_exponent:
        m, e = rse(arrays, slices=mine, strip_exponent=True) if mine else (0.0, float("-inf"))
        if world == 1:
            return m, e
        # (re, im, exponent): a complex mantissa travels as two doubles
        mc = complex(m)
        t = torch.tensor([mc.real, mc.imag, e if np.isfinite(e) else -1e300], dtype=torch.float64, device=cdev)
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t, group=group)
        trip = torch.stack(gathered).cpu().numpy()
        cplx = np.dtype(rse.dtype).kind == "c"
        return combine_pairs([((complex(a, b) if cplx else float(a)), float(c)) for a, b, c in trip if c > -1e299],
                             strip_exponent=True)
    part = np.ascontiguousarray(rse(arrays, slices=mine))
    if world == 1:
        return part
    t = torch.from_numpy(part.reshape(-1).copy())
    if t.is_complex():
        t = torch.view_as_real(t).reshape(-1)
    t = t.to(cdev)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    tot = torch.stack(gathered).sum(0).cpu().numpy()
    if np.dtype(part.dtype).kind == "c":
        tot = tot.reshape(-1, 2)
        tot = tot[:, 0] + 1j * tot[:, 1]
    return tot.astype(part.dtype).reshape(part.shape)
