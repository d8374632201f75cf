"""A minimal ``TensorNetwork`` carrying just the contraction dispatch of the
reference's class (quimb/tensor/tensor_core.py): the methods that sit ON the hot
path -- SURVEY.md section 8a rows a1, a2, a11 -- so that the parity tests can be
written the way quimb's are.  Everything else quimb's TensorNetwork does
(gauging, splitting, simplification, drawing ...) is out of scope.

* ``contract(tags=all, ...)``          tensor_core.py:9581-9716 (all tags and not inplace ->
                                        ``tensor_contract(*tensors, exponent=self.exponent)``)
* ``contract_tags(tags, which=)``      :9426-9577 (partition, contract, re-insert)
* ``contract_cumulative(tags_seq)``    :9720-9800 (``>>``)
* ``isel`` / ``cut_iter``              :9215-9244 / :9291-9328
* ``apply_to_arrays`` / ``to_device``  :5304
* ``&``, ``^``, ``>>``                 :9958 and friends
"""

import itertools

from .array import asarray
from .contract import Tensor, tensor_contract


def _tags_of(tags):
    if tags is None:
        return ()
    if isinstance(tags, str):
        return (tags,)
    return tuple(tags)


class TensorNetwork:
    def __init__(self, tensors=(), exponent=0.0):
        self.tensors = []
        self.exponent = exponent
        for t in tensors:
            self.add(t)

    # ---- construction --------------------------------------------------------
    def add(self, t):
        if isinstance(t, TensorNetwork):
            for x in t.tensors:
                self.add(x)
            self.exponent += t.exponent
        elif isinstance(t, Tensor):
            self.tensors.append(t)
        else:
            raise TypeError("can only add Tensor or TensorNetwork")

    def copy(self):
        return TensorNetwork([Tensor(t.data, t.inds, t.tags) for t in self.tensors], self.exponent)

    def __and__(self, other):
        tn = self.copy()
        tn.add(other)
        return tn

    def __iter__(self):
        return iter(self.tensors)

    def __len__(self):
        return len(self.tensors)

    @property
    def tags(self):
        return tuple(dict.fromkeys(tg for t in self.tensors for tg in t.tags))

    @property
    def tag_map(self):
        out = {}
        for i, t in enumerate(self.tensors):
            for tg in t.tags:
                out.setdefault(tg, []).append(i)
        return out

    def ind_sizes(self):
        return {ix: d for t in self.tensors for ix, d in zip(t.inds, t.shape)}

    def outer_inds(self):
        cnt = {}
        for t in self.tensors:
            for ix in t.inds:
                cnt[ix] = cnt.get(ix, 0) + 1
        return tuple(ix for ix, c in cnt.items() if c == 1)

    def inner_inds(self):
        cnt = {}
        for t in self.tensors:
            for ix in t.inds:
                cnt[ix] = cnt.get(ix, 0) + 1
        return tuple(ix for ix, c in cnt.items() if c >= 2)

    # ---- array handling --------------------------------------------------------
    def apply_to_arrays(self, fn):
        for t in self.tensors:
            t.data = fn(t.data)
        return self

    def to_device(self, dtype=None):
        """Move every tensor's data into HBM (``tn.apply_to_arrays(quimb_amd.asarray)``)."""
        return self.apply_to_arrays(lambda x: asarray(x, dtype=dtype))

    # ---- selection ----------------------------------------------------------------
    def _select(self, tags, which="all"):
        tags = _tags_of(tags)
        picked, rest = [], []
        for t in self.tensors:
            has = [tg in t.tags for tg in tags]
            hit = all(has) if which == "all" else any(has)
            (picked if hit else rest).append(t)
        return picked, rest

    def isel(self, selectors):
        """Fix index values (amplitudes, slices): every tensor holding a selected index
        is sliced at that value (Tensor.isel, tensor_core.py:2260-2348)."""
        new = []
        for t in self.tensors:
            if any(ix in selectors for ix in t.inds):
                key = tuple(int(selectors[ix]) if ix in selectors else slice(None) for ix in t.inds)
                data = t.data[key]
                inds = tuple(ix for ix in t.inds if ix not in selectors)
                new.append(Tensor(data, inds, t.tags))
            else:
                new.append(t)
        return TensorNetwork(new, self.exponent)

    def cut_iter(self, *inds):
        """Generator of networks with ``inds`` fixed to every combination of values; their
        contractions sum to the original's (tensor_core.py:9291-9328)."""
        sizes = self.ind_sizes()
        for vals in itertools.product(*[range(sizes[ix]) for ix in inds]):
            yield self.isel(dict(zip(inds, vals)))

    # ---- contraction ------------------------------------------------------------------
    def contract_tags(self, tags, which="any", inplace=False, **opts):
        picked, rest = self._select(tags, which)
        if not picked:
            raise ValueError("No tags were found - nothing to contract.")
        # indices shared with the untouched tensors must survive
        outside = {ix for t in rest for ix in t.inds}
        inner = {}
        for t in picked:
            for ix in t.inds:
                inner[ix] = inner.get(ix, 0) + 1
        out_inds = tuple(ix for ix, c in inner.items() if c == 1 or ix in outside)
        opts.setdefault("output_inds", out_inds)
        if not rest:
            res = tensor_contract(*picked, exponent=self.exponent if self.exponent else None, **opts)
            if inplace and isinstance(res, Tensor):
                self.tensors = [res]
                return self
            return res
        res = tensor_contract(*picked, preserve_tensor=True, **opts)
        tn = self if inplace else TensorNetwork((), self.exponent)
        tn.tensors = rest + [res]
        return tn

    def contract(self, tags=all, output_inds=None, optimize=None, backend=None, inplace=False,
                 strip_exponent=False, **opts):
        if tags is all or tags is Ellipsis:
            kw = dict(output_inds=output_inds, optimize=optimize, backend=backend, strip_exponent=strip_exponent)
            kw.update(opts)
            if self.exponent:
                kw["exponent"] = self.exponent
            res = tensor_contract(*self.tensors, **kw)
            if inplace:
                if isinstance(res, Tensor):
                    self.tensors = [res]
                return self
            return res
        return self.contract_tags(tags, inplace=inplace, optimize=optimize, backend=backend, **opts)

    def contract_cumulative(self, tags_seq, inplace=False, **opts):
        tn = self if inplace else self.copy()
        acc = ()
        for tags in tags_seq:
            acc = acc + _tags_of(tags)
            res = tn.contract_tags(acc, which="any", inplace=True, **opts)
            if not isinstance(res, TensorNetwork):
                return res
        if len(tn.tensors) == 1 and not inplace:
            return tn.tensors[0]
        return tn

    def __xor__(self, tags):
        return self.contract(tags)

    def __ixor__(self, tags):
        res = self.contract(tags, inplace=True)
        return res if isinstance(res, TensorNetwork) else self

    def __rshift__(self, tags_seq):
        return self.contract_cumulative(tags_seq)

    def __irshift__(self, tags_seq):
        res = self.contract_cumulative(tags_seq, inplace=True)
        return res if isinstance(res, TensorNetwork) else self

    def __repr__(self):
        return f"TensorNetwork(tensors={len(self.tensors)}, exponent={self.exponent})"
