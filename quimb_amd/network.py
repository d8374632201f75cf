"""A minimal ``TensorNetwork`` carrying just the contraction dispatch of the
reference's class (quimb/tensor/tensor_core.py): the methods that sit ON the hot
path -- SURVEY.md section 8a rows a1, a2, a11 -- so that the parity tests can be
written the way quimb's are.  Everything else quimb's TensorNetwork does
(gauging, splitting, simplification, drawing ...) is out of scope.

* ``contract(tags=all, ...)``          tensor_core.py:9581-9716 (all tags and not inplace ->
                                        ``tensor_contract(*tensors, exponent=self.exponent)``)
* ``contract_tags(tags, which=)``      :9426-9577 (partition, contract, re-insert; ``strip_exponent`` /
                                        ``equalize_norms`` bookkeeping in ``tn.exponent``)
* ``equalize_norms`` / ``strip_exponent``  :10801-10880
* ``contract_cumulative(tags_seq)``    :9720-9800 (``>>``)
* ``contract_structured(site_tags)``   tn1d/core.py:502-557 (blocks of ``structure_bsz`` sites, cumulative)
* ``isel`` / ``cut_iter``              :9215-9244 / :9291-9328
* ``apply_to_arrays`` / ``to_device``  :5304
* ``&``, ``^``, ``>>``                 :9958 and friends
"""

import itertools
import math

from . import ops
from .array import asarray
from .contract import Tensor, tensor_contract


def _scalar(t):
    """0-d Tensor -> python scalar, real if the imaginary part vanishes (``maybe_realify_scalar``)."""
    from .contract import _realify_scalar

    return _realify_scalar(asarray(t.data).to_numpy().item())


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
    def contract_tags(self, tags, which="any", output_inds=None, optimize=None, backend=None,
                      strip_exponent=False, equalize_norms="auto", preserve_tensor=False, inplace=False, **opts):
        """Contract the tensors matching ``tags`` (``all`` / ``...``: every tensor) and put the result back
        (tensor_core.py:9426-9577).  ``equalize_norms`` strips the scale inside the contraction and accumulates
        it in ``tn.exponent`` ("auto": follow ``strip_exponent``); ``strip_exponent`` returns
        ``(result, exponent)`` when everything was contracted, ``result * 10**exponent`` being the value."""
        if tags is all or tags is Ellipsis:
            picked, rest = list(self.tensors), []
        else:
            picked, rest = self._select(tags, which)
        if not picked:
            raise ValueError("No tags were found - nothing to contract.")
        if output_inds is None:
            # indices shared with the untouched tensors must survive
            outside = {ix for t in rest for ix in t.inds}
            inner = {}
            for t in picked:
                for ix in t.inds:
                    inner[ix] = inner.get(ix, 0) + 1
            output_inds = tuple(ix for ix, c in inner.items() if c == 1 or ix in outside)
        if equalize_norms == "auto":
            equalize_norms = strip_exponent
        keep_tensor = preserve_tensor or inplace or bool(rest)
        t = tensor_contract(*picked, output_inds=output_inds, optimize=optimize, backend=backend,
                            strip_exponent=bool(equalize_norms), preserve_tensor=keep_tensor, **opts)
        if equalize_norms:
            t, exponent = t                       # scale taken out step by step inside the contraction
        elif strip_exponent:
            if isinstance(t, Tensor):             # take it out of the finished result
                nrm = ops.norm_fro(t.data)
                t = Tensor(t.data / nrm, t.inds, t.tags)
            else:
                nrm = abs(t)
                t = t / nrm
            exponent = math.log10(nrm)
        else:
            exponent = None
        if not rest and not inplace:
            total = self.exponent + (exponent or 0.0)
            if strip_exponent:
                return t, total
            if total:
                t = Tensor(t.data * 10.0**total, t.inds, t.tags) if isinstance(t, Tensor) else t * 10.0**total
            return t
        tn = self if inplace else TensorNetwork((), self.exponent)
        tn.tensors = rest + [t]
        if exponent is not None:
            tn.exponent = tn.exponent + exponent
        return tn

    def contract(self, tags=all, output_inds=None, optimize=None, backend=None, inplace=False,
                 strip_exponent=False, equalize_norms="auto", **opts):
        return self.contract_tags(tags, output_inds=output_inds, optimize=optimize, backend=backend,
                                  strip_exponent=strip_exponent, equalize_norms=equalize_norms, inplace=inplace,
                                  **opts)

    # ---- norms / exponent bookkeeping (tensor_core.py:10801-10841) ------------------------------------
    # ---- local contractions (tensor_core.py:6206-6289) ------------------------------------------------------
    def _contracted_inds(self, picked, rest, output_inds=None):
        """Indices the contraction of ``picked`` must keep: those still used by ``rest`` or wanted as output
        (``compute_contracted_inds``)."""
        if output_inds is None:
            cnt = {}
            for t in self.tensors:
                for ix in dict.fromkeys(t.inds):
                    cnt[ix] = cnt.get(ix, 0) + 1
            output_inds = {ix for ix, c in cnt.items() if c == 1}
        else:
            output_inds = set(output_inds)
        outside = {ix for t in rest for ix in t.inds}
        keep = []
        for t in picked:
            for ix in t.inds:
                if ix not in keep and (ix in outside or ix in output_inds):
                    keep.append(ix)
        return tuple(keep)

    def _unique(self, tags):
        picked, _ = self._select(tags, which="all")
        if len(picked) != 1:
            raise ValueError(f"tags {tags!r} must identify exactly one tensor, found {len(picked)}")
        return picked[0]

    def contract_between(self, tags1, tags2, output_inds=None, equalize_norms=False, **contract_opts):
        """Contract the two tensors identified by ``tags1`` / ``tags2`` in place (no-op if they are the same
        tensor) -- ``contract_between`` / ``_contract_between_tids``, tensor_core.py:6206-6262."""
        t1, t2 = self._unique(tags1), self._unique(tags2)
        if t1 is t2:
            return self
        rest = [t for t in self.tensors if t is not t1 and t is not t2]
        out = self._contracted_inds([t1, t2], rest, output_inds)
        t12 = tensor_contract(t1, t2, output_inds=out, preserve_tensor=True, **contract_opts)
        pos = self.tensors.index(t2)
        self.tensors[pos] = t12
        self.tensors.remove(t1)
        if equalize_norms:
            self.strip_exponent(t12, equalize_norms)
        return self

    def contract_ind(self, ind, output_inds=None, **contract_opts):
        """Contract every tensor carrying ``ind`` into one, in place (``contract_ind``, tensor_core.py:6264-6289);
        ``ind`` survives if it is an output index."""
        picked = [t for t in self.tensors if ind in t.inds]
        if not picked:
            raise ValueError(f"index {ind!r} not found")
        rest = [t for t in self.tensors if not any(t is p for p in picked)]
        out = self._contracted_inds(picked, rest, output_inds)
        tnew = tensor_contract(*picked, output_inds=out, preserve_tensor=True, **contract_opts)
        pos = self.tensors.index(picked[0])
        self.tensors[pos] = tnew
        for p in picked[1:]:
            self.tensors.remove(p)
        return self

    # ---- whole-network queries (tensor_core.py:9880-10030) ------------------------------------------------------
    def contraction_tree(self, optimize=None, output_inds=None, **kwargs):
        from .contract import array_contract_tree

        return array_contract_tree([t.inds for t in self.tensors], output_inds, shapes=[t.shape for t in self.tensors],
                                   optimize=optimize, **kwargs)

    def contraction_width(self, optimize=None, **kwargs):
        return self.contraction_tree(optimize, **kwargs).contraction_width()

    def contraction_cost(self, optimize=None, **kwargs):
        return self.contraction_tree(optimize, **kwargs).contraction_cost()

    def trace(self, left_inds, right_inds, **contract_opts):
        """Trace over ``left_inds`` joined with ``right_inds`` (``TensorNetwork.trace``): the right indices are
        renamed onto the left ones and the network is contracted."""
        left = (left_inds,) if isinstance(left_inds, str) else tuple(left_inds)
        right = (right_inds,) if isinstance(right_inds, str) else tuple(right_inds)
        ren = dict(zip(right, left))
        tn = TensorNetwork([t.reindex(ren) for t in self.tensors], self.exponent)
        return tn.contract(output_inds=contract_opts.pop("output_inds", None) or
                           tuple(ix for ix in tn.outer_inds() if ix not in left), **contract_opts)

    @property
    def arrays(self):
        return tuple(t.data for t in self.tensors)

    def strip_exponent(self, tensor, value=None):
        """Scale ``tensor`` so that its norm is ``value`` (default 1) and move the factor, log10, into
        ``self.exponent``."""
        value = 1.0 if value is None or value is True else float(value)
        i = next(k for k, t in enumerate(self.tensors) if t is tensor)
        factor = ops.norm_fro(asarray(tensor.data)) / value
        self.tensors[i] = Tensor(asarray(tensor.data) / factor, tensor.inds, tensor.tags)
        self.exponent = self.exponent + math.log10(factor)

    def equalize_norms(self, value=None, inplace=False):
        """Make every tensor's norm the same: ``value`` given -> that norm, the factors accumulated in
        ``exponent``; ``value=None`` -> the geometric mean, exponent untouched (tensor_core.py:10843-10880)."""
        tn = self if inplace else self.copy()
        norms = [ops.norm_fro(asarray(t.data)) for t in tn.tensors]
        if value is None:
            target = 10.0 ** (sum(math.log10(n) for n in norms) / len(norms))
        else:
            target = float(value)
            tn.exponent = tn.exponent + sum(math.log10(n / target) for n in norms)
        tn.tensors = [Tensor(asarray(t.data) * (target / n), t.inds, t.tags) for t, n in zip(tn.tensors, norms)]
        return tn

    def equalize_norms_(self, value=None):
        return self.equalize_norms(value, inplace=True)

    def contract_cumulative(self, tags_seq, output_inds=None, strip_exponent=False, equalize_norms="auto",
                            inplace=False, **opts):
        """Contract ``tags_seq[0]``, then that with ``tags_seq[1]``, ... (tensor_core.py:9720-9800); the scale of
        every partial result goes into the exponent when ``equalize_norms``."""
        tn = self if inplace else self.copy()
        if equalize_norms == "auto":
            equalize_norms = strip_exponent
        seq = list(tags_seq)
        acc = ()
        for k, tags in enumerate(seq):
            acc = acc + _tags_of(tags)
            tn.contract_tags(acc, which="any", inplace=True, equalize_norms=equalize_norms,
                             output_inds=output_inds if k == len(seq) - 1 else None, **opts)
        if len(tn.tensors) != 1 or inplace:
            if strip_exponent and not equalize_norms and len(tn.tensors) == 1:
                tn.strip_exponent(tn.tensors[0])
            return tn
        t = tn.tensors[0]
        if strip_exponent:
            if not equalize_norms:
                tn.strip_exponent(t)
                t = tn.tensors[0]
            if not t.inds:
                t = _scalar(t)
            return t, tn.exponent
        if tn.exponent:
            t = Tensor(asarray(t.data) * 10.0**tn.exponent, t.inds, t.tags)
        return _scalar(t) if not t.inds else t

    def to_dense(self, *inds_seq, **opts):
        """Contract everything and fuse the open indices into one axis per group (``TensorNetwork.to_dense``,
        tensor_core.py:10230-10260)."""
        out = tuple(ix for g in inds_seq for ix in g)
        t = self.contract(all, output_inds=out, preserve_tensor=True, **opts)
        if isinstance(t, tuple):
            raise ValueError("to_dense does not combine with strip_exponent")
        return t.to_dense(*inds_seq)

    def aslinearoperator(self, left_inds, right_inds, **opts):
        """This network as an operator from ``right_inds`` to ``left_inds`` without forming the matrix
        (``TensorNetwork.aslinearoperator``, tensor_core.py:10262-10290)."""
        from .linop import TNLinearOperator

        lo = TNLinearOperator(self.tensors, left_inds, right_inds, **opts)
        if self.exponent:
            raise ValueError("aslinearoperator: fold tn.exponent into a tensor first (distribute it)")
        return lo

    def contract_structured(self, site_tags, structure_bsz=5, **opts):
        """1D structured contraction (``TensorNetwork1D.contract_structured``, quimb/tensor/tn1d/core.py:502-557):
        the site tags present in the network, in the given order, grouped ``structure_bsz`` at a time and
        contracted cumulatively."""
        present = self.tags
        seq = [t for t in site_tags if t in present]
        if structure_bsz > 1:
            seq = [tuple(seq[i:i + structure_bsz]) for i in range(0, len(seq), structure_bsz)]
        return self.contract_cumulative(seq, **opts)

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
