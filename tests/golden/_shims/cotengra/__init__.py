"""Minimal numpy-only stand-in for cotengra==0.8.2 (not vendored in the
reference): the call surface quimb's tensor core imports, with the published
pairwise algorithm restated (greedy path; per step tensordot/einsum; optional
strip_exponent).  Only used by tests/golden/make_golden.py."""
import itertools
import math

import numpy as np

from autoray import do

from . import utils  # noqa: F401

_BASE = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"


def get_symbol(i):
    if i < 52:
        return _BASE[i]
    return chr(i + 140)


def get_symbol_map(inputs):
    syms = {}
    for t in inputs:
        for ix in t:
            if ix not in syms:
                syms[ix] = get_symbol(len(syms))
    return syms


class PathOptimizer:
    pass


class HyperOptimizer(PathOptimizer):
    def __init__(self, *a, **k):
        raise NotImplementedError("hyper-optimisation is not part of the golden-generation shim")


ReusableHyperOptimizer = HyperCompressedOptimizer = ReusableHyperCompressedOptimizer = HyperOptimizer


class ContractionTree:
    """Just enough of a tree: inputs/output/size_dict + linear path."""

    def __init__(self, inputs, output, size_dict, path):
        self.inputs, self.output, self.size_dict, self._path = inputs, output, size_dict, path
        self.sliced_inds = ()

    def get_path(self):
        return self._path

    def contraction_cost(self):
        inputs = [tuple(t) for t in self.inputs]
        cost = 0
        for con in self._path:
            con = tuple(sorted(con, reverse=True))
            ops = [inputs.pop(p) for p in con]
            allinds = set().union(*ops)
            rest = set(self.output).union(*inputs) if inputs else set(self.output)
            cost += math.prod(self.size_dict[i] for i in allinds)
            inputs.append(tuple(i for i in dict.fromkeys(itertools.chain(*ops[::-1])) if i in rest))
        return cost

    def contraction_width(self):
        inputs = [tuple(t) for t in self.inputs]
        w = max((math.prod(self.size_dict[i] for i in t) for t in inputs), default=1)
        for con in self._path:
            con = tuple(sorted(con, reverse=True))
            ops = [inputs.pop(p) for p in con]
            rest = set(self.output).union(*inputs) if inputs else set(self.output)
            new = tuple(i for i in dict.fromkeys(itertools.chain(*ops[::-1])) if i in rest)
            w = max(w, math.prod(self.size_dict[i] for i in new))
            inputs.append(new)
        return math.log2(w)


class ContractionTreeCompressed(ContractionTree):
    pass


def _greedy_path(inputs, output, size_dict):
    inputs = [tuple(t) for t in inputs]
    ids = list(range(len(inputs)))
    terms = dict(enumerate(inputs))
    path = []
    while len(ids) > 1:
        best = None
        for a, b in itertools.combinations(range(len(ids)), 2):
            ta, tb = terms[ids[a]], terms[ids[b]]
            shared = set(ta) & set(tb)
            rest = set(output)
            for k, idk in enumerate(ids):
                if k not in (a, b):
                    rest.update(terms[idk])
            new = tuple(i for i in dict.fromkeys(ta + tb) if i in rest)
            sz = lambda t: math.prod(size_dict[i] for i in t)
            score = (0 if shared else 1, sz(new) - sz(ta) - sz(tb))
            if best is None or score < best[0]:
                best = (score, a, b, new)
        _, a, b, new = best
        path.append((a, b))
        nid = max(terms) + 1
        terms[nid] = new
        for p in sorted((a, b), reverse=True):
            ids.pop(p)
        ids.append(nid)
    return path


def _resolve_path(inputs, output, size_dict, optimize):
    if isinstance(optimize, ContractionTree):
        return optimize.get_path()
    if isinstance(optimize, (list, tuple)):
        return [tuple(c) for c in optimize]
    return _greedy_path(inputs, output, size_dict)


def _infer_output(inputs):
    counts = {}
    for t in inputs:
        for ix in t:
            counts[ix] = counts.get(ix, 0) + 1
    return tuple(ix for ix, c in counts.items() if c == 1)


def array_contract_tree(inputs, output=None, size_dict=None, shapes=None, optimize="auto", **kwargs):
    inputs = tuple(tuple(t) for t in inputs)
    if output is None:
        output = _infer_output(inputs)
    if size_dict is None:
        size_dict = {ix: d for t, s in zip(inputs, shapes) for ix, d in zip(t, s)}
    return ContractionTree(inputs, tuple(output), size_dict, _resolve_path(inputs, output, size_dict, optimize))


def array_contract_path(*args, **kwargs):
    return array_contract_tree(*args, **kwargs).get_path()


def _pair_step(ops, tis, new, backend, implementation):
    """One pairwise step the way cotengra executes it: ``tensordot`` (+ a transpose into the kept order) when
    no index is batched / repeated, ``einsum`` otherwise; through the caller's ``implementation=(tensordot,
    einsum)`` pair if one was given, else through autoray on ``backend`` (default: the operands' own)."""
    la, lb = tis
    shared = [ix for ix in la if ix in lb]
    plain = len(set(la)) == len(la) and len(set(lb)) == len(lb) and not any(ix in new for ix in shared)
    if isinstance(implementation, tuple):
        td, es = implementation
        tr = lambda x, perm: do("transpose", x, perm)
    else:
        td = lambda a, b, axes: do("tensordot", a, b, axes, like=backend)
        es = lambda eq, *xs: do("einsum", eq, *xs, like=backend)
        tr = lambda x, perm: do("transpose", x, perm)
    if plain and isinstance(implementation, tuple):
        axes = ([la.index(ix) for ix in shared], [lb.index(ix) for ix in shared])
        x = td(ops[0], ops[1], axes)
        got = tuple(ix for ix in la if ix not in shared) + tuple(ix for ix in lb if ix not in shared)
        if got != tuple(new):
            x = tr(x, tuple(got.index(ix) for ix in new))
        return x
    syms = get_symbol_map(list(tis) + [new])
    eq = ",".join("".join(syms[i] for i in t) for t in tis) + "->" + "".join(syms[i] for i in new)
    return es(eq, *ops)


def _contract(arrays, inputs, output, path, strip_exponent, backend=None, implementation=None):
    arrays = list(arrays)
    inputs = [tuple(t) for t in inputs]
    exponent = 0.0
    for con in path:
        con = tuple(sorted(con, reverse=True))
        ops = [arrays.pop(p) for p in con][::-1]
        tis = [inputs.pop(p) for p in con][::-1]
        rest = set(output)
        for t in inputs:
            rest.update(t)
        new = tuple(i for i in dict.fromkeys(itertools.chain(*tis)) if i in rest)
        if len(ops) == 2 and (backend is not None or implementation is not None):
            x = _pair_step(ops, tis, new, backend, implementation)
        else:
            syms = get_symbol_map(tis + [new])
            eq = ",".join("".join(syms[i] for i in t) for t in tis) + "->" + "".join(syms[i] for i in new)
            x = do("einsum", eq, *ops)                 # autoray dispatch on the operands' backend, as cotengra does
        if strip_exponent:
            f = do("max", do("abs", x))
            f = float(f.item() if hasattr(f, "item") else f)
            if f > 0:
                x = x / f
                exponent += math.log10(f)
        arrays.append(x)
        inputs.append(new)
    syms = get_symbol_map(inputs + [tuple(output)])
    eq = ",".join("".join(syms[i] for i in t) for t in inputs) + "->" + "".join(syms[i] for i in output)
    if isinstance(implementation, tuple):
        x = implementation[1](eq, *arrays)
    else:
        x = do("einsum", eq, *arrays, like=backend)
    return (x, exponent) if strip_exponent else x


class _Expression:
    def __init__(self, tree, strip_exponent, constants):
        self.tree, self.strip_exponent, self.constants = tree, strip_exponent, constants or {}

    def __call__(self, *arrays, backend=None):
        if self.constants:
            it = iter(arrays)
            arrays = [self.constants[i] if i in self.constants else next(it) for i in range(len(self.tree.inputs))]
        return _contract(arrays, self.tree.inputs, self.tree.output, self.tree.get_path(), self.strip_exponent)


def array_contract_expression(inputs, output=None, size_dict=None, shapes=None, optimize="auto",
                              constants=None, strip_exponent=False, **kwargs):
    tree = array_contract_tree(inputs, output, size_dict, shapes, optimize)
    return _Expression(tree, strip_exponent, constants)


def array_contract(arrays, inputs, output=None, optimize="auto", backend=None, strip_exponent=False,
                   implementation=None, **kwargs):
    """``backend``: explicit backend for the pairwise calls (cotengra: "by default determined from the input
    arrays"); ``implementation``: "auto" / a (tensordot, einsum) pair of callables (cotengra's documented kwarg)."""
    shapes = [np.shape(a) for a in arrays]
    tree = array_contract_tree(inputs, output, shapes=shapes, optimize=optimize)
    if isinstance(implementation, str):
        implementation = None
    return _contract(arrays, tree.inputs, tree.output, tree.get_path(), strip_exponent, backend, implementation)


def get_hypergraph(*a, **k):
    raise NotImplementedError


def register_preset(*a, **k):
    pass
