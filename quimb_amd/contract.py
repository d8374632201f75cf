"""quimb-facing contraction interface (host mirror of the reference's glue).

Names, argument meaning and error behaviour follow the reference so that the
parity tests read like quimb's own:

* ``array_contract`` / ``array_contract_expression`` / ``array_contract_tree`` /
  ``array_contract_path``   <- quimb/tensor/contraction.py:272-313
* ``get/set_contract_strategy``, ``contract_strategy``, ``get/set_contract_backend``,
  ``contract_backend``, ``get/set_tensor_linop_backend``, ``tensor_linop_backend``
  (per-thread option stacks)  <- quimb/tensor/contraction.py:11-269
* ``tensor_contract`` and the minimal ``Tensor`` holder <- quimb/tensor/tensor_core.py:158-358

Where the reference forwards to cotengra, this module drives ``TreeExecutor``.
"""

import contextlib
import os
import threading
from collections import Counter

import numpy as np

from .array import Array, asarray
from .executor import TreeExecutor
from .pathfind import find_path, find_slices
from .tree import ContractionTree  # noqa: F401  (re-exported: quimb_amd.contract.ContractionTree)


#: a tree goes to the one-launch walker when it has at least this many steps and no intermediate above this size
_MICRO_MIN_STEPS = 24
_MICRO_MAX_ELEMS = 1 << 12


class _OptionStack:
    """A process-wide default plus a per-thread stack of temporary overrides."""

    def __init__(self, default):
        self.default = default
        self._tls = threading.local()

    def get(self):
        st = getattr(self._tls, "stack", None)
        return st[-1] if st else self.default

    def set(self, value):
        self.default = value

    @contextlib.contextmanager
    def override(self, value, set_globally=False):
        if set_globally:
            old, self.default = self.default, value
            try:
                yield
            finally:
                self.default = old
            return
        st = getattr(self._tls, "stack", None)
        if st is None:
            st = self._tls.stack = []
        st.append(value)
        try:
            yield
        finally:
            st.pop()


_STRATEGY = _OptionStack("greedy")
_BACKEND = _OptionStack(None)
_LINOP_BACKEND = _OptionStack(None)

get_contract_strategy, set_contract_strategy, contract_strategy = _STRATEGY.get, _STRATEGY.set, _STRATEGY.override
get_contract_backend, set_contract_backend, contract_backend = _BACKEND.get, _BACKEND.set, _BACKEND.override
get_tensor_linop_backend, set_tensor_linop_backend, tensor_linop_backend = (
    _LINOP_BACKEND.get,
    _LINOP_BACKEND.set,
    _LINOP_BACKEND.override,
)

_KNOWN_BACKENDS = (None, "auto", "quimb_amd", "hip")


def _check_backend(backend):
    if backend not in _KNOWN_BACKENDS:
        raise ValueError(
            f"quimb_amd.array_contract only executes on the 'quimb_amd' backend, got backend={backend!r}"
        )


def _gen_output_inds(all_inds):
    """Indices appearing exactly once, in order of first appearance; an index
    appearing more than twice is an error unless output_inds is given
    (reference: quimb/tensor/tensor_core.py:158-170)."""
    freq = Counter(all_inds)
    for ind, f in freq.items():
        if f > 2:
            raise ValueError(
                f"The index {ind} appears more than twice! If this is intentionally a 'hyper' "
                "tensor network you will need to explicitly supply `output_inds` when contracting for example."
            )
        if f == 1:
            yield ind


def _size_dict(inputs, shapes):
    size = {}
    for t, s in zip(inputs, shapes):
        if len(t) != len(s):
            raise ValueError(f"indices {t} do not match shape {s}")
        for ix, d in zip(t, s):
            if size.setdefault(ix, int(d)) != int(d):
                raise ValueError(f"index {ix!r} has inconsistent sizes {size[ix]} and {d}")
    return size


def array_contract_tree(inputs, output=None, size_dict=None, shapes=None, optimize=None, slicing=None, **_):
    """Find (or adopt) a contraction tree. ``slicing``: dict passed to
    ``find_slices`` (``target_slices`` / ``target_size``)."""
    inputs = tuple(tuple(t) for t in inputs)
    if output is None:
        output = tuple(_gen_output_inds(ix for t in inputs for ix in t))
    if size_dict is None:
        size_dict = _size_dict(inputs, shapes)
    if optimize is None:
        optimize = get_contract_strategy()
    tree = find_path(inputs, tuple(output), size_dict, optimize)
    if slicing:
        tree = find_slices(tree, **slicing)
    return tree


def array_contract_path(*args, **kwargs):
    return array_contract_tree(*args, **kwargs).get_path()


class ContractExpression:
    """Callable ``expr(*arrays, backend=None)`` bound to one tree + dtype."""

    def __init__(self, tree, dtype, strip_exponent=False, constants=None):
        self.tree = tree
        self.executor = TreeExecutor(tree, dtype)
        self.strip_exponent = strip_exponent
        self.constants = dict(constants or {})
        self._const_dev = {k: asarray(v).astype(dtype) for k, v in self.constants.items()}
        # trees of many small tensors (circuit amplitudes) are dispatch-bound step by step: let the device walk
        # them in one launch (MicroTree) -- same plan, same arithmetic order per step; QAMD_MICROTREE=0 opts out
        self._micro = None
        if (not strip_exponent and tree.nslices == 1 and len(tree.steps) >= _MICRO_MIN_STEPS
                and os.environ.get("QAMD_MICROTREE", "1") != "0"):
            try:
                from .microtree import MicroTree

                self._micro = MicroTree(tree, dtype, max_elems=_MICRO_MAX_ELEMS)
            except ValueError:
                self._micro = None

    def __call__(self, *arrays, backend=None, slices=None):
        _check_backend(backend)
        if self._const_dev:
            it = iter(arrays)
            n = len(self.tree.inputs)
            arrays = [self._const_dev[i] if i in self._const_dev else next(it) for i in range(n)]
        host_in = not any(isinstance(a, Array) for a in arrays if not any(a is c for c in self._const_dev.values()))
        if self._micro is not None and slices is None:
            out = self._micro(arrays)
            return out.to_numpy() if host_in else out
        out = self.executor(arrays, strip_exponent=self.strip_exponent, slices=slices)
        if self.strip_exponent:
            out, e = out
            return (out.to_numpy() if host_in else out), e
        return out.to_numpy() if host_in else out


_EXPR_CACHE = {}
_EXPR_CACHE_MAX = 4096


def array_contract_expression(
    inputs, output=None, size_dict=None, shapes=None, optimize=None, dtype="float64",
    strip_exponent=False, constants=None, cache=True, slicing=None, **_
):
    inputs = tuple(tuple(t) for t in inputs)
    if output is None:
        output = tuple(_gen_output_inds(ix for t in inputs for ix in t))
    output = tuple(output)
    if size_dict is None:
        size_dict = _size_dict(inputs, shapes)
    if optimize is None:
        optimize = get_contract_strategy()
    key = None
    if cache and constants is None:
        try:
            # the optimizer object itself is part of the key (it stays alive with the cache entry, so its
            # identity cannot be recycled); unhashable ones (lists) raise TypeError below -> not cached
            okey = optimize if isinstance(optimize, str) else ("obj", optimize)
            hash(okey)
            key = (inputs, output, tuple(sorted(size_dict.items(), key=repr)), okey, np.dtype(dtype).name,
                   bool(strip_exponent), repr(slicing))
            hit = _EXPR_CACHE.get(key)
            if hit is not None:
                return hit
        except TypeError:
            key = None
    tree = array_contract_tree(inputs, output, size_dict, optimize=optimize, slicing=slicing)
    expr = ContractExpression(tree, dtype, strip_exponent, constants)
    if key is not None:
        if len(_EXPR_CACHE) >= _EXPR_CACHE_MAX:
            _EXPR_CACHE.clear()
        _EXPR_CACHE[key] = expr
    return expr


def array_contract(arrays, inputs, output=None, optimize=None, backend=None, strip_exponent=False,
                   slicing=None, **kwargs):
    """Contract ``arrays`` labelled by ``inputs`` into ``output`` on the MI355X.

    Same call shape as ``quimb.tensor.contraction.array_contract``
    (contraction.py:272-292): ``optimize`` falls back to the thread's contract
    strategy, ``backend`` to the thread's contract backend.  numpy inputs give a
    numpy result (0-d for scalars, so quimb's ``maybe_realify_scalar`` unwraps
    them, tensor_core.py:215-221); ``Array`` inputs stay on the device."""
    if backend is None:
        backend = get_contract_backend()
    _check_backend(backend)
    arrays = list(arrays)
    if not arrays:
        raise ValueError("nothing to contract")
    shapes = [tuple(np.shape(a)) if not isinstance(a, Array) else a.shape for a in arrays]
    dt = np.result_type(*[a.dtype if hasattr(a, "dtype") else np.asarray(a).dtype for a in arrays])
    if dt.kind in "iub":
        dt = np.dtype("float64")
    expr = array_contract_expression(
        inputs, output, shapes=shapes, optimize=optimize, dtype=dt, strip_exponent=strip_exponent,
        slicing=slicing, **kwargs
    )
    return expr(*arrays)


# ---------------------------------------------------------------------------
# minimal Tensor holder + tensor_contract (reference: tensor_core.py:224-358)
# ---------------------------------------------------------------------------
class Tensor:
    """Labelled array: ``data`` + ``inds`` + ``tags`` (just what
    ``tensor_contract`` needs from quimb's ``Tensor``, tensor_core.py:1919)."""

    __slots__ = ("data", "inds", "tags")

    def __init__(self, data, inds, tags=None):
        self.data = data
        self.inds = tuple(inds)
        if len(self.inds) != len(np.shape(data) if not isinstance(data, Array) else data.shape):
            raise ValueError(f"Wrong number of inds, {self.inds}, supplied for array of shape {self.shape}.")
        if tags is None:
            tags = ()
        elif isinstance(tags, str):
            tags = (tags,)
        self.tags = tuple(dict.fromkeys(tags))

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def ndim(self):
        return len(self.inds)

    @property
    def dtype(self):
        return self.data.dtype

    def __matmul__(self, other):
        return tensor_contract(self, other)

    def __and__(self, other):
        from .network import TensorNetwork

        return TensorNetwork((self,)) & other

    def __repr__(self):
        return f"Tensor(shape={self.shape}, inds={self.inds}, tags={self.tags})"


def _realify_scalar(x, imag_tol=1e-12):
    if isinstance(x, complex) or np.iscomplexobj(x):
        return x.real if abs(x.imag) < abs(x.real) * imag_tol else x
    return x


def tensor_contract(*tensors, output_inds=None, optimize=None, backend=None, preserve_tensor=False,
                    drop_tags=False, strip_exponent=False, exponent=None, get=None, **contract_opts):
    """Contract labelled tensors; semantics of the reference's
    ``tensor_contract`` (tensor_core.py:224-358): output indices are those that
    appear once, in order of first appearance; a fully contracted result is
    returned as a python scalar unless ``preserve_tensor``; tags are unioned;
    ``strip_exponent`` returns ``(mantissa, exponent)`` with
    ``mantissa * 10**exponent == value``."""
    inds = tuple(t.inds for t in tensors)
    shapes = tuple(t.shape for t in tensors)
    arrays = tuple(t.data for t in tensors)
    if output_inds is None:
        inds_out = tuple(_gen_output_inds(ix for t in inds for ix in t))
    else:
        inds_out = tuple(output_inds)
    if get is not None:
        if get == "tree":
            return array_contract_tree(inds, inds_out, shapes=shapes, optimize=optimize, **contract_opts)
        if get == "path":
            return array_contract_path(inds, inds_out, shapes=shapes, optimize=optimize, **contract_opts)
        if get == "expression":
            dt = np.result_type(*[a.dtype for a in arrays])
            return array_contract_expression(inds, inds_out, shapes=shapes, optimize=optimize, dtype=dt, **contract_opts)
        raise ValueError(f"unsupported get={get!r}: options are 'tree', 'path', 'expression'")

    data_out = array_contract(arrays, inds, inds_out, optimize=optimize, strip_exponent=strip_exponent,
                              backend=backend, **contract_opts)
    result_exponent = None
    if strip_exponent:
        data_out, result_exponent = data_out
        if exponent is not None:
            result_exponent = result_exponent + exponent
    elif exponent is not None:
        data_out = data_out * 10**exponent

    if not inds_out and not preserve_tensor:
        if isinstance(data_out, Array):
            data_out = data_out.to_numpy()
        result = _realify_scalar(np.asarray(data_out).item())
    else:
        tags_out = None if drop_tags else tuple(dict.fromkeys(tg for t in tensors for tg in t.tags))
        result = Tensor(data_out, inds_out, tags_out)
    if strip_exponent:
        return result, result_exponent
    return result
