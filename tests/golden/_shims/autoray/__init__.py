"""Minimal numpy-only stand-in for autoray==0.10.1 (not vendored in the
reference): just the dispatch surface quimb's tensor core imports.  Only used by
tests/golden/make_golden.py to run the REAL quimb sources on the CPU."""
import functools
import importlib
import numbers

import numpy as np

from . import lazy  # noqa: F401

_REGISTRY = {}
_BACKEND_ALIASES = {"builtins": "numpy", "quimb": "numpy"}      # qarray is an ndarray subclass


def infer_backend(x):
    if isinstance(x, (numbers.Number, list, tuple)):
        return "numpy"
    mod = type(x).__module__.split(".")[0]
    return _BACKEND_ALIASES.get(mod, mod)


def infer_backend_multi(*arrays):
    return infer_backend(arrays[0])


def register_function(backend, name, fn, wrap=False):
    _REGISTRY[backend, name] = fn


def _np_fn(name):
    if (("numpy", name)) in _REGISTRY:
        return _REGISTRY["numpy", name]
    if name.startswith("scipy."):          # autoray resolves ``scipy.linalg.*`` for the numpy backend from scipy itself
        obj = importlib.import_module(name.rsplit(".", 1)[0])
        return getattr(obj, name.rsplit(".", 1)[1])
    obj = np
    for part in name.split("."):
        obj = getattr(obj, part)
    return obj


# autoray names numpy does not have (autoray translates them for the numpy backend)
_CUSTOM = {
    "complex": lambda re, im: np.asarray(re) + 1j * np.asarray(im),
}


def get_lib_fn(backend, name):
    if (backend, name) in _REGISTRY:
        return _REGISTRY[backend, name]
    if name in _CUSTOM:
        return _CUSTOM[name]
    if backend in ("numpy", "builtins"):
        return _np_fn(name)
    mod = importlib.import_module(backend)
    obj = mod
    for part in name.split("."):
        obj = getattr(obj, part)
    return obj


def do(fn, *args, like=None, **kwargs):
    if like is None:
        if fn == "einsum" and len(args) > 1 and isinstance(args[0], str):
            backend = infer_backend(args[1])        # the equation comes first (autoray's einsum dispatcher)
        else:
            backend = infer_backend(args[0]) if args else "numpy"
    elif isinstance(like, str):
        backend = like
    else:
        backend = infer_backend(like)
    return get_lib_fn(backend, fn)(*args, **kwargs)


class _Namespace:
    def __init__(self, backend):
        self._backend = backend

    def __getattr__(self, name):
        if name in ("linalg", "scipy", "random"):
            return _SubNamespace(self._backend, name)
        return get_lib_fn(self._backend, name)


class _SubNamespace:
    def __init__(self, backend, prefix):
        self._backend, self._prefix = backend, prefix

    def __getattr__(self, name):
        if self._prefix == "scipy" and name == "linalg":
            return _SubNamespace(self._backend, "scipy.linalg")
        return get_lib_fn(self._backend, f"{self._prefix}.{name}")


def get_namespace(like=None):
    if isinstance(like, str):
        return _Namespace(like)
    return _Namespace(infer_backend(like) if like is not None else "numpy")


class DoFunc:
    pass


def compose(fn=None, *, name=None):
    """Make ``fn`` dispatchable via do(fn.__name__) with .register overrides."""
    def deco(f):
        nm = name or f.__name__
        overrides = {}

        @functools.wraps(f)
        def wrapper(*args, like=None, **kwargs):
            backend = like if isinstance(like, str) else infer_backend(like if like is not None else args[0])
            impl = overrides.get(backend, f)
            return impl(*args, **kwargs)

        def register(backend, g=None):
            if g is None:
                def inner(h):
                    overrides[backend] = h
                    return h
                return inner
            overrides[backend] = g
            return g

        wrapper.register = register
        _CUSTOM[nm] = lambda *a, **k: wrapper(*a, **k)
        return wrapper

    return deco(fn) if fn is not None else deco


def conj(x):
    return np.conj(x)


def dag(x):
    return np.conj(np.swapaxes(x, -1, -2)) if np.ndim(x) >= 2 else np.conj(x)


def reshape(x, shape):
    return np.reshape(x, shape)


def shape(x):
    return tuple(np.shape(x))


def ndim(x):
    return np.ndim(x)


def size(x):
    return np.size(x)


def transpose(x, perm=None):
    return np.transpose(x, perm)


def real(x):
    return np.real(x)


def imag(x):
    return np.imag(x)


def to_numpy(x):
    b = infer_backend(x)
    if (b, "to_numpy") in _REGISTRY:
        return _REGISTRY[b, "to_numpy"](x)
    return np.asarray(x)


def astype(x, dtype):
    b = infer_backend(x)
    if b != "numpy":                      # autoray.astype dispatches on the array's backend
        return get_lib_fn(b, "astype")(x, dtype)
    return np.asarray(x).astype(dtype)


def get_dtype_name(x):
    return str(x.dtype) if hasattr(x, "dtype") else np.asarray(x).dtype.name


def get_common_dtype(*arrays):
    return np.result_type(*[a.dtype for a in arrays]).name


def to(x, like=None, backend=None, dtype=None, device=None):
    """autoray.to: convert every array leaf of a pytree to a target backend / dtype (``TensorNetwork.to``,
    quimb/tensor/tensor_core.py:5312-5356)."""
    if isinstance(like, str):            # "backend-dtype-device", each part optional
        for part in like.split("-"):
            if part.startswith(("float", "complex", "int")):
                dtype = dtype or part
            elif ":" in part or part in ("cpu", "cuda"):
                device = device or part
            elif backend is None:
                backend = part

    def conv(a):
        if not hasattr(a, "shape"):
            return a
        if backend is not None and infer_backend(a) != backend:
            a = do("asarray", a, like=backend)
        if dtype is not None:
            a = astype(a, dtype)
        return a

    return tree_map(conv, x)


def tree_map(f, tree, is_leaf=None):
    if isinstance(tree, (list, tuple)):
        return type(tree)(tree_map(f, t) for t in tree)
    if isinstance(tree, dict):
        return {k: tree_map(f, v) for k, v in tree.items()}
    return f(tree)


def tree_flatten(tree, get_ref=False):
    leaves = []
    def rec(t):
        if isinstance(t, (list, tuple)):
            return type(t)(rec(x) for x in t)
        if isinstance(t, dict):
            return {k: rec(v) for k, v in t.items()}
        leaves.append(t)
        return None
    ref = rec(tree)
    return (leaves, ref) if get_ref else leaves


def tree_unflatten(leaves, ref):
    it = iter(leaves)
    def rec(t):
        if isinstance(t, (list, tuple)):
            return type(t)(rec(x) for x in t)
        if isinstance(t, dict):
            return {k: rec(v) for k, v in t.items()}
        return next(it)
    return rec(ref)


def tree_apply(f, tree):
    tree_map(f, tree)


def tree_iter(tree):
    return iter(tree_flatten(tree))


def autojit(fn=None, **kwargs):
    if fn is None:
        return lambda f: f
    return fn


class backend_like:
    def __init__(self, like, set_globally="auto"):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


numpy = np
