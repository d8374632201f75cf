"""Module-level L0 array functions -- the surface quimb's ``do(name, ...,
like="quimb_amd")`` resolves to.

``tensordot`` / ``einsum`` / ``matmul`` run on the GETT kernel with the operand
permutes folded into addressing; ``transpose`` / ``fuse`` / ``take`` use the
tiled permute kernel; ``reshape`` is free.  Reference call sites:
quimb/tensor/tensor_core.py:3793 (``do("tensordot")``), :3152-3159
(``Tensor.gate``), quimb/tensor/array_ops.py:148-182 (``fuse``),
tensor_core.py:2332 (``take``).
"""

import builtins
import numbers

import numpy as np

from .options import get_options

from .array import Array, asarray, _coerce_dtype, _REAL_OF
from .pairwise import (
    GettSpec,
    PermuteSpec,
    ReduceSpec,
    contig_strides,
    parse_einsum,
    plan_pair,
    prod,
    tensordot_inds,
)


def _common(a, b):
    a, b = asarray(a), asarray(b)
    dt = _coerce_dtype(np.result_type(a.dtype, b.dtype))
    return a.astype(dt), b.astype(dt), dt


def _apply_pre(x, ops):
    """Run the single-operand preprocessing a PairStep asks for."""
    dev = x._dev
    for op in ops:
        if isinstance(op, ReduceSpec):
            out = Array.empty(op.keep_shape, x.dtype, dev)
            if out.size:
                dev.reduce_sum(out._buf, x._buf, op.keep_shape, op.keep_strides, op.red_shape, op.red_strides, x.dtype)
            x = out
        elif isinstance(op, PermuteSpec):
            out = Array.empty(op.shape, x.dtype, dev)
            if out.size:
                dev.permute(out._buf, x._buf, op.shape, op.strides, 0, x.dtype)
            x = out
        else:  # pragma: no cover
            raise TypeError(op)
    return x


def run_pair_step(step, a, b, out=None, ep=None):
    """Execute a planned pairwise step on device arrays of a common dtype.
    ``ep`` = (slots_a, slots_b, slots_out) in CALLER operand order enables the
    fused exponent-stripping epilogue of the GETT kernels."""
    a = _apply_pre(a, step.pre[0])
    b = _apply_pre(b, step.pre[1])
    dev = a._dev
    if out is None:
        out = Array.empty(step.out_shape, a.dtype, dev)
    if out.size == 0:
        return out
    if step.kind == "binary":
        s = step.spec
        dev.binary(out._buf, a._buf, s.sa, b._buf, s.sb, s.shape, "mul", a.dtype)
        return out
    ka, kb = (b, a) if step.swapped else (a, b)
    if ep is not None and step.swapped:
        ep = (ep[1], ep[0], ep[2])
    if a.dtype.kind == "c":
        _complex_gett(dev, step.spec, ka, kb, out)
    else:
        dev.contract_pair(step.spec, a.dtype, ka._buf, kb._buf, out._buf, ep)
    return out


def _complex_spec(spec):
    """Real GETT spec of a complex contraction: every stride doubles (interleaved
    re/im), the expanded small operand's strides quadruple, K gains an innermost
    group over A's component c, N one over the output component d."""
    b = tuple((d, 2 * sa, 4 * sb, 2 * sc) for d, sa, sb, sc in spec.b)
    m = tuple((d, 2 * sa, None, 2 * sc) for d, sa, _, sc in spec.m)
    n = tuple((d, None, 4 * sb, 2 * sc) for d, _, sb, sc in spec.n) + ((2, None, 1, 1),)
    k = tuple((d, 2 * sa, 4 * sb, None) for d, sa, sb, _ in spec.k) + ((2, 1, 2, None),)
    return GettSpec(b=b, m=m, n=n, k=k)


def _complex_gett(dev, spec, ka, kb, out):
    """Complex pairwise contraction on the REAL MFMA kernels (the 4-multiply form):
    the small operand is expanded once to 2x2 real blocks, the big operand and the
    result are used in place through their interleaved real views."""
    rdt = _REAL_OF[ka.dtype]
    kb2 = dev.empty(4 * kb.size, rdt)
    dev.complex_expand(kb2, kb._buf, kb.size, ka.dtype)
    # (the 2 x 2 real blocks interleave re / im along K as an innermost K group of two: under the opt-in split products
    # (Options.join_arith = "f16x3") the library centres such operands per PARITY of k -- csrc/gemmh.hip, SplitArgs.period)
    dev.contract_pair(_complex_spec(spec), rdt, dev.as_real(ka._buf), kb2, dev.as_real(out._buf))
    if hasattr(dev, "release_temp"):
        dev.release_temp(kb2)


def einsum_pair(a, a_inds, b, b_inds, out_inds, out_fixed=True, death=None):
    """``einsum`` of two operands with arbitrary hashable index labels.
    Returns (array, out_inds)."""
    a, b, _ = _common(a, b)
    try:
        step = plan_pair(tuple(a_inds), a.shape, tuple(b_inds), b.shape, tuple(out_inds), out_fixed, death)
    except NotImplementedError:
        if not out_fixed:
            raise
        # the requested order interleaves more index groups than one launch can address: contract into the
        # kernel's own order, then one permute pass
        step = plan_pair(tuple(a_inds), a.shape, tuple(b_inds), b.shape, tuple(out_inds), False, death)
        res = run_pair_step(step, a, b)
        perm = [step.out_inds.index(ix) for ix in out_inds]
        return transpose(res, perm), tuple(out_inds)
    return run_pair_step(step, a, b), step.out_inds


def tensordot(a, b, axes=2):
    """numpy-semantics ``tensordot`` (output = free axes of a, then of b)."""
    a, b, _ = _common(a, b)
    ai, bi, oi = tensordot_inds(a.ndim, b.ndim, axes)
    out, _ = einsum_pair(a, ai, b, bi, oi, True)
    return out


def matmul(a, b):
    a, b, _ = _common(a, b)
    if a.ndim == 0 or b.ndim == 0:
        raise ValueError("matmul: input operand does not have enough dimensions")
    if a.ndim == 1 and b.ndim == 1:
        return tensordot(a, b, 1)
    if a.ndim == 1:
        return matmul(a.reshape(1, -1), b).reshape(b.shape[:-2] + b.shape[-1:])
    if b.ndim == 1:
        return matmul(a, b.reshape(-1, 1)).reshape(a.shape[:-1])
    # broadcast batch dims
    ba, bb = a.shape[:-2], b.shape[:-2]
    bshape = np.broadcast_shapes(ba, bb)
    nb = len(bshape)
    a_inds = [("B", nb - len(ba) + i) if d != 1 or bshape[nb - len(ba) + i] == 1 else ("a1", i) for i, d in enumerate(ba)]
    b_inds = [("B", nb - len(bb) + i) if d != 1 or bshape[nb - len(bb) + i] == 1 else ("b1", i) for i, d in enumerate(bb)]
    a_inds += ["m", "k"]
    b_inds += ["k", "n"]
    if any(isinstance(i, tuple) and i[0] in ("a1", "b1") for i in a_inds + b_inds):
        # size-1 broadcast dims: drop them via reshape, then restore
        a2 = a.reshape([d for d, ix in zip(a.shape, a_inds) if not (isinstance(ix, tuple) and ix[0] == "a1")])
        b2 = b.reshape([d for d, ix in zip(b.shape, b_inds) if not (isinstance(ix, tuple) and ix[0] == "b1")])
        a_inds = [ix for ix in a_inds if not (isinstance(ix, tuple) and ix[0] == "a1")]
        b_inds = [ix for ix in b_inds if not (isinstance(ix, tuple) and ix[0] == "b1")]
        a, b = a2, b2
    out_inds = [("B", i) for i in range(nb)] + ["m", "n"]
    out, _ = einsum_pair(a, a_inds, b, b_inds, out_inds, True)
    return out


dot = matmul


def einsum(eq, *operands, **kwargs):
    """numpy-style ``einsum`` for one or two operands (what cotengra's executor
    issues per step); more operands go through the tree executor."""
    if not isinstance(eq, str):
        raise TypeError("einsum: interleaved format is not supported")
    ops = [asarray(x) for x in operands]
    inputs, output = parse_einsum(eq, len(ops))
    if len(ops) == 1:
        return _einsum_single(ops[0], inputs[0], output)
    if len(ops) == 2:
        out, _ = einsum_pair(ops[0], inputs[0], ops[1], inputs[1], output, True)
        return out
    from .contract import array_contract

    return array_contract(ops, inputs, output)


def _einsum_single(x, inds, out_inds):
    """Single-operand einsum: diagonals (repeated labels), sums, permutation."""
    from .pairwise import _view

    v = _view(inds, x.shape)
    sz = dict(zip(v.inds, v.shape))
    st = dict(zip(v.inds, v.strides))
    for ix in out_inds:
        if ix not in sz:
            raise ValueError(f"einsum: output index {ix!r} not in input")
    if len(set(out_inds)) != len(out_inds):
        raise ValueError("einsum: repeated output index")
    red = [ix for ix in v.inds if ix not in out_inds]
    oshape = [sz[ix] for ix in out_inds]
    ostr = [st[ix] for ix in out_inds]
    out = Array.empty(oshape, x.dtype, x._dev)
    if not out.size:
        return out
    if red:
        x._dev.reduce_sum(out._buf, x._buf, oshape, ostr, [sz[ix] for ix in red], [st[ix] for ix in red], x.dtype)
    else:
        x._dev.permute(out._buf, x._buf, oshape, ostr, 0, x.dtype)
    return out


# ---- layout ---------------------------------------------------------------
def transpose(x, axes=None):
    x = asarray(x)
    return x.transpose() if axes is None else x.transpose(axes)


def reshape(x, shape):
    return asarray(x).reshape(shape)


def ravel(x):
    return asarray(x).ravel()


def fuse(x, *axes_groups):
    """Index fusion in ONE permute pass: same semantics as the reference's
    composed ``fuse`` (quimb/tensor/array_ops.py:95-182): each group of axes is
    fused into a single axis placed at the position of the group's minimum axis,
    groups in the order given, unfused axes keep their relative order."""
    x = asarray(x)
    groups = [tuple(int(a) % x.ndim for a in g) for g in axes_groups]
    if not any(groups):
        return x
    in_group = {a for g in groups for a in g}
    position = builtins.min(in_group)
    before = [ax for ax in range(position) if ax not in in_group]
    after = [ax for ax in range(position, x.ndim) if ax not in in_group]
    perm = before + [ax for g in groups for ax in g] + after
    new_shape = (
        [x.shape[a] for a in before] + [prod(x.shape[a] for a in g) for g in groups] + [x.shape[a] for a in after]
    )
    return x.transpose(perm).reshape(new_shape)


def take(x, indices, axis=None):
    x = asarray(x)
    if axis is None:
        x = x.ravel()
        axis = 0
    axis = int(axis) % x.ndim
    if isinstance(indices, numbers.Integral):
        key = [slice(None)] * x.ndim
        key[axis] = int(indices)
        return x[tuple(key)]
    idx = [int(i) for i in np.asarray(indices).reshape(-1)]
    parts = []
    for i in idx:
        key = [slice(None)] * x.ndim
        key[axis] = slice(i, i + 1) if i != -1 else slice(i, None)
        parts.append(x[tuple(key)])
    return concatenate(parts, axis=axis)


def swapaxes(x, axis1, axis2):
    x = asarray(x)
    perm = list(range(x.ndim))
    perm[axis1 % x.ndim], perm[axis2 % x.ndim] = perm[axis2 % x.ndim], perm[axis1 % x.ndim]
    return x.transpose(perm)


def moveaxis(x, source, destination):
    x = asarray(x)
    src = [source] if isinstance(source, numbers.Integral) else list(source)
    dst = [destination] if isinstance(destination, numbers.Integral) else list(destination)
    src, dst = [a % x.ndim for a in src], [a % x.ndim for a in dst]
    if len(src) != len(dst) or len(set(src)) != len(src) or len(set(dst)) != len(dst):
        raise ValueError("moveaxis: source and destination must be the same number of distinct axes")
    perm = [a for a in range(x.ndim) if a not in src]
    for d, s_ in sorted(zip(dst, src)):
        perm.insert(d, s_)
    return x.transpose(perm)


def concatenate(arrays, axis=0):
    """numpy.concatenate on the device: every piece is copied (one strided pass, its ``axis`` moved to the
    front) straight into its block of the result, which is moved back if ``axis`` is not the leading one."""
    arrays = [asarray(a) for a in arrays]
    if not arrays:
        raise ValueError("need at least one array to concatenate")
    a0 = arrays[0]
    arrays = [a.astype(a0.dtype) for a in arrays]
    nd = a0.ndim
    if nd == 0:
        raise ValueError("zero-dimensional arrays cannot be concatenated")
    axis = axis % nd
    rest = a0.shape[:axis] + a0.shape[axis + 1:]
    for a in arrays:
        if a.ndim != nd or a.shape[:axis] + a.shape[axis + 1:] != rest:
            raise ValueError("all the input array dimensions except for the concatenation axis must match exactly")
    if len(arrays) == 1:
        return a0
    total = builtins.sum(a.shape[axis] for a in arrays)
    out = Array.empty((total,) + rest, a0.dtype, a0._dev)
    block = prod(rest)
    pos = 0
    perm = [axis] + [i for i in range(nd) if i != axis]
    for a in arrays:
        if a.size:
            st = contig_strides(a.shape)
            a._dev.permute(out._buf[pos * block:], a._buf, [a.shape[i] for i in perm], [st[i] for i in perm], 0, a.dtype)
        pos += a.shape[axis]
    return out if axis == 0 else moveaxis(out, 0, axis)


def stack(arrays, axis=0):
    arrays = [asarray(a) for a in arrays]
    return concatenate([expand_dims(a, axis % (a.ndim + 1)) for a in arrays], axis % (arrays[0].ndim + 1))


def outer(a, b):
    return tensordot(asarray(a).ravel(), asarray(b).ravel(), axes=0)


def kron(a, b):
    """Kronecker product of two arrays of equal rank (numpy.kron for the common 1-d / 2-d cases and beyond)."""
    a, b = asarray(a), asarray(b)
    if a.ndim != b.ndim:
        raise ValueError("kron: operands need the same number of dimensions")
    nd = a.ndim
    y = tensordot(a, b, axes=0)                                    # (a0.., b0..)
    y = y.transpose([i // 2 + (i % 2) * nd for i in range(2 * nd)])  # (a0, b0, a1, b1, ...)
    return y.reshape([a.shape[i] * b.shape[i] for i in range(nd)])


def power(x, p):
    x = asarray(x)
    if isinstance(p, numbers.Integral) and p >= 0:
        out = None
        base, e = x, int(p)
        if e == 0:
            return Array.full(x.shape, 1.0, x.dtype, x._dev)
        while e:                                                     # square-and-multiply on the device
            if e & 1:
                out = base.copy() if out is None else out * base   # (x ** 1 is a new array, as numpy's is)
            e >>= 1
            if e:
                base = base * base
        return out
    if p == 0.5:
        return sqrt(x)
    return exp(log(x) * p)


def square(x):
    x = asarray(x)
    return x * x


def mean(x, axis=None):
    x = asarray(x)
    n = x.size if axis is None else prod([x.shape[a % x.ndim] for a in ((axis,) if isinstance(axis, numbers.Integral) else axis)])
    return sum(x, axis=axis) / n


def vdot(a, b):
    a, b = asarray(a).ravel(), asarray(b).ravel()
    return tensordot(a.conj(), b, axes=([0], [0]))


def squeeze(x, axis=None):
    x = asarray(x)
    if axis is None:
        return x.reshape([d for d in x.shape if d != 1])
    axes = {a % x.ndim for a in ((axis,) if isinstance(axis, numbers.Integral) else axis)}
    return x.reshape([d for i, d in enumerate(x.shape) if i not in axes])


def expand_dims(x, axis):
    x = asarray(x)
    shape = list(x.shape)
    shape.insert(axis % (x.ndim + 1), 1)
    return x.reshape(shape)


# ---- elementwise / reductions ------------------------------------------------
def multiply(a, b):
    if isinstance(a, numbers.Number):
        return asarray(b) * a
    return asarray(a) * b


def add(a, b):
    return asarray(a) + b


def subtract(a, b):
    return asarray(a) - b


def true_divide(a, b):
    return asarray(a) / b


divide = true_divide


def negative(x):
    return -asarray(x)


def conj(x):
    return asarray(x).conj()


conjugate = conj


def real(x):
    return asarray(x).real


def imag(x):
    return asarray(x).imag


def astype(x, dtype):
    return asarray(x).astype(dtype)


def sum(x, axis=None):
    x = asarray(x)
    if axis is None:
        axes = tuple(range(x.ndim))
    elif isinstance(axis, numbers.Integral):
        axes = (int(axis) % x.ndim,)
    else:
        axes = tuple(int(a) % x.ndim for a in axis)
    st = contig_strides(x.shape)
    keep = [i for i in range(x.ndim) if i not in axes]
    out = Array.empty([x.shape[i] for i in keep], x.dtype, x._dev)
    if out.size:
        if x.size == 0:
            x._dev.fill(out._buf, out.size, 0.0, out.dtype)
        else:
            x._dev.reduce_sum(
                out._buf, x._buf, [x.shape[i] for i in keep], [st[i] for i in keep],
                [x.shape[i] for i in axes], [st[i] for i in axes], x.dtype,
            )
    return out


def trace(x):
    x = asarray(x)
    return _einsum_single(x, ("i", "i"), ())


def _unary(x, op):
    if not isinstance(x, Array):          # python / numpy scalars and arrays keep numpy semantics
        return getattr(np, "abs" if op == "abs" else op)(x)
    if x.dtype.kind == "c" and op != "abs":
        raise TypeError(f"quimb_amd.{op}: complex arrays are not supported")
    out = Array.empty(x.shape, _REAL_OF[x.dtype], x._dev)
    if out.size:
        x._dev.unary(out._buf, x._buf, x.size, op, x.dtype)
    return out


def abs(x):  # noqa: A001  (autoray resolves do("abs") by this name)
    """Elementwise magnitude (real result for complex input) -- ``do("abs", x)`` of the
    strip_exponent / norm fallbacks (quimb/tensor/tensor_core.py:330-340, array_ops.py:257-263)."""
    return _unary(x, "abs")


absolute = abs


def sqrt(x):
    return _unary(x, "sqrt")


def exp(x):
    return _unary(x, "exp")


def log(x):
    return _unary(x, "log")


def log10(x):
    return _unary(x, "log10")


def _minmax(x, want_min, axis=None):
    if not isinstance(x, Array):
        return (np.min if want_min else np.max)(x, axis=axis)
    if axis is not None:
        raise NotImplementedError("quimb_amd.max / min reduce over the whole array only")
    if x.dtype.kind == "c":
        raise TypeError("quimb_amd.max / min: complex arrays are not ordered")
    if x.size == 0:
        raise ValueError("zero-size array has no maximum")
    return Array(x._dev, x._dev.minmax(x._buf, x.size, want_min, x.dtype), (), x.dtype)


def max(x, axis=None):  # noqa: A001
    """Maximum over all elements as a 0-d device array (``do("max", do("abs", x))``)."""
    return _minmax(x, False, axis)


def min(x, axis=None):  # noqa: A001
    return _minmax(x, True, axis)


amax, amin = max, min


def diagonal(x, offset=0, axis1=0, axis2=1):
    """numpy.diagonal semantics: the diagonal becomes the LAST axis (one strided copy)."""
    x = asarray(x)
    nd = x.ndim
    axis1, axis2 = axis1 % nd, axis2 % nd
    if axis1 == axis2:
        raise ValueError("axis1 and axis2 cannot be the same")
    st = contig_strides(x.shape)
    d1, d2 = x.shape[axis1], x.shape[axis2]
    if offset >= 0:
        n, off = builtins.max(builtins.min(d1, d2 - offset), 0), offset * st[axis2]
    else:
        n, off = builtins.max(builtins.min(d1 + offset, d2), 0), -offset * st[axis1]
    keep = [a for a in range(nd) if a not in (axis1, axis2)]
    shape = [x.shape[a] for a in keep] + [n]
    strides = [st[a] for a in keep] + [st[axis1] + st[axis2]]
    return x._strided_copy(shape, strides, off if n else 0)


def absmax(x):
    """max |x| as a python float (one device reduction + 8-byte read-back)."""
    x = asarray(x)
    return x._dev.absmax(x._buf, x.size, x.dtype)


def norm_fro(x):
    """Frobenius norm, registered for ``quimb.tensor.array_ops.norm_fro``
    (array_ops.py:257-274)."""
    x = asarray(x)
    v = x.ravel()
    out, _ = einsum_pair(v.conj(), ("i",), v, ("i",), (), True)
    return float(np.sqrt(builtins.abs(out.item())))


def zeros(shape, dtype="float64", **_):
    return Array.full(shape, 0.0, dtype)


def ones(shape, dtype="float64", **_):
    return Array.full(shape, 1.0, dtype)


def eye(n, dtype="float64", **_):
    return Array.from_numpy(np.eye(n), dtype=dtype)


def array(x, dtype=None, **_):
    return asarray(x, dtype=dtype)


def shape(x):
    return asarray(x).shape


def ndim(x):
    return asarray(x).ndim


def size(x):
    return asarray(x).size


def implementation_pair():
    """The ``(tensordot, einsum)`` pair for cotengra's ``implementation=`` injection point:
    ``tn.contract(..., implementation=quimb_amd.implementation_pair())``.  quimb forwards unknown contraction
    kwargs to ``cotengra.array_contract`` untouched (tensor_core.py:293-294, :327; contraction.py:279, :291), and
    cotengra then calls ``tensordot(a, b, axes)`` / ``einsum(eq, a, b)`` for every pairwise step of its tree.
    Host arrays are moved into HBM on first touch and every intermediate stays there (the result is an
    ``Array``)."""
    def _td(a, b, axes=2):
        return tensordot(asarray(a), asarray(b), axes)

    def _es(eq, *xs):
        return einsum(eq, *[asarray(x) for x in xs])

    return _td, _es
