"""CPU oracle: a numpy restatement of the reference's contraction path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``quimb_amd/`` imports this file; only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may -- and there only as the checker / timed baseline, never as the product.

What is restated (file:line under /root/reference unless stated):

* ``gen_output_inds``            quimb/tensor/tensor_core.py:158-170
* ``oracle_array_contract``      the arithmetic of ``ctg.array_contract`` as called at
                                 quimb/tensor/contraction.py:285.  cotengra==0.8.2 (pixi.lock:70)
                                 and autoray==0.10.1 (pixi.lock:67) are NOT vendored in the
                                 reference tree; their published algorithm is restated: walk the
                                 contraction path, one pairwise ``einsum`` (== tensordot + transpose,
                                 or batched matmul for hyper indices) per step, optional
                                 ``strip_exponent`` (after every step divide by max|x| and accumulate
                                 log10), slices summed.
* ``oracle_tensor_contract``     quimb/tensor/tensor_core.py:224-358 (output-index inference,
                                 scalar unwrapping, tag union, exponent handling)
* ``oracle_fuse``                quimb/tensor/array_ops.py:95-182
* ``oracle_contract_boundary_2d`` TensorNetwork2D.contract_boundary quimb/tensor/tn2d/core.py:2502-2642 ->
                                 _contract_interleaved_boundary_sequence :2322-2500 -> _contract_boundary_core
                                 :1355-1484 (absorb a row, QR gauge sweep backwards, truncated-SVD sweep
                                 forwards; ``tensor_split`` cutoff rule tensor_core.py:400, decomp.py:759-760).
                                 PINNED against the real quimb's own values (tests/golden/boundary.npz,
                                 agreement 1e-14) and the 16x16 Ising known-answer test at chi=8.
* builders                       TN2D_from_fill_fn quimb/tensor/tensor_builder.py:1345-1369 (index order
                                 l, r, u, d); classical Ising tensors tensor_builder.py:2318-2466 and
                                 TN2D_classical_ising_partition_function :2687-2811; MPS order l, r, p
                                 quimb/tensor/tn1d/core.py:1881-1901

Pinning: the restatement is checked in ``tests/test_oracle.py`` against (a) the
reference's only fixed-number known-answer test on this path -- the 16x16 Ising
partition function 8.459419593253275e100
(tests/test_tensor/test_tn2d/test_core.py:309-335) -- via an independent exact
transfer-matrix evaluation, (b) ``numpy.einsum`` over the whole network, and (c)
golden vectors produced by importing the real quimb from /root/reference
(``tests/golden/make_golden.py``).
"""

import itertools
import math

import numpy as np

_SYMS = [chr(c) for c in range(ord("a"), ord("z") + 1)] + [chr(c) for c in range(ord("A"), ord("Z") + 1)]
_SYMS += [chr(c) for c in range(192, 192 + 5000)]


def gen_output_inds(all_inds):
    from collections import Counter

    freq = Counter(all_inds)
    out = []
    for ind, f in freq.items():
        if f > 2:
            raise ValueError(f"The index {ind} appears more than twice!")
        if f == 1:
            out.append(ind)
    return tuple(out)


def _pair_result_inds(ia, ib, others, output):
    keep = []
    rest = set(output)
    for t in others:
        rest.update(t)
    for ix in tuple(ia) + tuple(ib):
        if ix in rest and ix not in keep:
            keep.append(ix)
    return tuple(keep)


def _einsum_inds(arrays, inds_list, out_inds):
    sym = {}
    for t in list(inds_list) + [out_inds]:
        for ix in t:
            if ix not in sym:
                sym[ix] = _SYMS[len(sym)]
    eq = ",".join("".join(sym[ix] for ix in t) for t in inds_list) + "->" + "".join(sym[ix] for ix in out_inds)
    return np.einsum(eq, *arrays)


def _pair_contract(ops, tis, keep):
    """One pairwise step the way cotengra's executor issues it: ``tensordot``
    (transpose-copies + BLAS gemm inside numpy) when there are no batch / hyper
    indices, ``einsum`` otherwise.  Returns (array, its index order)."""
    if len(ops) == 2:
        ia, ib = tis
        sa, sb, sk = set(ia), set(ib), set(keep)
        shared = [ix for ix in ia if ix in sb]
        simple = len(sa) == len(ia) and len(sb) == len(ib) and not any(ix in sk for ix in shared)
        simple = simple and all(ix in sk for ix in ia if ix not in sb) and all(ix in sk for ix in ib if ix not in sa)
        if simple:
            ax_a = [ia.index(ix) for ix in shared]
            ax_b = [ib.index(ix) for ix in shared]
            x = np.tensordot(ops[0], ops[1], axes=(ax_a, ax_b))
            order = tuple(ix for ix in ia if ix not in sb) + tuple(ix for ix in ib if ix not in sa)
            return x, order
    return _einsum_inds(ops, tis, keep), tuple(keep)


def _refuse_if_huge(ops, tis, keep):
    import os

    dims = {ix: d for t, a in zip(tis, ops) for ix, d in zip(t, np.shape(a))}
    n = 1
    for ix in keep:
        n *= dims[ix]
    nbytes = n * max(np.asarray(a).dtype.itemsize for a in ops)
    limit = int(os.environ.get("QAMD_ORACLE_MAX_BYTES", MAX_INTERMEDIATE_BYTES))
    if nbytes > limit:
        raise MemoryError(f"oracle: this path builds an intermediate of {nbytes / 2**30:.1f} GiB (> {limit / 2**30:.1f} GiB); "
                          f"pass an explicit path (e.g. a lattice sweep) or raise QAMD_ORACLE_MAX_BYTES")


def oracle_contract_core(arrays, inputs, output, path, strip_exponent=False):
    """Pairwise evaluation along a linear (opt_einsum style) path."""
    arrays = list(arrays)
    inputs = [tuple(t) for t in inputs]
    exponent = 0.0
    for con in path:
        con = tuple(sorted(con, reverse=True))
        ops = [arrays.pop(p) for p in con][::-1]
        tis = [inputs.pop(p) for p in con][::-1]
        if len(ops) == 1:
            keep = _pair_result_inds(tis[0], (), inputs, output)
        else:
            keep = _pair_result_inds(tis[0], tis[1], inputs, output)
        _refuse_if_huge(ops, tis, keep)
        x, keep = _pair_contract(ops, tis, keep)
        if strip_exponent:
            f = np.max(np.abs(x))
            if f > 0:
                x = x / f
                exponent += math.log10(f)
        arrays.append(x)
        inputs.append(keep)
    if len(arrays) != 1:
        # leftover disconnected pieces: multiply them out
        x = _einsum_inds(arrays, inputs, tuple(output))
    else:
        x = _einsum_inds(arrays, inputs, tuple(output))
    return (x, exponent) if strip_exponent else x


def naive_path(n):
    return [(0, 1)] * (n - 1)


#: a single intermediate larger than this many bytes is refused (MemoryError) instead of attempted: a path that builds a
#: 6^14-element tensor on the way to a scalar is a mistake of the caller, and on a memory-limited box the attempt takes
#: the whole container down with it (round 3 / 4: an un-pathed 6x6 D=6 call -- 627 GB -- lost four GPU boxes).  The
#: full-size 10x10 D=6 sweep needs 6^11 doubles = 2.9 GB; override with QAMD_ORACLE_MAX_BYTES.
MAX_INTERMEDIATE_BYTES = 8 << 30


def small_first_path(inputs, output, size_dict):
    """Default path when the caller names none: repeatedly contract the pair of tensors SHARING an index whose
    result is smallest (ties: first in list order), outer products only once nothing shares an index -- a greedy
    "smallest intermediate first" order in opt_einsum's linear format.  The value of a network does not depend on the
    path (a sum of products), so any order is the reference's arithmetic; this one keeps un-pathed test calls cheap."""
    live = [tuple(t) for t in inputs]
    out = set(output)
    path = []
    vol = lambda t: math.prod(size_dict[ix] for ix in t)
    while len(live) > 1:
        holders = {}
        for pos, t in enumerate(live):
            for ix in set(t):
                holders.setdefault(ix, []).append(pos)
        pairs = {(i, j) for hs in holders.values() for a, i in enumerate(hs) for j in hs[a + 1:]}
        if not pairs:                                   # disconnected pieces: outer product of the two smallest
            i, j = sorted(sorted(range(len(live)), key=lambda q: (vol(live[q]), q))[:2])
            pairs = {(i, j)}
        best = None
        for i, j in sorted(pairs):
            shared = set(live[i]) & set(live[j])
            keep = tuple(ix for ix in dict.fromkeys(live[i] + live[j])
                         if ix in out or len(holders[ix]) > (2 if ix in shared else 1))
            key = (vol(keep), i, j)
            if best is None or key < best[0]:
                best = (key, i, j, keep)
        _, i, j, keep = best
        path.append((i, j))
        live.pop(j)
        live.pop(i)
        live.append(keep)
    return path


def oracle_array_contract(arrays, inputs, output=None, path=None, strip_exponent=False, sliced_inds=(),
                          size_dict=None, dtype=None):
    """The reference arithmetic: numpy pairwise contraction in path order,
    optionally sliced (slices are summed) and exponent-stripped."""
    arrays = [np.asarray(a) if dtype is None else np.asarray(a, dtype=dtype) for a in arrays]
    inputs = [tuple(t) for t in inputs]
    if output is None:
        output = gen_output_inds(ix for t in inputs for ix in t)
    if path is None:
        sd = {ix: d for t, a in zip(inputs, arrays) for ix, d in zip(t, np.shape(a))}
        path = small_first_path([tuple(ix for ix in t if ix not in sliced_inds) for t in inputs], output, sd)
    if not sliced_inds:
        return oracle_contract_core(arrays, inputs, output, path, strip_exponent)
    if size_dict is None:
        size_dict = {ix: d for t, a in zip(inputs, arrays) for ix, d in zip(t, a.shape)}
    total, total_e = None, None
    for vals in itertools.product(*[range(size_dict[ix]) for ix in sliced_inds]):
        fix = dict(zip(sliced_inds, vals))
        arrs, ins = [], []
        for a, t in zip(arrays, inputs):
            key = tuple(fix[ix] if ix in fix else slice(None) for ix in t)
            arrs.append(a[key])
            ins.append(tuple(ix for ix in t if ix not in fix))
        r = oracle_contract_core(arrs, ins, output, path, strip_exponent)
        if strip_exponent:
            x, e = r
            if total is None:
                total, total_e = x, e
            else:
                en = max(total_e, e)
                total = total * 10 ** (total_e - en) + x * 10 ** (e - en)
                total_e = en
        else:
            total = r if total is None else total + r
    return (total, total_e) if strip_exponent else total


def realify_scalar(x, imag_tol=1e-12):
    if isinstance(x, complex):
        return x.real if abs(x.imag) < abs(x.real) * imag_tol else x
    return x


def oracle_tensor_contract(tensors, output_inds=None, path=None, strip_exponent=False, exponent=None,
                           preserve_tensor=False):
    """``tensors``: sequence of (data, inds, tags).  Returns scalar or
    (data, inds, tags); with ``strip_exponent`` a pair (result, exponent)."""
    arrays = [np.asarray(t[0]) for t in tensors]
    inds = [tuple(t[1]) for t in tensors]
    if output_inds is None:
        out = gen_output_inds(ix for t in inds for ix in t)
    else:
        out = tuple(output_inds)
    data = oracle_array_contract(arrays, inds, out, path, strip_exponent)
    e = None
    if strip_exponent:
        data, e = data
        if exponent is not None:
            e = e + exponent
    elif exponent is not None:
        data = data * 10**exponent
    if not out and not preserve_tensor:
        res = realify_scalar(np.asarray(data).item())
    else:
        tags = tuple(dict.fromkeys(tg for t in tensors for tg in (t[2] if len(t) > 2 and t[2] else ())))
        res = (data, out, tags)
    return (res, e) if strip_exponent else res


def oracle_fuse(x, *axes_groups):
    x = np.asarray(x)
    groups = tuple(tuple(g) for g in axes_groups)
    if not any(groups):
        return x
    ndim = x.ndim
    ax2g = {ax: g for g, axes in enumerate(groups) for ax in axes}
    pos = min(a for g in groups for a in g)
    before = [ax for ax in range(pos) if ax not in ax2g]
    after = [ax for ax in range(pos, ndim) if ax not in ax2g]
    perm = before + [ax for g in groups for ax in g] + after
    new_shape = [x.shape[a] for a in before] + [int(np.prod([x.shape[a] for a in g])) for g in groups]
    new_shape += [x.shape[a] for a in after]
    return np.transpose(x, perm).reshape(new_shape)


# ---------------------------------------------------------------------------
# builders (restated input generators)
# ---------------------------------------------------------------------------
def tn2d_inds(Lx, Ly):
    """Index labels of an open Lx x Ly lattice, per site in row-major order,
    each in the reference's l, r, u, d order (u = towards row i+1)."""

    def bond(a, b):
        a, b = sorted((a, b))
        return ("b", a, b)

    inputs = []
    for i, j in itertools.product(range(Lx), range(Ly)):
        inds = []
        if j > 0:
            inds.append(bond((i, j), (i, j - 1)))
        if j < Ly - 1:
            inds.append(bond((i, j), (i, j + 1)))
        if i < Lx - 1:
            inds.append(bond((i, j), (i + 1, j)))
        if i > 0:
            inds.append(bond((i, j), (i - 1, j)))
        inputs.append(tuple(inds))
    return inputs


def tn2d_from_fill_fn(fill_fn, Lx, Ly, D):
    inputs = tn2d_inds(Lx, Ly)
    arrays = [fill_fn((D,) * len(t)) for t in inputs]
    return arrays, inputs


def tn2d_rand(Lx, Ly, D, seed=0, low=-0.1, high=1.0, dtype="float32", normalize=True):
    """'Mostly positive' uniform fill as the reference's own 2D accuracy tests use
    (tests/test_tensor/test_tn2d/test_core.py:243-247).  ``normalize`` rescales
    every tensor by 1/(mean * D^(deg/2)) so the network value stays O(1)-ish in
    fp32 instead of ~1e105 (SURVEY.md section 7 'fp32 dynamic range')."""
    rng = np.random.default_rng(seed)
    mean = 0.5 * (low + high)

    def fill(shape):
        x = rng.uniform(low, high, size=shape)
        if normalize:
            x = x / (mean * D ** (len(shape) / 2.0))
        return x.astype(dtype)

    return tn2d_from_fill_fn(fill, Lx, Ly, D)


def ising_sqrtS(beta, j=1.0):
    c, s = math.cosh(j * beta) ** 0.5, math.sinh(j * beta) ** 0.5
    return np.array([[c + s, c - s], [c - s, c + s]]) / 2**0.5


def ising_T(beta, ndir, h=0.0):
    """T[d1..dn] = sum_i prod_k sqrtS[i, d_k] * H[i]"""
    S = ising_sqrtS(beta)
    H = np.array([math.exp(-beta * h), math.exp(beta * h)])
    syms = "abcdefgh"[:ndir]
    eq = ",".join("i" + s for s in syms) + ",i->" + syms
    return np.einsum(eq, *([S] * ndir), H)


def tn2d_classical_ising(Lx, Ly, beta, h=0.0):
    inputs = tn2d_inds(Lx, Ly)
    arrays = [ising_T(beta, len(t), h) for t in inputs]
    return arrays, inputs


def ising_partition_exact(Lx, Ly, beta, j=1.0):
    """Exact open-boundary 2D Ising partition function by row transfer over the
    2^Ly row configurations -- independent of any tensor-network code."""
    n = 1 << Ly
    spins = 1 - 2 * ((np.arange(n)[:, None] >> np.arange(Ly)[None, :]) & 1)  # (n, Ly) in {+1,-1}
    horiz = np.exp(beta * j * np.sum(spins[:, :-1] * spins[:, 1:], axis=1))  # within-row bonds
    w = np.array([[math.exp(beta * j), math.exp(-beta * j)], [math.exp(-beta * j), math.exp(beta * j)]])
    v = horiz.astype(np.float64).copy()
    log_scale = 0.0
    for _ in range(Lx - 1):
        t = v.reshape((2,) * Ly)
        for ax in range(Ly):
            t = np.moveaxis(np.tensordot(w, t, axes=([1], [ax])), 0, ax)
        v = t.reshape(n) * horiz
        m = v.max()
        v /= m
        log_scale += math.log10(m)
    return float(v.sum()) * 10**log_scale if log_scale < 300 else (float(v.sum()), log_scale)


def mps_rand(L, chi, d=2, seed=0, dtype="float64"):
    """Open-boundary MPS tensors in the reference's l, r, p index order."""
    rng = np.random.default_rng(seed)
    arrays, inputs = [], []
    for i in range(L):
        inds, shape = [], []
        if i > 0:
            inds.append(("b", i - 1)); shape.append(chi)
        if i < L - 1:
            inds.append(("b", i)); shape.append(chi)
        inds.append(("k", i)); shape.append(d)
        x = rng.normal(size=shape)
        x = x / np.linalg.norm(x) ** (1.5 / x.ndim)
        arrays.append(x.astype(dtype))
        inputs.append(tuple(inds))
    return arrays, inputs


# ---------------------------------------------------------------------------------------------------------
# boundary contraction of a 2D network (TensorNetwork2D.contract_boundary, quimb/tensor/tn2d/core.py:2502-2642;
# _contract_interleaved_boundary_sequence :2322-2500; _contract_boundary_core :1355-1484)
# ---------------------------------------------------------------------------------------------------------
def _svals_to_keep(s, max_bond, cutoff):
    """``tensor_split``'s default ``cutoff_mode="rel"`` (quimb/tensor/tensor_core.py:400): keep s_i > cutoff * s_0
    (quimb/tensor/decomp.py:759-760, :908-909), at least one, then the ``max_bond`` cap (:990-1001)."""
    s = np.abs(np.asarray(s, dtype=np.float64))
    n = max(int(np.sum(s > cutoff * s[0])), 1) if cutoff > 0.0 else len(s)
    if max_bond is not None and max_bond > 0:
        n = min(n, max_bond)
    return n


def oracle_contract_boundary_2d(arrays, Lx, Ly, max_bond=None, cutoff=1e-10, canonize=True, sequence=None):
    """Restated in the reference's own orientation: per absorbed line, contract each boundary tensor with
    the site next to it (``contract_((tag1, tag2))``, :1393), then ``canonize_plane`` sweeping from the last
    column back to the first (:1455-1466, QR, absorb towards the sweep direction) and ``compress_plane``
    sweeping forwards (:1472-1484, truncated SVD, ``absorb="right"``).  Sides alternate xmin, xmax (ymin,
    ymax when Lx < Ly, :2398-2409) until adjacent; the rest is contracted exactly.  float64 throughout.
    Returns (mantissa, exponent) with value = mantissa * 10**exponent."""
    sequence = tuple(sequence) if sequence is not None else ("xmin", "xmax")
    g = [[None] * Ly for _ in range(Lx)]
    for i in range(Lx):
        for j in range(Ly):
            a = np.asarray(arrays[i * Ly + j], dtype=np.float64 if np.asarray(arrays[0]).dtype.kind != "c" else np.complex128)
            have = (j > 0, j < Ly - 1, i < Lx - 1, i > 0)
            it = iter(a.shape)
            g[i][j] = a.reshape([next(it) if h else 1 for h in have])       # l r u d
    if Lx < Ly:    # sweep over columns: relabel so that the code below always sweeps rows
        g = [[g[i][j].transpose(3, 2, 1, 0) for i in range(Lx)] for j in range(Ly)]
        Lx, Ly = Ly, Lx

    def compress_line(t):
        # t[j]: (l, r, x) with x the inward leg
        if canonize:
            for j in range(Ly - 1, 0, -1):                                   # QR from the far end backwards
                l, r, x = t[j].shape
                q, rr = np.linalg.qr(t[j].transpose(1, 2, 0).reshape(r * x, l))   # (r x, k) (k, l)
                k = q.shape[1]
                t[j] = q.reshape(r, x, k).transpose(2, 0, 1)
                t[j - 1] = np.einsum("lrx,kr->lkx", t[j - 1], rr)
        for j in range(Ly - 1):                                               # truncated SVD forwards
            l, r, x = t[j].shape
            u, s, vh = np.linalg.svd(t[j].transpose(0, 2, 1).reshape(l * x, r), full_matrices=False)
            k = _svals_to_keep(s, max_bond, cutoff)
            t[j] = u[:, :k].reshape(l, x, k).transpose(0, 2, 1)
            t[j + 1] = np.einsum("kl,lrx->krx", s[:k, None] * vh[:k], t[j + 1])

    lo = [w[:, :, :, 0] for w in g[0]]                   # (l, r, u)
    hi = [w[:, :, 0, :] for w in g[-1]]                  # (l, r, d)
    ilo, ihi, turn = 0, Lx - 1, 0
    exponent = 0.0
    truncate = (max_bond is not None and max_bond > 0) or cutoff > 0.0
    while ihi - ilo > 1:
        side = sequence[turn % len(sequence)]
        turn += 1
        if side == "xmin":
            ilo += 1
            line = lo
            for j in range(Ly):
                y = np.einsum("LRx,lrux->LlRru", line[j], g[ilo][j])
                line[j] = y.reshape(y.shape[0] * y.shape[1], y.shape[2] * y.shape[3], y.shape[4])
        else:
            ihi -= 1
            line = hi
            for j in range(Ly):
                y = np.einsum("LRx,lrxd->LlRrd", line[j], g[ihi][j])
                line[j] = y.reshape(y.shape[0] * y.shape[1], y.shape[2] * y.shape[3], y.shape[4])
        if truncate:
            compress_line(line)
        for j in range(Ly):                               # keep every tensor O(1): value is unchanged
            nrm = np.linalg.norm(line[j])
            if nrm > 0:
                line[j] = line[j] / nrm
                exponent += math.log10(nrm)
    env = np.ones((1, 1), dtype=lo[0].dtype)
    for b, t in zip(lo, hi):
        env = np.einsum("ab,acx,bdx->cd", env, b, t)
        nrm = np.linalg.norm(env)
        if nrm > 0:
            env = env / nrm
            exponent += math.log10(nrm)
    return env.reshape(()).item(), exponent
