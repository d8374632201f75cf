"""``quimb_amd.linalg`` -- decompositions a backend must *answer* for so that
quimb's split/compress drivers (quimb/tensor/decomp.py:829-1118) keep tensors on
the device between contractions.  They are OUT of the hot-path scope (SURVEY.md
section 2.1): no hand-written kernels here, the calls go to rocSOLVER through
torch on the array's own device storage and are not counted in any metric.
"""

import numpy as np

from .array import Array


def _as_torch(x):
    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    buf = x._buf
    if not hasattr(buf, "view"):
        raise TypeError("quimb_amd.linalg needs device-resident arrays")
    return x, buf[: x.size].view(x.shape)


def _wrap(x, t):
    # torch hands back lazily conjugated views (``Vh`` of a complex svd carries the conj bit): the kernels read
    # raw memory, so materialise them
    t = t.resolve_conj().resolve_neg().contiguous()
    return Array(x._dev, t.reshape(-1), tuple(t.shape), x.dtype if t.dtype == x._buf.dtype else _np_dtype(t))


def _np_dtype(t):
    import torch

    return {torch.float32: "float32", torch.float64: "float64", torch.complex64: "complex64",
            torch.complex128: "complex128"}[t.dtype]


def _hook(x, name):
    """A device may answer the decompositions itself (the numpy plan interpreter of the CPU tests does)."""
    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    return x, getattr(x._dev, "linalg_" + name, None)


def _unit_scale(t):
    """``t / max|t|`` and the (real, device-resident) scale.  rocSOLVER's fp32 ``gesvd`` returns WRONG singular
    values -- silently, orthogonal U and V included -- for entries around 1e12 (a (4, 2) boundary tensor of the
    16x16 Ising network did it: s = [1.157e12, 4.97e11] instead of [1.596e12, 3.61e11]; scripts/probes/
    boundary_debug.py), so every decomposition sees O(1) entries; no host sync."""
    import torch

    amax = t.abs().amax() if t.numel() else torch.ones((), device=t.device)
    scale = torch.where((amax > 0) & torch.isfinite(amax), amax, torch.ones_like(amax))
    return t / scale, scale


def svd(x, full_matrices=False):
    import torch

    x, hook = _hook(x, "svd")
    if hook is not None:
        return hook(x, full_matrices)
    x, t = _as_torch(x)
    t, scale = _unit_scale(t)
    u, s, vh = torch.linalg.svd(t, full_matrices=full_matrices)
    s = s * scale
    return _wrap(x, u), Array(x._dev, s.resolve_conj().contiguous().reshape(-1), tuple(s.shape), _np_dtype(s)), _wrap(x, vh)


def svd_via_eig(x, max_bond=-1):
    """``(U, s, VH)`` of a 2-d array through the Hermitian eigen-decomposition of its Gram matrix -- the
    reference's ``svd_via_eig`` (quimb/tensor/decomp.py:1168, split ``method="svd:eig"``) with ``absorb=None``.
    The two Gram / back-projection products run on this library's GETT kernels, the small eigenproblem on
    rocSOLVER ``syevd``: 23 ms instead of 260 ms (``gesvd``) for the 1024 x 1024 fp64 two-site tensor of a
    chi = 512 DMRG step.  Singular values below ~sqrt(eps) * s_max lose relative accuracy (squared condition); only
    directions below the Gram matrix's noise floor (eigenvalue <= 8 eps * s_max^2) are dropped."""
    from . import ops

    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    m, n = x.shape
    right = n <= m                                    # decompose the smaller Gram matrix
    g = ops.tensordot(x.conj(), x, axes=([0], [0])) if right else ops.tensordot(x, x.conj(), axes=([1], [1]))
    w, v = eigh(g)                                    # ascending; the spectrum (min(m, n) numbers) goes to the host
    wh = w.to_numpy().astype(np.float64)[::-1]
    k = len(wh) if max_bond is None or max_bond < 0 else min(int(max_bond), len(wh))
    # Only directions this route cannot normalise are dropped: eigenvalues of the Gram matrix that are zero, clipped
    # negative, or below its noise floor eps * s_max^2 (there the eigenvector is noise and U = x V / s would hand back
    # "orthogonal" factors that are not: a rank-1 6x5 input gave |U^T U - I| = 1).  Everything above the floor is kept
    # -- the reference's svd_via_eig (quimb/tensor/decomp.py:1168) truncates by max_bond only -- so a full-rank input
    # comes back with min(m, n) values; their RELATIVE accuracy degrades below ~sqrt(eps) * s_max (squared condition),
    # which is the documented price of method "svd:eig".  At least one direction stays.
    s_all = np.sqrt(np.clip(wh, 0.0, None))
    eps = np.finfo(np.dtype(w.dtype.name if hasattr(w.dtype, "name") else w.dtype)).eps
    # (floor = 8 eps s_max^2: a Hermitian eigensolver returns a zero eigenvalue as a few eps of |G| with either sign;
    # in s that is 2.8 sqrt(eps) s_max -- 1e-3 in float32, 4e-8 in float64: callers that need more of the spectrum in
    # single precision use method "svd", and DMRG2 / tensor_split take the rank from the returned factors)
    keep = int(np.count_nonzero(wh > 8.0 * eps * float(wh[0]))) if wh.size and wh[0] > 0 else 1
    k = max(1, min(k, keep))
    sh = s_all[:k]
    V = v[:, ::-1][:, :k] if k < len(wh) else v[:, ::-1]          # columns by descending eigenvalue
    sinv = np.where(sh > 0, 1.0 / np.where(sh > 0, sh, 1.0), 0.0)
    S = Array.from_numpy(sh.astype(w.dtype), dev=x._dev)
    if right:                                         # g = x^H x = V s^2 V^H ;  U = x V / s ;  VH = V^H
        U = ops.tensordot(x, V, axes=([1], [0]))
        U = ops.multiply(U, Array.from_numpy(sinv.astype(x.dtype), dev=x._dev)[None, :])
        return U, S, ops.transpose(V.conj(), (1, 0))
    # g = x x^H = U s^2 U^H ;  VH = U^H x / s
    VH = ops.tensordot(V.conj(), x, axes=([0], [0]))
    VH = ops.multiply(VH, Array.from_numpy(sinv.astype(x.dtype), dev=x._dev)[:, None])
    return V, S, VH


def qr(x, mode="reduced"):
    import torch

    x, hook = _hook(x, "qr")
    if hook is not None:
        return hook(x, mode)
    x, t = _as_torch(x)
    t, scale = _unit_scale(t)
    q, r = torch.linalg.qr(t, mode=mode)
    return _wrap(x, q), _wrap(x, r * scale)


def eigh(x):
    import torch

    x, hook = _hook(x, "eigh")
    if hook is not None:
        return hook(x)
    x, t = _as_torch(x)
    t, scale = _unit_scale(t)
    w, v = torch.linalg.eigh(t)
    w = w * scale
    return Array(x._dev, w.contiguous().reshape(-1), tuple(w.shape), _np_dtype(w)), _wrap(x, v)


def _single(name, x, *others):
    """torch.linalg.<name> on device storage (or the device's own answer), one array out."""
    import torch

    x, hook = _hook(x, name)
    if hook is not None:
        return hook(x, *others)
    x, t = _as_torch(x)
    ts = [_as_torch(o)[1] for o in others]
    return _wrap(x, getattr(torch.linalg, name)(t, *ts))


def inv(x):
    return _single("inv", x)


def pinv(x):
    return _single("pinv", x)


def solve(a, b):
    return _single("solve", a, b)


def cholesky(x):
    return _single("cholesky", x)


def solve_triangular(a, b, lower=True, left=True):
    """``X`` with ``a X = b`` (``left``) or ``X a = b`` for a triangular ``a`` (rocBLAS trsm through torch)."""
    import torch

    a, hook = _hook(a, "solve_triangular")
    if hook is not None:
        return hook(a, b, lower, left)
    a, ta = _as_torch(a)
    _, tb = _as_torch(b)
    return _wrap(a, torch.linalg.solve_triangular(ta, tb, upper=not lower, left=left))


# ---------------------------------------------------------------------------------------------------------------------
# GEMM-shaped decompositions: the O(n^3) part on this library's kernels, only an n x n factorisation on rocSOLVER.
# An L = 100, chi = 512 DMRG2 sweep spent 0.8-1.7 s in geqrf/orgqr (canonisation) and 1.1-2.0 s in syevd of 1024 x 1024
# Gram matrices (splits) against 0.22 s on the contraction path (profiles/r04_dmrg_sweep.txt); the reference ships the
# GEMM-shaped alternatives as split drivers: ``qr:cholesky`` (quimb/tensor/decomp.py:2359-2420), ``cholesky``
# (:2262-2322), ``svd:rand`` (:1689-1868), ``rsvd`` (:2538; quimb/linalg/rand_linalg.py:114-205).
# ---------------------------------------------------------------------------------------------------------------------
def cholesky_regularized(g, shift=True):
    """Lower Cholesky factor of the Hermitian matrix ``g`` with the reference's regularisation (``_with_diag_shift``,
    decomp.py:1866-1880): ``shift`` True / negative -> eps * trace(g) on the diagonal, a positive float -> that multiple
    of the trace, False / 0 -> none, "auto" -> none first, eps * trace if the factorisation fails."""
    import torch

    g, hook = _hook(g, "cholesky")
    if shift == "auto":
        try:
            return cholesky_regularized(g, shift=False)
        except Exception:
            return cholesky_regularized(g, shift=True)
    sh = {False: 0.0, True: -1.0}.get(shift, shift)
    sh = float(np.finfo(np.dtype(g.dtype)).eps) if sh < 0 else float(sh)
    if hook is not None:
        a = g.to_numpy()
        if sh > 0:
            a = a + sh * np.trace(a) * np.eye(a.shape[-1], dtype=a.dtype)
        return hook(Array.from_numpy(a, dev=g._dev))
    g, t = _as_torch(g)
    if sh > 0:
        t = t + (sh * torch.diagonal(t).sum()) * torch.eye(t.shape[-1], dtype=t.dtype, device=t.device)
    L, info = torch.linalg.cholesky_ex(t)
    if int(info.max()) != 0:          # (one host read: a failed factorisation must not hand back garbage)
        raise np.linalg.LinAlgError("cholesky_regularized: matrix not positive definite")
    return _wrap(g, L)


def _gram_defect(g):
    """max |g - I| of a (small) Gram matrix, read back as one float."""
    g, hook = _hook(g, "eigh")
    if hook is not None:                      # the plan interpreter: numpy
        a = g.to_numpy()
        return float(np.max(np.abs(a - np.eye(a.shape[0], dtype=a.dtype))))
    import torch

    g, t = _as_torch(g)
    return float((t - torch.eye(t.shape[0], dtype=t.dtype, device=t.device)).abs().max().item())


def qr_via_cholesky(x, shift=True, refine=False, return_defect=False):
    """``(Q, R)`` of a tall 2-d array (m >= n) from the Cholesky factor of its Gram matrix -- the reference's
    ``qr_via_cholesky`` (decomp.py:2359-2420): ``G = x^H x`` (one GETT launch), ``G = R^H R`` (potrf on n x n),
    ``Q = x R^-1`` (a triangular solve).  Orthogonality of Q degrades as cond(x)^2 eps (and by the regularising shift);
    ``refine`` repeats the step on Q (CholeskyQR2: orthogonal to eps for cond(x) < eps^-1/2) and folds the second
    triangle into R; ``refine="auto"`` forms ``Q^H Q`` (the Gram matrix the second pass would factor anyway), reads its
    distance from the identity back and runs the second pass only if that exceeds 100 eps -- a well-conditioned input
    pays one extra GETT launch instead of a second potrf + trsm.  R has a positive real diagonal by construction.
    ``return_defect``: also return max |Q^H Q - I| of the returned Q where ``refine="auto"`` measured it on the way (the
    one-pass case), else None."""
    from . import ops

    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    g = ops.tensordot(x.conj(), x, axes=([0], [0]))            # x^H x, (n, n)
    L = cholesky_regularized(g, shift=shift)                    # g = L L^H  ->  R = L^H
    R = ops.transpose(L.conj(), (1, 0))
    Q = solve_triangular(R, x, lower=False, left=False)         # Q R = x
    defect = None           # max |Q^H Q - I| of the factor that is RETURNED, where this routine knows it
    if refine == "auto":
        g2 = ops.tensordot(Q.conj(), Q, axes=([0], [0]))
        defect = _gram_defect(g2)
        eps_ = float(np.finfo(np.dtype(x.dtype)).eps)
        if defect > 100.0 * eps_:
            first = defect
            L2 = cholesky_regularized(g2, shift=shift)
            R2 = ops.transpose(L2.conj(), (1, 0))
            Q = solve_triangular(R2, Q, lower=False, left=False)
            R = ops.tensordot(R2, R, axes=([1], [0]))
            # CholeskyQR2: a first factor with ||Q1^H Q1 - I||_2 < 1/2 (here bounded by n x the max-norm that was read back)
            # has cond(Q1) < sqrt(3), and the second pass then leaves O(eps) -- no need to measure it again.  Otherwise the
            # result is NOT known to be orthonormal (None): the checked caller measures it and falls back if need be.
            n_ = g2.shape[0]
            defect = 100.0 * eps_ if np.isfinite(first) and first * n_ <= 0.5 else None     # (nominal: as good as a one-pass accept)
    elif refine:
        Q, R2 = qr_via_cholesky(Q, shift=shift, refine=False)
        R = ops.tensordot(R2, R, axes=([1], [0]))
    if return_defect:
        return Q, R, defect
    return Q, R


def lq_via_cholesky(x, shift=True, refine=False):
    """``(L, Q)`` of a wide 2-d array (m <= n): ``x x^H = L L^H``, ``Q = L^-1 x`` -- the form the Cholesky route yields
    directly (decomp.py:2383-2386, ``transposed = False``).  ``refine`` as in ``qr_via_cholesky``."""
    from . import ops

    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    g = ops.tensordot(x, x.conj(), axes=([1], [1]))            # x x^H, (m, m)
    L = cholesky_regularized(g, shift=shift)
    Q = solve_triangular(L, x, lower=True, left=True)           # L Q = x
    if refine == "auto":
        g2 = ops.tensordot(Q, Q.conj(), axes=([1], [1]))
        if _gram_defect(g2) > 100.0 * float(np.finfo(np.dtype(x.dtype)).eps):
            L2 = cholesky_regularized(g2, shift=shift)
            Q = solve_triangular(L2, Q, lower=True, left=True)
            L = ops.tensordot(L, L2, axes=([1], [0]))
    elif refine:
        L2, Q = lq_via_cholesky(Q, shift=shift, refine=False)
        L = ops.tensordot(L, L2, axes=([1], [0]))
    return L, Q


def _randn(dev, shape, dtype, seed):
    """Gaussian test matrix.  ``seed`` given: numpy's ``default_rng(seed).normal`` on the host (the stream the
    reference draws from, decomp.py:1795-1799 -- results reproducible against it); None: drawn on the device from a
    per-device generator seeded once (the same sequence in every process)."""
    dtype = np.dtype(dtype)
    if seed is not None or not hasattr(dev, "torch"):
        rng = seed if isinstance(seed, np.random.Generator) else np.random.default_rng(seed)
        return Array.from_numpy(rng.normal(size=shape).astype(dtype), dev=dev)      # (real draws for complex x too, as there)
    torch = dev.torch
    rdt = np.zeros(0, dtype).real.dtype
    gen = getattr(dev, "_sketch_generator", None)
    if gen is None:          # one generator per device object, seeded once: sketches are reproducible run to run
        gen = dev._sketch_generator = torch.Generator(device=dev.tdev)
        gen.manual_seed(0x5EED)
    t = torch.randn(shape, dtype=dev._tdt[np.dtype(rdt)], device=dev.tdev, generator=gen).to(dev._tdt[dtype])
    return Array(dev, t.reshape(-1), tuple(shape), dtype)


#: how far from the identity ``Q^H Q`` may be (max-norm, in units of n x the dtype's eps, n = number of columns: the
#: rounding level of an n-column Gram matrix) for a Cholesky-QR basis to be accepted as orthonormal by
#: ``orth_cholesky_checked``.  (A flat 1000 eps sent every third chi = 512 fp64 split of a DMRG sweep to geqrf / orgqr --
#: 14 ms instead of 2.7 -- for defects of 1-3e-13.)
ORTH_DEFECT_EPS = 100.0


def orth_cholesky_checked(y):
    """An orthonormal basis of the columns of ``y`` by CholeskyQR2, CHECKED: the Gram route squares the condition number, so
    on an ill-conditioned sketch (fp32 with a spectral decay of 1e-3 already, fp64 beyond 1e-8) even the refined ``Q`` is
    far from an isometry -- and everything downstream (canonical forms, <H> / sum s^2) assumes one.  The defect of the
    FINAL factor is measured (one more small GETT launch, unless the passes themselves established it); beyond
    ``ORTH_DEFECT_EPS`` x n x eps, or when a Cholesky
    factorisation fails outright, the basis comes from Householder QR instead (``linalg.qr``: rocSOLVER geqrf / orgqr).
    Returns ``(Q, used_fallback)``."""
    from . import ops

    y = y if isinstance(y, Array) else Array.from_numpy(np.asarray(y))
    eps = float(np.finfo(np.dtype(y.dtype)).eps)
    try:
        # (a well-conditioned sketch was measured on the way -- one pass, its defect already read back: nothing extra)
        Q, _, defect = qr_via_cholesky(y, shift=True, refine="auto", return_defect=True)
        if defect is None:
            defect = _gram_defect(ops.tensordot(Q.conj(), Q, axes=([0], [0])))
        if np.isfinite(defect) and defect <= ORTH_DEFECT_EPS * y.shape[1] * eps:
            return Q, False
    except (np.linalg.LinAlgError, RuntimeError, FloatingPointError):
        pass          # potrf refused the (shifted) Gram matrix: not positive definite to working precision
    return qr(y)[0], True


def _orth(y, method):
    if method in ("qr:cholesky", "cholesky"):
        return orth_cholesky_checked(y)[0]
    if method == "qr":
        return qr(y)[0]
    if method in ("svd:eig", "eig", "svd"):
        # as the reference: ``array_split(y, absorb="lorthog", method=...)`` with ITS defaults (decomp.py:1806), i.e. the
        # basis is truncated at the relative cutoff 1e-10 ("rsum2") -- after power iterations that drops directions
        from .split import array_split

        return array_split(y, method, "lorthog")[0]
    raise ValueError(f"unknown orthogonalisation method {method!r}")


def svd_rand(x, k, oversample=10, num_iterations=2, method_lorthog="qr", method_reduced="svd", right=None,
             stabilize=False, seed=None, factors_only=False):
    """Rank-``k`` randomised SVD ``(U, s, VH)`` by sketching -- the reference's ``svd_rand_truncated`` (decomp.py:1689-
    1868) with ``absorb=None``: ``y = x w`` for a Gaussian ``w`` with ``k + oversample`` columns, ``num_iterations``
    rounds of ``y <- x (x^H y)`` (plain, as the reference; ``stabilize`` re-orthogonalises y after every product -- what
    quimb's other randomised driver does, rand_linalg.py:180-188 -- so that directions below eps^(1/(2q+1)) s_max
    survive the powers), ``Q`` = an orthonormal basis of y (``method_lorthog``), the small factor ``B = Q^H x``
    decomposed by ``method_reduced`` and truncated to k, ``U = Q U_B``.  ``right`` False sketches the row space instead
    (default: the shorter side is reduced: ``right = m > n``).  Every product is a GETT launch of this library; with
    ``method_lorthog="qr:cholesky"`` and ``method_reduced="svd:eig"`` the only LAPACK work left is a potrf and a syevd
    of size k + oversample.

    ``factors_only``: the reference's shortcut for ``absorb`` "right" / "left" when the sketch is no wider than the
    target rank (``k >= k_sketch``, decomp.py:1808-1815 / :1836-1843): no decomposition of the reduced factor at all --
    ``(Q, None, Q^H x)`` (``right``) or ``(x Q, None, Q^H)`` -- an isometry times the rest, which is all a sweep needs
    to move its orthogonality centre.  Raises if the sketch is wider than k."""
    from . import ops

    x = x if isinstance(x, Array) else Array.from_numpy(np.asarray(x))
    m, n = x.shape
    k = min(m, n) if k is None or k < 0 else min(int(k), m, n)
    ks = min(m, n, k + int(oversample))
    if right is None:
        right = m > n
    mm = lambda a, b: ops.tensordot(a, b, axes=([1], [0]))
    xh = None
    if right:
        y = mm(x, _randn(x._dev, (n, ks), x.dtype, seed))                     # (m, ks)
        for _ in range(int(num_iterations)):
            if stabilize:
                y = _orth(y, method_lorthog)
            y = ops.tensordot(x.conj(), y, axes=([0], [0]))                     # x^H y, (n, ks)
            if stabilize:
                y = _orth(y, method_lorthog)
            y = mm(x, y)
        Q = _orth(y, method_lorthog)                                             # (m, ks)
        B = ops.tensordot(Q.conj(), x, axes=([0], [0]))                         # Q^H x, (ks, n)
        if factors_only:
            if k < ks:
                raise ValueError("factors_only needs k >= k + oversample (oversample=0)")
            return Q, None, B
    else:
        w = _randn(x._dev, (ks, m), x.dtype, seed)
        y = mm(w, x)                                                             # (ks, n)
        for _ in range(int(num_iterations)):
            if stabilize:
                y = ops.transpose(_orth(ops.transpose(y, (1, 0)), method_lorthog), (1, 0))
            y = ops.tensordot(y, x.conj(), axes=([1], [1]))                     # y x^H, (ks, m)
            if stabilize:
                y = ops.transpose(_orth(ops.transpose(y, (1, 0)), method_lorthog), (1, 0))
            y = mm(y, x)
        Q = _orth(ops.transpose(y.conj(), (1, 0)), method_lorthog)              # (n, ks): basis of the row space
        B = mm(x, Q)                                                             # (m, ks)
        if factors_only:
            if k < ks:
                raise ValueError("factors_only needs k >= k + oversample (oversample=0)")
            return B, None, ops.transpose(Q.conj(), (1, 0))
    if method_reduced in ("svd:eig", "eig"):
        ub, s, vbh = svd_via_eig(B)
    elif method_reduced == "svd":
        ub, s, vbh = svd(B)
    else:
        raise ValueError(f"unknown method_reduced {method_reduced!r}")
    kk = min(k, s.shape[0])
    ub, vbh = ub[:, :kk], vbh[:kk, :]
    s = Array.from_numpy(s.to_numpy()[:kk], dev=x._dev)
    if right:
        return mm(Q, ub), s, vbh
    return ub, s, ops.tensordot(vbh, Q.conj(), axes=([1], [1]))                  # VH = VH_B Q^H


def rsvd(x, k, q=2, p=0, seed=None, method_lorthog="qr:cholesky", method_reduced="svd:eig"):
    """Halko's randomised SVD with STABILISED power iterations (``rsvd_core``, quimb/linalg/rand_linalg.py:114-205: the
    sketch is re-orthogonalised after every product; q = 2 power iterations, no oversampling by default) -- the engine of
    the reference's ``rsvd`` split driver (decomp.py:2538) for a fixed target rank."""
    return svd_rand(x, k, oversample=p, num_iterations=q, method_lorthog=method_lorthog, method_reduced=method_reduced,
                    stabilize=True, seed=seed)


def eigvalsh(x):
    w, _ = eigh(x)
    return w


def norm(x, ord=None):
    from .ops import norm_fro

    if ord not in (None, "fro", 2):
        raise NotImplementedError("only the Frobenius / vector 2-norm is provided")
    return norm_fro(x)
