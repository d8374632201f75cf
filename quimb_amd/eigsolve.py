"""Device-resident Lanczos for the lowest / highest eigenpair of a Hermitian operator.

Caller side of the hot path: DMRG's local solve
(``DMRG._eigs`` -> ``eigh(A, k=1, which=..., v0=..., ncv=..., tol=..., maxiter=...)``,
quimb/tensor/tn1d/dmrg.py:626-645, answered in the reference by
``quimb.linalg.base_linalg.eigensystem_partial`` -> scipy ARPACK, base_linalg.py:80) calls
``A.matvec`` 10-30 times per site.  With ARPACK the Krylov vectors live on the host, so
every matvec of a device operator pays two PCIe copies of the (chi*d*d*chi) vector; here
the basis ``Q[m, n]``, the matvecs (``TNLinearOperator``: cached contraction expression
on the GETT kernels) and the re-orthogonalisation GEMVs all stay in HBM, and only the
(j+1) projection coefficients and one norm per iteration cross to the host, where the
tiny tridiagonal problem is solved.

Same call shape as the reference's ``eigh`` for the arguments DMRG passes; returns
``(eigenvalues, eigenvectors)`` with eigenvectors as a device ``Array`` of shape (n, k).
"""

import numpy as np

from . import ops
from .array import Array, asarray


def _matvec(A, x):
    """``(A x, owned)``.  ``owned``: the solver may update the product IN PLACE.  Only ``matvec_borrow`` is that
    contract (the operator lends a buffer that is consumed before the next product is asked for); whatever a plain
    ``.matvec`` returns may be a buffer the operator keeps -- a stored array, a cached result, its own argument -- and
    is copied before the first write."""
    if hasattr(A, "matvec_borrow"):
        return asarray(A.matvec_borrow(x)), True
    if hasattr(A, "matvec"):
        return asarray(A.matvec(x)), False
    return ops.matmul(A, x), True            # a fresh array by construction (ops are pure)


def eigh_lanczos(A, k=1, which="SA", v0=None, ncv=None, tol=1e-10, maxiter=None, return_vecs=True, miniter=0):
    """``k`` extremal eigenpairs of the Hermitian operator ``A`` (anything with ``.shape``, ``.dtype`` and
    ``.matvec`` on device arrays -- a ``TNLinearOperator`` -- or a dense device matrix).

    Lanczos with full re-orthogonalisation in a basis of ``ncv`` device vectors and explicit restart
    from the current Ritz vector(s); converged when every wanted Ritz pair has residual
    ``|beta_m s_m| <= tol * max(|theta|, 1)`` (``tol = 0`` means machine precision, as in scipy).
    ``which``: "SA" (algebraically smallest) or "LA" (largest).  ``miniter``: Lanczos steps taken before the
    residual test may stop the first cycle -- an implicitly restarted solver (ARPACK, the reference's default)
    always completes one ``ncv``-step cycle, so a loose ``tol`` still improves a good starting vector; DMRG
    passes ``miniter=ncv`` for the same effect."""
    if which not in ("SA", "LA"):
        raise ValueError("which must be 'SA' or 'LA'")
    n = int(A.shape[0])
    dtype = np.dtype(A.dtype)
    rdt = np.zeros(0, dtype).real.dtype
    eps = np.finfo(rdt).eps
    tol = float(tol) if tol else 10 * eps
    k = int(k)
    if not 1 <= k < n:
        raise ValueError("need 1 <= k < n")
    m = int(ncv) if ncv else max(2 * k + 1, 20)
    m = max(min(m, n), k + 1)
    maxiter = int(maxiter) if maxiter else 10 * n
    if v0 is None:
        rng = np.random.default_rng(0)
        v0 = rng.standard_normal(n).astype(rdt)
        if dtype.kind == "c":
            v0 = v0 + 1j * rng.standard_normal(n).astype(rdt)
    q = asarray(v0).astype(dtype).reshape(n)
    dev = q._dev
    Q = Array.full((m + 1, n), 0.0, dtype, dev)       # basis rows; rows beyond the current step stay zero
    passes = 1 if rdt == np.float64 else 2   # classical Gram-Schmidt twice in single precision
    # the vector work of a step runs in three device launches per pass (csrc/krylov.hip): projection coefficients,
    # (alpha, beta) and the norm never visit the host inside a cycle -- they are read when a convergence test is due
    h = dev.empty(m + 1, dtype)
    h_sum = dev.empty(m + 1, dtype) if passes > 1 else None       # alpha = the coefficient summed over the passes
    ab = dev.empty(2 * (m + 1), np.float64)
    ws = dev.krylov_workspace(m + 1, n, dtype)

    def put(row, vec):      # Q[row] <- vec  (device-to-device, one strided copy)
        dev.permute(Q._buf[row * n:], vec._buf, (n,), (1,), 0, dtype)

    def norm(x):
        return float(np.sqrt(abs(ops.tensordot(x.conj(), x, axes=([0], [0])).item())))

    nmv = 0
    nrm = norm(q)
    if nrm == 0:
        raise ValueError("v0 is zero")
    put(0, q / nrm)
    theta = S = None
    first_test = max(k, min(int(miniter), m))          # no residual test (hence no host read) before this many steps
    while True:
        alphas, betas = [], []
        j_done = 0
        for j in range(m):
            qj = Array(dev, Q._buf[j * n:], (n,), dtype)
            w, owned = _matvec(A, qj)
            nmv += 1
            if w.dtype != dtype:
                w, owned = w.astype(dtype), True        # (a converted copy is ours)
            if w.size != n or w.ndim != 1:
                w = w.reshape(n)
            if not owned or dev.shares_storage(w._buf, Q._buf):
                w = w.copy()                            # w is updated in place below: never in somebody else's buffer
            # full re-orthogonalisation against the basis so far: h = Q^H w ; w -= Q^T h ; Q[j+1] = w / |w|
            for ps in range(passes):
                dev.krylov_project(h, h_sum, Q._buf, n, j + 1, w._buf, n, ps > 0, dtype, ws)
                dev.krylov_subtract(w._buf, Q._buf, n, j + 1, h, n, ps == passes - 1, dtype, ws)
            dev.krylov_extend(Q._buf[(j + 1) * n:], w._buf, n, (h if h_sum is None else h_sum)[j:], ab[2 * j:], eps, dtype, ws)
            j_done = j + 1
            if j_done < first_test and j_done < m:
                continue                                # (a breakdown in here zeroes the later rows: found below)
            got = np.asarray(dev.to_host(ab, 2 * j_done, np.float64), dtype=np.float64)
            alphas, betas = [float(x) for x in got[0::2]], [float(x) for x in got[1::2]]
            dead = [i for i in range(j_done) if betas[i] <= eps * max(abs(alphas[i]), 1.0)]
            if dead:                                    # invariant subspace found at step dead[0]
                j_done = dead[0] + 1
                alphas, betas = alphas[:j_done], betas[:j_done]
            beta = betas[-1]
            # Ritz values of the j_done x j_done tridiagonal matrix
            T = np.diag(alphas) + np.diag(betas[:-1], 1) + np.diag(betas[:-1], -1)
            evals, evecs = np.linalg.eigh(T)
            order = np.argsort(evals) if which == "SA" else np.argsort(-evals)
            kk = min(k, j_done)
            theta, S = evals[order[:kk]], evecs[:, order[:kk]]
            resid = np.abs(beta * S[-1, :])
            if j_done >= first_test and np.all(resid <= tol * np.maximum(np.abs(theta), 1.0)):
                break
            if dead or j + 1 == m:
                break                                  # invariant subspace found, or basis full
        Sfull = np.zeros((m + 1, S.shape[1]), dtype)
        Sfull[:j_done] = S
        X = ops.tensordot(asarray(Sfull), Q, axes=([0], [0]))               # (k, n) Ritz vectors
        converged = np.all(np.abs(betas[-1] * S[-1, :]) <= tol * np.maximum(np.abs(theta), 1.0)) or \
            betas[-1] <= eps * max(abs(alphas[-1]), 1.0)
        if converged or nmv >= maxiter or j_done == n:
            break
        # explicit restart from the (sum of the) wanted Ritz vectors
        x0 = X[0] if k == 1 else ops.sum(X, axis=0)
        x0 = x0 / norm(x0)
        dev.fill(Q._buf, Q.size, 0.0, dtype)
        put(0, x0)
    if not return_vecs:
        return theta.astype(rdt)
    return theta.astype(rdt), ops.transpose(X, (1, 0))
