"""Two-site DMRG on the device: the caller of the hot path that BASELINE config #5 names
(``DMRG2(MPO_ham_heis(100), bond_dims=512)``; reference quimb/tensor/tn1d/dmrg.py -- ``DMRG.solve`` :1032-1160,
``sweep_right`` / ``sweep_left`` :885-1020, ``_update_local_state_2site`` :803-870, ``form_local_ops``
:681-732, ``MovingEnvironment`` :105-443, defaults ``get_default_opts`` :19-100).

Per bond the reference (a) forms the effective Hamiltonian L-W-W-R as a ``TNLinearOperator``, (b) asks ``eigh``
for its lowest eigenpair starting from the current two-site tensor (``local_eig_tol=1e-3``, ``ncv=4``),
(c) splits the result by a truncated SVD (``max_bond``, ``cutoff`` with ``cutoff_mode="sum2"``) absorbing the
singular values in the sweep direction, (d) moves the environment one site on.  Here (a) is
``quimb_amd.TNLinearOperator`` (cached expression on the GETT kernels), (b) the device-resident Lanczos
``eigh_lanczos``, (c) ``quimb_amd.linalg`` (rocSOLVER; ``split="eig"`` uses the Gram-matrix route), (d) one cached
three-step contraction.  Nothing but eigenvalues, norms and singular values crosses to the host.

Site tensors are held as (left, phys, right) so that every matricisation is a free reshape; ``state`` hands them
back in the reference's MPS order (left, right, phys; quimb/tensor/tn1d/core.py:1881-1901).
"""

import itertools
import math

import numpy as np

from . import linalg, ops
from .array import Array, asarray
from .contract import array_contract
from .eigsolve import eigh_lanczos
from .linop import TNLinearOperator
from .split import svals_to_keep


def mpo_ham_heis(L, j=1.0, bz=0.0, dtype="float64"):
    """Spin-1/2 Heisenberg chain with open ends as MPO site arrays in the reference's order -- first (r, k, b),
    bulk (l, r, k, b), last (l, k, b) -- and its convention ``H = sum_i j S_i.S_{i+1} - bz sum_i Sz_i`` with
    spin operators, not Pauli matrices (``MPO_ham_heis``, quimb/tensor/tensor_builder.py:5474-5530; the bond
    dimension is 5 as there, the gauge of the bond is this builder's own)."""
    if L < 2:
        raise ValueError("need at least two sites")
    try:
        jx, jy, jz = j
    except (TypeError, ValueError):
        jx = jy = jz = j
    if jx != jy:
        raise NotImplementedError("only jx == jy (a real MPO) is provided")
    sp = np.array([[0.0, 1.0], [0.0, 0.0]])
    sm = sp.T.copy()
    sz = np.array([[0.5, 0.0], [0.0, -0.5]])
    eye = np.eye(2)
    W = np.zeros((5, 5, 2, 2))
    W[0, 0] = eye
    W[1, 0] = sm
    W[2, 0] = sp
    W[3, 0] = sz
    W[4, 0] = -bz * sz
    W[4, 1] = 0.5 * jx * sp
    W[4, 2] = 0.5 * jx * sm
    W[4, 3] = jz * sz
    W[4, 4] = eye
    out = [W[4].copy()] + [W.copy() for _ in range(L - 2)] + [W[:, 0].copy()]
    return [a.astype(dtype) for a in out]


class DMRG2:
    """Two-site DMRG for an open-boundary MPO Hamiltonian.

    ``ham``: MPO site arrays in the reference's order (see ``mpo_ham_heis``); ``bond_dims`` / ``cutoffs``: one
    value per sweep, the last one repeated (``_set_bond_dim_seq``, dmrg.py:596-604).  After ``solve``:
    ``energy``, ``energies`` (one per sweep), ``state`` (MPS arrays, reference order l, r, p)."""

    def __init__(self, ham, bond_dims=None, cutoffs=1e-8, which="SA", p0=None, dtype=None, seed=0, split="svd",
                 canonize="qr", split_opts=None):
        if bond_dims is None:
            bond_dims = [8, 16, 32, 64, 128, 256, 512, 1024]        # DMRG2's default schedule, dmrg.py:1168
        ws = [asarray(w) for w in ham]
        self.L = len(ws)
        if self.L < 2:
            raise ValueError("DMRG2 needs at least two sites")
        self.dtype = np.dtype(dtype) if dtype is not None else np.dtype(ws[0].dtype)
        ws = [w.astype(self.dtype) for w in ws]
        # pad the end tensors to four legs (l, r, k, b) with unit bonds
        ws[0] = ws[0].reshape((1,) + ws[0].shape)
        ws[-1] = ws[-1].reshape((ws[-1].shape[0], 1) + ws[-1].shape[1:])
        self._W = ws
        self.phys = [w.shape[2] for w in ws]
        if which not in ("SA", "LA"):
            raise ValueError("which must be 'SA' or 'LA'")
        if split not in ("svd", "eig", "rand"):
            raise ValueError("split must be 'svd', 'eig' or 'rand'")
        if canonize not in ("qr", "cholesky"):
            raise ValueError("canonize must be 'qr' or 'cholesky'")
        # split: "svd" -- rocSOLVER gesvd (the reference's default driver); "eig" -- the reference's ``svd:eig`` (Gram
        # matrix + syevd of size chi d); "rand" -- the reference's ``svd:rand`` (decomp.py:1689): a sketch of ``max_bond +
        # oversample`` columns, so the only LAPACK work is a potrf and a syevd of THAT size and everything O(n^3) runs on
        # the contraction kernels.  ``split_opts`` go to ``linalg.svd_rand`` (default: no power iterations -- the sketch of
        # a two-site tensor whose numerical rank is below max_bond + oversample is exact to the discarded tail, and plain
        # power iterations would push the small Schmidt directions a sweep relies on below the rounding floor).
        # canonize: "qr" -- geqrf/orgqr; "cholesky" -- the reference's ``qr:cholesky`` (decomp.py:2359) with one
        # refinement pass, double precision only (cond^2 must stay below 1/eps), Householder QR as the fallback.
        self.which, self.split, self.canonize = which, split, canonize
        self.split_opts = dict(split_opts or {})
        self.opts = {"default_sweep_sequence": "R", "local_eig_tol": 1e-3, "local_eig_ncv": 4,
                     "local_eig_maxiter": None}
        self._set_seq(bond_dims, cutoffs)
        self.energies, self.local_energies, self.total_energies = [], [], []
        if p0 is None:
            chi0 = self._bond_dim0
            rng = np.random.default_rng(seed)
            dims = [1] + [min(chi0, math.prod(self.phys[:i]), math.prod(self.phys[i:]))
                          for i in range(1, self.L)] + [1]
            p0 = []
            for i in range(self.L):
                x = rng.standard_normal((dims[i], self.phys[i], dims[i + 1]))
                if self.dtype.kind == "c":
                    x = x + 1j * rng.standard_normal(x.shape)
                p0.append(x)
            self._A = [asarray(x).astype(self.dtype) for x in p0]
        else:
            # reference MPS order: first (r, p), bulk (l, r, p), last (l, p)  ->  (l, p, r)
            xs = [asarray(x).astype(self.dtype) for x in p0]
            xs[0] = xs[0].reshape((1,) + xs[0].shape)
            xs[-1] = xs[-1].reshape((xs[-1].shape[0], 1, xs[-1].shape[1]))
            self._A = [ops.transpose(x, (0, 2, 1)) for x in xs]
        one = Array.full((1, 1, 1), 1.0, self.dtype)
        self._Lenv = [one] + [None] * self.L          # _Lenv[i]: everything left of site i, legs (ket, mpo, bra)
        self._Renv = [None] * self.L + [one]          # _Renv[i]: everything from site i on

    # ---- schedules --------------------------------------------------------------------------------------
    def _set_seq(self, bond_dims, cutoffs):
        self._set_bond_dim_seq(bond_dims)
        self._set_cutoff_seq(cutoffs)

    # the two schedules are independent, as in the reference (tn1d/dmrg.py:596-603): passing one to ``solve`` leaves
    # the other where it is (a scalar of either kind -- int, float, numpy scalar -- is a one-element schedule)
    def _set_bond_dim_seq(self, bond_dims):
        import numbers

        bds = (bond_dims,) if isinstance(bond_dims, numbers.Real) else tuple(bond_dims)
        self._bond_dim0 = int(bds[0])
        self._bond_dims = itertools.chain(bds, itertools.repeat(bds[-1]))

    def _set_cutoff_seq(self, cutoffs):
        import numbers

        cts = (cutoffs,) if isinstance(cutoffs, numbers.Real) else tuple(cutoffs)
        self._cutoffs = itertools.chain(cts, itertools.repeat(cts[-1]))

    @property
    def energy(self):
        return self.energies[-1]

    @property
    def state(self):
        out = [ops.transpose(a, (0, 2, 1)) for a in self._A]
        out[0] = out[0].reshape(out[0].shape[1:])
        out[-1] = out[-1].reshape((out[-1].shape[0], out[-1].shape[2]))
        return out

    def max_bond(self):
        return max(a.shape[2] for a in self._A[:-1])

    # ---- gauge and environments ----------------------------------------------------------------------------
    def _canonize_to(self, site):
        """Left-canonical left of ``site``, right-canonical right of it (QR sweeps from both ends)."""
        A = self._A
        chol = self.canonize == "cholesky" and self.dtype.itemsize >= 8 and self.dtype.kind in "fc" \
            and self.dtype not in (np.dtype("float32"), np.dtype("complex64"))

        def qr_tall(m):
            if chol and m.shape[0] >= m.shape[1]:
                try:
                    return linalg.qr_via_cholesky(m, shift=True, refine="auto")
                except np.linalg.LinAlgError:
                    pass
            return linalg.qr(m)

        def lq_wide(m):                      # (lower factor, isometry with orthonormal rows)
            if chol and m.shape[0] <= m.shape[1]:
                try:
                    return linalg.lq_via_cholesky(m, shift=True, refine="auto")
                except np.linalg.LinAlgError:
                    pass
            q, rr = linalg.qr(ops.transpose(m, (1, 0)))           # LQ through the QR of the transpose: A = (R^T)(Q^T)
            return ops.transpose(rr, (1, 0)), ops.transpose(q, (1, 0))

        for i in range(site):
            l, p, r = A[i].shape
            q, rr = qr_tall(A[i].reshape((l * p, r)))
            A[i] = q.reshape((l, p, q.shape[1]))
            A[i + 1] = ops.tensordot(rr, A[i + 1], axes=([1], [0]))
        for i in range(self.L - 1, site, -1):
            l, p, r = A[i].shape
            lo, q = lq_wide(A[i].reshape((l, p * r)))
            k = q.shape[0]
            A[i] = q.reshape((k, p, r))
            A[i - 1] = ops.tensordot(A[i - 1], lo, axes=([2], [0]))

    def _grow_left(self, i):
        """_Lenv[i + 1] from _Lenv[i] and site i:  L'[A, W, B] = L[a, w, b] A[a, s, A] W[w, W, t, s] conj(A)[b, t, B]
        (the MPO's upper index k meets the bra, its lower index b the ket: <psi|H|psi> = conj(psi)_k H_kb psi_b)."""
        a = self._A[i]
        self._Lenv[i + 1] = array_contract(
            [self._Lenv[i], a, self._W[i], a.conj()],
            [("a", "w", "b"), ("a", "s", "A"), ("w", "W", "t", "s"), ("b", "t", "B")], ("A", "W", "B"))

    def _grow_right(self, i):
        """_Renv[i] from _Renv[i + 1] and site i."""
        a = self._A[i]
        self._Renv[i] = array_contract(
            [self._Renv[i + 1], a, self._W[i], a.conj()],
            [("A", "W", "B"), ("a", "s", "A"), ("w", "W", "t", "s"), ("b", "t", "B")], ("a", "w", "b"))

    # ---- one bond ------------------------------------------------------------------------------------------
    def _update_local_state_2site(self, i, direction, max_bond, cutoff):
        A, W = self._A, self._W
        theta = ops.tensordot(A[i], A[i + 1], axes=([2], [0]))                 # (l, p1, p2, r)
        dims = theta.shape
        heff = TNLinearOperator(
            [(self._Lenv[i], ("A", "w", "a")), (W[i], ("w", "x", "s", "S")), (W[i + 1], ("x", "y", "t", "T")),
             (self._Renv[i + 2], ("B", "y", "b"))],            # env legs are (ket, mpo, bra): ket side = input
            ("a", "s", "t", "b"), ("A", "S", "T", "B"), dtype=self.dtype)
        n = heff.shape[0]
        if n <= 2:
            dense = heff.matmat(np.eye(n, dtype=self.dtype))
            dense = np.asarray(dense.to_numpy() if hasattr(dense, "to_numpy") else dense)
            w, v = np.linalg.eigh(0.5 * (dense + dense.conj().T))
            k = 0 if self.which == "SA" else -1
            loc_en, gs = float(w[k]), asarray(v[:, k].astype(self.dtype))
        else:
            ncv = max(self.opts["local_eig_ncv"], 3)
            evals, vecs = eigh_lanczos(heff, k=1, which=self.which, v0=theta.reshape((n,)), ncv=min(ncv, n),
                                       tol=self.opts["local_eig_tol"], maxiter=self.opts["local_eig_maxiter"] or 10 * ncv,
                                       miniter=ncv)
            loc_en, gs = float(evals[0]), vecs.reshape((n,))
        m = gs.reshape((dims[0] * dims[1], dims[2] * dims[3]))
        if self.split == "rand" and max_bond and self.split_opts.get("oversample", 10) == 0 and max_bond >= min(m.shape):
            # nothing to truncate and no singular values wanted (static bond): the isometry is the identity on the short
            # side, or one Cholesky-QR of the tall one -- no eigen-decomposition near the ends of the chain either
            right = direction == "right"
            mm_, nn_ = m.shape
            if right and mm_ <= nn_:
                lf, rf = Array.from_numpy(np.eye(mm_, dtype=self.dtype), dev=m._dev), m
            elif not right and nn_ <= mm_:
                lf, rf = m, Array.from_numpy(np.eye(nn_, dtype=self.dtype), dev=m._dev)
            elif right:
                lf, rf = linalg.qr_via_cholesky(m, shift=True, refine="auto") if self.dtype.itemsize >= 8 and self.dtype.kind == "f" \
                    or self.dtype == np.dtype("complex128") else linalg.qr(m)
            else:
                if self.dtype == np.dtype("float64") or self.dtype == np.dtype("complex128"):
                    lf, rf = linalg.lq_via_cholesky(m, shift=True, refine="auto")
                else:
                    q_, r_ = linalg.qr(ops.transpose(m, (1, 0)))
                    lf, rf = ops.transpose(r_, (1, 0)), ops.transpose(q_, (1, 0))
            k = lf.shape[1]
            A[i] = lf.reshape((dims[0], dims[1], k))
            A[i + 1] = rf.reshape((k, dims[2], dims[3]))
            th = ops.tensordot(A[i], A[i + 1], axes=([2], [0])).reshape((n,))
            hv = heff.matvec(th)
            nrm2 = float(np.real(ops.tensordot(th.conj(), th, axes=([0], [0])).item()))
            return loc_en, float(np.real(ops.tensordot(th.conj(), hv, axes=([0], [0])).item())) / nrm2
        if self.split == "rand" and max_bond and 0 < max_bond + self.split_opts.get("oversample", 10) < min(m.shape):
            # (the Gram route squares the condition number: double precision only, as in ``_canonize_to``; a single-precision
            # sketch takes Householder QR -- and ``linalg._orth`` checks whichever Cholesky basis it is given)
            dbl = self.dtype in (np.dtype("float64"), np.dtype("complex128"))
            so = dict(oversample=10, num_iterations=0, method_lorthog="qr:cholesky" if dbl else "qr", method_reduced="svd:eig")
            so.update(self.split_opts)
            if so["oversample"] == 0:
                # the reference's ``svd:rand`` with a sketch no wider than the bond (decomp.py:1808-1815, :1836-1843): the
                # reduced factor is NOT decomposed -- an isometry on the side the sweep leaves behind, everything else moves
                # on; static truncation to max_bond (the cutoff plays no part, as there), no singular values
                right = direction == "right"
                lf, _, rf = linalg.svd_rand(m, max_bond, right=right, factors_only=True, **so)
                k = lf.shape[1]
                A[i] = lf.reshape((dims[0], dims[1], k))
                A[i + 1] = rf.reshape((k, dims[2], dims[3]))
                th = ops.tensordot(A[i], A[i + 1], axes=([2], [0])).reshape((n,))
                hv = heff.matvec(th)
                nrm2 = float(np.real(ops.tensordot(th.conj(), th, axes=([0], [0])).item()))
                tot = float(np.real(ops.tensordot(th.conj(), hv, axes=([0], [0])).item())) / nrm2
                return loc_en, tot
            u, s, vh = linalg.svd_rand(m, max_bond, **so)
        else:
            u, s, vh = (linalg.svd if self.split == "svd" else linalg.svd_via_eig)(m)
        sh = s.to_numpy()
        k = svals_to_keep(sh, cutoff, "sum2", max_bond)          # bond_compress_cutoff_mode, dmrg.py:85
        sk = asarray(sh[:k].astype(u.dtype))
        if direction == "right":       # A[i] left-canonical, s.V^H moves on
            A[i] = u[:, :k].reshape((dims[0], dims[1], k))
            A[i + 1] = ops.multiply(vh[:k, :], sk[:, None]).reshape((k, dims[2], dims[3]))
        else:
            A[i] = ops.multiply(u[:, :k], sk[None, :]).reshape((dims[0], dims[1], k))
            A[i + 1] = vh[:k, :].reshape((k, dims[2], dims[3]))
        # the energy of the (truncated, un-normalised) state: <H> / <1> with the two-site tensor as the centre
        s2 = sh[:k].astype(np.float64) ** 2
        th = ops.tensordot(A[i], A[i + 1], axes=([2], [0])).reshape((n,))
        hv = heff.matvec(th)
        tot = float(np.real(ops.tensordot(th.conj(), hv, axes=([0], [0])).item())) / float(s2.sum())
        return loc_en, tot

    # ---- sweeps --------------------------------------------------------------------------------------------
    def sweep(self, direction, canonize=True, max_bond=None, cutoff=1e-9):
        """One pass over all bonds, "R" (left to right) or "L"; returns the energy after the last update."""
        L = self.L
        local = []
        if direction == "R":
            if canonize:
                self._canonize_to(0)
                for i in range(L - 1, 1, -1):
                    self._grow_right(i)
            for i in range(L - 1):
                en, tot = self._update_local_state_2site(i, "right", max_bond, cutoff)
                local.append(en)
                if i < L - 2:
                    self._grow_left(i)
        elif direction == "L":
            if canonize:
                self._canonize_to(L - 1)
                for i in range(L - 2):
                    self._grow_left(i)
            for i in range(L - 2, -1, -1):
                en, tot = self._update_local_state_2site(i, "left", max_bond, cutoff)
                local.append(en)
                if i > 0:
                    self._grow_right(i + 1)
        else:
            raise ValueError("direction must be 'R' or 'L'")
        self.local_energies.append(local)
        return tot

    def solve(self, tol=1e-4, bond_dims=None, cutoffs=None, sweep_sequence=None, max_sweeps=10, verbosity=0):
        """Sweep until the energy changes by less than ``tol`` between sweeps (absolute, dmrg.py:1024-1028) or
        ``max_sweeps`` is reached; returns whether it converged."""
        if bond_dims is not None:
            self._set_bond_dim_seq(bond_dims)
        if cutoffs is not None:
            self._set_cutoff_seq(cutoffs)
        seq = sweep_sequence or self.opts["default_sweep_sequence"]
        previous = "0"
        for k in range(max_sweeps):
            direction = seq[k % len(seq)]
            max_bond, cutoff = next(self._bond_dims), next(self._cutoffs)
            canonize = not (direction + previous in {"LR", "RL"})
            energy = self.sweep(direction, canonize=canonize, max_bond=max_bond, cutoff=cutoff)
            self.energies.append(energy)
            if verbosity:
                print(f"{len(self.energies)}, {direction}, max_bond=({self.max_bond()}/{max_bond}), "
                      f"cutoff:{cutoff}  Energy: {energy}")
            if len(self.energies) >= 2 and abs(self.energies[-2] - self.energies[-1]) < tol:
                return True
            previous = direction
        return False
