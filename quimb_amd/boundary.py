"""Boundary contraction of a 2D tensor network: the caller of the hot path that config #3's
*approximate* mode uses (``TensorNetwork2D.contract_boundary`` ->
``_contract_interleaved_boundary_sequence`` -> ``_contract_boundary_core``,
quimb/tensor/tn2d/core.py:2502-2642, :1355-1484).

The reference absorbs one lattice row into a boundary line by pairwise contractions (the hot path: here one
GETT launch per site that writes the result directly in the fused-bond layout, so ``fuse`` costs nothing),
then gauges the line with a QR sweep and truncates it with an SVD sweep in the opposite direction
(``canonize_plane`` / ``compress_plane``; truncation rule: keep at most ``max_bond`` singular values and
drop those below ``cutoff`` x the largest -- ``tensor_split``'s default ``cutoff_mode="rel"``).
By default the two shortest opposing sides move inwards alternately until they are adjacent and the
remaining two lines are contracted exactly.

Boundary tensors are kept in (left, inward, right) order: both matricisations the sweeps need --
(left*inward, right) for QR and (left, inward*right) for the SVD -- are then free reshapes.
The decompositions go through ``quimb_amd.linalg`` (rocSOLVER; plumbing, not measured); everything
else stays on this library's kernels and in HBM.  The norm of the line is moved into a log10 exponent
after every row (``equalize_norms`` in the reference), so fp32 survives values like 1e105.
"""

import math

import numpy as np

from . import linalg, ops
from .array import asarray
from .contract import array_contract
from .split import svals_to_keep


def _pad4(x, i, j, Lx, Ly):
    """Site array in the reference's l, r, u, d order (missing legs on the edges) -> all four legs."""
    have = (j > 0, j < Ly - 1, i < Lx - 1, i > 0)
    it = iter(x.shape)
    shape = tuple(next(it) if h else 1 for h in have)
    return x.reshape(shape)


class _Line:
    """One boundary: ``Ly`` tensors of shape (left, inward, right) and the log10 scale taken out of them."""

    def __init__(self, tensors):
        self.t = list(tensors)
        self.exponent = 0.0

    def absorb(self, row, inward_axis):
        """Contract the inward leg of every boundary tensor with the matching leg of the next lattice row
        (4-leg arrays, l r u d); ``inward_axis`` = 2 (u) when moving up from xmin, 3 (d) when moving down."""
        out_leg = "u" if inward_axis == 2 else "d"
        in_leg = "d" if inward_axis == 2 else "u"
        for j, w in enumerate(row):
            b = self.t[j]
            y = array_contract(
                [b, w],
                [("L", in_leg, "R"), ("l", "r", "u", "d")],
                ("L", "l", out_leg, "R", "r"),
            )
            L, l, x, R, r = y.shape
            self.t[j] = y.reshape((L * l, x, R * r))

    def canonize(self):
        """QR sweep left -> right: every tensor but the last becomes an isometry (left*inward -> right)."""
        t = self.t
        for j in range(len(t) - 1):
            l, x, r = t[j].shape
            q, rr = linalg.qr(t[j].reshape((l * x, r)))
            k = q.shape[1]
            t[j] = q.reshape((l, x, k))
            t[j + 1] = ops.tensordot(rr, t[j + 1], axes=([1], [0]))

    def compress(self, max_bond, cutoff, method="svd"):
        """SVD sweep right -> left, truncating every bond; U*s is absorbed towards the left."""
        t = self.t
        svd = linalg.svd if method == "svd" else linalg.svd_via_eig
        for j in range(len(t) - 1, 0, -1):
            l, x, r = t[j].shape
            u, s, vh = svd(t[j].reshape((l, x * r)))
            sh = s.to_numpy()
            # tensor_split's default relative cutoff (tensor_core.py:400), then the max_bond cap
            k = svals_to_keep(sh, cutoff, "rel", max_bond)
            t[j] = vh[:k, :].reshape((k, x, r))
            us = ops.multiply(u[:, :k], asarray(sh[:k].astype(u.dtype))[None, :])
            t[j - 1] = ops.tensordot(t[j - 1], us, axes=([2], [0]))

    def equalize(self):
        """Move the line's norm (it sits in tensor 0 after ``compress``) into the exponent."""
        nrm = float(ops.norm_fro(self.t[0]))
        if nrm > 0.0 and math.isfinite(nrm):
            self.t[0] = self.t[0] / nrm
            self.exponent += math.log10(nrm)

    def max_bond(self):
        return max(max(x.shape[0], x.shape[2]) for x in self.t)


def contract_boundary_2d(arrays, Lx, Ly, max_bond=None, cutoff=1e-10, canonize=True, sequence=None,
                         strip_exponent=False, dtype=None, method="svd"):
    """Value of an open ``Lx`` x ``Ly`` network given as the row-major list of site arrays in the
    reference's l, r, u, d leg order (``TN2D_from_fill_fn``, quimb/tensor/tensor_builder.py:1345-1369;
    ``u`` points to row i+1).

    ``max_bond=None`` with ``cutoff=0`` is exact.  ``sequence`` (subset of ("xmin", "xmax"), default both)
    names the sides that move inwards, alternately, until the two lines are adjacent.  ``method``: "svd"
    (rocSOLVER ``gesvd``) or "eig" (``linalg.svd_via_eig``: Gram matrix on the GETT kernels + ``syevd`` -- the
    reference's ``method="svd:eig"`` split; singular values below ~sqrt(eps) x the largest lose accuracy, which
    is harmless for values that are truncated anyway).  Returns the scalar,
    or ``(mantissa, exponent)`` with ``mantissa * 10**exponent`` the value when ``strip_exponent``."""
    if len(arrays) != Lx * Ly:
        raise ValueError(f"expected {Lx * Ly} site arrays, got {len(arrays)}")
    if Lx < 1 or Ly < 1 or Lx * Ly < 2:
        raise ValueError("need at least two sites")
    sequence = tuple(sequence) if sequence is not None else ("xmin", "xmax")
    if method not in ("svd", "eig"):
        raise ValueError("method must be 'svd' or 'eig'")
    if not sequence or any(s not in ("xmin", "xmax") for s in sequence):
        raise ValueError("sequence must name 'xmin' and / or 'xmax' (rows are the sweep direction here)")
    xs = [asarray(a) if dtype is None else asarray(a).astype(dtype) for a in arrays]
    grid = [[_pad4(xs[i * Ly + j], i, j, Lx, Ly) for j in range(Ly)] for i in range(Lx)]
    if Lx < Ly:
        # the reference moves the two SHORTEST opposing sides (tn2d/core.py:2398-2409): sweep over columns by
        # transposing the lattice -- new (l, r, u, d) = old (d, u, r, l)
        grid = [[ops.transpose(grid[i][j], (3, 2, 1, 0)) for i in range(Lx)] for j in range(Ly)]
        Lx, Ly = Ly, Lx
    # the reference gauges a line from its far end backwards and truncates forwards (``canonize_plane`` with
    # ``yreverse=True`` then ``compress_plane``, tn2d/core.py:1455-1484); the sweeps below run QR forwards and SVD
    # backwards on free reshapes, so the lattice is mirrored left <-> right once, here: same bonds, same order
    rows = [[ops.transpose(w, (1, 0, 2, 3)) for w in reversed(row)] for row in grid]
    # (l, r, u, d=1) -> (l, u, r)   /   (l, r, u=1, d) -> (l, d, r)
    lo = _Line([ops.transpose(w.reshape(w.shape[:3]), (0, 2, 1)) for w in rows[0]])
    hi = _Line([ops.transpose(w.reshape((w.shape[0], w.shape[1], w.shape[3])), (0, 2, 1)) for w in rows[-1]])
    ilo, ihi = 0, Lx - 1
    truncate = (max_bond is not None and max_bond > 0) or cutoff > 0.0
    turn = 0
    while ihi - ilo > 1:
        side = sequence[turn % len(sequence)]
        turn += 1
        if side == "xmin":
            ilo += 1
            line = lo
            line.absorb(rows[ilo], 2)
        else:
            ihi -= 1
            line = hi
            line.absorb(rows[ihi], 3)
        if truncate:
            if canonize:
                line.canonize()
            line.compress(max_bond, cutoff, method)
        line.equalize()
    # the two lines are adjacent: contract the ladder exactly, left to right
    env = None
    exponent = lo.exponent + hi.exponent
    for b, t in zip(lo.t, hi.t):
        if env is None:
            env = array_contract([b, t], [("a", "x", "c"), ("b", "x", "d")], ("a", "b", "c", "d"))
            env = env.reshape(env.shape[2:])
        else:
            env = array_contract([env, b, t], [("a", "b"), ("a", "x", "c"), ("b", "x", "d")], ("c", "d"))
        nrm = float(ops.norm_fro(env))
        if nrm > 0.0 and math.isfinite(nrm):
            env = env / nrm
            exponent += math.log10(nrm)
    mantissa = env.reshape(()).item()
    if strip_exponent:
        # the reference hands back a unit-modulus mantissa and everything else in the exponent
        # (``strip_exponent`` is forwarded to the final ``tn.contract``, tn2d/core.py:2493-2498)
        mag = abs(mantissa)
        if mag > 0.0:
            mantissa, exponent = mantissa / mag, exponent + math.log10(mag)
        return mantissa, exponent
    return mantissa * 10.0**exponent
