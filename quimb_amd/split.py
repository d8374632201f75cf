"""Truncated splits of device arrays: the hand-off between contractions that boundary contraction and DMRG sit on
(``array_split``, quimb/tensor/decomp.py:35-174; ``tensor_split``, quimb/tensor/tensor_core.py:390-640).

The decompositions themselves are rocSOLVER's (through ``quimb_amd.linalg``; ``method="svd:eig"`` runs the Gram
products on this library's kernels); what is restated here is the reference's *policy*: how many singular values
survive (``cutoff`` / ``cutoff_mode`` / ``max_bond``, decomp.py:901-937 and :744-790), the optional renormalisation
of the kept ones (:940-966) and where the singular values are absorbed (``absorb``, :264-298, :690-725).
"""

import numpy as np

from . import linalg, ops
from .array import asarray

_CUTOFF_MODES = {"abs": 1, "rel": 2, "sum2": 3, "rsum2": 4, "sum1": 5, "rsum1": 6}
_ABSORB = {
    None: None, "U,s,VH": None,
    "s": "s", "svals": "s",
    "lsqrt": "lsqrt", "Usq": "lsqrt",
    "VH": "rorthog", "rorthog": "rorthog",
    "Us": "lfactor", "lfactor": "lfactor",
    "Us,VH": "left", "left": "left",
    "Usq,sqVH": "both", "both": "both",
    "U,sVH": "right", "right": "right",
    "U": "lorthog", "lorthog": "lorthog",
    "sVH": "rfactor", "rfactor": "rfactor",
    "sqVH": "rsqrt", "rsqrt": "rsqrt",
}


def svals_to_keep(s, cutoff=1e-10, cutoff_mode="rsum2", max_bond=None):
    """Number of leading singular values that survive -- never fewer than one.

    ``abs``: s_i > cutoff;  ``rel``: s_i > cutoff * s_0;  ``sum2`` / ``sum1``: drop the longest tail whose summed
    s^2 (s) stays <= cutoff;  ``rsum2`` / ``rsum1``: the same relative to the total (decomp.py:901-937)."""
    if cutoff_mode not in _CUTOFF_MODES:
        raise ValueError(f"unknown cutoff_mode {cutoff_mode!r}")
    s = np.abs(np.asarray(s, dtype=np.float64))
    n = len(s)
    if n and cutoff is not None and cutoff > 0.0:
        if cutoff_mode == "abs":
            n = int(np.sum(s > cutoff))
        elif cutoff_mode == "rel":
            n = int(np.sum(s > cutoff * s[0]))
        else:
            sp = s**2 if cutoff_mode in ("sum2", "rsum2") else s
            target = cutoff * (sp.sum() if cutoff_mode in ("rsum2", "rsum1") else 1.0)
            acc = 0.0
            for i in range(n - 1, -1, -1):
                if not np.isnan(sp[i]):
                    acc += sp[i]
                if acc > target:
                    break
                n -= 1
        n = max(n, 1)
    if max_bond is not None and max_bond > 0:
        n = min(n, int(max_bond))
    return n


def renorm_factor(s, n_keep, renorm):
    """Factor that restores the kept values' sum of ``s**renorm`` to that of all of them (decomp.py:940-966)."""
    s = np.abs(np.asarray(s, dtype=np.float64))
    p = s**renorm if renorm >= 2 else s
    keep, lose = p[:n_keep].sum(), p[n_keep:].sum()
    f = (keep + lose) / keep
    return f ** (1.0 / renorm) if renorm >= 2 else f


def _scale_cols(u, v):
    return ops.multiply(u, asarray(v.astype(u.dtype))[None, :])


def _scale_rows(v, vh):
    return ops.multiply(vh, asarray(v.astype(vh.dtype))[:, None])


def array_split(x, method="svd", absorb="both", max_bond=None, cutoff=1e-10, cutoff_mode="rsum2", renorm=None,
                stabilized=True, **opts):
    """``(left, s, right)`` of a 2-d device array, entries ``None`` where ``absorb`` does not ask for them.

    ``method``: "svd" (rocSOLVER gesvd), "svd:eig" / "eig" (Gram matrix + syevd), "qr", "lq" (no truncation), and the
    GEMM-shaped drivers of the reference (``quimb_amd.linalg``): "qr:cholesky" (QR- or LQ-like by ``absorb``; options
    ``shift``, ``refine``), "cholesky" (Hermitian positive definite input; ``absorb`` "both" / "lsqrt" / "rsqrt";
    ``shift``), "svd:rand" (needs ``max_bond``; options ``oversample``, ``num_iterations``, ``method_lorthog``,
    ``method_reduced``, ``right``, ``seed``, ``stabilize``) and "rsvd" (stabilised power iterations, then the cutoff
    policy on the ``max_bond`` values found; options ``q``, ``p``, ``seed``).
    ``absorb``: None / "U,s,VH", "both", "left", "right", "lorthog", "rorthog", "lfactor", "rfactor", "lsqrt",
    "rsqrt", "s" (aliases as in decomp.py:264-298).  ``renorm``: True -> the power matching ``cutoff_mode``.
    ``stabilized`` (qr / lq only): fix the phases so the triangular factor has a non-negative real diagonal."""
    x = asarray(x)
    if x.ndim != 2:
        raise ValueError("array_split needs a 2-d array")
    if absorb not in _ABSORB:
        raise ValueError(f"Invalid absorb mode: {absorb}")
    mode = _ABSORB[absorb]
    if method in ("qr", "lq"):
        if mode is None or mode in ("s", "both", "lsqrt", "rsqrt"):
            raise ValueError(f"You can't return the singular values separately when `method='{method}'`.")
        if method == "lq" or mode in ("left", "lfactor", "rorthog"):
            # x = L Q : through the QR of the transpose (x^T = Q' R' -> x = R'^T Q'^T)
            q, r = linalg.qr(ops.transpose(x, (1, 0)))
            left, right = ops.transpose(r, (1, 0)), ops.transpose(q, (1, 0))
        else:
            left, right = linalg.qr(x)
        if stabilized:
            # make the triangular factor's diagonal real and non-negative (``qr_stabilized``, decomp.py:2112-2126):
            # the phases go into the isometric factor, the product is unchanged, the decomposition unique
            iso_left = not (method == "lq" or mode in ("left", "lfactor", "rorthog"))
            tri = right if iso_left else left
            d = ops.diagonal(tri).to_numpy()
            mag = np.abs(d)
            phase = np.where(mag > 0, d / np.where(mag > 0, mag, 1.0), 1.0)
            if np.any(phase != 1.0):
                if iso_left:
                    left, right = _scale_cols(left, phase), _scale_rows(np.conj(phase), right)
                else:
                    left, right = _scale_cols(left, np.conj(phase)), _scale_rows(phase, right)
        return (left if mode in ("left", "right", "lfactor", "lorthog") else None, None,
                right if mode in ("left", "right", "rfactor", "rorthog") else None)
    if method == "qr:cholesky":
        # QR-like ("right" family) or LQ-like ("left" family) from the Gram matrix's Cholesky factor (decomp.py:2359-2420)
        if mode in ("right", "lorthog", "rfactor"):
            left, right = linalg.qr_via_cholesky(x, **opts)
        elif mode in ("left", "rorthog", "lfactor"):
            left, right = linalg.lq_via_cholesky(x, **opts)
        else:
            raise ValueError(f"Invalid absorb mode for qr_via_cholesky: {absorb}")
        return (left if mode in ("left", "right", "lfactor", "lorthog") else None, None,
                right if mode in ("left", "right", "rfactor", "rorthog") else None)
    if method == "cholesky":
        if mode not in ("both", "lsqrt", "rsqrt"):
            raise ValueError(f"Invalid absorb={absorb} in cholesky_regularized. Should be one of 'both', 'lsqrt' or 'rsqrt'.")
        L = linalg.cholesky_regularized(x, **opts)
        return (L if mode != "rsqrt" else None, None,
                ops.transpose(L.conj(), (1, 0)) if mode != "lsqrt" else None)
    if method == "svd":
        u, s, vh = linalg.svd(x)
    elif method in ("svd:eig", "eig"):
        u, s, vh = linalg.svd_via_eig(x)
    elif method == "svd:rand":
        # static truncation (``max_bond`` only, decomp.py:1703-1706): the cutoff policy below still sees the k values found
        if max_bond is None or max_bond < 0:
            import warnings

            warnings.warn("Using 'svd:rand' without `max_bond` is inefficient, consider simply using 'svd' or 'svd:eig' instead.")
        if opts.get("right") is None and mode in ("right", "lorthog", "rfactor", "left", "rorthog", "lfactor"):
            opts = dict(opts, right=mode in ("right", "lorthog", "rfactor"))
        kk = min(x.shape) if max_bond is None or max_bond < 0 else min(int(max_bond), *x.shape)
        if kk >= min(min(x.shape), kk + int(opts.get("oversample", 10))) and mode in ("right", "lorthog", "rfactor") \
                and opts.get("right") is not False:
            # no truncation below the sketch: the reduced factor is not decomposed (decomp.py:1808-1815)
            q, _, b = linalg.svd_rand(x, max_bond, **dict(opts, right=True, factors_only=True))
            return (q if mode != "rfactor" else None), None, (b if mode != "lorthog" else None)
        if kk >= min(min(x.shape), kk + int(opts.get("oversample", 10))) and mode in ("left", "rorthog", "lfactor") \
                and opts.get("right") is not True:
            b, _, qh = linalg.svd_rand(x, max_bond, **dict(opts, right=False, factors_only=True))
            return (b if mode != "rorthog" else None), None, (qh if mode != "lfactor" else None)
        u, s, vh = linalg.svd_rand(x, max_bond, **opts)
        cutoff = 0.0
    elif method == "rsvd":
        if max_bond is None or max_bond < 0:
            raise ValueError("method 'rsvd' on the device needs a target rank (max_bond)")
        u, s, vh = linalg.rsvd(x, max_bond, **opts)
    else:
        raise ValueError(f"unknown split method {method!r}")
    sh = s.to_numpy().astype(np.float64)
    if renorm is True:
        renorm = {"sum2": 2, "rsum2": 2, "sum1": 1, "rsum1": 1}.get(cutoff_mode, 0)
    renorm = int(renorm or 0)
    n = svals_to_keep(sh, cutoff if cutoff is not None else 0.0, cutoff_mode, max_bond)
    if n < len(sh):
        f = renorm_factor(sh, n, renorm) if renorm > 0 else 1.0
        sh = sh[:n] * f
        u, vh = u[:, :n], vh[:n, :]
    rdt = s.dtype
    if mode is None:
        return u, asarray(sh.astype(rdt)), vh
    if mode == "s":
        return None, asarray(sh.astype(rdt)), None
    sq = np.sqrt(sh)
    left = {"both": lambda: _scale_cols(u, sq), "lsqrt": lambda: _scale_cols(u, sq),
            "left": lambda: _scale_cols(u, sh), "lfactor": lambda: _scale_cols(u, sh),
            "right": lambda: u, "lorthog": lambda: u}.get(mode)
    right = {"both": lambda: _scale_rows(sq, vh), "rsqrt": lambda: _scale_rows(sq, vh),
             "right": lambda: _scale_rows(sh, vh), "rfactor": lambda: _scale_rows(sh, vh),
             "left": lambda: vh, "rorthog": lambda: vh}.get(mode)
    return (left() if left else None), None, (right() if right else None)


def tensor_split(T, left_inds, method="svd", get=None, absorb="both", max_bond=None, cutoff=1e-10,
                 cutoff_mode="rel", renorm=None, right_inds=None, bond_ind=None, ltags=None, rtags=None, **opts):
    """Split a ``quimb_amd.Tensor`` across ``left_inds | right_inds`` (reference ``tensor_split``,
    tensor_core.py:390-640: matricise with ``to_dense``-style fusing, ``array_split``, un-fuse, new bond last on
    the left factor and first on the right one).  ``get``: None -> (Tensor, Tensor) -- or (Tensor, s Tensor, Tensor)
    when ``absorb=None`` -- , "arrays" -> the raw arrays, "values" -> ALL singular values (no truncation)."""
    from .contract import Tensor

    left_inds = tuple(left_inds)
    if right_inds is None:
        right_inds = tuple(ix for ix in T.inds if ix not in left_inds)
    else:
        right_inds = tuple(right_inds)
    if set(left_inds) & set(right_inds) or set(left_inds) | set(right_inds) != set(T.inds):
        raise ValueError("left_inds and right_inds must partition the tensor's indices")
    data = asarray(T.data)
    perm = [T.inds.index(ix) for ix in left_inds + right_inds]
    x = ops.transpose(data, perm)
    ldims, rdims = x.shape[: len(left_inds)], x.shape[len(left_inds):]
    x = x.reshape((int(np.prod(ldims, dtype=np.int64)), int(np.prod(rdims, dtype=np.int64))))
    if get == "values":      # the whole spectrum, untruncated (``array_svals``, decomp.py:177-197)
        return array_split(x, method, "s", None, 0.0, cutoff_mode, None)[1]
    left, s, right = array_split(x, method, absorb, max_bond, cutoff, cutoff_mode, renorm, **opts)
    if left is not None:
        left = left.reshape(tuple(ldims) + (left.shape[1],))
    if right is not None:
        right = right.reshape((right.shape[0],) + tuple(rdims))
    if get == "arrays":
        return tuple(a for a in (left, s, right) if a is not None) if _ABSORB[absorb] is not None else (left, s, right)
    if bond_ind is None:
        bond_ind = ("_split_bond", id(T) & 0xFFFFFF)
    tags = tuple(T.tags)
    out = []
    if left is not None:
        out.append(Tensor(left, left_inds + (bond_ind,), tuple(ltags) if ltags else tags))
    if s is not None:
        out.append(Tensor(s, (bond_ind,), tags))
    if right is not None:
        out.append(Tensor(right, (bond_ind,) + right_inds, tuple(rtags) if rtags else tags))
    return tuple(out)
