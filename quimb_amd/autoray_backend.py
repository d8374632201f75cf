"""Registration with autoray / quimb, for an environment where they are
installed (they are not in the build image -- see INTEGRATION.md).

autoray infers a backend from the top-level module of an array's class, so
``quimb_amd.Array`` instances dispatch to the module-level functions of
``quimb_amd`` with no registration at all.  ``register()`` adds the few names
whose spelling differs and the composed quimb functions that allow per-backend
overrides (quimb/tensor/array_ops.py:148-274):

    import quimb_amd.autoray_backend as qab; qab.register()
    tn.apply_to_arrays(quimb_amd.asarray)          # tensor_core.py:5304
    tn.contract(all, optimize=tree)                # every do(...) now lands on HIP kernels
    # or, leaving numpy data in place:
    with qtn.contract_backend("quimb_amd"): ...
"""


def register():
    import autoray as ar

    import quimb_amd as qa
    from . import linalg

    ar.register_function("quimb_amd", "to_numpy", qa.to_numpy)
    ar.register_function("quimb_amd", "asarray", qa.asarray)
    ar.register_function("quimb_amd", "array", qa.asarray)
    ar.register_function("quimb_amd", "astype", qa.astype)
    ar.register_function("quimb_amd", "linalg.svd", linalg.svd)
    ar.register_function("quimb_amd", "linalg.qr", linalg.qr)
    ar.register_function("quimb_amd", "linalg.eigh", linalg.eigh)
    ar.register_function("quimb_amd", "linalg.norm", linalg.norm)
    try:  # one-pass index fusion + fused norm instead of the composed defaults
        from quimb.tensor import array_ops

        array_ops.fuse.register("quimb_amd")(lambda x, *groups, backend=None: qa.fuse(x, *groups))
        array_ops.norm_fro.register("quimb_amd")(qa.norm_fro)
    except Exception:  # quimb not importable: the autoray part above still stands
        pass
    try:  # quimb's split drivers (``array_split`` dispatches to them by backend): policy in split.py, LAPACK in rocSOLVER
        from quimb.tensor import decomp

        decomp.svd_truncated.register("quimb_amd")(svd_truncated)
        decomp.qr_stabilized.register("quimb_amd")(qr_stabilized)
        if hasattr(decomp, "svd_via_eig_truncated"):
            decomp.svd_via_eig_truncated.register("quimb_amd")(svd_via_eig_truncated)
        # the GEMM-shaped drivers (decomp.py:2359 qr_via_cholesky, :2262 cholesky_regularized, :1689 svd_rand_truncated,
        # :2538 rsvd): Gram / sketch products on this library's kernels, potrf / syevd of the reduced size on rocSOLVER
        for name, fn in (("qr_via_cholesky", qr_via_cholesky), ("cholesky_regularized", cholesky_regularized),
                         ("svd_rand_truncated", svd_rand_truncated)):
            drv = getattr(decomp, name, None)
            if drv is not None and hasattr(drv, "register"):
                drv.register("quimb_amd")(fn)
    except Exception:
        pass
    return "quimb_amd"


# quimb's numeric codes (quimb/tensor/decomp.py:264-298)
_CUTOFF_MODE_NAMES = {1: "abs", 2: "rel", 3: "sum2", 4: "rsum2", 5: "sum1", 6: "rsum1"}
_ABSORB_NAMES = {None: None, 2: "s", -12: "lsqrt", -11: "rorthog", -10: "lfactor", -1: "left", 0: "both", 1: "right",
                 10: "lorthog", 11: "rfactor", 12: "rsqrt"}


def _split(method, x, cutoff, cutoff_mode, max_bond, absorb, renorm, info):
    from .split import array_split

    left, s, right = array_split(
        x, method, _ABSORB_NAMES.get(absorb, absorb), max_bond if max_bond and max_bond > 0 else None,
        cutoff if cutoff and cutoff > 0 else 0.0, _CUTOFF_MODE_NAMES.get(cutoff_mode, cutoff_mode), renorm or None)
    if info is not None and "error" in info:
        info["error"] = None          # the discarded weight is not tracked on the device path
    return left, s, right


def svd_truncated(x, cutoff=-1.0, cutoff_mode=4, max_bond=-1, absorb=0, renorm=0, info=None, **_):
    """Drop-in for ``quimb.tensor.decomp.svd_truncated`` (decomp.py:831) on device arrays."""
    return _split("svd", x, cutoff, cutoff_mode, max_bond, absorb, renorm, info)


def svd_via_eig_truncated(x, cutoff=-1.0, cutoff_mode=4, max_bond=-1, absorb=0, renorm=0, info=None, **_):
    return _split("svd:eig", x, cutoff, cutoff_mode, max_bond, absorb, renorm, info)


def qr_stabilized(x, absorb=1, stabilized=True, **_):
    """Drop-in for ``quimb.tensor.decomp.qr_stabilized`` (decomp.py:2057): QR (or LQ, by ``absorb``) with the
    triangular factor's diagonal made non-negative."""
    from .split import array_split

    return array_split(x, "qr", _ABSORB_NAMES.get(absorb, absorb), None, 0.0, "rel", None, stabilized=stabilized)


def qr_via_cholesky(x, absorb=-1, shift=True, solve_triangular=True, **_):
    """Drop-in for ``quimb.tensor.decomp.qr_via_cholesky`` (decomp.py:2359-2420; default absorb there: "left")."""
    from .split import array_split

    return array_split(x, "qr:cholesky", _ABSORB_NAMES.get(absorb, absorb), None, 0.0, "rel", None, shift=shift)


def cholesky_regularized(x, absorb=0, shift=True, **_):
    """Drop-in for ``quimb.tensor.decomp.cholesky_regularized`` (decomp.py:2262-2322)."""
    from .split import array_split

    return array_split(x, "cholesky", _ABSORB_NAMES.get(absorb, absorb), None, 0.0, "rel", None, shift=shift)


def svd_rand_truncated(x, max_bond, absorb=0, oversample=10, num_iterations=2, method_lorthog="qr", method_reduced="svd",
                       right=None, lorthog_opts=None, reduced_opts=None, seed=None, **_):
    """Drop-in for ``quimb.tensor.decomp.svd_rand_truncated`` (decomp.py:1689-1868)."""
    from .split import array_split

    return array_split(x, "svd:rand", _ABSORB_NAMES.get(absorb, absorb), max_bond if max_bond and max_bond > 0 else None, 0.0,
                       "rel", None, oversample=oversample, num_iterations=num_iterations, method_lorthog=method_lorthog,
                       method_reduced=method_reduced, right=right, seed=seed)
