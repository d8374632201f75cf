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
    return "quimb_amd"
