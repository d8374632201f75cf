"""ctypes binding of ``libquimb_amd.so`` (the C-ABI declared in
``include/quimb_amd.h``).  The product path has no CPU fallback: if the
library is missing this module raises at first use.
"""

import ctypes as C
import os

QAMD_MAX_GROUPS = 8
QAMD_MAX_NDIM = 32
QAMD_F32, QAMD_F64, QAMD_C64, QAMD_C128 = 0, 1, 2, 3

_G = QAMD_MAX_GROUPS
_I64G = C.c_int64 * _G


class PairPlanStruct(C.Structure):
    """Mirror of ``qamd_pair_plan``."""

    _fields_ = [
        ("dtype", C.c_int32),
        ("nb", C.c_int32),
        ("nm", C.c_int32),
        ("nn", C.c_int32),
        ("nk", C.c_int32),
        ("conj_a", C.c_int32),
        ("conj_b", C.c_int32),
        ("kernel", C.c_int32),
        ("dim_b", _I64G), ("sa_b", _I64G), ("sb_b", _I64G), ("sc_b", _I64G),
        ("dim_m", _I64G), ("sa_m", _I64G), ("sc_m", _I64G),
        ("dim_n", _I64G), ("sb_n", _I64G), ("sc_n", _I64G),
        ("dim_k", _I64G), ("sa_k", _I64G), ("sb_k", _I64G),
        ("tile_cfg", C.c_int32),
        ("split_k", C.c_int32),
        ("vec_a", C.c_int32),
        ("vec_b", C.c_int32),
        ("a_kcontig", C.c_int32),
        ("b_kcontig", C.c_int32),
        ("c_ncontig", C.c_int32),
        ("vec_c", C.c_int32),
    ]


class Chain2PlanStruct(C.Structure):
    """Mirror of ``qamd_chain2_plan``."""

    _fields_ = [
        ("dtype", C.c_int32), ("D", C.c_int32), ("nm", C.c_int32), ("flags", C.c_int32),
        ("dim_m", _I64G), ("sa_m", _I64G), ("sc_m", _I64G),
        ("sa_v", C.c_int64),
        ("w1_strides", C.c_int64 * 4), ("w2_strides", C.c_int64 * 4),
    ]


class RowpassPlanStruct(C.Structure):
    """Mirror of ``qamd_rowpass_plan``."""

    _fields_ = [
        ("dtype", C.c_int32), ("D", C.c_int32), ("nsites", C.c_int32), ("nS", C.c_int32),
        ("sv", C.c_int64 * 5), ("sd", C.c_int64 * 5), ("sh", C.c_int64),
        ("dim_s", C.c_int64 * 4), ("sa_s", C.c_int64 * 4), ("sc_s", C.c_int64 * 4),
        ("w_strides", (C.c_int64 * 4) * 5),
        ("ed", C.c_int32 * 5), ("eh", C.c_int32), ("kernel", C.c_int32), ("pad_", C.c_int32),
    ]


class Epilogue(C.Structure):
    """Mirror of ``qamd_epilogue``."""

    _fields_ = [("scale_a", C.c_void_p), ("scale_b", C.c_void_p), ("absmax_out", C.c_void_p)]


ABSMAX_SLOTS = 64

#: every symbol ``include/quimb_amd.h`` declares: (name, restype, argtypes)
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_pi64 = C.POINTER(C.c_int64)
_pplan = C.POINTER(PairPlanStruct)
SYMBOLS = [
    ("qamd_abi_version", C.c_int, []),
    ("qamd_build_info", C.c_char_p, []),
    ("qamd_pair_plan_finalize", C.c_int, [_pplan, _i64, _i64, _i64]),
    ("qamd_pair_ktab_len", _i64, [_pplan]),
    ("qamd_pair_build_ktab", C.c_int, [_pplan, _vp, _vp]),
    ("qamd_pair_workspace_bytes", _i64, [_pplan]),
    ("qamd_contract_pair", C.c_int, [_pplan, _vp, _vp, _vp, _vp, _vp, _i64, _vp]),
    ("qamd_pair_describe", C.c_int, [_pplan, C.c_char_p, _i32]),
    ("qamd_contract_pair_ex", C.c_int, [_pplan, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(Epilogue), _vp]),
    ("qamd_chain2_chunk", C.c_int, [_i32, _i32]),
    ("qamd_chain2_describe", C.c_int, [C.POINTER(Chain2PlanStruct), C.c_char_p, _i32]),
    ("qamd_contract_chain2", C.c_int, [C.POINTER(Chain2PlanStruct), _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("qamd_absmax_log10_sum", C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    ("qamd_div_by_absmax", C.c_int, [_vp, _i64, _vp, _i32, _vp]),
    ("qamd_permute", C.c_int, [_vp, _vp, _i32, _pi64, _pi64, _i64, _i32, _vp]),
    ("qamd_reduce_sum", C.c_int, [_vp, _vp, _i32, _pi64, _pi64, _i32, _pi64, _pi64, _i32, _vp]),
    ("qamd_binary", C.c_int, [_vp, _vp, _pi64, _vp, _pi64, _i32, _pi64, _i32, _i32, _vp]),
    ("qamd_scale", C.c_int, [_vp, _i64, _dbl, _dbl, _i32, _vp]),
    ("qamd_axpby", C.c_int, [_vp, _vp, _i64, _dbl, _dbl, _i32, _vp]),
    ("qamd_axpby_exp", C.c_int, [_vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    ("qamd_conj", C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    ("qamd_cast", C.c_int, [_vp, _i32, _vp, _i32, _i64, _vp]),
    ("qamd_fill", C.c_int, [_vp, _i64, _dbl, _dbl, _i32, _vp]),
    ("qamd_complex_expand", C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    ("qamd_strip_exponent", C.c_int, [_vp, _i64, _i32, _vp, _vp, _vp]),
    ("qamd_absmax", C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    ("qamd_microtree_run", C.c_int, [_i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32, _vp]),
    ("qamd_microtree_run_ex", C.c_int, [_i32, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _i64, _vp, _i64, _i32, _i32, _vp]),
    ("qamd_unary", C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    ("qamd_minmax", C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    ("qamd_absmax_log10_sum_add", C.c_int, [_vp, _i64, _i32, _vp, _vp]),
    # a join consumed by one inner product (gemmk.hip, DOT variant)
    ("qamd_pair_dot_workspace_bytes", _i64, [_vp]),
    ("qamd_rowpass_supported", C.c_int, [_i32, _i32, _i32]),
    ("qamd_contract_rowpass", C.c_int, [C.POINTER(RowpassPlanStruct), _vp, C.POINTER(C.c_void_p), _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp]),
    ("qamd_contract_pair_dot", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    # the vector work of a Lanczos step (krylov.hip)
    ("qamd_krylov_workspace_bytes", _i64, [_i32, _i64, _i32]),
    ("qamd_krylov_project", C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp]),
    ("qamd_krylov_subtract", C.c_int, [_vp, _vp, _i64, _i32, _vp, _i64, _i32, _i32, _vp, _vp]),
    ("qamd_krylov_extend", C.c_int, [_vp, _vp, _i64, _vp, _vp, C.c_double, _i32, _vp, _vp]),
    # launch programs (record once, replay with one host call)
    ("qamd_program_create", _vp, [_i32]),
    ("qamd_program_destroy", None, [_vp]),
    ("qamd_program_record_begin", C.c_int, [_vp]),
    ("qamd_program_set_lane", C.c_int, [_vp, _i32]),
    ("qamd_program_wait", C.c_int, [_vp, _i32, _i32]),
    ("qamd_program_mark", C.c_int, [_vp, _i32]),
    ("qamd_program_record_end", C.c_int, [_vp]),
    ("qamd_program_bind_inputs", C.c_int, [_vp, _i32, C.POINTER(C.c_void_p), _pi64]),
    ("qamd_program_num_ops", _i32, [_vp]),
    ("qamd_program_num_launches", _i32, [_vp]),
    ("qamd_program_num_marks", _i32, [_vp]),
    ("qamd_program_run", C.c_int, [_vp, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _i32]),
    ("qamd_program_mark_ms", C.c_int, [_vp, _i32, _i32, C.POINTER(C.c_int32), C.POINTER(C.c_float)]),
]

_ERRORS = {
    -1: "QAMD_EINVAL (malformed plan / argument)",
    -2: "QAMD_EUNSUPPORTED",
    -3: "QAMD_EWORKSPACE (workspace too small)",
    -4: "QAMD_ELAUNCH (kernel launch failed)",
}


class QamdError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise QamdError(f"{what} failed: {_ERRORS.get(rc, rc)}")


def library_path():
    # QAMD_LIBRARY: an experiment build of the SAME library (e.g. -DQAMD_CHAIN2_TIMING), see scripts/README.md
    return os.environ.get("QAMD_LIBRARY") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libquimb_amd.so")


_LIB = None


def load():
    """Load the shared library and bind every declared symbol. Raises
    ``QamdError`` (never falls back) if it is missing or incomplete."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise QamdError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C quimb_amd/csrc` -- quimb_amd has no CPU fallback"
        )
    # torch bundles its own libamdhip64; import it FIRST so that this library binds
    # to the same already-loaded HIP runtime (two runtimes in one process cannot
    # share streams / allocations and every launch fails).
    import torch  # noqa: F401

    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise QamdError(f"{path} does not export {name}") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.qamd_abi_version() != 1:
        raise QamdError("libquimb_amd.so ABI version mismatch")
    _LIB = lib
    return lib
