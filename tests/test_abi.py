"""The C-ABI shared library loads (no GPU needed) and exports every symbol
``include/quimb_amd.h`` declares; host-only entry points behave."""

import ctypes as C
import os
import re


from quimb_amd import _lib
from quimb_amd.device import fill_plan_struct
from quimb_amd.pairwise import plan_pair

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ensure_built():
    if not os.path.exists(_lib.library_path()):
        import __graft_entry__ as g

        g.build()


def test_header_symbols_exported():
    _ensure_built()
    hdr = open(os.path.join(ROOT, "include", "quimb_amd.h")).read()
    declared = set(re.findall(r"\b(qamd_[a-z_0-9]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_dtype")}
    lib = C.CDLL(_lib.library_path())
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, missing
    assert {n for n, _, _ in _lib.SYMBOLS} == declared


def test_load_and_plan_finalize():
    _ensure_built()
    lib = _lib.load()
    assert lib.qamd_abi_version() == 1
    assert b"gfx950" in lib.qamd_build_info()
    # a boundary-sweep step in the executor's death-ordered layout:
    #   A[h, v, m] (m one contiguous run)  x  S[h, x, v, y]  ->  C[x, m, y]
    death = (("x", 1), ("m", 5), ("y", 9))
    step = plan_pair(("h", "v", "m"), (6, 6, 46656), ("h", "x", "v", "y"), (6, 6, 6, 6), ("m", "x", "y"), False, death)
    assert step.out_inds == ("x", "m", "y")
    p = fill_plan_struct(step.spec, _lib.QAMD_F32)
    assert lib.qamd_pair_plan_finalize(C.byref(p), 16, 16, 16) == 0
    assert p.vec_a == 4 and p.a_kcontig == 0 and p.c_ncontig == 1
    # big x small, A's stride-1 index in M, C = [.., m, y] -> streaming kernel with transposed stores
    assert p.kernel == 2 and p.vec_c == 4
    assert lib.qamd_pair_ktab_len(C.byref(p)) == 2 * 64  # K = 36 padded to the largest k-tile (32)
    assert lib.qamd_pair_workspace_bytes(C.byref(p)) == 0
    # caller can force the tiled kernel
    p.kernel = -1
    assert lib.qamd_pair_plan_finalize(C.byref(p), 16, 16, 16) == 0 and p.kernel == 0
    # same operands, numpy-style output order [m, x, y]: M-contiguous C is impossible, N block is 36
    step2 = plan_pair(("h", "v", "m"), (6, 6, 46656), ("h", "x", "v", "y"), (6, 6, 6, 6), ("m", "x", "y"), True)
    p2 = fill_plan_struct(step2.spec, _lib.QAMD_F32)
    assert lib.qamd_pair_plan_finalize(C.byref(p2), 16, 16, 16) == 0 and p2.kernel == 2
    # A[m, k] row-major (k contiguous) cannot stream -> tiled kernel
    step3 = plan_pair(("m", "k"), (46656, 36), ("k", "n"), (36, 36), ("m", "n"), True)
    p3 = fill_plan_struct(step3.spec, _lib.QAMD_F32)
    assert lib.qamd_pair_plan_finalize(C.byref(p3), 16, 16, 16) == 0 and p3.kernel == 0 and p3.a_kcontig == 1
    # malformed plan is rejected, not crashed on
    p.nm = 99
    assert lib.qamd_pair_plan_finalize(C.byref(p), 16, 16, 16) == -1


def test_struct_layout_matches_header():
    # int32 x8 + 13 int64[8] arrays + int32 x8
    assert C.sizeof(_lib.PairPlanStruct) == 8 * 4 + 13 * 8 * 8 + 8 * 4
    assert C.sizeof(_lib.Epilogue) == 24


def test_library_reads_no_environment():
    """"no hidden global state".  The shared library has no reference to getenv / secure_getenv at all --
    kernel choices are steered by plan inputs only (qamd_pair_plan.kernel / tile_cfg / split_k, QAMD_CHAIN2_FORCE_* flags)."""
    import subprocess

    from quimb_amd import _lib

    out = subprocess.run(["nm", "-D", "--undefined-only", _lib.library_path()], capture_output=True, text=True)
    if out.returncode != 0:
        pytest.skip("nm not available")
    assert "getenv" not in out.stdout, [ln for ln in out.stdout.splitlines() if "getenv" in ln]
