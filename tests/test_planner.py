"""The host planner of the C-ABI library (qamd_pair_plan_finalize: pure host code, no GPU needed): which kernel a
pairwise contraction is sent to.  Pins the round-3 rules for the k-outer MFMA kernel (gemmk.hip): the headline joins
and their rank shards take it with the tile the measurements favour, everything it does not cover keeps the older
kernels."""
import ctypes as C
import os

import numpy as np
import pytest


def _describe(a_inds, a_shape, b_inds, b_shape, out, dtype="float32", env=None, aligns=(16, 16, 16)):
    from quimb_amd import _lib
    from quimb_amd.device import dtype_code, fill_plan_struct
    from quimb_amd.pairwise import plan_pair

    lib = _lib.load()
    st = plan_pair(tuple(a_inds), tuple(a_shape), tuple(b_inds), tuple(b_shape), tuple(out), True)
    p = fill_plan_struct(st.spec, dtype_code(np.dtype(dtype)))
    p.tile_cfg, p.split_k, p.kernel = -1, 0, 0
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        assert lib.qamd_pair_plan_finalize(C.byref(p), *aligns) == 0
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    buf = C.create_string_buffer(200)
    lib.qamd_pair_describe(C.byref(p), buf, 200)
    return buf.value.decode(), p


@pytest.mark.parametrize("m,n,k,want", [
    (7776, 7776, 7776, "gemmk_kernel<3, 4, 3, 1>"),      # the joins of the 10x10 D=6 quadrant tree: 1271 tiles = 4.96 rounds
    (3888, 1944, 7776, "gemmk_kernel<2, 4, 3, 1>"),      # one rank of eight
    (3888, 3888, 7776, None),                            # one rank of four: any gemmk tile
    (8192, 8192, 8192, "gemmk_kernel<4, 4, 3, 1>"),
    (4096, 4096, 4096, "gemmk_kernel<4, 4, 3, 1>"),
    (7776, 7776, 216, None),                             # K % 16 == 8: the leading half tile
])
def test_k_outer_joins_take_gemmk(m, n, k, want):
    name, p = _describe("km", (k, m), "kn", (k, n), "mn")
    assert p.kernel == 5 and name.startswith("gemmk_kernel<"), name
    if want:
        assert name == want
    ta, tb = p.tile_cfg // 16, p.tile_cfg % 16
    assert 2 <= ta <= 4 and 2 <= tb <= 4 and p.split_k == 1


def test_what_gemmk_does_not_cover_keeps_the_older_kernels():
    k5 = lambda *a, **kw: _describe(*a, **kw)[1].kernel == 5
    assert k5("km", (512, 2048), "kn", (512, 2048), "mn")
    assert not k5("km", (512, 2048), "kn", (512, 2048), "mn", dtype="float64")          # fp32 only
    assert not k5("mk", (2048, 512), "kn", (512, 2048), "mn")                            # A contiguous along k
    assert not k5("km", (512, 2048), "nk", (2048, 512), "mn")                            # B contiguous along k
    assert not k5("km", (100, 2048), "kn", (100, 2048), "mn")                            # K % 8
    assert not k5("km", (32, 4096), "kn", (32, 4096), "mn")                              # K < 64
    assert not k5("km", (512, 2048), "kn", (512, 64), "mn")                              # N < 128: streaming / tiled kernels
    assert not k5("km", (512, 512), "kn", (512, 512), "mn")                              # 16 tiles: split-K kernels fill the chip
    assert not k5("km", (512, 2046), "kn", (512, 2048), "mn")                            # M % 4
    assert not k5("km", (512, 2048), "kn", (512, 2048), "mn", env={"QAMD_GEMMK": "0"})   # switched off
    assert not k5("km", (512, 2048), "kn", (512, 2048), "mn", aligns=(8, 16, 16))        # operand not 16-byte aligned
    name, p = _describe("km", (512, 512), "kn", (512, 512), "mn", env={"QAMD_GEMMK_TILE": "22"})
    assert p.kernel == 5 and name == "gemmk_kernel<2, 2, 3, 2>"                          # pinning overrides the tile floor
    # C contiguous along m: the operands swap roles inside the launch, the plan is the same kind
    assert k5("km", (512, 2048), "kn", (512, 4096), "nm")
    # tensor addressing: free bundles of two groups each, batch bundle
    assert k5("kab", (256, 32, 64), "kcd", (256, 64, 32), "acbd")
    assert k5("bkm", (3, 512, 1024), "bkn", (3, 512, 1024), "bmn")
