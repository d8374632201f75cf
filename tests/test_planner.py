"""The host planner of the C-ABI library (qamd_pair_plan_finalize: pure host code, no GPU needed): which kernel a
pairwise contraction is sent to.  Pins the round-3 rules for the k-outer MFMA kernel (gemmk.hip): the headline joins
and their rank shards take it with the tile the measurements favour, everything it does not cover keeps the older
kernels."""
import ctypes as C
import os

import numpy as np
import pytest


def _describe(a_inds, a_shape, b_inds, b_shape, out, dtype="float32", pin=None, aligns=(16, 16, 16)):
    """``pin``: (kernel, tile_cfg) set on the plan BEFORE finalize -- the library's only steering input (it reads no
    environment variable): kernel -2 = no MFMA GEMM kernels, -5 / -6 = gemmk / gemmd with tile 16 ta + tb."""
    from quimb_amd import _lib
    from quimb_amd.device import dtype_code, fill_plan_struct
    from quimb_amd.pairwise import plan_pair

    lib = _lib.load()
    st = plan_pair(tuple(a_inds), tuple(a_shape), tuple(b_inds), tuple(b_shape), tuple(out), True)
    p = fill_plan_struct(st.spec, dtype_code(np.dtype(dtype)))
    p.tile_cfg, p.split_k, p.kernel = -1, 0, 0
    if pin is not None:
        p.kernel, p.tile_cfg = pin
    assert lib.qamd_pair_plan_finalize(C.byref(p), *aligns) == 0
    buf = C.create_string_buffer(200)
    lib.qamd_pair_describe(C.byref(p), buf, 200)
    return buf.value.decode(), p


@pytest.mark.parametrize("m,n,k,want", [
    (7776, 7776, 7776, "gemmk_kernel<3, 4, 2, 2>"),      # the joins of the 10x10 D=6 quadrant tree: 1271 tiles = 4.96 rounds
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
    assert not k5("km", (512, 2048), "kn", (512, 2048), "mn", pin=(-2, -1))               # pinned away from it
    assert not k5("km", (512, 2048), "kn", (512, 2048), "mn", aligns=(8, 16, 16))        # operand not 16-byte aligned
    name, p = _describe("km", (512, 512), "kn", (512, 512), "mn", pin=(-5, 16 * 2 + 2))
    assert p.kernel == 5 and name == "gemmk_kernel<2, 2, 3, 2>"                          # pinning overrides the tile floor
    # C contiguous along m: the operands swap roles inside the launch, the plan is the same kind
    assert k5("km", (512, 2048), "kn", (512, 4096), "nm")
    # tensor addressing: free bundles of two groups each, batch bundle
    assert k5("kab", (256, 32, 64), "kcd", (256, 64, 32), "acbd")
    assert k5("bkm", (3, 512, 1024), "bkn", (3, 512, 1024), "bmn")


# ---- round 4: the fp64 kernel on the LDS-DMA ring (gemmd.hip) ------------------------------------------------------------
def test_fp64_gemm_shapes_take_gemmd():
    """The two GEMM-shaped products of the chi = 512 effective-Hamiltonian matvec (BASELINE config #5) in the layouts the
    executor gives them, and square GEMMs in every operand layout: the tile that makes ONE round of 256 CUs, k slabs where
    the grid is under-filled, the store order re-arranged for a k-contiguous operand."""
    chi, w, d = 512, 5, 2
    # first product: L[a, p, A] . x[A, S1, S2, B]  (A k-contiguous in L), result written as [S1, p, S2, B, a]
    name, p = _describe("apA", (chi, w, chi), "AstB", (chi, d, d, chi), "sptBa", dtype="float64")
    assert p.kernel == 6 and name == "gemmd_kernel<5, 2, true, false, true> split_k=1", name      # 160 x 128: 16 x 16 tiles
    assert (p.dim_m[p.nm - 1], p.sc_m[p.nm - 1]) == (chi, 1)       # a -- C's stride-1 index -- moved innermost in M: lanes store runs
    # last product: T[r, B, a, s1, s2] . R[b, r, B]: K = (r, B), 64 tiles of 128 x 128 -> 4 k slabs
    name, p = _describe("rBast", (w, chi, chi, d, d), "brB", (chi, w, chi), "astb", dtype="float64")
    assert p.kernel == 6 and name == "gemmd_kernel<4, 2, false, true, false> split_k=4", name
    for a_inds, b_inds in (("mk", "kn"), ("km", "kn"), ("mk", "nk"), ("km", "nk")):
        shape = lambda t: tuple(4096 for _ in t)
        name, p = _describe(a_inds, shape(a_inds), b_inds, shape(b_inds), "mn", dtype="float64")
        assert p.kernel == 6 and name.startswith("gemmd_kernel<4, 2,") and p.split_k == 1, name
        assert (p.a_kcontig, p.b_kcontig) == (int(a_inds == "mk"), int(b_inds == "nk"))


def test_what_gemmd_does_not_cover():
    k6 = lambda *a, **kw: _describe(*a, dtype=kw.pop("dtype", "float64"), **kw)[1].kernel == 6
    assert k6("mk", (2048, 512), "kn", (512, 2048), "mn")
    assert not k6("mk", (2048, 512), "kn", (512, 2048), "mn", dtype="float32")            # fp64 only
    assert not k6("mk", (2048, 520), "kn", (520, 2048), "mn")                              # K % 16
    assert not k6("mk", (2048, 48), "kn", (48, 2048), "mn")                                # K < 64
    assert not k6("mk", (2047, 512), "kn", (512, 2048), "mn")                              # odd M: granules of two elements
    assert not k6("mk", (2048, 512), "kn", (512, 2048), "mn", pin=(-2, -1))                # pinned away from it
    assert not k6("mk", (2048, 512), "kn", (512, 2048), "mn", aligns=(8, 16, 16))         # operand not 16-byte aligned
    assert not k6("mk", (128, 512), "kn", (512, 128), "mn")                                # tiny grid: the generic kernels
    # K in two groups that cannot fuse (their order differs between the operands): the innermost one holds the k-tiles
    assert not k6("muk", (2048, 32, 24), "kun", (24, 32, 2048), "mn")                      # ... 24: not a multiple of 16
    assert k6("muk", (2048, 24, 32), "kun", (32, 24, 2048), "mn")                          # ... 32: fine, the outer group is free
    name, p = _describe("mk", (128, 512), "kn", (512, 128), "mn", dtype="float64", pin=(-6, 16 * 2 + 1))
    assert p.kernel == 6 and name.startswith("gemmd_kernel<2, 1, true, false, false>")     # pinning overrides the floors


def test_join_dot_workspace_follows_the_gemmk_tiles():
    """qamd_contract_pair_dot (the second join + closing inner product in one launch) needs one double per workgroup of
    the k-outer kernel; plans on other kernels report 0 = not supported."""
    from quimb_amd import _lib

    lib = _lib.load()
    name, p = _describe("km", (7776, 7776), "kn", (7776, 7776), "mn")
    assert name == "gemmk_kernel<3, 4, 2, 2>"
    assert lib.qamd_pair_dot_workspace_bytes(C.byref(p)) == 8 * 41 * 31
    name, p = _describe("km", (7776, 3888), "kn", (7776, 1944), "mn")          # one rank of eight: 128 x 256 tiles
    assert lib.qamd_pair_dot_workspace_bytes(C.byref(p)) == 8 * 31 * 8
    name, p = _describe("km", (512, 2048), "kn", (512, 2048), "mn", dtype="float64")
    assert p.kernel != 5 and lib.qamd_pair_dot_workspace_bytes(C.byref(p)) == 0
    # without a device nothing can be launched, but the refusal of an unsupported plan is host logic
    one = (C.c_float * 4)()
    assert lib.qamd_contract_pair_dot(C.byref(p), one, one, one, one, None, 0, None, None, None) == -2      # QAMD_EUNSUPPORTED


def test_split_product_joins_are_opt_in():
    """Kernel 7 (gemmh.hip: fp32 joins as three exact fp16 products on the f16 matrix pipe) is never the planner's own choice:
    only plan.kernel = -7 on input selects it, for the shapes that pay for the split pass; everything else falls through to
    the automatic choice.  The workspace holds the operands' split images."""
    from quimb_amd import _lib

    lib = _lib.load()
    name, p = _describe("km", (7776, 7776), "kn", (7776, 7776), "mn")
    assert p.kernel == 5                                                                  # default: the fp32 MFMA kernel
    name, p = _describe("km", (7776, 7776), "kn", (7776, 7776), "mn", pin=(-7, -1))
    assert p.kernel == 7 and name == "gemmh8_kernel<4, 4> f16x3", name                    # 961 tiles of 256 x 256 on the eight-wave kernel
    mpad, npad, kpad = 31 * 256, 31 * 256, 7776
    img = lambda x: 2 * (kpad // 8) * x * 16       # both fp16 halves, 16 bytes per (k-group of 8, column)
    means = lambda x: (2 + 6 * 32) * x * 8 + 2 * 4 * x   # per parity of k: a column's mean (double), 3 x 32 partial sums, the fp32 constant subtracted
    assert lib.qamd_pair_workspace_bytes(C.byref(p)) == 1024 + means(mpad) + means(npad) + img(mpad) + img(npad)
    assert lib.qamd_pair_dot_workspace_bytes(C.byref(p)) == (8 * 31 * 31 + 255) // 256 * 256 + lib.qamd_pair_workspace_bytes(C.byref(p))
    # without its workspace the call is refused before anything is launched (host logic: no device needed)
    one = (C.c_float * 4)()
    assert lib.qamd_contract_pair_ex(C.byref(p), one, one, one, None, None, 0, None, None) == -3          # QAMD_EWORKSPACE
    assert lib.qamd_contract_pair_dot(C.byref(p), one, one, one, one, None, 0, None, None, None) == -3
    name, p = _describe("km", (7776, 3888), "kn", (7776, 1944), "mn", pin=(-7, -1))       # one rank of eight: one round of 128 x 256
    assert name in ("gemmh8_kernel<4, 2> f16x3", "gemmh8_kernel<2, 4> f16x3"), name
    name, p = _describe("km", (8192, 8192), "kn", (8192, 8192), "mn", pin=(-7, 16 * 4 + 4))
    assert name == "gemmh8_kernel<4, 4> f16x3"
    name, p = _describe("km", (300, 7000), "kn", (300, 5000), "mn", pin=(-8, -1))         # K % 32 != 0: zero-padded images (not a gemmk join: -8)
    assert p.kernel == 7 and lib.qamd_pair_workspace_bytes(C.byref(p)) > 0
    # not covered -> the automatic choice, as if the pin were 0
    # kernel = -8: any operand layout (the split pass gathers with the operands' own strides): k-contiguous operands, K in groups;
    # kernel = -7 leaves those to their blocked fp32 kernels
    for args in (("mk", (2048, 512), "kn", (512, 2048), "mn"), ("km", (512, 2048), "nk", (2048, 512), "mn"),
                 ("mk", (2048, 512), "nk", (2048, 512), "nm"), ("muk", (2048, 24, 32), "kun", (32, 24, 2048), "mn")):
        name8, p8 = _describe(*args, pin=(-8, -1))
        assert p8.kernel == 7 and name8.startswith("gemmh"), (args, name8)
        name7, p7 = _describe(*args, pin=(-7, -1))
        name0, p0 = _describe(*args)
        assert p7.kernel != 7 and (p7.kernel, name7) == (p0.kernel, name0), (args, name7, name0)
    name8, p8 = _describe("km", (7776, 7776), "kn", (7776, 7776), "mn", pin=(-8, -1))
    assert p8.kernel == 7 and name8 == "gemmh8_kernel<4, 4> f16x3"
    # -7 is granted only where the default would have been the chain kernel gemmk
    name7, p7 = _describe("km", (512, 512), "kn", (512, 512), "mn", pin=(-7, -1))         # 16 tiles: split-K kernels by default
    assert p7.kernel != 7 and _describe("km", (512, 512), "kn", (512, 512), "mn", pin=(-8, -1))[1].kernel == 7
    for args, kw in ((("km", (128, 4096), "kn", (128, 4096), "mn"), {}),                  # K < 256: not worth the split pass
                     (("km", (512, 2048), "kn", (512, 200), "mn"), {}),                   # N < 256
                     (("hvm", (6, 6, 46656), "hxvy", (6, 6, 6, 6), "mxy"), {}),           # big x small: the streaming kernels
                     (("bkm", (3, 512, 512), "bkn", (3, 512, 512), "bmn"), {}),           # batch bundle
                     (("km", (512, 2048), "kn", (512, 2048), "mn"), dict(dtype="float64"))):
        name7, p7 = _describe(*args, pin=(-7, -1), **kw)
        name0, p0 = _describe(*args, **kw)
        assert p7.kernel != 7 and (p7.kernel, p7.tile_cfg, p7.split_k, name7) == (p0.kernel, p0.tile_cfg, p0.split_k, name0), (args, name7, name0)


def test_join_arith_option_pins_kernel_minus_seven():
    import quimb_amd as qa
    from quimb_amd.options import Options

    assert Options().join_arith == "f32"
    assert Options.from_env({"QAMD_JOIN_ARITH": "f16x3"}).join_arith == "f16x3"
    assert Options(join_arith="f16x3-all").join_arith == "f16x3-all"
    with qa.exec_options(join_arith="f16x3") as o:
        assert o.join_arith == "f16x3" and qa.get_options().join_arith == "f16x3"
    assert qa.get_options().join_arith == "f32"
    with pytest.raises(ValueError):
        Options(join_arith="bf16")
    with pytest.raises(ValueError):
        qa.get_options().replace(join_arith="fast")
