"""Golden vectors from the real quimb (tests/golden/) against (a) the oracle,
(b) the host logic on the numpy plan interpreter, (c) the HIP path (-m gpu)."""

import numpy as np
import pytest

import checks
from oracle import np_oracle as orc


def _product_fns():
    import quimb_amd as qa

    def contract(arrays, inputs, output):
        return qa.array_contract(arrays, inputs, output)

    def tensor_contract(ts):
        r = qa.tensor_contract(*[qa.Tensor(*t) for t in ts])
        if isinstance(r, qa.Tensor):
            return (np.asarray(r.data), r.inds, r.tags)
        return r

    return dict(
        contract=contract,
        tensor_contract=tensor_contract,
        fuse=lambda x, *g: qa.fuse(qa.asarray(x), *g).to_numpy(),
        transpose=lambda x, p: qa.transpose(qa.asarray(x), p).to_numpy(),
        getitem=lambda x, k: qa.asarray(x)[k].to_numpy(),
    )


def test_oracle_matches_golden():
    checks.check_golden(
        contract=lambda a, i, o: orc.oracle_array_contract(a, i, o),
        tensor_contract=lambda ts: orc.oracle_tensor_contract(ts),
        fuse=orc.oracle_fuse,
        transpose=np.transpose,
        getitem=lambda x, k: x[k],
    )


def test_host_logic_matches_golden(emu):
    checks.check_golden(**_product_fns())


@pytest.mark.gpu
def test_hip_matches_golden(hip):
    checks.check_golden(**_product_fns(), rtol=1e-10)


def test_boundary_oracle_matches_real_quimb():
    """The restated boundary contraction reproduces the real quimb's ``contract_boundary`` values (made by
    tests/golden/make_golden.py through the stand-ins) to rounding level, square and rectangular lattices,
    plain and stripped."""

    def fn(arrs, Lx, Ly, strip_exponent=False, **kw):
        m, e = orc.oracle_contract_boundary_2d(arrs, Lx, Ly, **kw)
        if strip_exponent:
            return m / abs(m), e + np.log10(abs(m))
        return m * 10.0**e

    checks.check_boundary_golden(fn, 1e-12)


def test_boundary_host_logic_matches_real_quimb(emu):
    checks.check_boundary("float64")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_boundary_hip_matches_real_quimb(hip, dtype):
    checks.check_boundary(dtype)


def test_dmrg2_host_logic_matches_real_quimb(emu):
    """Same MPO tensors, same schedule as a run of the real quimb's DMRG2: same converged energy and bond."""
    checks.check_dmrg("float64")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_dmrg2_hip_matches_real_quimb(hip, dtype):
    checks.check_dmrg(dtype)


def test_split_policy_matches_real_quimb(emu):
    """Kept rank, singular values and factor products of ``tensor_split`` for every cutoff mode / absorb / method
    of the golden grid (tests/golden/split.npz, made by the real quimb)."""
    checks.check_split("float64")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_split_hip_matches_real_quimb(hip, dtype):
    checks.check_split(dtype)


def test_decomp_drivers_match_real_quimb(emu):
    """"qr:cholesky" / "cholesky" / "svd:rand" / "rsvd" (host logic on the plan interpreter) vs the real quimb's drivers."""
    checks.check_decomp_drivers("float64")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_decomp_drivers_hip_match_real_quimb(hip, dtype):
    checks.check_decomp_drivers(dtype)


def test_decomp_drivers_at_chi512_host_logic(emu):
    checks.check_decomp_full_chi()


@pytest.mark.gpu
def test_decomp_drivers_at_chi512(hip):
    """Cholesky-QR of a 1024 x 512 site matrix and the sketch split of a 1024 x 1024 two-site tensor on the device."""
    checks.check_decomp_full_chi()


def test_circuits_host_logic_match_real_quimb(emu):
    """``Circuit`` (dense state, amplitudes, batched amplitudes) and ``CircuitMPS`` (exact, truncated to chi = 4,
    non-local gates through swaps) reproduce the real quimb's states on the same gate lists."""
    checks.check_circuits("complex128")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", ["complex128", "complex64"])
def test_circuits_hip_match_real_quimb(hip, dtype):
    checks.check_circuits(dtype)


def test_local_ops_match_real_quimb(emu):
    """Tensor.gate / contract_between / contract_ind / trace of the mirrors vs the REAL quimb (tests/golden/local.npz)."""
    checks.check_golden_local()


@pytest.mark.gpu
def test_local_ops_hip_match_real_quimb(hip):
    checks.check_golden_local(rtol=1e-11)


def test_round3_trees_match_real_quimb_values(emu):
    """The real quimb's values of two golden lattices (``TN2D_rand(4,4,3,seed=42).contract(all)``, the 6x6 Ising
    partition function) through the round-3 ways of contracting them: the four-quadrant tree, a tree FOUND by recursive
    bisection / the time objective, range slices chosen by cost, and the quadrant tree sharded into rank blocks
    (every rank's share evaluated here, summed)."""
    import quimb_amd as qa
    from quimb_amd.quadrants import QuadrantRank, QuadrantSharding, combine_pairs
    from quimb_amd.rangeslice import RangeSliced, RangeSlicedExecutor, find_range_slices

    for name, L in (("tn2d_rand_4x4_D3", 4), ("ising_6x6_b044", 6)):
        g = checks.load_golden(name)
        arrays, inputs, want = g["arrays"], [tuple(t) for t in g["inputs"]], g["value"].item()
        size = {ix: d for t, a in zip(inputs, arrays) for ix, d in zip(t, a.shape)}
        # quimb's builders order a site's indices by direction, not by neighbour: rebuild the row-major site order check
        assert len(arrays) == L * L
        quad = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(L, L))
        assert qa.TreeExecutor(quad, "float64")(arrays).to_numpy().item() == pytest.approx(want, rel=1e-11)
        for strategy in ("bisection", "auto-time"):
            found = qa.find_path(inputs, (), size, strategy)
            assert qa.TreeExecutor(found, "float64")(arrays).to_numpy().item() == pytest.approx(want, rel=1e-11)
        d_cut = min(size.values())
        for n in (2, d_cut):
            rse = RangeSlicedExecutor(RangeSliced(quad, find_range_slices(quad, n)), "float64")
            assert rse(arrays).item() == pytest.approx(want, rel=1e-11)
        for world in (2, 4) if d_cut % 2 == 0 else (3,):
            sh = QuadrantSharding(inputs, size, L, L, world)
            pairs = []
            for r in range(world):
                m, e = QuadrantRank(sh, r, "float64")(sh.shard(arrays, r))
                pairs.append((m.to_numpy().item(), e))
            assert combine_pairs(pairs) == pytest.approx(want, rel=1e-11)
