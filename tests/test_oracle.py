"""Pin the CPU oracle before trusting it: the reference's fixed-number KAT, an
independent exact evaluation, numpy.einsum over whole networks, and the
reference's fuse / output-index rules."""

import numpy as np
import pytest

from oracle import np_oracle as orc


def test_ising_16x16_known_answer():
    """tests/test_tensor/test_tn2d/test_core.py:309-335 in the reference:
    TN2D_classical_ising_partition_function(16, 16, 0.44) == 8.459419593253275e100."""
    arrays, inputs = orc.tn2d_classical_ising(16, 16, 0.44)
    import quimb_amd as qa  # host-only use: path helper

    path = qa.sweep_path_2d(16, 16)
    m, e = orc.oracle_array_contract(arrays, inputs, (), path=path, strip_exponent=True)
    Z = m.item() * 10.0**e
    assert Z == pytest.approx(8.459419593253275e100, rel=1e-10)
    # independent of any tensor-network code: row transfer over 2^16 configurations
    assert orc.ising_partition_exact(16, 16, 0.44) == pytest.approx(8.459419593253275e100, rel=1e-10)


@pytest.mark.parametrize("Lx,Ly,beta", [(3, 4, 0.3), (5, 5, 0.44), (2, 7, 1.0)])
def test_ising_small_vs_transfer_matrix(Lx, Ly, beta):
    arrays, inputs = orc.tn2d_classical_ising(Lx, Ly, beta)
    Z = orc.oracle_array_contract(arrays, inputs, ())
    assert Z.item() == pytest.approx(orc.ising_partition_exact(Lx, Ly, beta), rel=1e-11)


def test_pairwise_path_equals_global_einsum():
    rng = np.random.default_rng(0)
    inputs = [("a", "b", "c"), ("c", "d"), ("d", "e", "a"), ("e", "f"), ("b", "g")]
    size = dict(a=2, b=3, c=4, d=5, e=3, f=2, g=4)
    arrays = [rng.normal(size=[size[i] for i in t]) for t in inputs]
    want = np.einsum("abc,cd,dea,ef,bg->fg", *arrays)
    for path in [None, [(0, 1), (0, 1), (0, 1), (0, 1)], [(3, 4), (0, 1), (0, 1), (0, 1)], [(1, 2), (0, 2), (0, 1), (0, 1)]]:
        got = orc.oracle_array_contract(arrays, inputs, ("f", "g"), path=path)
        np.testing.assert_allclose(got, want, rtol=1e-12)
    m, e = orc.oracle_array_contract(arrays, inputs, ("f", "g"), strip_exponent=True)
    np.testing.assert_allclose(m * 10**e, want, rtol=1e-12)
    assert np.max(np.abs(m)) == pytest.approx(1.0)
    # slices sum to the whole
    got = orc.oracle_array_contract(arrays, inputs, ("f", "g"), sliced_inds=("c", "e"))
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_output_index_rule_and_scalar_unwrap():
    assert orc.gen_output_inds([0, 1, 2, 1, 2, 3]) == (0, 3)
    with pytest.raises(ValueError):
        orc.gen_output_inds([1, 1, 1])
    rng = np.random.default_rng(1)
    a, b = rng.normal(size=(2, 3, 4)), rng.normal(size=(3, 4, 2))
    s = orc.oracle_tensor_contract([(a, (0, 1, 2)), (b, (1, 2, 0))])
    assert isinstance(s, float) and s == pytest.approx(np.einsum("abc,bca->", a, b))
    data, inds, tags = orc.oracle_tensor_contract([(a, (0, 1, 2), ("red",)), (b, (1, 2, 3), ("blue",))])
    assert inds == (0, 3) and tags == ("red", "blue") and data.shape == (2, 2)


def test_fuse_rule():
    x = np.arange(2 * 3 * 4 * 5 * 6).reshape(2, 3, 4, 5, 6)
    # groups are inserted at the minimum fused axis, in the order given (array_ops.py:150-163)
    y = orc.oracle_fuse(x, (4, 2), (3, 0))
    assert y.shape == (24, 10, 3)
    np.testing.assert_array_equal(y, np.transpose(x, (4, 2, 3, 0, 1)).reshape(24, 10, 3))
    np.testing.assert_array_equal(orc.oracle_fuse(x, (1, 2)), x.reshape(2, 12, 5, 6))


def test_bench_generator_matches_oracle_generator():
    """bench.py builds its input itself (the product bench must not depend on the checker); the tensors have to be
    the oracle generator's, bit for bit -- tests/golden/full_size_oracle.json is keyed on them."""
    import importlib.util
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    from oracle import np_oracle as orc

    for (Lx, Ly, D, seed, low) in [(3, 4, 3, 7, -0.1), (4, 3, 2, 0, -0.6), (2, 2, 6, 3, -0.1)]:
        a1, i1 = bench.tn2d_rand(Lx, Ly, D, seed=seed, low=low, dtype="float32")
        a2, i2 = orc.tn2d_rand(Lx, Ly, D, seed=seed, low=low, dtype="float32")
        assert [t.shape for t in a1] == [t.shape for t in a2]
        assert all(np.array_equal(x, y) for x, y in zip(a1, a2))
        # same network structure (index names may differ): same sharing pattern
        rel = lambda ins: [[[j for j, u in enumerate(ins) if ix in u] for ix in t] for t in ins]
        assert rel(i1) == rel([tuple(t) for t in i2])


def test_oracle_default_path_is_bounded_and_huge_paths_are_refused():
    """Rounds 3 and 4 lost GPU boxes (container OOM-kill) to an oracle call WITHOUT a path on a 6x6 D=6 lattice: the old
    default order built a 6^14-element intermediate (627 GB).  The default is now a smallest-intermediate-first order,
    and any path -- the caller's too -- that would materialise more than ``MAX_INTERMEDIATE_BYTES`` raises instead."""
    import quimb_amd as qa
    from oracle import np_oracle as orc

    arrays, inputs = orc.tn2d_rand(6, 6, 6, seed=12, dtype="float64")
    inputs = [tuple(t) for t in inputs]
    size = {ix: 6 for t in inputs for ix in t}
    path = orc.small_first_path(inputs, (), size)
    tree = qa.ContractionTree(inputs, (), size, path=path)
    assert 2 ** tree.contraction_width() * 8 < 64 << 20               # < 64 MB where the old default needed 627 GB
    want = orc.oracle_array_contract(arrays, inputs, (), path=qa.sweep_path_2d(6, 6)).item()
    assert orc.oracle_array_contract(arrays, inputs, ()).item() == pytest.approx(want, rel=1e-12)
    with pytest.raises(MemoryError, match="intermediate"):
        orc.oracle_array_contract(arrays, inputs, (), path=orc.naive_path(36))
    # open and hyper indices, disconnected pieces: same values as one whole-network einsum
    rng = np.random.default_rng(0)
    xs = [rng.normal(size=s) for s in ((3, 4), (4, 5, 3), (5, 2), (2,), (6,))]
    ins = [("a", "b"), ("b", "c", "a"), ("c", "d"), ("d",), ("z",)]
    np.testing.assert_allclose(orc.oracle_array_contract(xs, ins, ("z",)),
                               np.einsum("ab,bca,cd,d,z->z", *xs), rtol=1e-12)
