"""Range slicing (quimb_amd/rangeslice.py): sliced indices in RANGES, chosen by cost, for any tree -- the identity the
reference pins for single-value slices (sum over slices == full contraction, tests/test_tensor/test_tensor_core.py:325-330)
restated for ranges, on the plan interpreter; and the choice it makes on the headline tree."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("path_kind", ["quadrant", "sweep", "greedy"])
@pytest.mark.parametrize("nslices", [1, 2, 3, 4, 6, 8, 12])
def test_range_slices_sum_to_the_network(emu, path_kind, nslices):
    from oracle import np_oracle as orc
    import quimb_amd as qa
    from quimb_amd.rangeslice import RangeSliced, RangeSlicedExecutor, find_range_slices

    arrays, inputs = orc.tn2d_rand(4, 5, 4, seed=2, dtype="float64")
    inputs = [tuple(t) for t in inputs]
    size = {ix: 4 for t in inputs for ix in t}
    want = orc.oracle_array_contract(arrays, inputs, ()).item()
    if path_kind == "greedy":
        tree = qa.find_path(inputs, (), size, "greedy")
    else:
        tree = qa.ContractionTree(inputs, (), size, path=(qa.quadrant_path_2d if path_kind == "quadrant" else qa.sweep_path_2d)(4, 5))
    parts = find_range_slices(tree, nslices)
    rs = RangeSliced(tree, parts)
    assert rs.nslices == nslices and int(np.prod(list(parts.values()) or [1])) == nslices
    rse = RangeSlicedExecutor(rs, "float64")
    assert rse(arrays).item() == pytest.approx(want, rel=1e-12)
    m, e = rse(arrays, strip_exponent=True)
    assert m * 10.0**e == pytest.approx(want, rel=1e-12)
    # a subset of the slices is a partial sum: two disjoint halves add up
    if nslices > 1:
        a = rse(arrays, slices=range(0, nslices, 2)).item()
        b = rse(arrays, slices=range(1, nslices, 2)).item()
        assert a + b == pytest.approx(want, rel=1e-12)
    # equal ranges share ONE plan
    if all(4 % k == 0 for k in parts.values()):
        assert len(rse._by_sizes) == 1


def test_range_slices_with_open_and_hyper_indices(emu):
    """Output indices are never split; a hyper index (three tensors) may be; a tensor output sums element-wise."""
    import quimb_amd as qa
    from quimb_amd.rangeslice import RangeSliced, RangeSlicedExecutor, find_range_slices

    rng = np.random.default_rng(3)
    inputs = [("a", "b", "h"), ("b", "c", "h"), ("c", "d", "h"), ("d", "e")]
    size = dict(a=6, b=4, c=6, d=4, e=3, h=2)
    arrays = [rng.uniform(-0.5, 1.0, size=[size[i] for i in t]) for t in inputs]
    out = ("a", "e")
    want = np.einsum("abh,bch,cdh,de->ae", *arrays)
    tree = qa.find_path(inputs, out, size, "greedy")
    for n in (2, 4, 6):
        parts = find_range_slices(tree, n)
        assert not set(parts) & set(out)
        got = RangeSlicedExecutor(RangeSliced(tree, parts), "float64")(arrays)
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    with pytest.raises(ValueError):
        RangeSliced(tree, {"a": 2})                       # an output index
    with pytest.raises(ValueError):
        RangeSliced(tree, {"h": 3})                       # more ranges than values
    with pytest.raises(ValueError):
        find_range_slices(tree, 6 * 4 * 6 * 4 * 2 * 7)    # does not fit


def test_range_slices_choose_the_cut_bonds_of_the_headline_tree():
    """Cost alone leads to the geometry ``QuadrantSharding`` writes down: on the 10x10 D=6 quadrant tree 2 / 4 / 8 range
    slices take halves of the cut bonds in the outermost columns, with the same inflation and busiest-slice fraction;
    the same number of range slices of the SWEEP costs 6.3x (single values of three bonds, round 1: 140x)."""
    from oracle import np_oracle as orc
    import quimb_amd as qa
    from quimb_amd.quadrants import QuadrantSharding
    from quimb_amd.rangeslice import RangeSliced, find_range_slices

    _, inputs = orc.tn2d_rand(10, 10, 2, seed=0)
    inputs = [tuple(t) for t in inputs]
    size = {ix: 6 for t in inputs for ix in t}
    quad = qa.ContractionTree(inputs, (), size, path=qa.quadrant_path_2d(10, 10))
    for world in (2, 4, 8):
        parts = find_range_slices(quad, world)
        rep = RangeSliced(quad, parts).cost_report()
        ref = QuadrantSharding(inputs, size, 10, 10, world).cost_report()
        assert set(parts) == set(QuadrantSharding(inputs, size, 10, 10, world).sliced)
        assert rep["inflation"] == pytest.approx(ref["inflation"], rel=1e-12)
        assert rep["largest_slice_fraction"] == pytest.approx(ref["busiest_rank_fraction"], rel=1e-12)
    sweep = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(10, 10))
    assert 5.0 < RangeSliced(sweep, find_range_slices(sweep, 8)).cost_report()["inflation"] < 8.0


def _worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist

    import quimb_amd as qa
    import quimb_amd.device as qd
    from emu_device import EmuDevice
    from oracle import np_oracle as orc
    from quimb_amd.rangeslice import RangeSliced, RangeSlicedExecutor, contract_range_sliced, find_range_slices

    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        qd.set_default_device(EmuDevice())
        arrays, inputs = orc.tn2d_rand(4, 4, 4, seed=6, dtype="float64")
        inputs = [tuple(t) for t in inputs]
        size = {ix: 4 for t in inputs for ix in t}
        tree = qa.find_path(inputs, (), size, "bisection")
        rse = RangeSlicedExecutor(RangeSliced(tree, find_range_slices(tree, 4)), "float64")     # 4 slices over 2 / 3 ranks
        val = contract_range_sliced(rse, arrays)
        m, e = contract_range_sliced(rse, arrays, strip_exponent=True)
        np.save(os.path.join(outdir, f"r{rank}.npy"), np.asarray([np.asarray(val).item(), m * 10.0**e]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_range_sliced_over_ranks_gloo(tmp_path, world):
    import torch.multiprocessing as mp

    from oracle import np_oracle as orc

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    arrays, inputs = orc.tn2d_rand(4, 4, 4, seed=6, dtype="float64")
    want = orc.oracle_array_contract(arrays, inputs, ()).item()
    for r in range(world):
        got = np.load(tmp_path / f"r{r}.npy")
        assert got[0] == pytest.approx(want, rel=1e-11) and got[1] == pytest.approx(want, rel=1e-11)
