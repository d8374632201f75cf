"""CPU tests of the host logic (planner, executor, slicing, interface mirror)
against the oracle, with the device ops interpreted in numpy (tests/emu_device.py)."""

import numpy as np
import pytest

import checks


@pytest.mark.parametrize("dtype", ["float32", "float64", "complex64", "complex128"])
def test_pairwise(emu, dtype):
    checks.check_pairwise(dtype)
    assert emu.calls["contract_pair"] > 0


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_tensordot_matmul(emu, dtype):
    checks.check_tensordot_matmul(dtype)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_layout_ops(emu, dtype):
    checks.check_layout_ops(dtype)


@pytest.mark.parametrize("dtype", ["float32", "float64", "complex128"])
def test_tree_executor(emu, dtype):
    checks.check_tree_executor(dtype)


def test_hyper_network(emu):
    checks.check_hyper_network("float64")


@pytest.mark.parametrize("dtype", ["float32", "float64", "complex128"])
def test_strip_exponent(emu, dtype):
    checks.check_strip_exponent(dtype)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_sliced(emu, dtype):
    checks.check_sliced(dtype)


def test_tensor_contract_semantics(emu):
    checks.check_tensor_contract_semantics()


def test_option_stacks(emu):
    checks.check_option_stacks()


def test_mps_dense(emu):
    checks.check_mps_dense()


def test_ising_small(emu):
    from oracle import np_oracle as orc

    Z = checks.check_ising(5, 6, 0.44)
    assert Z == pytest.approx(orc.ising_partition_exact(5, 6, 0.44), rel=1e-10)


def test_sweep_needs_no_permutes(emu):
    """The executor's layout choice makes a 2D boundary sweep permute-free."""
    from oracle import np_oracle as orc
    import quimb_amd as qa

    arrays, inputs = orc.tn2d_rand(5, 5, 3, seed=1, dtype="float64")
    size = {ix: 3 for t in inputs for ix in t}
    tree = qa.ContractionTree(inputs, (), size, path=qa.sweep_path_2d(5, 5))
    out = qa.TreeExecutor(tree, "float64")(arrays)
    want = orc.oracle_array_contract(arrays, inputs, (), path=tree.get_path())
    checks.assert_close(out.to_numpy(), want, "float64")
    assert emu.calls["permute"] == 0
    assert emu.calls["contract_pair"] == 24


def test_stream_shapes_on_emulator(emu):
    checks.check_stream_kernels("float64")
